#!/usr/bin/env python3
"""One-off deep fuzz of the HIP path against the oracle (many seeds); same generator as tests/test_search_gpu.py::test_fuzz_many_small_indexes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tests.test_search_gpu as T
import finito_amd as fa

def main(n_seeds=12):
    import types
    src = T.test_fuzz_many_small_indexes
    for seed in range(n_seeds):
        # re-run the test body with a different seed
        orig = np.random.default_rng
        np.random.default_rng = lambda s, _o=orig, _seed=seed: _o(1000 + 7919 * _seed)
        try:
            for kern in (4, 3, 2, 0):
                fa.lib().fin_set_option(b"kernel", kern)
                src()
            fa.lib().fin_set_option(b"kernel", 4)
            for ptab, prepass in ((-1, 1), (0, 1), (3, 1), (6, 0), (-1, 0)):   # walk mode, cold restarts and probes of the default kernel
                T.test_fuzz_walks_restarts_and_probes(ptab, prepass)
        finally:
            np.random.default_rng = orig
        print("seed", seed, "ok", flush=True)
    fa.lib().fin_set_option(b"kernel", 4)

if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 12)
