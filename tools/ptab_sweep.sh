#!/bin/bash
# Prefix-table depth against step time (is T = 15 still load-bearing once looks and error bridging go through the k-mer table and
# the string filter?).  usage: tools/ptab_sweep.sh [workload...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
for W in ${@:-chr1}; do for T in 15 14 13 12; do
  FINITO_PTAB_T=$T timeout -k 10 300 python bench.py --workload $W --steps 10 --warmup 2 --no-e2e --no-cpu 2> /dev/null | python -c "
import json,sys; d=json.load(sys.stdin); c=d['config']; print('$W T=$T', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['kernel_ms_parts'].items()}, 'tables B/base', c['derived_tables_bytes_per_indexed_base'], 'fast', d['roofline'].get('reads_finished_by_the_fast_path'))"
done; done
