#!/bin/bash
# usage: tools/exp_libs.sh <workload> lib1.so lib2.so ...   (bench one workload with prebuilt libraries of different commits)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
W=$1; shift
cp finito_amd/libfinito_amd.so /tmp/lib_keep.so
for L in "$@"; do
  cp $L finito_amd/libfinito_amd.so
  python bench.py --workload $W --steps 3 --warmup 1 --no-cpu --no-e2e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('LIB [$L]', 'k-mers/s %.4g' % d['value'], {k: round(v, 2) for k, v in d['roofline'].get('kernel_ms_parts').items()}, d['roofline']['pipeline_queue_slots']['stream_rounds'][:3])"
done
cp /tmp/lib_keep.so finito_amd/libfinito_amd.so
