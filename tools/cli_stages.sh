#!/bin/bash
# Stage timings of `finito search-fmin` (FINITO_TIMING=1) on the files tools/cli_e2e.sh leaves in /tmp/fin_e2e, for three sinks.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
T=/tmp/fin_e2e
[ -f $T/r.fq ] || tools/cli_e2e.sh > /dev/null 2>&1
now() { date +%s.%N; }
el() { python3 -c "print('%.2f' % ($(now) - $1))"; }
for SINK in $T/out_s.txt /dev/shm/fin_out_s.txt /dev/null; do
  for MODE in gpu host; do
    [ $MODE == host ] && export FINITO_HOST_FORMAT=1 || unset FINITO_HOST_FORMAT
    S=$(now); FINITO_TIMING=1 finito_amd/finito search-fmin -i $T/idx -q $T/r.fq -o $SINK 2>&1 | grep -E "timing|us/query"; echo "== sink $SINK, text by $MODE: wall $(el $S) s"
  done
done
rm -f /dev/shm/fin_out_s.txt $T/out_s.txt
