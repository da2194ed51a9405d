#!/bin/bash
# usage: abk.sh <rounds> <variant> [<variant> ...]   alternating bench runs of the default library and tools/expbuild/libpr_amd_<variant>.so:
# per-launch times of the SC matcher (structure | binary), step, planted top-1
n=$1; shift
for i in $(seq $n); do
  for v in "" "$@"; do
    lib=${v:+tools/expbuild/libpr_amd_$v.so}
    PR_AMD_LIB=$lib python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); L = d['roofline']['launches']; print('%-10s' % '${v:-default}', 'structure %.3f ms' % L[0]['ms'], 'binary %.3f ms' % L[1]['ms'], 'step %.2f' % d['ms_per_step'], 'top1', d['parity']['planted_top1_correct'])
"
  done
done
