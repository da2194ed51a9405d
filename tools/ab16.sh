#!/bin/bash
# like tools/ab.sh in PR_SC_ARITH_F16: alternating runs of the default library and tools/expbuild/libpr_amd_<variant>.so
v=$1; n=${2:-3}
for i in $(seq $n); do
  for lib in "" tools/expbuild/libpr_amd_$v.so; do
    PR_AMD_LIB=$lib python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline --sc-arith f16 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('${lib:-default}', 'ms/launch %.2f' % d['roofline']['ms_per_launch'], 'step %.2f' % d['ms_per_step'], 'top1', d['parity']['planted_top1_correct'])
"
  done
done
