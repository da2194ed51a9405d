#!/bin/bash
# usage: abn.sh <rounds> <variant> [<variant> ...]   alternating bench runs of the default library and tools/expbuild/libpr_amd_<variant>.so
n=$1; shift
for i in $(seq $n); do
  for v in "" "$@"; do
    lib=${v:+tools/expbuild/libpr_amd_$v.so}
    PR_AMD_LIB=$lib python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); p = d['per_rank'][0]['phases_ms']; print('${v:-default}', 'step %.2f' % d['ms_per_step'], 'select %.3f' % p['select'], 'pack %.3f' % p['pack(db)'], 'top1', d['parity']['planted_top1_correct'])
"
  done
done
