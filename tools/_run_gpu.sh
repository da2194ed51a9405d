mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "pack" 2>&1 | tail -3
echo "once:"; python tools/bench_pack.py 2>&1 | tail -1
echo "col:"; PR_SC_PACK=col python tools/bench_pack.py 2>&1 | tail -1
python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('step %.2f' % d['ms_per_step'], d['value'], d['per_rank'][0]['phases_ms'])
"
