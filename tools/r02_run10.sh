mkdir -p gpurun_out
(time python -m pytest tests -m gpu -q -x --durations=5) > gpurun_out/r02_t10.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|^real|^E  " gpurun_out/r02_t10.log | cut -c1-300 | head -20
for fx in "" "--force-exchange"; do
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra $fx 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fx=[$fx]', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'matcher ms', round(d['roofline']['ms_per_launch'],3), 'rest', round(d['ms_per_step']-d['roofline']['ms_per_launch'],3), d['parity'])"
done
