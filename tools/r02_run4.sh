mkdir -p gpurun_out
python -m pytest tests/test_cli.py -m gpu -q -k config1_end_to_end 2>&1 | grep -E "^E  " | cut -c1-900 | head -12
python bench.py --steps 10 --warmup 2 > gpurun_out/r02_b4.json 2> gpurun_out/r02_b4.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02_b4.json').read().strip().splitlines()[-1])
print(json.dumps(d.get('extra'),indent=1)); print(d['value'], d['ms_per_step'], d['cpu_baseline'])
PY
