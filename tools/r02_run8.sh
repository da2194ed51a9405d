mkdir -p gpurun_out
for fx in "" "--force-exchange" "" "--force-exchange"; do
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra $fx 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fx=[$fx]', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'matcher ms', round(d['roofline']['ms_per_launch'],3), 'rest', round(d['ms_per_step']-d['roofline']['ms_per_launch'],3))"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/fx_prof -o fx -- python /root/repo/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra --force-exchange > /root/repo/gpurun_out/fx_prof.log 2>&1
cd /root/repo; python profiles/summarize.py gpurun_out/fx_prof/fx_results.db | cut -c1-150 | head -24; rm -rf gpurun_out/fx_prof
