mkdir -p gpurun_out
python tools/probe_bias.py > gpurun_out/r02_bias.txt 2>&1; cat gpurun_out/r02_bias.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r02_prof1 -o r02 -- python /root/repo/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/r02_prof1.log 2>&1
cd /root/repo
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r02_prof1/**/*kernel_stats.csv',recursive=True)
print(f)
for r in list(csv.DictReader(open(f[0])))[:14]: print(r['Name'][:60], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
PY
