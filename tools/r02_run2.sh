mkdir -p gpurun_out
(time python -m pytest tests -m gpu -q --durations=8) > gpurun_out/r02_t2.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|^real" gpurun_out/r02_t2.log
python bench.py --steps 10 --warmup 2 > gpurun_out/r02_b2.json 2> gpurun_out/r02_b2.err; tail -n 2 gpurun_out/r02_b2.json; tail -n 3 gpurun_out/r02_b2.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r02_prof2 -o r02 -- python /root/repo/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/r02_prof2.log 2>&1
cd /root/repo; python profiles/summarize.py gpurun_out/r02_prof2/r02_results.db | cut -c1-170 | head -14
