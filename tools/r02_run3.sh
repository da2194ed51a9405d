mkdir -p gpurun_out
(time python -m pytest tests -m gpu -q --durations=5) > gpurun_out/r02_t3.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|^real" gpurun_out/r02_t3.log
python bench.py --steps 10 --warmup 2 > gpurun_out/r02_b3.json 2> gpurun_out/r02_b3.err; tail -n 2 gpurun_out/r02_b3.json; tail -n 3 gpurun_out/r02_b3.err
bash tools/profile_round.sh r02_mid > /dev/null 2>&1
grep -E "kernel trace|avg=" gpurun_out/r02_mid.txt | cut -c1-150 | head -12
