timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r05_gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_gputest.log
grep -E "passed|failed|rc=" gpurun_out/r05_gputest.log
