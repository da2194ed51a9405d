mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r05_gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_gputest.log
grep -E "passed|failed|rc=" gpurun_out/r05_gputest.log
for m in 1 8; do echo "m=$m: $(python tools/latency_probe.py $m 2>/dev/null | tail -1)"; done
PR_SC_BINARY=0 python tools/latency_probe.py 1 2>/dev/null | tail -1
PR_SC_BINARY=0 PR_SC_ONLINE=h python tools/latency_probe.py 1 2>/dev/null | tail -1
