for m in 9 12 16 24 32; do echo "m=$m: $(python tools/latency_probe.py $m 2>/dev/null | tail -1)"; done
timeout 900 python tools/fuzz_all.py 61 30 match,matcher,group 2>&1 | grep -E "^BAD|fuzz_all"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_binary.py -q 2>&1 | grep -E "passed|failed"
