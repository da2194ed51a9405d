for m in 12 16 24 32; do echo "f16 m=$m: $(python tools/latency_probe.py $m f16 2>/dev/null | tail -1)"; done
PR_SC_BINARY=0 python tools/latency_probe.py 16 2>/dev/null | tail -1
timeout 1500 python -m pytest tests/test_gpu_f16.py tests/test_gpu_parity.py tests/test_gpu_binary.py -q 2>&1 | grep -E "passed|failed"
timeout 600 python tools/fuzz_all.py 95 30 match,matcher,fused 2>&1 | grep -E "^BAD|fuzz_all:"
