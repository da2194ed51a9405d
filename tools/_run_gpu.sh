bash tools/ab.sh hot 2
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "hipgraph" 2>&1 | tail -2
