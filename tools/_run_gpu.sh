timeout 1200 python -m pytest tests/test_dist_gloo.py -m gpu -x -q 2>&1 | tail -5
