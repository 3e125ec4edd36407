timeout 600 python -m pytest tests/test_hip_distributed.py -m gpu -x -q 2>&1 | tail -25
