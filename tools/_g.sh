python -m pytest tests/test_hip_bucket.py -x -q 2>&1 | tail -30
