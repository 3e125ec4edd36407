python -m pytest tests/test_hip_bucket.py -x -q -k "long_episodes or ragged_last" 2>&1 | tail -12
