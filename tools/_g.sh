python -m pytest tests/test_hip_fullsize.py -m gpu -x -q -k "c4_full" 2>&1 | tail -25
