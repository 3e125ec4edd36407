python -m pytest tests/test_hip_bucket.py -m gpu -x -q 2>&1 | tail -15
bash tools/profile_bench.sh r02c --steps 50 --warmup 5 2>&1 | tail -3
