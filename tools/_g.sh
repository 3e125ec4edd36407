python -m pytest tests/test_hip_bucket.py tests/test_hip_graph.py -m gpu -x -q 2>&1 | tail -3
tools/variant_time.sh base "k_bucket_learn|k_bucket_rollout"
python tools/step_probe.py --steps 300
python tools/step_probe.py --steps 300 --batch-log2 17
