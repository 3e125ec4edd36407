python -m pytest tests/test_hip_graph.py tests/test_hip_bucket.py -m gpu -x -q 2>&1 | tail -25
python tools/step_probe.py --steps 200
python tools/step_probe.py --steps 200 --batch-log2 17
