python -m pytest tests/test_hip_bucket.py tests/test_hip_graph.py tests/test_hip_fullsize.py -x -q 2>&1 | tail -12
python tools/step_probe.py --steps 1000 2>&1 | tail -1
bash tools/step_kernels.sh 2>&1 | grep -E "rollout|learn|records|sum of"
