python -m pytest tests/test_hip_bucket.py tests/test_hip_graph.py tests/test_hip_fullsize.py -x -q 2>&1 | tail -4
bash tools/step_kernels.sh 2>&1 | grep -E "learn|records|sum of"
