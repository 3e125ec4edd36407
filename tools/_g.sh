python -m pytest tests/test_hip_graph.py tests/test_hip_curve.py -x -q 2>&1 | tail -3
bash tools/step_kernels.sh 2>&1 | grep -E "optim|sum of"
python tools/step_probe.py --steps 1000 2>&1 | tail -1
