python -m pytest tests -m gpu -x -q 2>&1 | tail -3
bash tools/step_kernels.sh 2>&1 | grep -E "finish|upper|zero|sum of"
python tools/step_probe.py --steps 1000 2>&1 | tail -1
