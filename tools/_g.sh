python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python tools/step_probe.py --steps 300
python tools/step_probe.py --steps 300 --batch-log2 17
