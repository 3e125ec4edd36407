for lvl in 1 2 3; do echo "level=$lvl"; RNAD_BUCKET_LEVEL=$lvl python tools/step_probe.py --steps 300 | tail -1;  RNAD_BUCKET_LEVEL=$lvl python tools/step_probe.py --steps 300 --batch-log2 17 | tail -1; done
RNAD_BUCKET_LEVEL=3 RNAD_BUCKET_CHUNK=128 python tools/step_probe.py --steps 300 | tail -1
RNAD_BUCKET_LEVEL=3 RNAD_BUCKET_CHUNK=512 python tools/step_probe.py --steps 300 | tail -1
