#!/bin/bash
# learner / rollout time of the default step for several partition depths and item sizes, after `--pre` training steps
for lvl in 3 4; do for ch in 256 1024; do
  echo "level=$lvl chunk=$ch"; RNAD_BUCKET_LEVEL=$lvl RNAD_BUCKET_CHUNK=$ch python tools/step_probe.py --steps 300 "$@" | tail -1
done; done
