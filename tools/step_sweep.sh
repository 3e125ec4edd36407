#!/bin/bash
# step time of the default mode for several bucket-table sizes (RNAD_BUCKET_ROWS) and work-item sizes: tools/step_sweep.sh [probe args]
for rows in 8 16 32 64 128 256 512; do for ch in 256; do
  echo -n "rows=$rows chunk=$ch  "; RNAD_BUCKET_ROWS=$rows RNAD_BUCKET_CHUNK=$ch python tools/step_probe.py --steps 200 "$@" | tail -1
done; done
