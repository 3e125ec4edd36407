#!/bin/bash
# What k_bucket_learn_c's time is made of: the kernel rebuilt with parts removed (RNAD_ABLATE bit mask in csrc/bucket.hip: 1 = no LDS
# atomics, 8 = no phase 1 (the per-lane steps below the cut), 16 = no phase 2 (the steps the workgroup shares)), timed by rocprofv3 over the
# frozen-weights probe (the ablated kernels write garbage gradients).
#   tools/ablate_learn.sh [probe args]     (on the GPU box; the variants are built here if missing)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in 1 8 16 24 9 17; do
  [ -f r-nad_amd/csrc/_variants/abl$v.so ] || tools/build_variant.sh abl$v bucket.hip -DRNAD_ABLATE=$v > /dev/null 2>&1
done
tools/variant_time.sh base 'k_bucket_learn' --freeze "$@"
for v in 1 8 16 24 9 17; do
  tools/variant_time.sh r-nad_amd/csrc/_variants/abl$v.so 'k_bucket_learn' --freeze "$@"
done
