#!/bin/bash
# What k_bucket_learn's time is made of: the kernel rebuilt with parts removed (RNAD_ABLATE bit mask in csrc/bucket.hip: 1 = no LDS
# atomics, 2 = every record gather hits one line, 4 = no V-trace / NeuRD arithmetic), timed by rocprofv3 over the frozen-weights probe.
#   tools/ablate_learn.sh [probe args]     (on the GPU box; the variants are built here if missing)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in 1 2 4 3 5 6 7; do
  [ -f r-nad_amd/csrc/_variants/abl$v.so ] || tools/build_variant.sh abl$v bucket.hip -DRNAD_ABLATE=$v > /dev/null 2>&1
done
tools/variant_time.sh base 'k_bucket_learn' --freeze "$@"
for v in 1 2 4 3 5 6 7; do
  tools/variant_time.sh r-nad_amd/csrc/_variants/abl$v.so 'k_bucket_learn' --freeze "$@"
done
