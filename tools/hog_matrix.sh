#!/bin/bash
# Step ms vs CUs held by another kernel, for the strip-launch variants (gpurun -- bash tools/hog_matrix.sh > profiles/...).
R=${GRAFT_REPO_ROOT:-$(dirname $(dirname $(readlink -f $0)))}; cd $R
for v in "" "FRHIP_SPLIT_STRIPS=1"; do
  for k in 0 4 8 16 32; do
    env $v python tools/cu_hog.py --hog $k --steps 20 2>/dev/null | tail -1
  done
done
