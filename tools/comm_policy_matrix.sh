#!/bin/bash
# Step ms with the gradient collectives of a data-parallel run emulated on ONE GPU (tools/cu_hog.py --comm): launch policy x
# CUs held per collective x emulated all-reduce rate.  gpurun -- bash tools/comm_policy_matrix.sh > profiles/...
R=${GRAFT_REPO_ROOT:-$(dirname $(dirname $(readlink -f $0)))}; cd $R
python tools/cu_hog.py --hog 0 --steps 30 2>/dev/null | tail -1
for gbps in 150 75; do
  for k in 8 32; do
    for pol in 2 1 0; do
      FRHIP_DP_OVERLAP=$pol python tools/cu_hog.py --comm $k --comm-gbps $gbps --steps 30 2>/dev/null | tail -1
    done
  done
done
