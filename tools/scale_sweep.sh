#!/bin/bash
# 1 -> 8 GPU weak-scaling sweep of the headline workload, ready for whoever gets a multi-GPU node (none in this project's
# pool: never run here).  bench.py --gpus N starts its own ranks.  Prints images/s per N and the efficiency against N = 1,
# for the gradient-exchange variants: default (bucketed all-reduce under the backward pass, enqueued once backward has left the
# one-workgroup-per-CU layers), FRHIP_DP_OVERLAP=1 (every bucket as soon as it is complete), =0 (exchange after backward),
# FRHIP_DP_GATE_HO=56 (a later gate), FRHIP_SPLIT_STRIPS=1 (half-channel strip workgroups).
#   bash tools/scale_sweep.sh [steps] > profiles/rNN_scale_sweep.txt
R=${GRAFT_REPO_ROOT:-$(dirname $(dirname $(readlink -f $0)))}; cd $R
STEPS=${1:-50}
NG=$(python -c "import torch; print(torch.cuda.device_count())")
for v in "" "FRHIP_DP_OVERLAP=1" "FRHIP_DP_OVERLAP=0" "FRHIP_DP_GATE_HO=56" "FRHIP_SPLIT_STRIPS=1"; do
  base=""
  for n in 1 2 4 8; do
    [ $n -le $NG ] || continue
    out=$(env $v python bench.py --gpus $n --steps $STEPS --warmup 10 --no-roofline --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1)
    ips=$(echo "$out" | python -c "import sys,json; print(json.loads(sys.stdin.readline())['value'])")
    [ -z "$base" ] && base=$ips
    python -c "print('%-24s N=%d  %10.1f img/s  %7.3f ms/step  efficiency %.3f' % ('${v:-default}', $n, $ips, 256.0*$n/$ips*1e3, $ips/($n*$base)))"
  done
done
