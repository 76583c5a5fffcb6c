#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/pmc_round.sh r04 > gpurun_out/r4_pmc.log 2>&1; tail -45 gpurun_out/r4_pmc.log | cut -c1-260
