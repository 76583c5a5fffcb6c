#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/profile_round.sh r04 > gpurun_out/r4_prof.log 2>&1; tail -12 gpurun_out/r4_prof.log | cut -c1-300
bash tools/profile_config3.sh > gpurun_out/r4_prof_c3.log 2>&1; tail -2 gpurun_out/r4_prof_c3.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
