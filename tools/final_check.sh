#!/bin/bash
# Round-end check on the GPU box: the GPU suite, smoke(), and the default bench line with its wall time.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -x -q -m gpu ) > gpurun_out/final_suite.log 2>&1; tail -6 gpurun_out/final_suite.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time python bench.py ) > gpurun_out/final_bench.log 2>&1; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"traffic_source": "[^"]*"\|real.*' gpurun_out/final_bench.log | head -12
