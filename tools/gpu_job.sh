#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -x -q -m gpu --durations=6 ) > gpurun_out/r4_full_suite5.log 2>&1; tail -14 gpurun_out/r4_full_suite5.log | cut -c1-200
