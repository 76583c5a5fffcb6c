#!/bin/bash
# round 4, eighth GPU pass: reworked Linear kernels + tile-width test + pending model tests + kernel stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/prof_r4c
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "linear or flatten_order or tile_width" > gpurun_out/r4_t3.log 2>&1; tail -4 gpurun_out/r4_t3.log
timeout 2400 python -m pytest tests/test_gpu_model.py -x -q -k "bn2_backward or configs0 or bench_size or two_sgd or full_step_matches" > gpurun_out/r4_t5.log 2>&1; tail -6 gpurun_out/r4_t5.log; grep -h "vs oracle" gpurun_out/r4_t5.log | cut -c1-420
cd /tmp
FRHIP_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r4c/single -o r -- python3 $R/bench.py --steps 12 --warmup 4 --no-roofline --no-cpu-baseline --no-other-configs > $R/gpurun_out/prof_r4c/single.log 2>&1
cd $R
for m in single; do db=$(find gpurun_out/prof_r4c/$m -name "*.db" | head -1); python tools/trace_gaps.py $db --csv gpurun_out/prof_r4c/kernel_stats_$m.csv --timeline gpurun_out/prof_r4c/timeline_$m.txt > gpurun_out/prof_r4c/gaps_$m.txt 2>&1; head -3 gpurun_out/prof_r4c/gaps_$m.txt; done
find gpurun_out/prof_r4c -name "*.db" -delete
grep -i "linear\|dropout_cm\|conv_wgrad_kernel<unsigned short, 128, 128" gpurun_out/prof_r4c/kernel_stats_single.csv | cut -c1-160
b() { printf "%-50s " "$*"; env "$@" timeout 600 python bench.py --steps 40 --warmup 10 --no-roofline --no-cpu-baseline --no-other-configs 2>gpurun_out/r4_b.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f ms  loss %.3f' % (d['ms_per_step'], d['config'].get('final_loss', 0)))" || tail -5 gpurun_out/r4_b.err; }
r3() { printf "%-50s " "round-3 tree $*"; (cd _r3 && env "$@" timeout 600 python bench.py --steps 40 --warmup 10 --no-roofline --no-cpu-baseline --no-other-configs 2>../gpurun_out/r4_b3.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f ms  loss %.3f' % (d['ms_per_step'], d['config'].get('final_loss', 0)))" || tail -5 ../gpurun_out/r4_b3.err); }
{ r3 A=1; b A=1; b FRHIP_LINEAR_CM=0; r3 A=1; b A=1; } 2>&1 | tee gpurun_out/r4_ab6.log
bash tools/hog_matrix.sh 2>&1 | tee gpurun_out/r4_hog_matrix.txt
