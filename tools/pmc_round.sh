#!/bin/bash
# PMC evidence for the kernels of one training step (run on the GPU box: gpurun -- bash tools/pmc_round.sh [tag]).
# Three SEPARATE rocprofv3 passes over tools/kbench.py suite (counters only with --kernel-trace, as the pool requires):
#   f: FETCH_SIZE   w: WRITE_SIZE   s: SQ_* + GRBM_GUI_ACTIVE   i: instruction mix (SQ_INSTS_*: vector ALU / LDS / vector
#   memory instructions per MFMA -- the SIMD issue port is what the MFMA kernels compete for)
# plus one un-profiled run for the event-timed durations.  Output: gpurun_out/pmc_$tag/{f,w,s}/..., times.log;
# tools/pmc_summary.py turns them into profiles/$TAG_pmc_kernels.json.
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
python3 $R/tools/kbench.py suite --iters 20 > $O/times.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- python3 $R/tools/kbench.py suite --iters 3 > $O/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- python3 $R/tools/kbench.py suite --iters 3 > $O/w.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/s -o s -- python3 $R/tools/kbench.py suite --iters 3 > $O/s.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD --output-format csv -d $O/i -o i -- python3 $R/tools/kbench.py suite --iters 3 > $O/i.log 2>&1
# keep only our kernels' rows (the ATen fill / rand kernels have kilobyte-long names)
for d in f w s i; do for c in $(find $O/$d -name "*counter_collection.csv"); do
  python3 - "$c" <<'PY'
import csv, sys
p = sys.argv[1]
rows = list(csv.DictReader(open(p)))
keep = [r for r in rows if "at::native" not in r["Kernel_Name"] and "rocclr" not in r["Kernel_Name"]]
with open(p, "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()) if rows else [])
    w.writeheader()
    w.writerows(keep)
PY
done; find $O/$d -name "*kernel_trace.csv" -delete; done
cd $R && python3 tools/pmc_summary.py $O gpurun_out/pmc_$TAG/summary.json && tail -40 gpurun_out/pmc_$TAG/summary.txt
