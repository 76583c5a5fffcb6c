#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output) of tools/kbench.py into the per-launch
HBM traffic record bench.py reports as roofline.traffic.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_f -o f -- \
        python3 $R/tools/kbench.py strip 256 256 14 --mode 1 --pro 0 --epi 2 --iters 5
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_w -o w -- (same command)
    python tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w "conv3x3_strip<256,256,14,PRO=0>" conv3x3_strip_kernel

Corrections per MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are in KB; on gfx950
FETCH_SIZE reports half of the bytes of a wide (16 B per lane) coalesced streaming read -> doubled; WRITE_SIZE exact.
"""
import csv
import glob
import json
import os
import sys


def per_launch(folder, counter, kernel_substr):
    vals = []
    for path in glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == counter and kernel_substr in row["Kernel_Name"]:
                    vals.append(float(row["Counter_Value"]))
    if not vals:
        raise SystemExit("no %s rows for %s under %s" % (counter, kernel_substr, folder))
    vals.sort()
    return vals[len(vals) // 2], len(vals)


def main():
    fdir, wdir, label, substr = sys.argv[1:5]
    B = 256
    fetch_kb, nf = per_launch(fdir, "FETCH_SIZE", substr)
    write_kb, nw = per_launch(wdir, "WRITE_SIZE", substr)
    hbm = int(fetch_kb * 1024 * 2 + write_kb * 1024)
    # 256->256 @14x14, B=256, data gradient with the PReLU-backward epilogue: g strip + aux (y1) + output + weights
    act = B * 14 * 14 * 256 * 2
    algorithmic = 3 * act + 256 * 256 * 9 * 2
    rec = {"kernel": label, "launches_sampled": [nf, nw], "FETCH_SIZE_KB_per_launch": fetch_kb,
           "WRITE_SIZE_KB_per_launch": write_kb,
           "correction": "gfx950: FETCH_SIZE counts half of a 16-B/lane streaming read -> doubled; WRITE_SIZE exact",
           "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": algorithmic,
           "note": "algorithmic = input strip + aux + output (25.7 MB each at B=256) + weights 1.2 MB"}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r01_pmc_dominant_kernel.json")
    with open(out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
