#!/bin/bash
# Resource usage of every kernel of one csrc file (VGPRs, AGPRs, scratch, LDS, occupancy), from hipcc's own remarks:
#   tools/kres.sh conv3x3_s2_strip.hip [extra hipcc flags]
# No GPU needed (cross-compiles for gfx950).
SRC=$1; shift
CS=$(dirname "$0")/../stylegan-for-facerec_amd/frhip/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$CS/../../../include -Wno-unused-value "$@" \
  -Rpass-analysis=kernel-resource-usage -c $CS/$SRC -o /tmp/kres_$$.o 2>&1 | python3 -c '
import re, sys, subprocess
cur = None
rows = []
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    m = re.search(r"remark:\s+([A-Za-z ]+(?:\[[A-Za-z/ ]*\])?): (\d+)", line)
    if cur is not None and m:
        cur[m.group(1).strip()] = int(m.group(2))
    elif "error" in line:
        sys.stdout.write(line)
names = subprocess.run(["/usr/bin/c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines() if rows else []
for r, n in zip(rows, names):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"\(.*", "", n); n = n.replace("void ", "")
    print("%-66s vgpr %3d agpr %3d scratch %5d lds %6d occ %d" % (n[:66], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("ScratchSize [bytes/lane]", -1), r.get("LDS Size [bytes/block]", -1), r.get("Occupancy [waves/SIMD]", -1)))
'
rm -f /tmp/kres_$$.o
