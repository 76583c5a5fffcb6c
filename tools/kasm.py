"""Instruction census of one kernel in a hipcc -S listing: counts by class, and how many SGPR-spill lane ops / waits sit
between the first and the last MFMA.  usage: python tools/kasm.py file.s <substring of the mangled name>"""
import re
import sys


def kernel_body(path, sub):
    lines = open(path).read().split("\n")
    start = [i for i, l in enumerate(lines) if l.startswith("_Z") and sub in l and l.rstrip().split(":")[0].endswith(("i", "s", "l"))
             or (l.startswith("_Z") and sub in l and ": " in l)]
    if not start:
        raise SystemExit("no kernel matches %r" % sub)
    i = start[0]
    j = i
    while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
        j += 1
    return lines[i], lines[i + 1:j]


if __name__ == "__main__":
    name, body = kernel_body(sys.argv[1], sys.argv[2])
    ins = [l.strip().split()[0] for l in body if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    cnt = {}
    for k in ins:
        cnt[k] = cnt.get(k, 0) + 1
    mf = [i for i, l in enumerate(body) if "v_mfma" in l]
    inside = body[mf[0]:mf[-1]] if mf else []
    c = lambda ls, k: sum(1 for l in ls if k in l)
    print(name.split(":")[0][:100])
    print("instructions %d  mfma %d  ds_read %d  global_load %d  global_store %d  s_load %d  scratch %d" % (
        len(ins), c(body, "v_mfma"), c(body, "ds_read") + c(body, "ds_load"), c(body, "global_load"), c(body, "global_store"),
        c(body, "s_load"), c(body, "scratch_")))
    print("lane spill ops total %d, between first and last MFMA %d; s_waitcnt inside %d; valu inside %d" % (
        c(body, "v_readlane") + c(body, "v_writelane"), c(inside, "v_readlane") + c(inside, "v_writelane"),
        c(inside, "s_waitcnt"), sum(1 for l in inside if re.match(r"\s+v_(?!mfma)", l))))
