"""Register / scratch / LDS footprint of every kernel in a hipcc -S device listing (and a diff of two listings).

usage: python tools/kregs.py new.s [old.s] [name filter ...]
Build a listing with: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -S --cuda-device-only file.hip -o file.s
"""
import re
import subprocess
import sys


def regs(path):
    s = open(path).read()
    out = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, flags=re.S):
        body = m.group(2)
        f = lambda k: int(re.search(k + r" (\d+)", body).group(1))
        out[m.group(1)] = (f("next_free_vgpr"), f("next_free_sgpr"), f("private_segment_fixed_size"),
                           f("group_segment_fixed_size"))
    return out


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
        return dict(zip(names, r.stdout.split("\n")))
    except OSError:
        return {n: n for n in names}


if __name__ == "__main__":
    paths = [a for a in sys.argv[1:] if a.endswith(".s")]
    filt = [a for a in sys.argv[1:] if not a.endswith(".s")]
    new = regs(paths[0])
    old = regs(paths[1]) if len(paths) > 1 else {}
    dn = demangle(list(new))
    print("# vgpr sgpr scratch lds  (old -> new)")
    for k, v in new.items():
        name = re.sub(r"\(anonymous namespace\)::", "", dn[k])[:110]
        if filt and not any(f in name for f in filt):
            continue
        print(("%s -> " % (old[k],) if k in old else "") + "%s  %s" % (v, name))
