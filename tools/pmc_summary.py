#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/pmc_round.sh into one record per kernel instance.

    python tools/pmc_summary.py gpurun_out/pmc_r02 profiles/r02_pmc_kernels.json

Per kernel (B = 256, bf16): event-timed duration (un-profiled run), HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE:
MI355X_MICROARCH.md, HBM section -- FETCH_SIZE counts half of a 16-B/lane streaming read on gfx950, both are in KB),
algorithmic bytes and FLOPs, the HBM and MFMA floors, and from the SQ pass:
  mfma_util   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)   (busy cycles count 16 per 16x16x32 MFMA)
  wait_lds    = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES ; wait_any / wait_inst / active likewise (quad-cycle units cancel)
"""
import csv
import glob
import json
import os
import re
import sys

B = 256
HBM_ACHIEVABLE = 6.3e12   # MI355X_MICROARCH.md: 8 TB/s spec, 6.29 measured
MFMA_PEAK = 2.5e15


def act(W, C):
    return B * W * W * C * 2


# label -> (kernel-name regex, algorithmic bytes, FLOPs)
def table():
    w3 = lambda ci, co: ci * co * 9 * 2  # noqa: E731
    f3 = lambda W, ci, co: 2.0 * B * W * W * ci * co * 9  # noqa: E731
    dw = lambda ci, co: ci * co * 9 * 4  # noqa: E731
    return {
        "strip_64_64_112_fwd": (r"conv3x3_roll64_kernel<112, 1, false>|conv3x3_strip_kernel<64, 64, 112, \d+, \d+, \d+, 1, 1,", 2 * act(112, 64) + w3(64, 64), f3(112, 64, 64)),
        "strip_64_64_112_dgrad": (r"conv3x3_roll64_kernel<112, 0, true>|conv3x3_strip_kernel<64, 64, 112, \d+, \d+, \d+, 1, 0,", 3 * act(112, 64) + w3(64, 64), f3(112, 64, 64)),
        "s2_64_56_fwd": (r"conv3x3_s2_roll64_kernel<0, 2>|conv3x3_s2_kernel<64, 64, 56, \d+, \d+, \d+, 0, 2[,>]", act(112, 64) + act(56, 64) + w3(64, 64), f3(56, 64, 64)),
        "s2_64_56_dgrad": (r"conv3x3_s2_roll64_kernel<1, 0>|conv3x3_s2_kernel<64, 64, 56, \d+, \d+, \d+, 1, 0[,>]", act(56, 64) + 2 * act(112, 64) + w3(64, 64), f3(56, 64, 64)),
        "s2_128_28_fwd": (r"conv3x3_s2_ws_kernel<128, 28, \d+, \d+, 2>|conv3x3_s2_kernel<128, \d+, 28, \d+, \d+, \d+, 0, 2[,>]", act(56, 128) + act(28, 128) + w3(128, 128), f3(28, 128, 128)),
        "s2_128_28_dgrad": (r"conv3x3_s2_kernel<128, \d+, 28, \d+, \d+, \d+, 1, 0[,>]", act(28, 128) + 2 * act(56, 128) + w3(128, 128), f3(28, 128, 128)),
        "s2_256_14_fwd": (r"conv3x3_s2_ws_kernel<256, 14, \d+, \d+, 2>|conv3x3_s2_kernel<256, \d+, 14, \d+, \d+, \d+, 0, 2[,>]", act(28, 256) + act(14, 256) + w3(256, 256), f3(14, 256, 256)),
        "s2_256_14_dgrad": (r"conv3x3_s2_kernel<256, \d+, 14, \d+, \d+, \d+, 1, 0[,>]", act(14, 256) + 2 * act(28, 256) + w3(256, 256), f3(14, 256, 256)),
        "s2_512_7_fwd": (r"conv3x3_s2_ws_kernel<512, 7, \d+, \d+, 2>|conv3x3_s2_kernel<512, \d+, 7, \d+, \d+, \d+, 0, 2[,>]", act(14, 512) + act(7, 512) + w3(512, 512), f3(7, 512, 512)),
        "s2_512_7_dgrad": (r"conv3x3_s2_kernel<512, \d+, 7, \d+, \d+, \d+, 1, 0[,>]", act(7, 512) + 2 * act(14, 512) + w3(512, 512), f3(7, 512, 512)),
        "strip_64_64_56_fwd_bn": (r"conv3x3_roll64_kernel<56, 1, false>|conv3x3_strip_kernel<64, 64, 56, \d+, \d+, \d+, 1, 1,", 2 * act(56, 64) + w3(64, 64), f3(56, 64, 64)),
        "strip_64_64_56_fwd_prelu": (r"conv3x3_roll64_kernel<56, 2, false>|conv3x3_strip_kernel<64, 64, 56, \d+, \d+, \d+, 1, 2,", 2 * act(56, 64) + w3(64, 64), f3(56, 64, 64)),
        "strip_64_64_56_dgrad": (r"conv3x3_roll64_kernel<56, 0, true>|conv3x3_strip_kernel<64, 64, 56, \d+, \d+, \d+, 1, 0,", 3 * act(56, 64) + w3(64, 64), f3(56, 64, 64)),
        # weight gradients: algorithmic bytes = both operands once + dW (fp32) once; the slabs are overhead
        "wgs_64_64_112": (r"conv_wgrad_vr_kernel<112, 1>|conv_wgrad_strip_kernel<112,", 2 * act(112, 64) + dw(64, 64), f3(112, 64, 64)),
        "wgs_64_64_56": (r"conv_wgrad_vr_kernel<56, 2>|conv_wgrad_strip_kernel<56, \d+, \d+, \d+, 2, false", 2 * act(56, 64) + dw(64, 64), f3(56, 64, 64)),
        "strip_128_128_28_fwd": (r"conv3x3_strip_kernel<128, 128, 28, \d+, \d+, \d+, 1, 1,", 2 * act(28, 128) + w3(128, 128), f3(28, 128, 128)),
        "strip_256_256_14_fwd_bn": (r"conv3x3_strip_kernel<256, 256, 14, \d+, \d+, \d+, 1, 1,", 2 * act(14, 256) + w3(256, 256), f3(14, 256, 256)),
        "strip_256_256_14_fwd_prelu": (r"conv3x3_strip_kernel<256, 256, 14, \d+, \d+, \d+, 1, 2,", 2 * act(14, 256) + w3(256, 256), f3(14, 256, 256)),
        # FR_PRO_RESBN: two sources in, the residual sum + the convolution output out
        "strip_256_256_14_fwd_resbn": (r"conv3x3_strip_kernel<256, 256, 14, \d+, \d+, \d+, 1, 4,", 4 * act(14, 256) + w3(256, 256), f3(14, 256, 256)),
        "strip_256_256_14_dgrad": (r"conv3x3_strip_kernel<256, 256, 14, \d+, \d+, \d+, 1, 0,", 3 * act(14, 256) + w3(256, 256), f3(14, 256, 256)),
        "wgs_256_256_14": (r"conv_wgrad_roll_kernel<14, 2>|conv_wgrad_strip_kernel<14,", 2 * act(14, 256) + dw(256, 256), f3(14, 256, 256)),
        "wgs_256_256_14_bn": (r"conv_wgrad_roll_kernel<14, 1>", 2 * act(14, 256) + dw(256, 256), f3(14, 256, 256)),
        "wgs_128_128_28": (r"conv_wgrad_roll_kernel<28, 2>|conv_wgrad_strip_kernel<28,", 2 * act(28, 128) + dw(128, 128), f3(28, 128, 128)),
        "wgs_512_512_7": (r"conv_wgrad_roll_kernel<7, 2>|conv_wgrad_strip_kernel<7, 7, 4,", 2 * act(7, 512) + dw(512, 512), f3(7, 512, 512)),
        "strip_512_512_7_fwd": (r"conv3x3_strip_kernel<512, (128|256), 7,", 2 * act(7, 512) + w3(512, 512), f3(7, 512, 512)),
        # stride-2 weight gradients: the high-res input + the low-res gradient once + dW
        "wgs_s2_64_56": (r"conv_wgrad_s2roll_kernel<56, 2>|conv_wgrad_strip_kernel<56, 2, 1, 8, 2, true", act(112, 64) + act(56, 64) + dw(64, 64), f3(56, 64, 64)),
        "wgs_s2_128_28": (r"conv_wgrad_s2roll_kernel<28, 2>|conv_wgrad_strip_kernel<28, 4, 1, 8, 2, true", act(56, 128) + act(28, 128) + dw(128, 128), f3(28, 128, 128)),
        "wgs_s2_256_14": (r"conv_wgrad_s2roll_kernel<14, 2>|conv_wgrad_strip_kernel<14, 7, 1, 8, 2, true", act(28, 256) + act(14, 256) + dw(256, 256), f3(14, 256, 256)),
        "wgs_s2_512_7": (r"conv_wgrad_s2roll_kernel<7, 2>|conv_wgrad_strip_kernel<7, 7, 2, 8, 2, true", act(14, 512) + act(7, 512) + dw(512, 512), f3(7, 512, 512)),
    }


def load(folder):
    """kernel name -> counter -> [values per dispatch]"""
    out = {}
    for path in glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                out.setdefault(r["Kernel_Name"], {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return out


def med(v):
    v = sorted(v)
    return v[len(v) // 2]


def pick(data, rx, counter):
    vals = [x for name, c in data.items() if re.search(rx, name) for x in c.get(counter, [])]
    return med(vals) if vals else None


def main():
    src, dst = sys.argv[1], sys.argv[2]
    times = {}
    for line in open(os.path.join(src, "times.log")):
        if line.startswith("KBENCH_SUITE "):
            times = json.loads(line[len("KBENCH_SUITE "):])
    f, w, s = load(os.path.join(src, "f")), load(os.path.join(src, "w")), load(os.path.join(src, "s"))
    im = load(os.path.join(src, "i")) if os.path.isdir(os.path.join(src, "i")) else {}
    recs, lines = [], []
    for label, (rx, alg_bytes, flops) in table().items():
        fk, wk = pick(f, rx, "FETCH_SIZE"), pick(w, rx, "WRITE_SIZE")
        sq = {c: pick(s, rx, c) for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
                                         "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY",
                                         "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE")}
        t = times.get(label, {})
        ms = t.get("ms")
        rec = {"kernel": label, "name_regex": rx, "ms_event_timed": ms, "tflops": t.get("tflops"),
               "algorithmic_bytes": alg_bytes, "flops": flops,
               "hbm_floor_us": round(alg_bytes / HBM_ACHIEVABLE * 1e6, 1),
               "mfma_floor_us_at_2.5PF": round(flops / MFMA_PEAK * 1e6, 1)}
        if fk is not None and wk is not None:
            rec["FETCH_SIZE_KB"], rec["WRITE_SIZE_KB"] = fk, wk
            rec["hbm_bytes"] = int(fk * 1024 * 2 + wk * 1024)
            rec["traffic_over_algorithmic"] = round(rec["hbm_bytes"] / alg_bytes, 3)
            if ms:
                rec["hbm_GBps"] = round(rec["hbm_bytes"] / (ms * 1e-3) / 1e9, 1)
        if ms:
            rec["x_hbm_floor"] = round(ms * 1e3 / rec["hbm_floor_us"], 2)
            rec["frac_mfma_peak"] = round(flops / (ms * 1e-3) / MFMA_PEAK, 3)
        if sq["SQ_VALU_MFMA_BUSY_CYCLES"] is not None and sq["GRBM_GUI_ACTIVE"]:
            rec["mfma_util"] = round(sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * sq["GRBM_GUI_ACTIVE"] / 8.0), 3)
        if sq["SQ_WAVE_CYCLES"]:
            wc = sq["SQ_WAVE_CYCLES"]
            for k, c in (("wait_any", "SQ_WAIT_ANY"), ("wait_inst", "SQ_WAIT_INST_ANY"), ("wait_lds", "SQ_WAIT_INST_LDS"),
                         ("active", "SQ_ACTIVE_INST_ANY")):
                if sq[c] is not None:
                    rec[k] = round(sq[c] / wc, 3)
            rec["sq_raw"] = sq
        mix = {c: pick(im, rx, c) for c in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD",
                                            "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU")}
        if mix["SQ_INSTS_MFMA"]:
            # SQ_INSTS_VALU counts the MFMAs too: vector-ALU instructions proper = VALU - MFMA
            mf = mix["SQ_INSTS_MFMA"]
            rec["valu_per_mfma"] = round(((mix["SQ_INSTS_VALU"] or 0) - mf) / mf, 2)
            rec["lds_per_mfma"] = round((mix["SQ_INSTS_LDS"] or 0) / mf, 2)
            rec["vmem_per_mfma"] = round(((mix["SQ_INSTS_VMEM_RD"] or 0) + (mix["SQ_INSTS_VMEM_WR"] or 0)) / mf, 3)
            rec["inst_mix_raw"] = mix
        recs.append(rec)
        lines.append("%-28s %7s ms  x%-5s HBM floor  traffic/alg %-6s  %6s GB/s  mfma_util %-6s frac_peak %-6s wait_any %-6s wait_inst %-6s "
                     "wait_lds %-6s valu/mfma %-5s lds/mfma %-5s"
                     % (label, ms, rec.get("x_hbm_floor"), rec.get("traffic_over_algorithmic"), rec.get("hbm_GBps"),
                        rec.get("mfma_util"), rec.get("frac_mfma_peak"), rec.get("wait_any"), rec.get("wait_inst"),
                        rec.get("wait_lds"), rec.get("valu_per_mfma"), rec.get("lds_per_mfma")))
    out = {"batch": B, "dtype": "bf16", "corrections": "FETCH_SIZE, WRITE_SIZE in KB; FETCH_SIZE doubled (gfx950 counts half of a "
           "16-B/lane streaming read); separate --pmc passes (MI355X_MICROARCH.md HBM / rocprofv3 sections)",
           "hbm_achievable_Bps": HBM_ACHIEVABLE, "mfma_peak": MFMA_PEAK, "kernels": recs}
    with open(dst, "w") as fo:
        json.dump(out, fo, indent=1)
    with open(os.path.splitext(dst)[0] + ".txt", "w") as fo:
        fo.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
