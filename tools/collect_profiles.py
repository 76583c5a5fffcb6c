#!/usr/bin/env python3
"""Copy the artefacts of tools/profile_round.sh and tools/pmc_round.sh from gpurun_out/ (scratch) into profiles/
(tracked) under per-round names, and derive profiles/<tag>_mfma_util.json from the SQ pass.

    python tools/collect_profiles.py r02
"""
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    prof, pmc, dst = (os.path.join(REPO, "gpurun_out", "prof_" + tag), os.path.join(REPO, "gpurun_out", "pmc_" + tag),
                      os.path.join(REPO, "profiles"))
    pairs = [(os.path.join(prof, "bench.json"), tag + "_bench.json"),
             (os.path.join(prof, "event_table.json"), tag + "_bench_bs256_bf16_event_table.json"),
             (os.path.join(prof, "kernel_stats_single.csv"), tag + "_bench_bs256_bf16_kernel_stats_single_stream.csv"),
             (os.path.join(prof, "kernel_stats_two.csv"), tag + "_bench_bs256_bf16_kernel_stats_two_streams.csv"),
             (os.path.join(prof, "gaps_single.txt"), tag + "_trace_gaps_single_stream.txt"),
             (os.path.join(prof, "gaps_two.txt"), tag + "_trace_gaps_two_streams.txt"),
             (os.path.join(pmc, "summary.json"), tag + "_pmc_kernels.json"),
             (os.path.join(pmc, "summary.txt"), tag + "_pmc_kernels.txt"),
             (os.path.join(pmc, "f", "f_counter_collection.csv"), tag + "_pmc_fetch_size_counters.csv"),
             (os.path.join(pmc, "w", "w_counter_collection.csv"), tag + "_pmc_write_size_counters.csv"),
             (os.path.join(pmc, "s", "s_counter_collection.csv"), tag + "_pmc_sq_counters.csv"),
             (os.path.join(pmc, "i", "i_counter_collection.csv"), tag + "_pmc_inst_mix_counters.csv"),
             (os.path.join(prof, "clocks.txt"), tag + "_clocks.txt"),
             (os.path.join(prof, "bench_200.json"), tag + "_bench_200_steps.json"),
             (os.path.join(prof, "timeline_two.txt"), tag + "_timeline_two_streams.txt"),
             (os.path.join(prof, "timeline_single.txt"), tag + "_timeline_single_stream.txt")]
    for src, name in pairs:
        if os.path.exists(src):
            shutil.copyfile(src, os.path.join(dst, name))
            print("copied", name)
        else:
            print("MISSING", src)
    summ = os.path.join(pmc, "summary.json")
    if os.path.exists(summ):
        d = json.load(open(summ))
        out = {"source": tag + "_pmc_sq_counters.csv (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES ... GRBM_GUI_ACTIVE over "
                         "tools/kbench.py suite, B = 256, bf16)",
               "formula": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); SQ_VALU_MFMA_BUSY_"
                          "CYCLES counts 16 cycles per v_mfma_f32_16x16x32_bf16 summed over all SIMDs; GRBM_GUI_ACTIVE is summed "
                          "over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)",
               "kernels": [{k: r.get(k) for k in ("kernel", "ms_event_timed", "tflops", "frac_mfma_peak", "mfma_util",
                                                  "wait_any", "wait_inst", "wait_lds", "active", "valu_per_mfma", "lds_per_mfma",
                                                  "vmem_per_mfma", "hbm_GBps",
                                                  "traffic_over_algorithmic", "x_hbm_floor")} for r in d["kernels"]]}
        with open(os.path.join(dst, tag + "_mfma_util.json"), "w") as f:
            json.dump(out, f, indent=1)
        print("wrote", tag + "_mfma_util.json")


if __name__ == "__main__":
    main()
