"""Timeline analysis of a rocprofv3 --kernel-trace run (rocpd sqlite output): per-kernel totals, the time the GPU
is busy with >= 1 kernel, the idle gaps between kernels and the amount of two-stream overlap, for the LAST
training step in the trace (steps are delimited by the fr_sgd-style optimizer kernel).

    python tools/trace_gaps.py gpurun_out/prof/r_results.db [--csv out.csv]
"""
import collections
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(Fr\w+Args\)$", "", name)
    return name[:90]


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, start, end, queue_id, grid_x, workgroup_x from kernels order by start").fetchall()
    # step boundaries: the multi-tensor SGD kernel ends a step
    ends = [i for i, r in enumerate(rows) if "sgd" in r[0]]
    per_step = collections.Counter()
    # two sgd launches per step (backbone, head): find the last complete step = between the -3rd and -1st sgd
    lo, hi = ends[-3] + 1, ends[-1] + 1
    step = rows[lo:hi]
    t0, t1 = step[0][1], max(r[2] for r in step)
    print("step: %d kernels, wall %.3f ms" % (len(step), (t1 - t0) / 1e6))
    # union of busy intervals and overlap
    ev = []
    for r in step:
        ev.append((r[1], 1))
        ev.append((r[2], -1))
    ev.sort()
    busy = over = 0
    depth, last = 0, t0
    for t, d in ev:
        if depth >= 1:
            busy += t - last
        if depth >= 2:
            over += t - last
        depth += d
        last = t
    ksum = sum(r[2] - r[1] for r in step)
    print("sum of kernel durations %.3f ms, busy (>=1 kernel) %.3f ms, idle %.3f ms, >=2 kernels %.3f ms"
          % (ksum / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, over / 1e6))
    # idle gaps: where no kernel is running
    srt = sorted(step, key=lambda r: r[1])
    gaps, cur_end, cur_name = [], srt[0][2], srt[0][0]
    for r in srt[1:]:
        if r[1] > cur_end:
            gaps.append((r[1] - cur_end, short(cur_name), short(r[0])))
        if r[2] > cur_end:
            cur_end, cur_name = r[2], r[0]
    gaps.sort(reverse=True)
    print("gaps: %d, total %.3f ms, median %.2f us" % (len(gaps), sum(g[0] for g in gaps) / 1e6,
                                                        gaps[len(gaps) // 2][0] / 1e3 if gaps else 0))
    for g in gaps[:12]:
        print("   %7.1f us  after %-50s before %-50s" % (g[0] / 1e3, g[1][:50], g[2][:50]))
    agg = collections.defaultdict(lambda: [0, 0])
    for r in step:
        a = agg[short(r[0])]
        a[0] += 1
        a[1] += r[2] - r[1]
    out = sorted(agg.items(), key=lambda kv: -kv[1][1])
    for k, (n, ns) in out[:70]:
        print("%-92s n=%3d %8.1f us  avg %7.1f" % (k, n, ns / 1e3, ns / 1e3 / n))
    if "--timeline" in sys.argv:
        # every kernel of the step in start order: offset from the step start, duration, queue -- to read the two-stream
        # schedule (who runs beside whom, who waits for whom)
        path = sys.argv[sys.argv.index("--timeline") + 1]
        qids = sorted({r[3] for r in step})
        with open(path, "w") as f:
            f.write("# start_us dur_us end_us queue grid name\n")
            for r in sorted(step, key=lambda r: r[1]):
                f.write("%9.1f %7.1f %9.1f q%d %5d %s\n" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, (r[2] - t0) / 1e3,
                                                           qids.index(r[3]), r[4] // max(r[5], 1), short(r[0])[:70]))
    if "--csv" in sys.argv:
        path = sys.argv[sys.argv.index("--csv") + 1]
        with open(path, "w") as f:
            f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage"\n')
            for k, (n, ns) in out:
                f.write('"%s",%d,%d,%.1f,%.2f\n' % (k, n, ns, ns / n, 100.0 * ns / ksum))
            f.write('"__step_wall_ns",1,%d,%d,0\n"__busy_ns",1,%d,%d,0\n"__overlap_ns",1,%d,%d,0\n'
                    % (t1 - t0, t1 - t0, busy, busy, over, over))


if __name__ == "__main__":
    main()
