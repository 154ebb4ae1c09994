"""Idle time of one graph-replayed train step from a `rocprofv3 --kernel-trace` CSV: the union of the
kernels' execution intervals against the step's span, the largest gaps with the kernels on either side,
and how much of the span has one, two or more kernels in flight.

    python tools/trace_gaps.py gpurun_out/trace_ov/.../tr_kernel_trace.csv"""
import csv
import sys

from trace_step import short


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
    spans = [(int(rows[b]["End_Timestamp"]) - int(rows[a + 1]["Start_Timestamp"]), a, b)
             for a, b in zip(ends[:-1], ends[1:]) if b - a >= 50]
    _, a, b = min(spans)
    step = rows[a + 1: b + 1]
    iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in step]
    t0, t1 = iv[0][0], max(e for _, e, _ in iv)
    # sweep
    ev = sorted([(s, 1) for s, _, _ in iv] + [(e, -1) for _, e, _ in iv])
    depth, last, at_depth = 0, t0, {}
    for t, d in ev:
        at_depth[depth] = at_depth.get(depth, 0) + (t - last)
        depth += d
        last = t
    print("span %.3f ms; in flight: " % ((t1 - t0) / 1e6) +
          ", ".join("%d kernels %.3f ms" % (k, v / 1e6) for k, v in sorted(at_depth.items())))
    # gaps: intervals with nothing in flight
    gaps, cur_end, prev = [], iv[0][1], iv[0][2]
    for s, e, n in iv[1:]:
        if s > cur_end:
            gaps.append((s - cur_end, prev, n, (cur_end - t0) / 1e3))
        if e > cur_end:
            cur_end, prev = e, n
    gaps.sort(reverse=True)
    print("%d gaps, %.3f ms in total; the largest:" % (len(gaps), sum(g[0] for g in gaps) / 1e6))
    for g, p, n, at in gaps[:20]:
        print("  %6.1f us at %8.1f us  after %-40s before %s" % (g / 1e3, at, p[:40], n[:40]))


if __name__ == "__main__":
    main()
