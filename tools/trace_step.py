"""Per-dispatch view of one train step from a `rocprofv3 --kernel-trace` CSV.

    python tools/trace_step.py gpurun_out/trace/tr_kernel_trace.csv [--sum]

Prints every kernel dispatch of the last complete step (the dispatches between the last two
AdamW launches) in order with its duration, or with --sum the per-kernel totals."""
import csv
import re
import sys
from collections import OrderedDict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+)(<.*>)?\(", name)
    if m:
        return m.group(1) + (m.group(2) or "")
    return name[:80]


def main():
    path = sys.argv[1]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
    if len(ends) < 2:
        sys.exit("fewer than two steps in the trace")
    # the shortest complete step: graph replays of the timed region, not the eager roofline pass
    spans = [(int(rows[b]["End_Timestamp"]) - int(rows[a + 1]["Start_Timestamp"]), a, b)
             for a, b in zip(ends[:-1], ends[1:]) if b - a >= 50]
    _, a, b = min(spans)
    step = rows[a + 1: b + 1]
    t0 = int(step[0]["Start_Timestamp"])
    busy = 0
    if "--sum" in sys.argv:
        tot = OrderedDict()
        for r in step:
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            k = short(r["Kernel_Name"])
            a = tot.setdefault(k, [0, 0])
            a[0] += d
            a[1] += 1
            busy += d
        for k, (d, n) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
            print("%9.1f us %4d x  %s" % (d / 1e3, n, k))
    else:
        for r in step:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            busy += e - s
            grid = "%sx%sx%s" % (r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?"))
            print("%9.1f  %8.1f us  %-16s %s" % ((s - t0) / 1e3, (e - s) / 1e3, grid, short(r["Kernel_Name"])))
    span = int(step[-1]["End_Timestamp"]) - t0
    print("step: %d dispatches, busy %.3f ms, span %.3f ms" % (len(step), busy / 1e6, span / 1e6))


if __name__ == "__main__":
    main()
