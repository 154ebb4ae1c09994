"""MFMA utilisation per kernel family from one `rocprofv3 --pmc` pass over bench.py.

    python tools/pmc_mfma.py <counter_collection.csv> [out.json]

Counters (one pass: 8 SQ + 1 GRBM): SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES
SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE.
SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe cycles summed over the SIMDs (32 per v_mfma_f32_32x32x16_bf16);
GRBM_GUI_ACTIVE is the sum over the 8 XCDs of the cycles the dispatch was active, so
    mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)
is the share of SIMD-cycles during the dispatch in which the matrix pipe was busy, at whatever clock the
chip held; lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE."""
import csv
import json
import sys
from collections import OrderedDict

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from pmc_traffic import family  # noqa: E402


def main():
    agg = OrderedDict()
    for r in csv.DictReader(open(sys.argv[1])):
        f = family(r["Kernel_Name"])
        if f is None:
            continue
        d = agg.setdefault(f, {})
        a = d.setdefault(r["Counter_Name"], [0.0, 0])
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    out = OrderedDict()
    for f, d in agg.items():
        m = {k: v[0] / max(v[1], 1) for k, v in d.items()}
        gui = m.get("GRBM_GUI_ACTIVE", 0.0)
        row = {"launches_seen": max(v[1] for v in d.values())}
        if gui > 0:
            row["mfma_util"] = round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8.0 * 1024.0), 4)
            row["active_cycles_per_xcd"] = round(gui / 8.0)
        if m.get("SQ_LDS_IDX_ACTIVE", 0.0) > 0:
            row["lds_conflict"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"], 4)
        for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY",
                  "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_VALU"):
            if k in m:
                row[k] = round(m[k])
        out[f] = row
        print("%-28s mfma_util %6.3f  lds_conflict %6.3f  (%d launches)" % (
            f, row.get("mfma_util", float("nan")), row.get("lds_conflict", float("nan")), row["launches_seen"]))
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
