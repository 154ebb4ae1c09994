"""Where the waves of each kernel family spend their cycles, from two `rocprofv3 --pmc` passes over bench.py.

    python tools/pmc_stalls.py <pass_a_counter_collection.csv> <pass_b_counter_collection.csv> [out.json]

Pass A: SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS
        SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU
Pass B: SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL
        SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE
All SQ cycle counters are sums over waves (SQ_WAVE_CYCLES: cycles a wave is resident), so the shares below are
per resident wave-cycle: wait_any = waiting for anything, wait_inst = waiting for an instruction to issue,
wait_lds = of which for LDS; issue_* = cycles an instruction of that kind was issuing."""
import csv
import json
import sys
from collections import OrderedDict

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from pmc_traffic import family  # noqa: E402


def load(path, agg):
    for r in csv.DictReader(open(path)):
        f = family(r["Kernel_Name"])
        if f is None:
            continue
        a = agg.setdefault(f, {}).setdefault(r["Counter_Name"], [0.0, 0])
        a[0] += float(r["Counter_Value"])
        a[1] += 1


def main():
    agg = OrderedDict()
    load(sys.argv[1], agg)
    load(sys.argv[2], agg)
    out = OrderedDict()
    for f, d in agg.items():
        m = {k: v[0] / max(v[1], 1) for k, v in d.items()}
        wc = m.get("SQ_WAVE_CYCLES", 0.0)
        if wc <= 0:
            continue
        row = OrderedDict()
        for name, key in (("wait_any", "SQ_WAIT_ANY"), ("wait_inst", "SQ_WAIT_INST_ANY"), ("wait_lds", "SQ_WAIT_INST_LDS"),
                          ("issue_any", "SQ_ACTIVE_INST_ANY"), ("issue_lds", "SQ_ACTIVE_INST_LDS"),
                          ("issue_vmem", "SQ_ACTIVE_INST_VMEM"), ("issue_valu", "SQ_ACTIVE_INST_VALU")):
            row[name] = round(m.get(key, 0.0) / wc, 4)
        gui = m.get("GRBM_GUI_ACTIVE", 0.0)
        if gui > 0:
            simd_cycles = gui / 8.0 * 1024.0
            row["mfma_busy"] = round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd_cycles, 4)
            row["mfma_coexec"] = round(m.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0.0) / simd_cycles, 4)
            for name, key in (("lds_cmd_fifo_full", "SQ_LDS_CMD_FIFO_FULL"), ("lds_data_fifo_full", "SQ_LDS_DATA_FIFO_FULL"),
                              ("vmem_addr_fifo_full", "SQ_VMEM_TA_ADDR_FIFO_FULL")):
                row[name] = round(m.get(key, 0.0) / (gui / 8.0 * 256.0), 4)      # per CU-cycle
        out[f] = row
        print("%-28s " % f + "  ".join("%s %.3f" % (k, v) for k, v in row.items()))
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
