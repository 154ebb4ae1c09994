"""Average counter values per kernel from rocprofv3 --pmc CSVs (conv kernels only)."""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if not any(s in k for s in ("patch_", "igemm_")):
            continue
        key = "%s grid=%s wg=%s vgpr=%s/%s lds=%s" % (k[:80], r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"],
                                                     r["Accum_VGPR_Count"], r["LDS_Block_Size"])
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-36s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
