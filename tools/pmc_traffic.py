"""HBM-side traffic per kernel launch from two `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE).

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> [out.json]

FETCH_SIZE / WRITE_SIZE are kilobytes at the L2's memory side.  On gfx950 FETCH_SIZE tallies
128-byte read requests as 64 bytes, so the read side is doubled.  Bytes per launch are averaged
over all dispatches of a kernel family; the JSON (family -> bytes per launch) is what bench.py
reports as roofline.traffic."""
import csv
import json
import re
import sys
from collections import OrderedDict


def family(name):
    """Kernel name (mangled or demangled) -> the label bench.py uses, or a short name."""
    m = re.search(r"patch_fwd.*?(\d), ?(\d), ?(\d), ?(\d), ?(\d)>|patch_fwdI\w+?Li(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)E", name)
    if "patch_fwd" in name:
        nums = [int(v) for v in re.findall(r"Li(\d+)E", name)] or [int(v) for v in re.findall(r"[ ,<](\d+)(?=[,>])", name)]
        wm, wn = (nums + [0, 0])[:2] if len(nums) >= 5 else (None, None)
        if len(nums) == 4:          # partially demangled: <..., E, WN, MI, NI, KW>
            wn = nums[0]
            wm = 4 // wn if wn else None
        if wn == 4:
            return "patch_fwd<256px x 128ch>" if wm == 2 else "patch_fwd<128px x 128ch>"
        if wn == 2:
            return "patch_fwd<256px x 64ch>"
        if wn == 1:
            return "patch_fwd<256px x 32ch>"
    for key in ("lowk_fwd", "patch_dgrad_s2", "patch_wgrad", "igemm_fwd_group", "igemm_fwd", "igemm_wgrad", "wgrad_reduce", "bn_act_bwd_dx",
                "bn_act_fwd", "colreduce", "upsample_taps_bwd", "upsample_taps_fwd", "upsample_pad_fwd", "upsample_pad_bwd",
                "adamw_kernel_dev", "weight_pack_batch", "reflect_fold", "per_image_colsum"):
        if key in name:
            return key
    return None


def read(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        f = family(r["Kernel_Name"])
        if f is None:
            continue
        a = out.setdefault(f, [0.0, 0])
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    return out


def main():
    fetch = read(sys.argv[1], "FETCH_SIZE")
    write = read(sys.argv[2], "WRITE_SIZE")
    res = OrderedDict()
    for f in sorted(set(fetch) | set(write)):
        fk, fn = fetch.get(f, [0.0, 1])
        wk, wn = write.get(f, [0.0, 1])
        rd = 2.0 * fk / max(fn, 1) * 1024.0
        wr = wk / max(wn, 1) * 1024.0
        res[f] = {"read_bytes": rd, "write_bytes": wr, "bytes": rd + wr, "launches_seen": max(fn, wn)}
        print("%-28s read %9.2f MB  write %9.2f MB  per launch (%d launches)" % (f, rd / 1e6, wr / 1e6, max(fn, wn)))
    if len(sys.argv) > 3:
        json.dump(res, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
