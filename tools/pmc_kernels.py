"""Reduce any rocprofv3 --pmc pass (kernel-trace only, its own run) to per-kernel averages of every counter collected.

Usage: python tools/pmc_kernels.py <out.json> <counter_collection.csv> [<counter_collection.csv> ...] [--min-calls N]
Kernel names are cut at the first '(' (template arguments kept); counters are averaged over the launches of a kernel, several passes
(one csv each: rocprofv3 takes a limited set per run) are merged by kernel name.  GRBM_GUI_ACTIVE counts per XCD (8 instances summed).
"""
import csv, json, re, sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    cut = name.find("(")
    return (name if cut < 0 else name[:cut])[:96]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    min_calls = int(sys.argv[sys.argv.index("--min-calls") + 1]) if "--min-calls" in sys.argv else 1
    if "--min-calls" in sys.argv:
        args.remove(sys.argv[sys.argv.index("--min-calls") + 1])
    dst, srcs = args[0], args[1:]
    acc = defaultdict(lambda: defaultdict(list))
    for src in srcs:
        for row in csv.DictReader(open(src)):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for k, cs in sorted(acc.items()):
        n = max(len(v) for v in cs.values())
        if n < min_calls:
            continue
        o = {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())}
        o["launches_seen"] = n
        cyc = o.get("GRBM_GUI_ACTIVE", 0) / 8
        if cyc > 0:
            o["derived_launch_cycles"] = round(cyc)
            if "SQ_ACTIVE_INST_VALU" in o:
                o["derived_valu_busy_frac"] = round(o["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * 1024), 3)
            if "SQ_LDS_IDX_ACTIVE" in o:
                o["derived_lds_busy_frac"] = round(o["SQ_LDS_IDX_ACTIVE"] / (cyc * 256), 3)      # one LDS per CU
        # HBM bytes per launch as MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE doubled on gfx950 (128-byte
        # requests tallied at 64 bytes); the two counters come from separate passes
        if "FETCH_SIZE" in o:
            o["derived_hbm_read_bytes"] = int(2.0 * o["FETCH_SIZE"] * 1024.0)
        if "WRITE_SIZE" in o:
            o["derived_hbm_write_bytes"] = int(o["WRITE_SIZE"] * 1024.0)
        out[k] = o
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)


main()
