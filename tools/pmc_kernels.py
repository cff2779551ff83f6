"""Reduce any rocprofv3 --pmc pass (kernel-trace only, its own run) to per-kernel averages of every counter collected.

Usage: python tools/pmc_kernels.py <out.json> <counter_collection.csv> [<counter_collection.csv> ...] [--min-calls N]
Kernel names are cut at the first '(' (template arguments kept); counters are averaged over the launches of a kernel, several passes
(one csv each: rocprofv3 takes a limited set per run) are merged by kernel name.  GRBM_GUI_ACTIVE counts per XCD (8 instances summed).

Cycle base of the derived fractions (round 5).  derived_launch_cycles = GRBM_GUI_ACTIVE / 8, and the kernel trace that rocprofv3 writes next
to every counter file gives the same launches' wall durations: derived_effective_ghz = cycles / duration must come out as a clock this chip
can run at (<= 2.4 GHz; lower under load, MI355X_MICROARCH.md "DVFS give-back") -- it is printed so that a wrong base shows.  At C4 size
round 4 reported a "VALU busy fraction" of 1.19 for the backward blend with a plausible 2.25 GHz: the base was right, the NUMERATOR is not a
fraction of an exclusive resource -- SQ_ACTIVE_INST_VALU x 4 counts the quad-cycles of every VALU instruction, and a kernel that keeps every
SIMD supplied (41 000 waves at C4 against 11 000 at C3: no tail) retires more than one fp32 instruction per 4 cycles per SIMD whenever
instructions of different waves overlap in the pipe (transcendentals and DPP moves run beside the main ALU).  So: derived_valu_issue_rate is
the raw figure (may exceed 1), derived_valu_busy_frac is that figure capped at 1.0, and a value the cap touched is flagged
(derived_valu_busy_saturated) instead of being read as "119 % busy".
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
    # kernel -> (launch duration in ns, GRBM_GUI_ACTIVE of THE SAME launch): a counter csv is paired with the kernel trace rocprofv3 wrote
    # for the same process (same file prefix: <pid>_counter_collection.csv / <pid>_kernel_trace.csv), and its rows with the trace's by
    # Dispatch_Id -- never with another pass's trace that happens to lie in the directory (another clock state), never twice
    paired = defaultdict(list)
    import os
    seen_traces = set()
    for src in srcs:
        rows = list(csv.DictReader(open(src)))
        for row in rows:
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        tr = src.replace("counter_collection.csv", "kernel_trace.csv")
        if tr == src or not os.path.exists(tr) or os.path.realpath(tr) in seen_traces:
            continue
        seen_traces.add(os.path.realpath(tr))
        dur = {}
        for row in csv.DictReader(open(tr)):
            try:
                dur[row["Dispatch_Id"]] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
            except (KeyError, ValueError):
                pass
        for row in rows:
            if row["Counter_Name"] == "GRBM_GUI_ACTIVE" and row.get("Dispatch_Id") in dur:
                paired[short(row["Kernel_Name"])].append((dur[row["Dispatch_Id"]], float(row["Counter_Value"]) / 8))
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
            if paired.get(k):
                w_ns = sum(d for d, _ in paired[k]) / len(paired[k])
                o["derived_wall_us"] = round(w_ns / 1e3, 2)
                # the sanity figure, launch by launch matched: only where the launch is long enough for its ~3 us of dispatch overhead (counted
                # by GRBM, not by the trace's timestamps) not to be the figure -- a 4 us kernel "runs at 12 GHz" otherwise
                if w_ns >= 20e3:
                    o["derived_effective_ghz"] = round(sum(c for _, c in paired[k]) / sum(d for d, _ in paired[k]), 3)
            if "SQ_ACTIVE_INST_VALU" in o:
                raw = o["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * 1024)
                o["derived_valu_issue_rate"] = round(raw, 3)
                o["derived_valu_busy_frac"] = round(min(raw, 1.0), 3)
                if raw > 1.0:
                    o["derived_valu_busy_saturated"] = True
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
