"""Reduce two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only) to HBM bytes per launch.

Usage: python tools/pmc_traffic.py <workload> <fetch_counter_collection.csv> <write_counter_collection.csv>
Updates profiles/pmc_traffic.json, which bench.py reads for roofline.traffic.

Units / corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE
tallies 128-byte requests at 64 bytes, so it is doubled.  WRITE_SIZE is taken as reported (uncalibrated).
"""
import csv, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {"render_bwd": "render_bwd_kernel", "render_fwd": "render_fwd_kernel", "preprocess_fwd": "preprocess_fwd_kernel",
           "preprocess_bwd": "preprocess_bwd_kernel"}


def per_launch(path, counter):
    acc = defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        for short, pat in KERNELS.items():
            if pat in row["Kernel_Name"]:
                acc[short].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    workload, fpath, wpath = sys.argv[1:4]
    fetch, write = per_launch(fpath, "FETCH_SIZE"), per_launch(wpath, "WRITE_SIZE")
    out_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    sys.path.insert(0, ROOT)
    from bench import kernel_source_digest
    data = json.load(open(out_path)) if os.path.exists(out_path) else {}
    if data.get("kernel_source_digest") != kernel_source_digest():
        data = {}             # figures of other kernel sources do not ride along under this tree's digest
    entry, detail = {}, {}
    for k in KERNELS:
        if k in fetch and k in write:
            rd, wr = 2.0 * fetch[k] * 1024.0, write[k] * 1024.0
            entry[k] = int(rd + wr)
            detail[k] = {"fetch_bytes_corrected_x2": int(rd), "write_bytes": int(wr)}
    data[workload] = entry
    data["kernel_source_digest"] = kernel_source_digest()     # bench.py reports the traffic only for these very sources
    data["measured_by"] = "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python3 bench.py --steps 5 --warmup 2 (tools/profile_round.sh)"
    data.setdefault("_detail", {})[workload] = detail
    data["_note"] = "bytes per launch; FETCH_SIZE KiB x2 (gfx950 128-B requests tallied at 64 B) + WRITE_SIZE KiB"
    json.dump(data, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps(data[workload]), json.dumps(detail))


if __name__ == "__main__":
    main()
