"""tools/spmv_time_reduce.py <kernel_trace.csv> <plan.json>: average duration of the product launches per subset of tools/spmv_time.py."""
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
plan = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
names = [r["Kernel_Name"] for r in rows]
start = max(i for i, n in enumerate(names) if n.startswith("cubemap_mip_fwd_kernel"))     # the marker
prod = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[start:] if r["Kernel_Name"].startswith("csr_spmv3_batched_kernel")]
ip = 0
for p in plan["plan"]:
    k = p["reps"]
    a = prod[ip:ip + k]; ip += k
    line = {"transpose": p["transpose"], "levels": p["levels"], "product_us": round(sum(a[5:]) / max(1, len(a[5:])) / 1000.0, 2)}
    print(json.dumps(line))
