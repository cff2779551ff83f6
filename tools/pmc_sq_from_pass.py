"""profiles/pmc_sq.json entry of a workload from the per-kernel reduction of tools/pmc_pass.sh (tools/pmc_kernels.py's JSON):
   python tools/pmc_sq_from_pass.py <workload> <pmc_<workload>.json>
the VALU issue figures of the two blend kernels, in the form bench.py's roofline.valu_issue reads (same definitions as tools/pmc_sq.py)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
workload, path = sys.argv[1:3]
red = json.load(open(path))
out_path = os.path.join(ROOT, "profiles", "pmc_sq.json")
data = json.load(open(out_path)) if os.path.exists(out_path) else {}
sys.path.insert(0, ROOT)
from bench import kernel_source_digest
if data.get("kernel_source_digest") != kernel_source_digest():
    data = {"kernel_source_digest": kernel_source_digest(), "measured_by": "tools/pmc_pass.sh (rocprofv3 --kernel-trace --pmc SQ_* -- python3 bench.py)"}
entry = {}
for short, pat in (("render_bwd", "render_bwd_kernel"), ("render_fwd", "render_fwd_kernel")):
    for name, k in red.items():
        if pat in name and "SQ_INSTS_VALU" in k and "derived_launch_cycles" in k:
            entry[short] = {"cycles_per_valu_inst_per_simd": round(k["derived_launch_cycles"] * 1024.0 / k["SQ_INSTS_VALU"], 2),
                            "valu_busy_frac": k.get("derived_valu_busy_frac"), "valu_insts_per_launch": int(k["SQ_INSTS_VALU"])}
            break
if entry:
    data[workload] = entry
    json.dump(data, open(out_path, "w"), indent=1, sort_keys=True)
print(workload, json.dumps(entry))
