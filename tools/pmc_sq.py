"""Reduce a rocprofv3 --pmc pass of SQ counters (kernel-trace only, its own run) to per-kernel averages and issue-rate figures.

Usage: python tools/pmc_sq.py <pmc_counter_collection.csv> <out.json> [<workload> <extract.json>]
With the last two arguments the issue-rate extract bench.py reads (profiles/pmc_sq.json) is updated for that workload and stamped with
the digest of the kernel sources.
Pass used for profiles/r1_v7_pmc_sq_C2.json:
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY
            SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline
derived_valu_busy_frac = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE x 1024 SIMDs x ...): see DESIGN.md section 4; the counters are
summed over the device, GRBM_GUI_ACTIVE counts cycles of one clock domain per XCD (8 instances).
"""
import csv, json, os, sys
from collections import defaultdict

KERNELS = {"render_bwd": "render_bwd_kernel", "render_fwd": "render_fwd_kernel", "preprocess_fwd": "preprocess_fwd_kernel",
           "preprocess_bwd": "preprocess_bwd_kernel", "blend_order": "blend_order_kernel", "onesweep_kernel<16>": "radix_onesweep_kernel<16>",
           "onesweep_kernel<8>": "radix_onesweep_kernel<8>", "tile_ranges": "tile_ranges_kernel"}
N_SIMD, N_XCD = 1024, 8


def main():
    src, dst = sys.argv[1:3]
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(src)):
        for short, pat in KERNELS.items():
            if pat in row["Kernel_Name"]:
                acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for k, cs in sorted(acc.items()):
        o = {c: round(sum(v) / len(v)) for c, v in sorted(cs.items())}
        cyc = o.get("GRBM_GUI_ACTIVE", 0) / N_XCD          # cycles of the launch
        if cyc > 0 and "SQ_ACTIVE_INST_VALU" in o:
            o["derived_valu_busy_frac"] = round(o["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * N_SIMD), 3)
            o["derived_cycles_per_valu_inst_per_simd"] = round(cyc * N_SIMD / max(o.get("SQ_INSTS_VALU", 1), 1), 2)
            o["derived_mean_resident_waves_per_simd"] = round(o.get("SQ_WAVE_CYCLES", 0) * 4 / (cyc * N_SIMD), 2) if "SQ_WAVE_CYCLES" in o else None
        out[k] = o
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    if len(sys.argv) >= 5:
        workload, ext_path = sys.argv[3:5]
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        from bench import kernel_source_digest
        ext = json.load(open(ext_path)) if os.path.exists(ext_path) else {}
        digest = kernel_source_digest()
        if ext.get("kernel_source_digest") != digest:
            ext = {}                                   # figures of other sources do not mix with these
        ext[workload] = {k: {"cycles_per_valu_inst_per_simd": out[k]["derived_cycles_per_valu_inst_per_simd"],
                             "valu_busy_frac": out[k]["derived_valu_busy_frac"], "valu_insts_per_launch": out[k]["SQ_INSTS_VALU"]}
                         for k in ("render_bwd", "render_fwd") if k in out and "derived_cycles_per_valu_inst_per_simd" in out[k]}
        ext["kernel_source_digest"] = digest
        ext["measured_by"] = "rocprofv3 --kernel-trace --pmc SQ_* GRBM_GUI_ACTIVE -- python3 bench.py --steps 5 --warmup 2 (tools/profile_round.sh)"
        ext["_note"] = "a wave64 fp32 VALU instruction occupies its SIMD for 4 cycles, so 4.0 cycles per instruction per SIMD is the issue peak"
        json.dump(ext, open(ext_path, "w"), indent=1, sort_keys=True)
    for k in ("render_bwd", "render_fwd"):
        if k in out:
            print(k, {c: out[k][c] for c in out[k] if c.startswith("derived") or c in ("GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU")})


main()
