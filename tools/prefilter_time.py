"""Developer timing on the GPU box: EnvLight.build_mips forward + backward alone (128 -> 16 chain, the reference's defaults), HIP events
around `reps` iterations, with the symmetric rows (default) and with the full matrices (MRGS_NO_SYMMETRIC_SPMV=1 in the environment).
    python tools/prefilter_time.py [reps]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from materialrefgs_amd import shading as sh       # noqa: E402
from bench import kernel_source_digest   # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    dev = torch.device("cuda:0")
    env = sh.EnvLight(device=dev, min_res=16, max_res=128, trainable=True)
    with torch.no_grad():
        env.base.copy_(torch.rand_like(env.base))
    ups = None
    out = {}
    for phase in ("fwd", "fwd+bwd"):
        for it in range(reps + 20):
            if it == 20:
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            with torch.no_grad():
                env.base.add_(0.0)             # (the parameter was written: the prefilter is due)
            env.build_mips()
            if phase == "fwd+bwd":
                if ups is None:
                    ups = [torch.randn_like(m) for m in env.specular]
                env.base.grad = None
                torch.autograd.backward(env.specular, ups)
        e1.record()
        torch.cuda.synchronize()
        out[phase + "_us"] = round(1000.0 * e0.elapsed_time(e1) / reps, 2)
    ops = [o for o in sh.CubemapFilterOp._cache.values()]
    out["operators"] = [{"res": o.res, "kind": o.kind, "nnz": o.nnz, "lanes": o.lanes, "symmetric": o.sym is not None,
                         "sym_error": o.sym_error, "blocks_full": int(o.row_ptr[-1]) if o.val.dim() == 2 else None,
                         "patches_sym": o.sym.patches if o.sym is not None else None, "tiles": o.sym.n_tiles if o.sym is not None else None,
                         "max_panel_blocks": (o.sym.max_panel, o.t_sym.max_panel) if o.sym is not None else None} for o in ops]
    out["kernel_source_digest"] = kernel_source_digest()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
