"""Developer timing on the GPU box, run under `rocprofv3 --kernel-trace`: the batched prefilter products of the reference's default chain
(128 / 0.08, 64 / 0.29, 32 / 0.5, 16 / 1.0) for several subsets of the levels, REPS launches each, forward then transposed.  Prints the
order of the subsets; tools/spmv_time_reduce.py groups the trace's csr_spmv3_batched_kernel dispatches by it.
    rocprofv3 --kernel-trace -f csv -d OUT -o spmv -- python3 tools/spmv_time.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from materialrefgs_amd import shading as sh       # noqa: E402

REPS = 40
SUBSETS = [(0, 1, 2, 3), (0,), (1,), (2,), (3,), (1, 2, 3)]


def main():
    dev = torch.device("cuda:0")
    ops = [sh.CubemapFilterOp.get(dev, 128, 0, 0.08, 0.99), sh.CubemapFilterOp.get(dev, 64, 0, 0.29, 0.99),
           sh.CubemapFilterOp.get(dev, 32, 0, 0.5, 0.99), sh.CubemapFilterOp.get(dev, 16, 0, 1.0, 0.99)]
    xs = [torch.randn(6, o.res, o.res, 3, device=dev) for o in ops]
    torch.cuda.synchronize()
    # marker: one launch of the mip kernel separates the build's launches from the timed ones
    sh._mip_forward(xs[0])
    plan = []
    for tr in (False, True):
        for sub in SUBSETS:
            vec = [xs[i] if i in sub else None for i in range(4)]
            for _ in range(REPS):
                sh._spmv_batched(ops, vec, transpose=tr)
            torch.cuda.synchronize()
            plan.append({"transpose": tr, "levels": list(sub), "reps": REPS})
    print(json.dumps({"plan": plan, "symmetric": [o.sym is not None for o in ops],
                      "panels": [(o.sym.n_tiles, o.sym.max_panel, o.t_sym.max_panel) if o.sym is not None else None for o in ops]}))


if __name__ == "__main__":
    main()
