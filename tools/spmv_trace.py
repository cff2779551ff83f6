"""Developer, GPU box, with MRGS_LIB=build/spmv_trace/libmrgs.so (a -DMRGS_SPMV_TRACE build): per-wave timestamps (100 MHz wall clock) of one
batched prefilter product -- start of the wave, first gathers back, end of its loop, end -- summarised per level."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from materialrefgs_amd import shading as sh       # noqa: E402


def main():
    dev = torch.device("cuda:0")
    ops = [sh.CubemapFilterOp.get(dev, 128, 0, 0.08, 0.99), sh.CubemapFilterOp.get(dev, 64, 0, 0.29, 0.99),
           sh.CubemapFilterOp.get(dev, 32, 0, 0.5, 0.99), sh.CubemapFilterOp.get(dev, 16, 0, 1.0, 0.99)]
    xs = [torch.randn(6, o.res, o.res, 3, device=dev) for o in ops]
    buf = torch.zeros((4096 * 8 * 4,), dtype=torch.int64, device=dev)
    os.environ["MRGS_SPMV_TRACE_BUF"] = hex(buf.data_ptr())
    for sub in ((0, 1, 2, 3), (1,), (2,), (3,)):
        vec = [xs[i] if i in sub else None for i in range(4)]
        for _ in range(5):
            buf.zero_()
            sh._spmv_batched(ops, vec, transpose=False)
            torch.cuda.synchronize()
        t = buf.cpu().numpy().reshape(-1, 8, 4)
        live = t[:, 0, 3] != 0
        idx = np.nonzero(live)[0]
        tt = t[live].astype(np.float64)
        t0 = tt[:, :, 0].min()
        us = (tt - t0) / 100.0                       # 100 MHz
        order = sorted((i for i in sub if ops[i].sym is not None), key=lambda i: -ops[i].nnz)
        first = 0
        out = {"levels": list(sub), "all_end_us": round(float(us[:, :, 3].max()), 2)}
        # the blocks of the symmetric levels come in the batch order (heaviest first); CSR levels write no stamps
        pos = 0
        blocks_before = 0
        all_order = sorted(sub, key=lambda i: (0, -ops[i].sym.max_panel) if ops[i].sym is not None else (1, -ops[i].nnz))
        for i in all_order:
            op = ops[i]
            nb = op.sym.n_tiles * 12 if op.sym is not None else (op.nrows * (64 if op.lanes == 64 else 4) + 511) // 512
            if op.sym is not None:
                sel = (idx >= blocks_before) & (idx < blocks_before + nb)
                u = us[sel]
                out[f"res{op.res}"] = {"workgroups": int(sel.sum()), "start_us": [round(float(u[:, :, 0].min()), 2), round(float(u[:, :, 0].max()), 2)],
                                        "first_gather_after_us": round(float((u[:, :, 1] - u[:, :, 0]).mean()), 2),
                                        "loop_us": round(float((u[:, :, 2] - u[:, :, 1]).mean()), 2), "loop_max_us": round(float((u[:, :, 2] - u[:, :, 1]).max()), 2),
                                        "tail_us": round(float((u[:, :, 3] - u[:, :, 2]).mean()), 2), "end_us": round(float(u[:, :, 3].max()), 2),
                                        "wave_life_mean_us": round(float((u[:, :, 3] - u[:, :, 0]).mean()), 2)}
            blocks_before += nb
        print(json.dumps(out))


if __name__ == "__main__":
    main()
