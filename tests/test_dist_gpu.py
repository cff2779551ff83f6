"""View-parallel step on real renders (-m gpu): `bench.py --gpus 2` over gloo on the ONE GPU of the test box.

Two ranks (fresh child processes started by bench.py itself, as the driver's `--gpus N` run does) render views 0 and 1 of the same scene
and push their gradients through the reducers of materialrefgs_amd/dist.py -- FactoredGradReducer for the raster workload, SurfelGradReducer
for render_surfel's parameter set -- i.e. the all-gather of the factored SH gradients, the local expansion in libmrgs.so and the flat
all-reduce of the rest.  What comes out must be the SUM of two single-rank renders of those views ("sum over ranks
equals sum of single-GPU results").  RCCL itself needs more than one GPU and is not exercised here; the collectives run through gloo on
device tensors, everything around them is the production path."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(tmp_path, name, *args):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MRGS_DIST_BACKEND"] = "gloo"
    env["MRGS_BENCH_VIEWS"] = "8"          # all eight orbit cameras whatever the warm-up: --dump-step k then names camera k in every run
    dump = str(tmp_path / f"{name}.npz")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args, "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-secondary",
                        "--dump-grads", dump], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0]), dict(np.load(dump))


@pytest.mark.parametrize("workload", ["tiny", "tinyfull"])
def test_two_rank_step_equals_the_sum_of_two_single_rank_renders(gpu_device, tmp_path, workload):
    line2, g2 = _bench(tmp_path, "two", "--gpus", "2", "--workload", workload, "--dump-step", "0")        # rank r renders view r
    assert line2["n_gpus"] == 2 and line2["config"]["parallelism"] == "view-parallel x2" and line2["config"]["views_per_step"] == 2
    assert line2["value"] > 0 and line2["scaling"] == "weak"
    _, v0 = _bench(tmp_path, "v0", "--gpus", "1", "--workload", workload, "--dump-step", "0")             # view 0 alone
    line1, v1 = _bench(tmp_path, "v1", "--gpus", "1", "--workload", workload, "--dump-step", "1")         # view 1 alone
    assert line1["n_gpus"] == 1 and line1["config"]["parallelism"] == "single GPU"
    assert set(g2) == set(v0) == set(v1) and len(g2) >= 7
    for k in sorted(g2):
        want = v0[k].astype(np.float64) + v1[k].astype(np.float64)
        m = float(np.abs(want).max())
        if m == 0.0:
            assert float(np.abs(g2[k]).max()) == 0.0, k
            continue
        err = float(np.abs(g2[k].reshape(want.shape) - want).max()) / m
        print(f"  {workload:9s} {k:16s} |reduced - (view0 + view1)| / max = {err:.2e}   max|g| {m:.3e}")
        assert err <= 2e-5, (k, err)
    xm = line2["exchange_model"]
    assert xm["V"] == 8 and xm["allreduce_bytes"] < xm["dense_allreduce_bytes_avoided"] and min(np.atleast_1d(xm["sh_expand_ms_at_V"])) > 0
    if workload == "tinyfull":      # BASELINE config 5's parameter set: 21 floats per gaussian on the wire instead of 111 (+ the cubemap)
        assert 110.9 < xm["floats_per_gaussian_dense"] - 6 * 128 * 128 * 3 / line2["config"]["P"] < 111.1
        assert xm["floats_per_gaussian_on_the_wire"] - 6 * 128 * 128 * 3 / line2["config"]["P"] < 21.1
