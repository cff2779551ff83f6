"""PLY layout of GaussianModel.save_ply / load_ply (scene/gaussian_model.py:462-529, 725-838)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from materialrefgs_amd import io as mio  # noqa: E402


def _model(P=7, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)   # noqa: E731
    return {"xyz": r(P, 3), "normal1": r(P, 3), "normal2": r(P, 3), "features_dc": r(P, 1, 3), "features_rest": r(P, 15, 3),
            "indirect_dc": r(P, 1, 3), "indirect_rest": r(P, 15, 3), "indirect_asg": r(P, 32, 5), "opacity": r(P, 1),
            "refl_strength": r(P, 1), "metalness": r(P, 1), "roughness": r(P, 1), "ori_color": r(P, 3), "diffuse_color": r(P, 3),
            "scaling": r(P, 2), "rotation": r(P, 4)}


def test_attribute_order_is_the_reference_order():
    m = _model()
    names = mio.attribute_names({k: tuple(v.shape) for k, v in m.items()})
    # the list GaussianModel.construct_list_of_attributes() returned when the reference's own class was run
    # (tests/golden/gen_reference_render_vectors.py -> reference_render.npz: G_attributes)
    want = [str(x) for x in np.load(os.path.join(ROOT, "tests", "golden", "reference_render.npz"))["G_attributes"]]
    assert names == want and len(names) == 281
    assert want[:9] == ["x", "y", "z", "nx", "ny", "nz", "nx2", "ny2", "nz2"] and want[-1] == "rot_3" and want[9 + 3 + 45 + 3 + 45] == "ind_asg_0"


def test_ply_round_trip_and_byte_layout(tmp_path):
    m = _model(11, 3)
    path = str(tmp_path / "point_cloud" / "iteration_7" / "point_cloud.ply")
    mio.save_ply(path, m)
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().split("\n")
    assert lines[0] == "ply" and lines[1] == "format binary_little_endian 1.0" and lines[2] == "element vertex 11"
    assert lines[3] == "property float x" and lines[-2] == "property float rot_3"
    rows = np.frombuffer(body, dtype="<f4").reshape(11, 281)
    np.testing.assert_array_equal(rows[:, 0:3], m["xyz"].numpy())
    # SH coefficients are written channel-major: f_rest_k = features_rest[:, k % 15, k // 15]  (transpose(1, 2).flatten, :494)
    np.testing.assert_array_equal(rows[:, 12 + 16], m["features_rest"][:, 1, 1].numpy())
    np.testing.assert_array_equal(rows[:, 9 + 3 + 45 + 3 + 45 + 7], m["indirect_asg"][:, 7, 0].numpy())
    back = mio.load_ply(path)
    assert set(back) == set(m)
    for k in m:
        assert back[k].shape == m[k].shape, k
        torch.testing.assert_close(back[k], m[k], rtol=0, atol=0)


def test_env_map_files_round_trip(tmp_path):
    class Env(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.base = torch.nn.Parameter(torch.randn(6, 8, 8, 3))
    e1, e2 = Env(), Env()
    path = str(tmp_path / "point_cloud.ply")
    mio.save_ply(path, _model(), env_map=e1, env_map_2=e2)
    assert os.path.exists(str(tmp_path / "point_cloud1.map")) and os.path.exists(str(tmp_path / "point_cloud2.map"))
    f1, f2 = Env(), Env()
    mio.load_env_maps(path, f1, f2)
    torch.testing.assert_close(f1.base, e1.base, rtol=0, atol=0)
    torch.testing.assert_close(f2.base, e2.base, rtol=0, atol=0)


def test_envlight_signature_and_hdr_loading(tmp_path):
    """EnvLight takes the reference's constructor arguments in the reference's order (scene/light.py:22: path first, then device,
    scale); the Radiance .hdr reader and latlong_to_cubemap (scene/light_utils.py:34-48) round-trip a known image."""
    import inspect
    import numpy as np
    import torch
    from materialrefgs_amd import shading as sh
    params = list(inspect.signature(sh.EnvLight.__init__).parameters)
    assert params[:9] == ["self", "path", "device", "scale", "min_res", "max_res", "min_roughness", "max_roughness", "trainable"]
    rng = np.random.default_rng(0)
    H, W = 16, 32
    img = (rng.random((H, W, 3)) * 3).astype(np.float32)
    e = np.ceil(np.log2(img.max(-1))).astype(int)
    mant = np.clip(np.floor(img / np.ldexp(1.0, e)[..., None] * 256), 0, 255).astype(np.uint8)
    rgbe = np.concatenate([mant, (e + 128).astype(np.uint8)[..., None]], -1)
    p = tmp_path / "env.hdr"
    with open(p, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (H, W))
        f.write(rgbe.tobytes())
    back = sh.read_radiance_hdr(str(p))
    assert back.dtype == np.float32 and float(np.abs(back - img).max()) <= img.max() / 128          # 8-bit mantissas
    # a constant lat-long image maps to a constant cube; the +y face looks at the top rows of the image (v = acos(y) / pi -> 0)
    const = torch.full((H, W, 3), 0.25)
    assert torch.allclose(sh.latlong_to_cubemap(const, [8, 8], "cpu"), torch.full((6, 8, 8, 3), 0.25))
    grad = torch.linspace(0, 1, H)[:, None, None].expand(H, W, 3).contiguous()      # brightness grows towards the bottom (-y)
    cm = sh.latlong_to_cubemap(grad, [8, 8], "cpu")
    assert float(cm[2].mean()) < float(cm[0].mean()) < float(cm[3].mean())
