"""On-disk formats of a trained model: the point-cloud PLY of GaussianModel.save_ply / load_ply
(scene/gaussian_model.py:462-529, 725-838) and the environment-map `.map` files next to it (`EnvLight.state_dict()` through
torch.save, :520-527).  Plain host I/O in numpy -- the reference goes through the `plyfile` package, which writes exactly this
layout: ASCII header, one `vertex` element, every property `float` (f4), little-endian rows in the attribute order of
`construct_list_of_attributes` (:462-487)."""
import os
from typing import Dict

import numpy as np
import torch

# tensor name -> (PLY prefix, transposed on disk?)  in the order save_ply concatenates them (:515)
_FIELDS = [("xyz", None), ("normal1", None), ("normal2", None), ("features_dc", "f_dc"), ("features_rest", "f_rest"), ("indirect_dc", "ind_dc"),
           ("indirect_rest", "ind_rest"), ("indirect_asg", "ind_asg"), ("opacity", "opacity"), ("refl_strength", "refl_strength"),
           ("metalness", "metalness"), ("roughness", "roughness"), ("ori_color", "ori_color"), ("diffuse_color", "diffuse_color"),
           ("scaling", "scale"), ("rotation", "rot")]
_SCALAR = {"opacity", "refl_strength", "metalness", "roughness"}
_CHANNEL_MAJOR = {"features_dc", "features_rest", "indirect_dc", "indirect_rest", "indirect_asg"}   # stored as [P, C, K] -> transpose(1, 2).flatten


def attribute_names(shapes: Dict[str, tuple]):
    """construct_list_of_attributes (:462-487) for tensors of the given shapes."""
    names = ["x", "y", "z", "nx", "ny", "nz", "nx2", "ny2", "nz2"]
    for key, prefix in _FIELDS[3:]:
        n = int(np.prod(shapes[key][1:]))
        if key in _SCALAR:
            names.append(prefix)
        else:
            names += [f"{prefix}_{i}" for i in range(n)]
    return names


def _flat(key, t):
    a = t.detach().cpu().float()
    if key in _CHANNEL_MAJOR:
        a = a.transpose(1, 2)
    return a.reshape(a.shape[0], -1).contiguous().numpy()


def save_ply(path: str, tensors: Dict[str, torch.Tensor], env_map=None, env_map_2=None):
    """tensors: GaussianModel layout -- xyz [P,3], normal1/2 [P,3], features_dc [P,1,3], features_rest [P,15,3], indirect_dc [P,1,3],
    indirect_rest [P,15,3], indirect_asg [P,32,5], opacity/refl_strength/metalness/roughness [P,1], ori_color/diffuse_color [P,3],
    scaling [P,2], rotation [P,4] (raw, pre-activation values as the reference stores them)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    cols = [_flat(k, tensors[k]) for k, _ in _FIELDS]
    names = attribute_names({k: tuple(tensors[k].shape) for k, _ in _FIELDS})
    data = np.ascontiguousarray(np.concatenate(cols, axis=1), dtype="<f4")
    assert data.shape[1] == len(names)
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {data.shape[0]}"] + [f"property float {n}" for n in names] + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(data.tobytes())
    if env_map is not None:
        torch.save(env_map.state_dict(), path.replace(".ply", "1.map"))
    if env_map_2 is not None:
        torch.save(env_map_2.state_dict(), path.replace(".ply", "2.map"))


def read_ply_vertices(path: str):
    """(property names, float32 array [P, n_properties]) of the vertex element of a binary little-endian or ASCII PLY whose vertex
    properties are all 4-byte floats (what save_ply writes)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        fmt, count, names, in_vertex = None, None, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError("PLY header not terminated")
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    count = int(tok[2])
                elif count is None:
                    raise ValueError("the vertex element must come first")
            elif tok[0] == "property" and in_vertex:
                if tok[1] not in ("float", "float32"):
                    raise ValueError(f"unsupported vertex property type {tok[1]}")
                names.append(tok[2])
            elif tok[0] == "end_header":
                break
        if fmt == "binary_little_endian":
            data = np.frombuffer(f.read(count * len(names) * 4), dtype="<f4").reshape(count, len(names))
        elif fmt == "ascii":
            data = np.loadtxt(f, dtype=np.float32, max_rows=count).reshape(count, len(names))
        else:
            raise ValueError(f"unsupported PLY format {fmt}")
    return names, np.array(data, dtype=np.float32)


def load_ply(path: str, max_sh_degree: int = 3, device="cpu") -> Dict[str, torch.Tensor]:
    """load_ply (:725-838): returns the tensors in the GaussianModel layout (see save_ply)."""
    names, data = read_ply_vertices(path)
    col = {n: i for i, n in enumerate(names)}

    def group(prefix):
        ks = sorted([n for n in names if n.startswith(prefix + "_")], key=lambda x: int(x.split("_")[-1]))
        return data[:, [col[k] for k in ks]]

    def cols(*ks):
        return data[:, [col[k] for k in ks]]

    P = data.shape[0]
    n_rest = 3 * (max_sh_degree + 1) ** 2 - 3
    f_rest, ind_rest = group("f_rest"), group("ind_rest")
    assert f_rest.shape[1] == n_rest and ind_rest.shape[1] == n_rest                  # :762, :775
    out = {
        "xyz": cols("x", "y", "z"), "normal1": cols("nx", "ny", "nz"), "normal2": cols("nx2", "ny2", "nz2"),
        "features_dc": group("f_dc").reshape(P, 3, 1).transpose(0, 2, 1),
        "features_rest": f_rest.reshape(P, 3, -1).transpose(0, 2, 1),
        "indirect_dc": group("ind_dc").reshape(P, 3, 1).transpose(0, 2, 1),
        "indirect_rest": ind_rest.reshape(P, 3, -1).transpose(0, 2, 1),
        "indirect_asg": group("ind_asg").reshape(P, 5, -1).transpose(0, 2, 1),          # :788
        "opacity": cols("opacity"), "refl_strength": cols("refl_strength"), "metalness": cols("metalness"), "roughness": cols("roughness"),
        "ori_color": group("ori_color"), "diffuse_color": group("diffuse_color"), "scaling": group("scale"), "rotation": group("rot"),
    }
    return {k: torch.tensor(np.ascontiguousarray(v), dtype=torch.float32, device=device) for k, v in out.items()}


def load_env_maps(path: str, env_map=None, env_map_2=None):
    """The two `.map` files next to a PLY (:804-812): state dicts of EnvLight (key `base`, [6, res, res, 3])."""
    for env, suffix in ((env_map, "1.map"), (env_map_2, "2.map")):
        p = path.replace(".ply", suffix)
        if env is not None and os.path.exists(p):
            env.load_state_dict(torch.load(p, map_location="cpu"))
