"""On-disk formats the reconstruction stages exchange: the Gaussian PLY point cloud and the
`(capture(), iteration)` checkpoint tuple.

Reference: gs-simp/scene/gaussian_model.py:177-208 (construct_list_of_attributes / save_ply), :267-313 (load_ply),
:61-93 (capture / restore), gs-simp/train.py:132, inpaint_rec.py:167 (`torch.save((gaussians.capture(), iteration), path)`), train.py:38 (load).
The reference writes PLY through the third-party `plyfile` package (absent from this image); this module writes the
same file with numpy alone: header `ply / format binary_little_endian 1.0 / element vertex N / property float <name>
...`, then N packed little-endian float32 records in the attribute order

    x y z  nx ny nz  f_dc_0..2  f_rest_0..(3(M-1)-1)  opacity  scale_0..2  rot_0..3

with features stored CHANNEL-major (`transpose(1, 2).flatten(1)`: f_rest_k = channel k // (M-1), coefficient
k % (M-1)), normals all zero, and every value the RAW (un-activated) parameter. Pure host code: no GPU involved.
"""
import os
from typing import Dict, List, Tuple

import numpy as np
import torch

_PLY_TYPES = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1",
              "char": "i1", "int8": "i1", "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2",
              "int": "<i4", "int32": "<i4", "uint": "<u4", "uint32": "<u4"}


def attribute_names(n_dc: int, n_rest: int, n_scale: int = 3, n_rot: int = 4) -> List[str]:
    """gaussian_model.py:177-189."""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(n_dc)]
    names += [f"f_rest_{i}" for i in range(n_rest)]
    names.append("opacity")
    names += [f"scale_{i}" for i in range(n_scale)]
    names += [f"rot_{i}" for i in range(n_rot)]
    return names


def _np(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def save_ply(path: str, xyz, features_dc, features_rest, opacity, scaling, rotation) -> None:
    """xyz [P,3], features_dc [P,1,3], features_rest [P,M-1,3], opacity [P,1], scaling [P,3], rotation [P,4]
    (raw parameters, tensors or arrays) -> binary little-endian PLY (gaussian_model.py:191-208)."""
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    xyz = _np(xyz).astype(np.float32)
    P = xyz.shape[0]
    f_dc = np.ascontiguousarray(np.transpose(_np(features_dc), (0, 2, 1))).reshape(P, -1)
    f_rest = np.ascontiguousarray(np.transpose(_np(features_rest), (0, 2, 1))).reshape(P, -1)
    cols = [xyz, np.zeros_like(xyz), f_dc, f_rest, _np(opacity).reshape(P, -1), _np(scaling).reshape(P, -1),
            _np(rotation).reshape(P, -1)]
    table = np.concatenate([c.astype(np.float32) for c in cols], axis=1)
    names = attribute_names(f_dc.shape[1], f_rest.shape[1], cols[5].shape[1], cols[6].shape[1])
    assert table.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\n" + f"element vertex {P}\n"
    header += "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(np.ascontiguousarray(table, dtype="<f4").tobytes())


def read_ply_vertices(path: str) -> Dict[str, np.ndarray]:
    """Minimal PLY reader for the files this pipeline exchanges: a leading `vertex` element of scalar properties,
    `binary_little_endian`, `binary_big_endian` or `ascii`. Elements declared after `vertex` (the `element face 0`
    MeshLab / Open3D / CloudCompare write) are ignored: the vertex block comes first in the body, only its n records
    are read — as plyfile, which the reference's load_ply uses, does. Returns {property name: [N] array}."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, n, props, in_vertex = None, None, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                if tok[1] == "vertex" and n is None:
                    in_vertex, n = True, int(tok[2])
                elif n is None:
                    raise ValueError(f"{path}: the first element must be `vertex`")
                else:
                    in_vertex = False                           # a later element: its properties are not ours
            elif tok[0] == "property":
                if in_vertex:
                    if tok[1] == "list":
                        raise ValueError(f"{path}: list properties on the vertex element are not supported")
                    props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if n is None:
            raise ValueError(f"{path}: no vertex element")
        if fmt in ("binary_little_endian", "binary_big_endian"):
            rec = np.dtype(props)
            if fmt == "binary_big_endian":
                rec = rec.newbyteorder(">")
            data = np.frombuffer(f.read(n * rec.itemsize), dtype=rec, count=n)
            return {name: np.ascontiguousarray(data[name]).astype(np.dtype(t), copy=False) for name, t in props}
        if fmt == "ascii":
            rows = np.loadtxt(f, dtype=np.float64, max_rows=n, ndmin=2)
            return {name: rows[:, i].astype(np.dtype(t)) for i, (name, t) in enumerate(props)}
        raise ValueError(f"{path}: unsupported PLY format {fmt}")


def load_ply(path: str, max_sh_degree: int) -> Dict[str, np.ndarray]:
    """gaussian_model.py:267-313: returns the raw parameters in the MODEL's layouts — xyz [P,3], features_dc [P,1,3],
    features_rest [P,M-1,3] (already transposed back from the file's channel-major order), opacity [P,1], scaling
    [P,3], rotation [P,4], float32. Property names are matched and sorted numerically like the reference does."""
    v = read_ply_vertices(path)
    xyz = np.stack([v["x"], v["y"], v["z"]], axis=1)
    P = xyz.shape[0]
    f_dc = np.stack([v["f_dc_0"], v["f_dc_1"], v["f_dc_2"]], axis=1)[:, :, None]          # [P,3,1]
    numbered = lambda prefix: sorted((k for k in v if k.startswith(prefix)), key=lambda s: int(s.split("_")[-1]))
    rest_names = numbered("f_rest_")
    M = (max_sh_degree + 1) ** 2
    assert len(rest_names) == 3 * M - 3, f"{path}: {len(rest_names)} f_rest properties, sh degree {max_sh_degree} needs {3 * M - 3}"
    f_rest = np.stack([v[k] for k in rest_names], axis=1).reshape(P, 3, M - 1) if rest_names else np.zeros((P, 3, 0))
    scales = np.stack([v[k] for k in numbered("scale_")], axis=1)
    rots = np.stack([v[k] for k in numbered("rot")], axis=1)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return dict(xyz=f32(xyz), features_dc=f32(np.transpose(f_dc, (0, 2, 1))), features_rest=f32(np.transpose(f_rest, (0, 2, 1))),
                opacity=f32(v["opacity"][:, None]), scaling=f32(scales), rotation=f32(rots))


CAPTURE_FIELDS = ("active_sh_degree", "_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity",
                  "max_radii2D", "xyz_gradient_accum", "denom", "optimizer_state_dict", "spatial_lr_scale")


def capture_tuple(state: Dict) -> Tuple:
    """The 12-tuple of GaussianModel.capture() (gaussian_model.py:61-75) from a dict keyed by CAPTURE_FIELDS."""
    return tuple(state[k] for k in CAPTURE_FIELDS)


def save_checkpoint(path: str, state: Dict, iteration: int) -> None:
    """train.py:132: torch.save((gaussians.capture(), iteration), path)."""
    torch.save((capture_tuple(state), iteration), path)


def load_checkpoint(path: str, map_location=None) -> Tuple[Dict, int]:
    """Inverse: (model_params, first_iter) = torch.load(path) (train.py:38) as a dict keyed by CAPTURE_FIELDS."""
    model_args, iteration = torch.load(path, map_location=map_location, weights_only=False)
    if len(model_args) != len(CAPTURE_FIELDS):
        raise ValueError(f"{path}: capture tuple has {len(model_args)} fields, expected {len(CAPTURE_FIELDS)}")
    return dict(zip(CAPTURE_FIELDS, model_args)), int(iteration)
