"""Wire formats of the reconstruction stages: the Gaussian PLY (gs-simp/scene/gaussian_model.py:177-208,
:267-313) and the (capture(), iteration) checkpoint tuple (:61-93, train.py:132). The reference writes through the absent
third-party `plyfile`, so the file layout is checked byte for byte against the PLY specification it follows (header text
+ packed little-endian float32 records) and through save -> load round trips in the model's own layouts."""
import struct

import numpy as np
import pytest
import torch

from multiview_inpaint_amd import gaussian_io as gio


def _model(P, deg, seed=0):
    g = torch.Generator().manual_seed(seed)
    M = (deg + 1) ** 2
    r = lambda *s: torch.randn(*s, generator=g)
    return dict(xyz=r(P, 3), features_dc=r(P, 1, 3), features_rest=r(P, M - 1, 3), opacity=r(P, 1), scaling=r(P, 3), rotation=r(P, 4))


@pytest.mark.parametrize("deg", [0, 1, 3])
def test_ply_layout_and_round_trip(tmp_path, deg):
    m = _model(37, deg)
    path = str(tmp_path / "point_cloud" / "iteration_7" / "point_cloud.ply")        # directories are created like mkdir_p
    gio.save_ply(path, **m)
    raw = open(path, "rb").read()
    M = (deg + 1) ** 2
    names = gio.attribute_names(3, 3 * (M - 1))
    assert names[:9] == ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"]
    assert names[-8:] == ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    header = ("ply\nformat binary_little_endian 1.0\nelement vertex 37\n" + "".join(f"property float {n}\n" for n in names)
              + "end_header\n").encode()
    assert raw.startswith(header)
    body = raw[len(header):]
    assert len(body) == 37 * 4 * len(names)
    rec5 = struct.unpack("<" + "f" * len(names), body[5 * 4 * len(names):6 * 4 * len(names)])
    np.testing.assert_array_equal(rec5[:3], m["xyz"][5].numpy())
    assert rec5[3:6] == (0.0, 0.0, 0.0)                                              # normals
    np.testing.assert_array_equal(rec5[6:9], m["features_dc"][5, 0].numpy())
    # channel-major: f_rest_k = channel k // (M-1), coefficient k % (M-1)
    want = m["features_rest"][5].transpose(0, 1).reshape(-1).numpy()
    np.testing.assert_array_equal(rec5[9:9 + 3 * (M - 1)], want)
    back = gio.load_ply(path, deg)
    for k, v in m.items():
        assert back[k].dtype == np.float32 and back[k].shape == tuple(v.shape), k
        np.testing.assert_array_equal(back[k], v.numpy())


def test_ply_reader_sorts_numbered_properties_and_reads_ascii(tmp_path):
    """load_ply sorts f_rest_* / scale_* / rot_* numerically (gaussian_model.py:281-302): a file whose properties are
    declared out of order, in ascii, with a comment line, loads to the same model."""
    names = ["x", "y", "z", "opacity", "rot_3", "rot_2", "rot_1", "rot_0", "scale_2", "scale_0", "scale_1",
             "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in reversed(range(9))]
    rows = np.arange(2 * len(names), dtype=np.float64).reshape(2, -1) / 8
    path = tmp_path / "a.ply"
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 2\n")
        f.write("".join(f"property float {n}\n" for n in names) + "end_header\n")
        for r in rows:
            f.write(" ".join(repr(float(x)) for x in r) + "\n")
    m = gio.load_ply(str(path), 1)
    col = {n: rows[:, i].astype(np.float32) for i, n in enumerate(names)}
    np.testing.assert_array_equal(m["rotation"], np.stack([col[f"rot_{i}"] for i in range(4)], 1))
    np.testing.assert_array_equal(m["scaling"], np.stack([col[f"scale_{i}"] for i in range(3)], 1))
    np.testing.assert_array_equal(m["features_rest"][:, :, 0], np.stack([col[f"f_rest_{i}"] for i in (0, 1, 2)], 1))
    np.testing.assert_array_equal(m["features_rest"][:, 1, :], np.stack([col[f"f_rest_{i}"] for i in (1, 4, 7)], 1))
    with pytest.raises(AssertionError):
        gio.load_ply(str(path), 3)                         # wrong number of f_rest properties for the degree


def test_ply_reader_rejects_other_files(tmp_path):
    p = tmp_path / "x.ply"
    p.write_bytes(b"plx\n")
    with pytest.raises(ValueError):
        gio.read_ply_vertices(str(p))
    p.write_bytes(b"ply\nformat binary_middle_endian 1.0\nelement vertex 0\nproperty float x\nend_header\n")
    with pytest.raises(ValueError):
        gio.read_ply_vertices(str(p))
    p.write_bytes(b"ply\nformat ascii 1.0\nelement face 0\nproperty list uchar int vertex_indices\nend_header\n")
    with pytest.raises(ValueError):
        gio.read_ply_vertices(str(p))                      # vertex must be the first element
    p.write_bytes(b"ply\nformat binary_little_endian 1.0\nelement vertex 0\nproperty float x\nend_header\n")
    assert gio.read_ply_vertices(str(p))["x"].shape == (0,)


def test_checkpoint_tuple_matches_capture_order(tmp_path):
    m = _model(11, 1)
    p = torch.nn.Parameter(m["xyz"].clone())
    opt = torch.optim.Adam([{"params": [p], "lr": 1e-3, "name": "xyz"}], lr=0.0, eps=1e-15)
    p.grad = torch.ones_like(p)
    opt.step()
    state = dict(active_sh_degree=1, _xyz=p, _features_dc=m["features_dc"], _features_rest=m["features_rest"],
                 _scaling=m["scaling"], _rotation=m["rotation"], _opacity=m["opacity"], max_radii2D=torch.zeros(11),
                 xyz_gradient_accum=torch.zeros(11, 1), denom=torch.zeros(11, 1), optimizer_state_dict=opt.state_dict(),
                 spatial_lr_scale=2.5)
    path = str(tmp_path / "chkpnt30000.pth")
    gio.save_checkpoint(path, state, 30000)
    (args, it) = torch.load(path, weights_only=False)                  # what train.py:38 does
    assert it == 30000 and len(args) == 12
    # positional order of GaussianModel.restore (gaussian_model.py:77-89)
    assert args[0] == 1 and args[11] == 2.5 and torch.equal(args[1], p) and torch.equal(args[4], m["scaling"])
    assert torch.equal(args[5], m["rotation"]) and torch.equal(args[6], m["opacity"])
    assert args[10]["param_groups"][0]["name"] == "xyz"
    st, it2 = gio.load_checkpoint(path)
    assert it2 == 30000 and torch.equal(st["_xyz"], p) and st["optimizer_state_dict"]["state"][0]["step"] == 1


def test_extend_optimizer_state_matches_cat_tensors_to_optimizer():
    """densification_postfix's optimizer surgery (gaussian_model.py:384-404) on torch.optim.Adam: parameters grow, moments
    get zero rows, stepping continues identically to the reference's own loop."""
    from multiview_inpaint_amd.train_ops import extend_optimizer_state
    shapes = {"xyz": (3,), "f_dc": (1, 3), "opacity": (1,)}

    def make():
        g = torch.Generator().manual_seed(1)
        ps = {k: torch.nn.Parameter(torch.randn(7, *s, generator=g)) for k, s in shapes.items()}
        opt = torch.optim.Adam([{"params": [p], "lr": 1e-2, "name": k} for k, p in ps.items()], lr=0.0, eps=1e-15)
        for p in ps.values():
            p.grad = torch.randn(p.shape, generator=g)
        opt.step()
        return opt
    g2 = torch.Generator().manual_seed(2)
    ext = {k: torch.randn(3, *s, generator=g2) for k, s in shapes.items()}
    a, b = make(), make()
    new = extend_optimizer_state(a, ext)
    for grp in b.param_groups:                               # the reference's loop
        e = ext[grp["name"]]
        st = b.state.get(grp["params"][0], None)
        st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(e)), dim=0)
        st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(e)), dim=0)
        del b.state[grp["params"][0]]
        grp["params"][0] = torch.nn.Parameter(torch.cat((grp["params"][0], e), dim=0).requires_grad_(True))
        b.state[grp["params"][0]] = st
    for ga, gb in zip(a.param_groups, b.param_groups):
        pa, pb = ga["params"][0], gb["params"][0]
        assert new[ga["name"]] is pa and pa.shape[0] == 10 and torch.equal(pa, pb)
        assert torch.equal(a.state[pa]["exp_avg"], b.state[pb]["exp_avg"]) and (a.state[pa]["exp_avg"][7:] == 0).all()
        pa.grad, pb.grad = torch.ones_like(pa), torch.ones_like(pb)
    a.step(); b.step()
    for ga, gb in zip(a.param_groups, b.param_groups):
        assert torch.equal(ga["params"][0], gb["params"][0])


def test_ply_reader_ignores_trailing_elements_and_reads_big_endian(tmp_path):
    """Files written by MeshLab / Open3D / CloudCompare declare `element face 0` (with a list property) after the vertex
    element; plyfile — what the reference's load_ply uses (gaussian_model.py:267-270) — reads them, and so must this
    reader. Big-endian bodies are converted."""
    xyz = np.arange(6, dtype=np.float32).reshape(2, 3) + 0.25
    head = ("ply\nformat {fmt} 1.0\nelement vertex 2\nproperty float x\nproperty float y\nproperty float z\n"
            "element face 0\nproperty list uchar int vertex_indices\nend_header\n")
    for fmt, body in (("binary_little_endian", xyz.astype("<f4").tobytes()), ("binary_big_endian", xyz.astype(">f4").tobytes())):
        p = tmp_path / f"{fmt}.ply"
        p.write_bytes(head.format(fmt=fmt).encode() + body)
        v = gio.read_ply_vertices(str(p))
        assert sorted(v) == ["x", "y", "z"]
        np.testing.assert_array_equal(np.stack([v["x"], v["y"], v["z"]], 1), xyz)
        assert v["x"].dtype == np.float32 and v["x"].dtype.byteorder in "=<|"
