"""How many DISTINCT Gaussians does render_forward ever stage? (round 4: sizing a lazy SH -> RGB evaluation)

render_forward stages a tile's depth-sorted list 256 entries per round and stops after the round in which every pixel of the
tile has saturated; only staged entries need a colour. Prints, for the bench scene: visible Gaussians, gradient support,
distinct Gaussians in the rounds the forward needs (max n_contrib of the tile rounded up to 256), in those rounds + the one
prefetched round, and in round 0 alone."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multiview_inpaint_amd import raster as R, synthetic as syn
W, H, N = 1920, 1080, 1500000
cam = syn.make_camera(W, H, 50.0); sc = syn.make_scene(N, cam, 3, seed=0)
d = "cuda"
t = {k: torch.tensor(v, device=d) for k, v in sc.items() if k != "sh_degree"}
rs = R.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.zeros(3, device=d),
                                     scale_modifier=1.0, viewmatrix=torch.tensor(cam["viewmatrix"], device=d),
                                     projmatrix=torch.tensor(cam["projmatrix"], device=d), sh_degree=3,
                                     campos=torch.tensor(cam["campos"], device=d), prefiltered=False)
kw = dict(shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
c, radii, dep, st = R.rasterize_forward(rs, t["means3D"], t["opacities"], prepare_backward=True, **kw)
g = R.rasterize_backward(rs, st, torch.randn(3, H, W, device=d), t["means3D"], **kw)
D = st.D
gx, gy = (W + 15) // 16, (H + 15) // 16
tiles = gx * gy
ranges = st.tensor("ranges", (tiles, 2), torch.int32).long()
plist = st.tensor("point_list", (D,), torch.int32).long()
nc = st.tensor("n_contrib", (H, W), torch.int32)
pad_h, pad_w = (-H) % 16, (-W) % 16
nct = torch.nn.functional.pad(nc, (0, pad_w, 0, pad_h)).reshape(gy, 16, gx, 16)
tile_max = nct.amax(dim=(1, 3)).reshape(-1).long()          # last contributing 1-based position per tile
lens = ranges[:, 1] - ranges[:, 0]
# the forward stops staging after the round in which the last pixel of the tile saturates; the position of a pixel's last
# contributor (n_contrib) sits right in front of its saturation point, so ceil(tile_max / 256) rounds is what a tile needs
# (every pixel of this scene saturates: final_T < 1e-2 everywhere)
rounds_all = (lens + 255) // 256
rounds_need = torch.minimum((tile_max + 255) // 256, rounds_all)
unsat = rounds_need >= rounds_all


def distinct(rounds):
    n = torch.minimum(rounds * 256, lens)
    tot = int(n.sum())
    tile_of = torch.repeat_interleave(torch.arange(tiles, device=d), n)
    off = torch.arange(tot, device=d) - torch.repeat_interleave(torch.cumsum(n, 0) - n, n)
    ids = plist[ranges[tile_of, 0] + off]
    return tot, int(torch.unique(ids).numel())


vis = int((radii > 0).sum())
sup = int(st.tensor("grad_support", (N,), torch.uint8).sum())
print(f"N {N} visible {vis} D {D} gradient support {sup} tiles {tiles} unsaturated tiles {int(unsat.sum())}")
for name, r in (("round 0 only", torch.minimum(torch.ones_like(rounds_all), rounds_all)), ("needed rounds", rounds_need),
                ("needed + 1 prefetched", torch.minimum(rounds_need + 1, rounds_all)), ("whole lists", rounds_all)):
    tot, u = distinct(r)
    print(f"{name}: staged entries {tot} ({tot / D:.3f} of D), distinct Gaussians {u} ({u / N:.4f} of N, {u / vis:.4f} of visible)")
