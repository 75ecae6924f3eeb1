import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from multiview_inpaint_amd.svd import bench_svd as b, hip_ops
dev = torch.device("cuda")
eng = b.build(dev, with_control=True)
x, cond, ind = b.inputs(dev)
def mk(name):
    def pre(m, args):
        shp = [tuple(a.shape) if torch.is_tensor(a) else type(a).__name__ for a in args]
        print("ENTER", name, type(m).__name__, shp, flush=True)
    return pre
for name, m in eng.named_modules():
    if len(list(m.children())) == 0:
        m.register_forward_pre_hook(mk(name))
_a, _g = hip_ops.attention, hip_ops.group_norm_silu
def a2(q, k, v, h):
    print("HIP attention", tuple(q.shape), tuple(k.shape), q.dtype, h, flush=True); r = _a(q, k, v, h); torch.cuda.synchronize(); return r
def g2(x, *a):
    print("HIP groupnorm", tuple(x.shape), x.dtype, x.is_contiguous(), flush=True); r = _g(x, *a); torch.cuda.synchronize(); return r
hip_ops.attention, hip_ops.group_norm_silu = a2, g2
s = torch.full((28,), 100.0, device=dev)
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    out = eng.denoise(x, s, cond, num_video_frames=14, image_only_indicator=ind)
torch.cuda.synchronize()
print("DONE", out.shape, torch.isfinite(out).all().item(), flush=True)
