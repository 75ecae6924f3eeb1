"""Which lines convert a reduced-precision CUDA tensor to fp32 during one forward of the small production-width nets (a conversion per call
is a launch per call: the wrappers cache the fp32 form of parameters). Prints a count per call site. GPU box: python tools/experiments/find_float_casts.py"""
import collections
import os
import sys
import traceback

import torch

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import svd_helpers as H  # noqa: E402
from multiview_inpaint_amd.svd import layers as LY  # noqa: E402
from multiview_inpaint_amd.svd.unet import ControlNet, ControlledVideoUNet  # noqa: E402

dtype = torch.bfloat16
cunet = ControlledVideoUNet(**H.SMALL_UNET320).eval().cuda().to(dtype)
cnet = ControlNet(**H.SMALL_CTRL320).eval().cuda().to(dtype)
inp = H.seeded_inputs(53, hw=H.LATENT_HW320, cfg=H.SMALL_UNET320)
inp = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in inp.items()}
kw = dict(num_video_frames=H.T_FRAMES, image_only_indicator=inp["image_only_indicator"])
xin = torch.cat([inp["x"], inp["concat"]], 1).to(dtype)
tt = 0.25 * inp["sigma"].log()
ctx, vec, hint = inp["crossattn"].to(dtype), inp["vector"].to(dtype), inp["control_hint"].to(dtype)
LY.CONV_N320_MIN_BLOCKS = 1
sites = collections.Counter()
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "multiview_inpaint_amd" in fr.filename and "find_float_casts" not in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line}"
    return "?"


class Watch(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        out = func(*args, **kwargs)
        name = str(func)
        if ("_to_copy" in name or "copy_" in name) and torch.is_tensor(out) and out.is_cuda and out.dtype == torch.float32:
            src = [a for a in args if torch.is_tensor(a) and a.dtype in (torch.bfloat16, torch.float16)]
            if src:
                sites[name + " " + site()] += 1
        # --big: every ATen op that writes a large tensor (a full-size pass over an activation that no HIP wrapper accounts for)
        if "--big" in sys.argv and torch.is_tensor(out) and out.is_cuda and out.numel() >= (1 << 22) and "aten.empty" not in name \
                and "view" not in name and "reshape" not in name and "permute" not in name and "expand" not in name and "slice" not in name \
                and "select" not in name and "transpose" not in name and "detach" not in name and "alias" not in name and "as_strided" not in name \
                and "unsqueeze" not in name and "squeeze" not in name and "split" not in name and "unbind" not in name and "chunk" not in name:
            sites[f"BIG {name} {out.numel() * out.element_size() / 1e6:.0f} MB  " + site()] += 1
        return out


if "--full" in sys.argv:                      # the 14 x 576x1024 step of bench.py through the engine
    from multiview_inpaint_amd.svd import bench_svd
    from multiview_inpaint_amd.svd.schedule import EDMDiscretization
    dev = torch.device("cuda", 0)
    bench_svd.use_shipped_miopen_db()
    bench_svd.enable_gemm_tuning()
    os.environ.setdefault("MVI_SVD_TWO_STREAMS", "0")
    eng = bench_svd.build(dev, with_control=True, dtype=torch.bfloat16)
    x, cond, ind = bench_svd.inputs(dev, 14, 72, 128)
    cond = {k: v.to(torch.bfloat16) for k, v in cond.items()}
    sig = EDMDiscretization(sigma_max=700.0)(25, device=dev)
    with torch.no_grad():
        for rep in range(3):
            if rep == 2:
                with Watch():
                    eng.denoise(x, sig[rep].expand(x.shape[0]), cond, num_video_frames=14, image_only_indicator=ind)
            else:
                eng.denoise(x, sig[rep].expand(x.shape[0]), cond, num_video_frames=14, image_only_indicator=ind)
    torch.cuda.synchronize()
    for s_, n in sites.most_common():
        print(n, s_)
    sys.exit(0)

with torch.no_grad():
    for rep in range(3):
        if rep == 2:
            with Watch():
                ctrls = cnet(xin, hint, tt, ctx, vec, tokens_out=True, **kw)
                yc = cunet(xin, tt, ctx, vec, control=list(ctrls), **kw)
        else:
            ctrls = cnet(xin, hint, tt, ctx, vec, tokens_out=True, **kw)
            yc = cunet(xin, tt, ctx, vec, control=list(ctrls), **kw)
torch.cuda.synchronize()
for s, n in sites.most_common():
    print(n, s)
