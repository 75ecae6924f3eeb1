"""View-sharded multi-view reconstruction — BASELINE.json configs[4], SURVEY.md §8e: the loop body of
gs-simp/inpaint_rec.py:96-163 with the camera stack (scene/__init__.py:415-453) sharded over the ranks of one node.

One process per GPU (torchrun; `torch.distributed`, backend "nccl" = RCCL over xGMI). The Gaussian parameters and the Adam state are
replicated; per step every rank draws the SAME `world` views from an identically seeded stack and renders ITS one; the view
gradients are summed by ONE exchange (dist.CompactedGradExchange: the supports travel as bit masks in one all-gather, then only the
rows inside the union of the supports move, the SH gradient as 3-float colour factors); the densification statistics — per-view
norms and radii, gaussian_model.py:482-484, inpaint_rec.py:146-150 — are reduced on the side; densify_and_prune / reset_opacity
(gaussian_model.py:466-480, :437 `torch.normal`) then run on every rank on identical inputs with an identically seeded generator, so
the replicas stay bit-identical through a change of P. No collective runs inside the rasterizer.

What stays the reference's: the model class and its methods (update_learning_rate, oneupSHdegree, densify_and_prune, reset_opacity,
capture), the camera objects, the argument classes. What is this package's: the rasterizer fed with the model's STORED parameters
(raster.rasterize_forward_raw / rasterize_backward_raw(sh_grad="factor"): activations and their chain rule inside the kernels), the
fused L1 + DSSIM loss incl. the mask of the non-inpainted views (inpaint_rec.py:117-123 -> train_ops.photometric_loss_forward_backward(
weight = 1 - mask)), FusedAdam (through dropin.patch_gs_simp), the exchange.

Semantics: `world` views per optimizer step instead of the reference's one ("Semantics caveat"). reduce="sum" (default)
steps on the SUM of the view gradients, reduce="mean" on their mean; at world = 1 both are the reference's step.

    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m multiview_inpaint_amd.train_views /path/to/gs-simp \\
        -s <scene> --scene_id <name> --n_mode 2 ...          (the arguments of gs-simp/inpaint_rec.py)

The launcher (main) pins the process to its GPU BEFORE anything touches a GPU, patches the gs-simp modules (dropin.patch_gs_simp)
before they are imported, builds the reference's InpaintScene / InpaintGaussianModel on every rank (read-only inputs) and lets rank 0
alone write renders, point clouds and checkpoints.
"""
import math
import os
import random
import sys
from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.distributed as td

PARAMS = (("xyz", "_xyz"), ("f_dc", "_features_dc"), ("f_rest", "_features_rest"), ("opacity", "_opacity"),
          ("scaling", "_scaling"), ("rotation", "_rotation"))


def _world(group=None):
    return td.get_world_size(group) if td.is_available() and td.is_initialized() else 1


def _rank(group=None):
    return td.get_rank(group) if td.is_available() and td.is_initialized() else 0


class ViewStack:
    """The reference's `viewpoint_stack` (inpaint_rec.py:104-108: refilled from getInpaintTrainCameras when empty, one random
    pop per iteration) drawn identically on every rank: a private, seeded random.Random, `world` pops per step, rank r keeps
    the r-th. `cameras_fn()` must return the same list in the same order on every rank (the reference shuffles it with the global
    `random`, which the launcher seeds identically — scene/__init__.py:451; general_utils.py:132)."""

    def __init__(self, cameras_fn: Callable[[], Sequence], world: int, rank: int, seed: int = 0):
        self.cameras_fn, self.world, self.rank = cameras_fn, int(world), int(rank)
        self.rng = random.Random(seed)
        self.stack: List = []

    def next_views(self) -> List:
        """The `world` views of one step (identical lists on every rank)."""
        out = []
        for _ in range(self.world):
            if not self.stack:
                self.stack = list(self.cameras_fn())
                if not self.stack:
                    raise RuntimeError("ViewStack: the camera list is empty")
            out.append(self.stack.pop(self.rng.randint(0, len(self.stack) - 1)))
        return out

    def next_view(self):
        return self.next_views()[self.rank]


def _settings(R, cam, bg, sh_degree, scaling_modifier=1.0):
    return R.GaussianRasterizationSettings(
        image_height=int(cam.image_height), image_width=int(cam.image_width), tanfovx=math.tan(cam.FoVx * 0.5),
        tanfovy=math.tan(cam.FoVy * 0.5), bg=bg, scale_modifier=scaling_modifier, viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform, sh_degree=int(sh_degree), campos=cam.camera_center, prefiltered=False)


class ViewShardedTrainer:
    """One optimizer step = `world` views. `gaussians`: the reference's (Inpaint)GaussianModel after training_setup() — the six
    stored parameter tensors, `optimizer` with one named group each (gaussian_model.py:154-163), xyz_gradient_accum / denom /
    max_radii2D, active_sh_degree and the methods named in the module docstring. `opt`: the reference's OptimizationParams
    (lambda_dssim, densify_from_iter, densify_until_iter, densification_interval, opacity_reset_interval, densify_grad_threshold,
    random_background, iterations). Works without a process group (world = 1: the reference's loop on this package's ops)."""

    def __init__(self, gaussians, opt, cameras_fn: Callable[[], Sequence], background: torch.Tensor, cameras_extent: float,
                 group=None, seed: int = 0, reduce: str = "sum", white_background: bool = False, world: Optional[int] = None,
                 rank: Optional[int] = None):
        """world / rank default to the process group's; world = 1 forces the single-process form inside an initialised group (the
        comparison runs of the tests)."""
        if reduce not in ("sum", "mean"):
            raise ValueError("reduce must be 'sum' or 'mean'")
        self.g, self.opt, self.group = gaussians, opt, group
        self.world = _world(group) if world is None else int(world)
        self.rank = (_rank(group) if self.world > 1 else 0) if rank is None else int(rank)
        self.background, self.extent = background, float(cameras_extent)
        self.reduce, self.white_background, self.seed = reduce, bool(white_background), int(seed)
        self.views = ViewStack(cameras_fn, self.world, self.rank, seed)
        self.exchange = None
        self.last = {}

    # ---- pieces -----------------------------------------------------------------------------------------------------------
    def params(self) -> Dict[str, torch.Tensor]:
        return {name: getattr(self.g, attr) for name, attr in PARAMS}

    def _exchange_for(self, P: int, M: int, dev):
        """The exchange object for the current model size (rebuilt when densification changed P)."""
        from . import dist as md
        ex = self.exchange
        if ex is None or ex.P != P or ex.M != M:
            ex = self.exchange = md.CompactedGradExchange(P, M, int(self.g.active_sh_degree), dev, group=self.group, split_sh=True)
        ex.deg = int(self.g.active_sh_degree)                  # (oneupSHdegree raises it every 1000 iterations)
        return ex

    def _bg(self, iteration: int, dev):
        if not getattr(self.opt, "random_background", False):
            return self.background
        gen = torch.Generator("cpu").manual_seed((self.seed * 1000003 + iteration) * 64 + self.rank)      # a view's own colour
        return torch.rand(3, generator=gen).to(dev)

    def view_gradients(self, cam, iteration: int):
        """Forward, loss and backward of ONE view on the stored parameters. Returns (st, rs, loss3, grads): grads in the
        exchange's buffers (world > 1: SH gradient as colour factor) or freshly allocated (world = 1: dense)."""
        from . import raster as R, train_ops as T
        p = self.params()
        dev = p["xyz"].device
        raw = [p[k].detach() for k in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")]
        rs = _settings(R, cam, self._bg(iteration, dev), self.g.active_sh_degree)
        image, radii, _depth, st = R.rasterize_forward_raw(rs, *raw, prepare_backward=True)
        gt = cam.original_image.to(dev)
        inpainted = bool(getattr(cam, "inpainted", True))
        mask = getattr(cam, "mask", None)
        weight = None if inpainted or mask is None else 1.0 - mask.to(dev, torch.float32)        # inpaint_rec.py:120-123
        up = 1.0 / self.world if self.reduce == "mean" else 1.0
        loss3, g_img = T.photometric_loss_forward_backward(image, gt, float(self.opt.lambda_dssim), weight, upstream=up)
        if self.world == 1:
            g = R.rasterize_backward_raw(rs, st, g_img, *raw)
            return st, rs, radii, loss3, g
        ex = self._exchange_for(st.P, st.M, dev)
        out = dict(xyz=ex.views["means3D"], opacity=ex.views["opacities"], scaling=ex.views["scales"], rotation=ex.views["rotations"],
                   sh_color_factor=ex.views["sh_color_factor"], means2D=ex.views["means2D"])
        g = R.rasterize_backward_raw(rs, st, g_img, *raw, out=out, sh_grad="factor")
        return st, rs, radii, loss3, g

    def _densification_stats(self, means2D_grad, radii):
        """add_densification_stats + the max-radii update (gaussian_model.py:482-484, inpaint_rec.py:146-150) over the step's views."""
        from . import dist as md
        vis = radii > 0
        if self.world == 1:
            self.g.max_radii2D[vis] = torch.max(self.g.max_radii2D[vis], radii[vis].to(self.g.max_radii2D.dtype))
            n = torch.norm(means2D_grad[:, :2], dim=-1, keepdim=True)
            self.g.xyz_gradient_accum += torch.where(vis[:, None], n, 0.0).to(self.g.xyz_gradient_accum.dtype)
            self.g.denom += vis[:, None].to(self.g.denom.dtype)
            return
        # reduce="mean" ran the backward with upstream = 1 / world: the statistic is the norm of the UNSCALED view gradient
        # (gaussian_model.py:482-484 against densify_grad_threshold), whatever the optimizer steps on
        md.reduce_densification_stats(means2D_grad, vis, radii, self.g.xyz_gradient_accum, self.g.denom, self.g.max_radii2D,
                                      group=self.group, norm_scale=float(self.world) if self.reduce == "mean" else 1.0)

    def _seed_densification(self, iteration: int):
        """torch.normal of densify_and_split (gaussian_model.py:437) must draw the same samples on every rank."""
        s = (self.seed * 1000003 + iteration) & 0x7FFFFFFF
        torch.manual_seed(s)
        if torch.cuda.is_available():
            torch.cuda.manual_seed(s)

    # ---- one step ---------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, iteration: int):
        """inpaint_rec.py:96-163 for `world` views. Returns the device tensor [loss, mean|x - y|, mean SSIM] of THIS rank's view."""
        g, opt = self.g, self.opt
        g.update_learning_rate(iteration)
        if iteration % 1000 == 0:
            g.oneupSHdegree()
        cam = self.views.next_view()
        st, rs, radii, loss3, grads = self.view_gradients(cam, iteration)
        p = self.params()
        if self.world > 1:
            ex = self.exchange
            got = ex.exchange_support(p["xyz"].detach(), rs.campos, st.tensor("grad_support", (st.P,), torch.uint8))
            summed = dict(xyz=got["means3D"], opacity=got["opacities"], scaling=got["scales"], rotation=got["rotations"],
                          f_dc=got["shs_dc"], f_rest=got["shs_rest"])
        else:
            summed = dict(xyz=grads["xyz"], opacity=grads["opacity"], scaling=grads["scaling"], rotation=grads["rotation"],
                          f_dc=grads["features_dc"], f_rest=grads["features_rest"])
        if iteration < opt.densify_until_iter:
            self._densification_stats(grads["means2D"], radii)
        self.finish_step(iteration, p, summed)
        self.last = dict(P=st.P, D=st.D, view=getattr(cam, "image_name", None))
        return loss3

    @torch.no_grad()
    def finish_step(self, iteration: int, params_of_grads: Dict[str, torch.Tensor], summed: Dict[str, torch.Tensor]):
        """The tail of an iteration once the statistics are in (inpaint_rec.py:152-163): densify_and_prune / reset_opacity on their
        intervals, then the optimizer step. `params_of_grads`: the tensor objects `summed` was computed for."""
        g, opt = self.g, self.opt
        if iteration < opt.densify_until_iter:
            if iteration > opt.densify_from_iter and iteration % opt.densification_interval == 0:
                size_threshold = 20 if iteration > opt.opacity_reset_interval else None
                self._seed_densification(iteration)
                g.densify_and_prune(opt.densify_grad_threshold, 0.005, self.extent, size_threshold)
            if iteration % opt.opacity_reset_interval == 0 or (self.white_background and iteration == opt.densify_from_iter):
                g.reset_opacity()
        if iteration < opt.iterations:
            # the surgery re-creates parameters (cat_tensors_to_optimizer / _prune_optimizer / replace_tensor_to_optimizer,
            # gaussian_model.py:314-404): in the reference their .grad is None afterwards and optimizer.step() skips them, while a
            # tensor the surgery left alone (every one but the opacity at a bare reset_opacity) is stepped. Same here.
            for name, t in self.params().items():
                t.grad = summed[name].view_as(t) if t is params_of_grads[name] else None
            g.optimizer.step()
            g.optimizer.zero_grad(set_to_none=True)

    @torch.no_grad()
    def render(self, cam, background: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """Forward-only render of one camera on the stored parameters — what the reference's render() returns for the evaluation
        renders (inpaint_rec.py:68-69, :169-172 `render_set`; :208-229 `training_report`): no collective, any rank may call it."""
        from . import raster as R
        p = self.params()
        raw = [p[k].detach() for k in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")]
        rs = _settings(R, cam, self.background if background is None else background, self.g.active_sh_degree)
        image, radii, depth, _ = R.rasterize_forward_raw(rs, *raw)
        return {"render": image, "depth": depth, "radii": radii, "visibility_filter": radii > 0}

    def replicas_identical(self) -> bool:
        """Debug / test aid: every rank holds bit-identical parameters (one small all-gather of per-tensor checksums)."""
        if self.world == 1:
            return True
        vals = []
        for t in self.params().values():
            d = t.detach()
            vals += [d.double().sum(), d.double().abs().sum(), torch.tensor(float(d.shape[0]), dtype=torch.float64, device=d.device)]
        digest = torch.stack(vals)
        every = [torch.zeros_like(digest) for _ in range(self.world)]
        td.all_gather(every, digest, group=self.group)
        return all(torch.equal(every[0], e) for e in every)

    def train(self, first_iter: int, iterations: int, on_iteration: Optional[Callable] = None):
        """Iterations first_iter .. iterations (inclusive, 1-based like the reference's loop). on_iteration(iteration, loss3) runs on
        every rank after the step (rank 0 logs / saves there; loss3 is a device tensor — read it sparingly)."""
        for iteration in range(first_iter, iterations + 1):
            loss3 = self.step(iteration)
            if on_iteration is not None:
                on_iteration(iteration, loss3)


# ---------------------------------------------------------------------------------------------------------------------------------
# launcher: the reference's inpaint_rec.py, view-sharded

def save_png(path: str, image: torch.Tensor):
    """image [3 or 1, H, W] in [0, 1] -> 8-bit PNG (what torchvision.utils.save_image writes at inpaint_rec.py:256-259: clamp, x 255 +
    0.5, truncate), with the standard library only — torchvision is not part of the ROCm image this runs on."""
    import struct
    import zlib
    a = (image.detach().float().clamp(0, 1) * 255 + 0.5).to(torch.uint8).permute(1, 2, 0).contiguous().cpu().numpy()
    h, w, c = a.shape
    rows = b"".join(b"\x00" + a[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    with open(path, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2 if c == 3 else 0, 0, 0, 0))
                 + chunk(b"IDAT", zlib.compress(rows, 1)) + chunk(b"IEND", b""))


def render_set(trainer: "ViewShardedTrainer", model_path: str, iteration: int, views: Sequence) -> int:
    """inpaint_rec.py:244-259: renders of `views` and their ground truth under <model_path>/ours_<iteration>/{renders,gt}/%05d.png.
    Forward-only calls of path A; the launcher runs it on rank 0 alone."""
    rp, gp = (os.path.join(model_path, f"ours_{iteration}", d) for d in ("renders", "gt"))
    os.makedirs(rp, exist_ok=True)
    os.makedirs(gp, exist_ok=True)
    for idx, view in enumerate(views):
        out = trainer.render(view)
        save_png(os.path.join(rp, f"{idx:05d}.png"), out["render"])
        save_png(os.path.join(gp, f"{idx:05d}.png"), view.original_image[0:3])
    return len(views)


def training_report(trainer: "ViewShardedTrainer", iteration: int, scene, say=print) -> Dict[str, Dict[str, float]]:
    """The evaluation half of inpaint_rec.py:200-241 at a test iteration (the tensorboard half is out of scope): mean L1 and PSNR of
    the clamped render against the clamped image over the test cameras and over five training cameras (indices 5, 10, ..., 25,
    modulo the list). Returns {'test' | 'train': {'l1', 'psnr'}} for the sets that exist."""
    train = scene.getTrainCameras()
    sets = (("test", scene.getTestCameras()), ("train", [train[i % len(train)] for i in range(5, 30, 5)] if train else []))
    res = {}
    for name, cams in sets:
        if not cams:
            continue
        l1 = psnr = 0.0
        for cam in cams:
            img = trainer.render(cam)["render"].clamp(0.0, 1.0)
            gt = cam.original_image.to(img.device).clamp(0.0, 1.0)[0:3]
            l1 += float((img - gt).abs().mean())
            mse = ((img - gt) ** 2).reshape(img.shape[0], -1).mean(1, keepdim=True)          # utils/image_utils.py: per channel, then mean
            psnr += float((20 * torch.log10(1.0 / torch.sqrt(mse))).mean())
        res[name] = dict(l1=l1 / len(cams), psnr=psnr / len(cams))
        say(f"\n[ITER {iteration}] Evaluating {name}: L1 {res[name]['l1']} PSNR {res[name]['psnr']}")
    return res


def _pin_device():
    """Before ANY GPU call: make this process see exactly its GPU, so the reference's 87 hard-coded "cuda" / .cuda() land on it
    (gaussian_renderer/__init__.py:26, gaussian_model.py:126-152, general_utils.py:135 `cuda:0`)."""
    lr = os.environ.get("LOCAL_RANK")
    if lr is None or "MVI_TRAIN_VIEWS_NO_PIN" in os.environ:
        return
    # ONE device id, derived from whichever list the job was given (HIP's wins when both are set), written to both variables:
    # two independently derived values could name different GPUs
    # MVI_TRAIN_VIEWS_DEVICES: an explicit rank -> device list that the PARENT (torchrun) does not see as a visibility variable — torchrun
    # itself counts devices, and refuses a HIP_VISIBLE_DEVICES list longer than ROCR_VISIBLE_DEVICES (e.g. "0,0": two ranks on one GPU)
    vis = os.environ.get("MVI_TRAIN_VIEWS_DEVICES") or os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")
    if vis:                                                   # a restricted list: this rank's entry of it
        ids = [v for v in vis.split(",") if v != ""]
        if int(lr) >= len(ids):
            raise RuntimeError(f"train_views: LOCAL_RANK {lr} but only {len(ids)} visible device(s) ({vis}): two ranks would share a GPU")
        dev = ids[int(lr)]
    else:
        dev = lr
    os.environ["HIP_VISIBLE_DEVICES"] = os.environ["CUDA_VISIBLE_DEVICES"] = dev


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help"):
        print(__doc__)
        return 2
    _pin_device()
    gs_dir = os.path.abspath(argv[0])
    sys.path.insert(0, gs_dir)
    os.chdir(gs_dir)                                          # the reference reads bds/, inpaint/ relative to its directory
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    backend = os.environ.get("MVI_TRAIN_VIEWS_BACKEND", "nccl")
    if world > 1:
        td.init_process_group(backend, device_id=torch.device("cuda", 0) if backend == "nccl" else None)
    from multiview_inpaint_amd.dropin import patch_gs_simp
    done = patch_gs_simp.install()
    if rank == 0:
        print("[multiview_inpaint_amd.train_views] patched: " + ", ".join(done), file=sys.stderr)
    from argparse import ArgumentParser
    from arguments import ModelParams, OptimizationParams, PipelineParams     # the reference's own argument classes
    import numpy as np
    parser = ArgumentParser(description="inpaint_rec.py, view-sharded")
    lp, op, pp = ModelParams(parser), OptimizationParams(parser), PipelineParams(parser)
    parser.add_argument("--test_iterations", nargs="+", type=int, default=[7_000, 30_000])
    parser.add_argument("--save_iterations", nargs="+", type=int, default=[1_000, 7_000, 30_000])
    parser.add_argument("--checkpoint_iterations", nargs="+", type=int, default=[])
    parser.add_argument("--quiet", action="store_true")
    parser.add_argument("--n_mode", type=int, default=2)
    parser.add_argument("--scene_id", default=None, type=str)
    parser.add_argument("--ctrl_id", default="-1", type=str)
    parser.add_argument("--reduce", choices=["sum", "mean"], default="sum")
    parser.add_argument("--dry-run", dest="dry_run", type=int, default=0, metavar="N",
                        help="rehearsal on a new node: N iterations instead of --iterations, with one test and one save iteration at the end")
    args = parser.parse_args(argv[1:])
    if args.dry_run > 0:
        args.iterations = args.dry_run
        args.test_iterations, args.save_iterations = [args.dry_run], [args.dry_run]
    args.save_iterations.append(args.iterations)
    # general_utils.safe_state's seeding (utils/general_utils.py:132-134) without its `cuda:0` pin and stdout wrapper
    random.seed(0)
    np.random.seed(0)
    torch.manual_seed(0)
    torch.cuda.set_device(0)
    from scene import InpaintScene, InpaintGaussianModel
    dataset, opt, pipe = lp.extract(args), op.extract(args), pp.extract(args)
    if not dataset.model_path:
        dataset.model_path = os.path.join("./output_rec/", str(args.scene_id))
    if rank == 0:
        os.makedirs(dataset.model_path, exist_ok=True)
    if world > 1:
        td.barrier()
    gaussians = InpaintGaussianModel(dataset.sh_degree)
    scene = InpaintScene(dataset, gaussians)
    gaussians.training_setup(opt)                             # (patched: FusedAdam over the reference's param groups)
    bg = torch.tensor([1, 1, 1] if dataset.white_background else [0, 0, 0], dtype=torch.float32, device="cuda")
    ctrl = int(args.ctrl_id)
    out_render = os.path.join(scene.model_path, f"ctrl_{ctrl}" if ctrl >= 0 else str(args.n_mode))
    if rank == 0:
        os.makedirs(out_render, exist_ok=True)
    trainer = ViewShardedTrainer(gaussians, opt, lambda: scene.getInpaintTrainCameras(args.n_mode, args.ctrl_id), bg, scene.cameras_extent,
                                 reduce=args.reduce, white_background=dataset.white_background)
    ema = [0.0]
    # inpaint_rec.py:68-69: the inpaint cameras rendered once before the loop — forward-only, rank 0 (the others meet it at the
    # first exchange)
    if rank == 0:
        n0 = render_set(trainer, out_render, 0, scene.getInpaintCameras(args.n_mode, args.ctrl_id))
        if not args.quiet:
            print(f"[ITER 0] rendered {n0} inpaint cameras to {out_render}", flush=True)

    def on_iteration(iteration, loss3):
        if iteration % 10 == 0 or iteration in args.save_iterations:
            tot = loss3[:1].clone()
            if world > 1:
                td.all_reduce(tot)
            ema[0] = 0.4 * float(tot) / world + 0.6 * ema[0]
            if rank == 0 and not args.quiet and iteration % 100 == 0:
                print(f"[ITER {iteration}] loss (mean over {world} views, ema) {ema[0]:.7f}  P = {trainer.last.get('P')}", flush=True)
        if rank == 0 and iteration in args.save_iterations:
            scene.save(iteration, out_render)
        if rank == 0 and iteration in args.checkpoint_iterations:
            torch.save((gaussians.capture(), iteration), out_render + "/chkpnt" + str(iteration) + ".pth")
        if rank == 0 and iteration in args.test_iterations:
            # inpaint_rec.py:139-140 (training_report) and :169-172 (the validation renders: first and last two inpaint cameras + two
            # training cameras)
            training_report(trainer, iteration, scene)
            ic = scene.getInpaintCameras(args.n_mode, args.ctrl_id)
            render_set(trainer, out_render, iteration, list(ic[:2]) + list(ic[-2:]) + list(scene.getTrainCameras()[:2]))

    trainer.train(1, opt.iterations, on_iteration)
    if world > 1:
        ok = trainer.replicas_identical()
        td.barrier()
        td.destroy_process_group()
        if not ok:
            print("[multiview_inpaint_amd.train_views] the replicas diverged", file=sys.stderr)
            return 1
    if rank == 0:
        print("\nTraining complete.")
    return 0


if __name__ == "__main__":
    sys.exit(main())
