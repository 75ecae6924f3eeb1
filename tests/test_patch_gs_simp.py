"""multiview_inpaint_amd.dropin.patch_gs_simp on a stand-in of the gs-simp module layout (utils/loss_utils.py, scene/gaussian_model.py,
a training script that imports the loss functions by name — the shapes of gs-simp/train.py:17 and gaussian_model.py:149-167):
the runner patches the modules before the script's imports run. CPU only: nothing is stepped, the test is about WHO gets called."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_runner_patches_loss_functions_and_optimizer_before_the_script_imports_them(tmp_path):
    (tmp_path / "utils").mkdir()
    (tmp_path / "scene").mkdir()
    (tmp_path / "utils" / "__init__.py").write_text("")
    (tmp_path / "scene" / "__init__.py").write_text("")
    (tmp_path / "utils" / "loss_utils.py").write_text(textwrap.dedent("""
        def l1_loss(network_output, gt):
            return "stand-in l1"
        def ssim(img1, img2, window_size=11, size_average=True):
            return "stand-in ssim"
    """))
    (tmp_path / "scene" / "gaussian_model.py").write_text(textwrap.dedent("""
        import torch
        class GaussianModel:
            def training_setup(self, training_args):
                self._xyz = torch.nn.Parameter(torch.zeros(4, 3))
                self._opacity = torch.nn.Parameter(torch.zeros(4, 1))
                l = [{'params': [self._xyz], 'lr': 0.5 * training_args, "name": "xyz"},
                     {'params': [self._opacity], 'lr': 0.05, "name": "opacity"}]
                self.optimizer = torch.optim.Adam(l, lr=0.0, eps=1e-15)
                return "set up"
            def add_densification_stats(self, viewspace_point_tensor, update_filter):
                return "stand-in stats"
            def prune_points(self, mask):
                return "stand-in prune"
        class Untouched:
            pass
    """))
    (tmp_path / "gaussian_renderer").mkdir()
    (tmp_path / "gaussian_renderer" / "__init__.py").write_text(textwrap.dedent("""
        def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):
            return ("stand-in render", scaling_modifier, override_color)
    """))
    (tmp_path / "train_like.py").write_text(textwrap.dedent("""
        import sys
        from utils.loss_utils import l1_loss, ssim
        from scene.gaussian_model import GaussianModel
        from gaussian_renderer import render
        import gaussian_renderer as gr
        import utils.loss_utils as lu
        g = GaussianModel()
        assert g.training_setup(2.0) == "set up"
        o = g.optimizer
        print("ARGV", sys.argv[1:])
        print("LOSS", l1_loss.__module__, ssim.__module__, lu._reference_l1_loss(0, 0), lu._reference_ssim(0, 0))
        # a call the fused form cannot serve (override_color) reaches the script's own render with its arguments
        print("RENDER", getattr(render, "_mvi_patched", False), render.__name__, gr._reference_render is not render,
              render(None, g, None, None, 0.5, "colours"))
        print("STATS", getattr(GaussianModel.add_densification_stats, "_mvi_patched", False), g.add_densification_stats(None, [1, 2]))
        print("PRUNE", getattr(GaussianModel.prune_points, "_mvi_patched", False), g.prune_points([True, False]))
        print("OPT", type(o).__module__, type(o).__name__, [(pg["name"], pg["lr"], pg["eps"], pg["betas"]) for pg in o.param_groups],
              o.param_groups[0]["params"][0] is g._xyz)
    """))
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable, "-m", "multiview_inpaint_amd.dropin.patch_gs_simp", str(tmp_path / "train_like.py"), "-s", "scene_dir"],
                       cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    out = dict(line.split(" ", 1) for line in p.stdout.strip().splitlines())
    assert out["ARGV"] == "['-s', 'scene_dir']"
    assert out["LOSS"] == "multiview_inpaint_amd.train_ops multiview_inpaint_amd.train_ops stand-in l1 stand-in ssim"
    assert out["OPT"] == ("multiview_inpaint_amd.train_ops FusedAdam [('xyz', 1.0, 1e-15, (0.9, 0.999)), "
                          "('opacity', 0.05, 1e-15, (0.9, 0.999))] True")
    assert out["RENDER"] == "True render True ('stand-in render', 0.5, 'colours')"
    assert "scene.gaussian_model.GaussianModel.training_setup" in p.stderr and "Untouched" not in p.stderr
    assert "gaussian_renderer.render" in p.stderr
    assert out["STATS"] == "True stand-in stats"               # (a filter the mask-free form does not serve: the script's own method ran)
    assert "scene.gaussian_model.GaussianModel.add_densification_stats" in p.stderr
    assert out["PRUNE"] == "True stand-in prune" and "scene.gaussian_model.GaussianModel.prune_points" in p.stderr
