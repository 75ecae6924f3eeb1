"""Host logic of multiview_inpaint_amd.train_views (no GPU): the view stack every rank draws identically, the device pinning of the
launcher, and — over gloo with two ranks — the reduced densification statistics and the bit-identical densification decisions the
loop relies on. The GPU half (the loop itself at two ranks on one GPU) is tests/test_dist_gpu_ranks.py."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from multiview_inpaint_amd import train_views as TV  # noqa: E402


def test_view_stack_gives_every_rank_the_same_draws_and_its_own_view():
    cams = [f"cam{i}" for i in range(7)]                     # 7 views, 3 ranks: the stack refills in the middle of a step
    stacks = [TV.ViewStack(lambda: list(cams), 3, r, seed=5) for r in range(3)]
    seen = []
    for _ in range(10):
        lists = [s.next_views() for s in stacks]
        assert lists[0] == lists[1] == lists[2] and len(lists[0]) == 3
        seen += lists[0]
    # every view is used before any is repeated (pops without replacement, inpaint_rec.py:104-108), over whole refills
    assert sorted(seen[:7]) == sorted(cams) and sorted(seen[7:14]) == sorted(cams)
    a, b = TV.ViewStack(lambda: list(cams), 3, 0, seed=5), TV.ViewStack(lambda: list(cams), 3, 2, seed=5)
    for _ in range(5):
        va, vb = a.next_view(), b.next_view()
        assert va != vb                                      # different ranks, different views of the same step
    other = TV.ViewStack(lambda: list(cams), 3, 0, seed=6)
    assert [other.next_views() for _ in range(3)] != [TV.ViewStack(lambda: list(cams), 3, 0, seed=5).next_views() for _ in range(3)]


def test_launcher_pins_each_rank_to_its_own_gpu(monkeypatch):
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "MVI_TRAIN_VIEWS_NO_PIN", "MVI_TRAIN_VIEWS_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv("LOCAL_RANK", "3")
    TV._pin_device()
    assert os.environ["HIP_VISIBLE_DEVICES"] == "3" and os.environ["CUDA_VISIBLE_DEVICES"] == "3"
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "4,5,6,7")        # a restricted list: rank 3 takes its entry
    monkeypatch.setenv("CUDA_VISIBLE_DEVICES", "4,5,6,7")
    TV._pin_device()
    assert os.environ["HIP_VISIBLE_DEVICES"] == "7" and os.environ["CUDA_VISIBLE_DEVICES"] == "7"
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "4,5,6,7")        # only ONE of the two preset: both end on the same device
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES")
    TV._pin_device()
    assert os.environ["HIP_VISIBLE_DEVICES"] == "7" and os.environ["CUDA_VISIBLE_DEVICES"] == "7"
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "4,5")            # a list shorter than the local world: refuse, never share a GPU silently
    monkeypatch.setenv("CUDA_VISIBLE_DEVICES", "4,5")
    import pytest
    with pytest.raises(RuntimeError):
        TV._pin_device()
    monkeypatch.setenv("MVI_TRAIN_VIEWS_DEVICES", "2,2,2,2")    # an explicit rank -> device map wins (two ranks on one GPU: the tests)
    TV._pin_device()
    assert os.environ["HIP_VISIBLE_DEVICES"] == "2" and os.environ["CUDA_VISIBLE_DEVICES"] == "2"
    monkeypatch.delenv("MVI_TRAIN_VIEWS_DEVICES")
    monkeypatch.delenv("LOCAL_RANK")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1")
    TV._pin_device()
    assert os.environ["HIP_VISIBLE_DEVICES"] == "0,1"            # not under torchrun: nothing is touched


def test_trainer_rejects_an_unknown_reduction():
    import pytest
    with pytest.raises(ValueError):
        TV.ViewShardedTrainer(object(), object(), lambda: [], torch.zeros(3), 1.0, reduce="median", world=1)


def test_render_sets_and_report_of_the_launcher(tmp_path):
    """render_set (inpaint_rec.py:244-259) and the evaluation half of training_report (:200-241) on a fake trainer: file layout
    ours_<iteration>/{renders,gt}/%05d.png, 8-bit values as torchvision.utils.save_image writes them, L1 / PSNR means over the test
    cameras and over the five training cameras 5, 10, ..., 25 (modulo the list)."""
    import struct
    import zlib

    class Cam:
        def __init__(self, k):
            self.original_image = torch.full((3, 4, 6), 0.1 * k)
            self.k = k

    class Trainer:
        def render(self, cam, background=None):
            return {"render": cam.original_image + 0.05}

    class Scene:
        def getTrainCameras(self):
            return [Cam(k) for k in range(3)]

        def getTestCameras(self):
            return [Cam(5), Cam(6)]

    cams = [Cam(k) for k in range(3)]
    assert TV.render_set(Trainer(), str(tmp_path), 7, cams) == 3
    for sub in ("renders", "gt"):
        assert sorted(os.listdir(tmp_path / "ours_7" / sub)) == ["00000.png", "00001.png", "00002.png"]
    raw = (tmp_path / "ours_7" / "renders" / "00002.png").read_bytes()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n" and struct.unpack(">II", raw[16:24]) == (6, 4)
    idat = raw[raw.index(b"IDAT") + 4:raw.index(b"IEND") - 8]
    px = zlib.decompress(idat)
    assert len(px) == 4 * (1 + 6 * 3) and px[1] == int(0.25 * 255 + 0.5)          # filter byte, then the first pixel's red
    said = []
    res = TV.training_report(Trainer(), 7, Scene(), say=said.append)
    assert set(res) == {"test", "train"} and abs(res["test"]["l1"] - 0.05) < 1e-6
    assert abs(res["test"]["psnr"] - 20 * torch.log10(torch.tensor(1 / 0.05)).item()) < 1e-3
    assert len(said) == 2 and "[ITER 7] Evaluating test: L1" in said[0] and "Evaluating train" in said[1]
