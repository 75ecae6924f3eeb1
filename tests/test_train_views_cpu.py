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
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "MVI_TRAIN_VIEWS_NO_PIN"):
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
    monkeypatch.delenv("LOCAL_RANK")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1")
    TV._pin_device()
    assert os.environ["HIP_VISIBLE_DEVICES"] == "0,1"            # not under torchrun: nothing is touched


def test_trainer_rejects_an_unknown_reduction():
    import pytest
    with pytest.raises(ValueError):
        TV.ViewShardedTrainer(object(), object(), lambda: [], torch.zeros(3), 1.0, reduce="median", world=1)
