"""SURVEY.md §8e on the one GPU the suite gets: TWO ranks of the view-sharded step on cuda:0 over gloo — the device path of
dist.CompactedGradExchange (union scan, windowed gather / scatter kernels, paging) with world_size > 1. The ranks are forked by a
forkserver that tests/conftest.py starts before anything in the pytest process can initialise the GPU (a process that has may not
start programs on this pool); the work and its checks are tests/dist_gpu_worker.py."""
import socket

import pytest

pytestmark = pytest.mark.gpu


def _run_ranks(rank_launcher, target, world, kwargs):
    if rank_launcher is None:
        pytest.skip("no forkserver was started for this session (no GPU visible at configure time)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    q = rank_launcher.Queue()
    procs = [rank_launcher.Process(target=target, args=(r, world, port, q, kwargs), daemon=True) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    try:
        for _ in range(world):
            rank, ok, pages, lines = q.get(timeout=300)
            results[rank] = (ok, pages, lines)
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.terminate()                                    # exactly the processes started here
    return results


def test_two_ranks_on_one_gpu_exchange_equals_the_plain_sum_of_their_views(rank_launcher):
    import dist_gpu_worker as W
    world = 2
    results = _run_ranks(rank_launcher, W.entry, world, dict(N=60_001, W=480, H=304, forced=(None, None, 2000, None)))
    for r in range(world):
        ok, pages, lines = results[r]
        print("\n".join(lines))
        assert ok, lines[-3:]
        assert len(pages) == 4 and pages[2] > 3 and pages[0] == 1, pages      # the forced capacity paged; the first step did not


def test_two_rank_view_sharded_training_equals_one_process_over_both_views(rank_launcher):
    """BASELINE configs[4] (per-GPU rasterize + exchange of the Gaussian gradients) at two ranks THROUGH THE SHIPPED MODULE
    (multiview_inpaint_amd.train_views.ViewShardedTrainer): fourteen iterations of the loop body of inpaint_rec.py:96-163 — own view of
    the step's two, fused L1 + DSSIM (masked for the non-inpainted views), backward on the stored parameters into the exchange's
    buffers, bit-mask all-gather + compacted exchange, reduced densification statistics, densify_and_prune twice (identically seeded
    torch.normal, surgery through patch_gs_simp's hooks), FusedAdam — against one process that renders both views of every step with
    the same pieces: same loss curve up to the first change of P (1e-4), close after it, same sizes, replicas bit-identical at the end."""
    import dist_gpu_worker as W
    world = 2
    results = _run_ranks(rank_launcher, W.entry_training, world, dict())
    for r in range(world):
        ok, _, lines = results[r]
        print("\n".join(lines))
        assert ok, lines[-3:]


@pytest.mark.parametrize("reduce", ["sum", "mean"])
def test_launcher_dry_run_under_torchrun_with_two_ranks(rank_launcher, reduce):
    """`torchrun --nproc-per-node 2 -m multiview_inpaint_amd.train_views <gs-simp dir> ... --dry-run 12` with MVI_TRAIN_VIEWS_BACKEND=gloo
    on the stand-in gs-simp directory (tests/gs_simp_standin): main() itself — the reference's argument groups on one parser, each
    rank pinned to its entry of HIP_VISIBLE_DEVICES before anything touches a GPU, patch_gs_simp.install(), InpaintScene /
    InpaintGaussianModel / training_setup, the inpaint cameras rendered before the loop on rank 0 (inpaint_rec.py:68-69), twelve
    view-sharded iterations incl. two densifications, training_report + the validation renders + save + checkpoint at the test / save
    iteration on rank 0 (:139-142, :165-172), replicas_identical, exit code 0."""
    import dist_gpu_worker as W
    results = _run_ranks(rank_launcher, W.entry_launcher, 1, dict(iterations=12, reduce=reduce))
    ok, _, lines = results[0]
    print("\n".join(lines))
    assert ok, lines[-2:]
