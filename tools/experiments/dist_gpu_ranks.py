"""The compacted gradient exchange's DEVICE path with more than one rank, on a ONE-GPU box: W ranks on cuda:0 over gloo (the work is
tests/dist_gpu_worker.py, which the GPU suite runs with 2 ranks). The parent never touches the GPU: the workers are started before
anything initialises it. Usage (through gpurun): python tools/experiments/dist_gpu_ranks.py [W]  -> one line per step and rank + verdict"""
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(rank, world, port):
    import dist_gpu_worker as W
    ok, _ = W.run_rank(rank, world, port, say=lambda s: print(s, flush=True))
    if rank == 0:
        print("VERDICT", "ok" if ok else "FAILED", f"({world} ranks on one GPU over gloo, device path of CompactedGradExchange)", flush=True)


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    import torch.multiprocessing as mp                           # (the parent imports torch but never initialises the GPU)
    mp.spawn(worker, args=(world, port), nprocs=world, join=True)
