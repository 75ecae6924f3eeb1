"""Is the training iteration bound by the host (Python + launches) or by the device? For each variant of bench_train: wall time per
iteration with the queue kept full (sync at the end only) and the host's own enqueue time per iteration (perf_counter around the loop,
before the final synchronize). host ~= wall -> the host is the limit."""
import sys, os, time, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multiview_inpaint_amd import bench_train as B

orig_event = torch.cuda.Event
res = []
for v in sys.argv[1:] or ["hip_raw", "patched"]:
    t = {}
    real_sync = torch.cuda.synchronize
    marks = []

    def sync(*a, **k):
        marks.append(time.perf_counter())
        real_sync(*a, **k)
        marks.append(time.perf_counter())
    torch.cuda.synchronize = sync
    try:
        out = B.run(v, 40, 5)
    finally:
        torch.cuda.synchronize = real_sync
    # marks: [before sync after warmup, after it, before final sync, after it]
    host = (marks[2] - marks[1]) / 40 * 1e3
    wall = (marks[3] - marks[1]) / 40 * 1e3
    res.append(dict(variant=v, event_ms=out["ms_per_iteration"], host_enqueue_ms=round(host, 3), wall_ms=round(wall, 3)))
print(json.dumps(res))
