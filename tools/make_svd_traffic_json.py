"""profiles/svd_traffic.json from tools/pmc_svd_traffic.sh's passes: HBM bytes per launch of the attention kernel and of the
implicit-GEMM convolution at their largest shapes of the 14 x 576x1024 step, next to the bytes those launches must move.
Usage: python tools/make_svd_traffic_json.py gpurun_out/<tag> <tag>
Correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE tallies the 128-B requests of
16-B-per-lane reads (plain loads and LDS-DMA alike — both kernels read nothing narrower from HBM) at 64 B: read bytes = 2 x
FETCH_SIZE x 1024; WRITE_SIZE x 1024 as is. Shapes that fit the 256 MiB Infinity Cache under-report re-reads served on-die."""
import json
import os
import re
import sys

d, tag = sys.argv[1], sys.argv[2]
txt = open(os.path.join(d, "pmc_svd_traffic.txt")).read()


def counters(section):
    out = {}
    for m in re.finditer(r"^\s+(\S+)\s+([\d.]+)\s+\(n=(\d+)\)", section, re.M):
        out[m.group(1)] = float(m.group(2))
    return out


att, conv = txt.split("== implicit-GEMM convolution")[0], txt.split("== implicit-GEMM convolution")[1]
res = {}
for name, sec, algo, what in (
        # q, k, v read once + out written once: 4 x 28 x 9216 x 320 x 2 B (K/V re-reads by the 36 query blocks of a head are L2 / MALL hits by design)
        ("attention", att, 4 * 28 * 9216 * 320 * 2, "B 28, H 5, S 9216, D 64 bf16: q, k, v read once, out written once"),
        # x [28, 72, 128, 640] read once, W [320][9 x 640] read once, out [28, 72, 128, 320] written once
        ("conv3x3_n320", conv, 28 * 72 * 128 * (640 + 320) * 2 + 320 * 9 * 640 * 2, "28 x 72x128, 640 -> 320 bf16: x and W read once, out written once")):
    c = counters(sec)
    rd, wr = 2.0 * c["FETCH_SIZE"] * 1024.0, c["WRITE_SIZE"] * 1024.0
    hit = c.get("TCC_HIT_sum", 0.0) / max(c.get("TCC_HIT_sum", 0.0) + c.get("TCC_MISS_sum", 0.0), 1.0)
    res[name] = dict(shape=what, algorithmic_bytes=int(algo), FETCH_SIZE_KiB=c["FETCH_SIZE"], WRITE_SIZE_KiB=c["WRITE_SIZE"],
                     hbm_read_bytes_corrected=int(rd), hbm_write_bytes=int(wr), hbm_bytes_corrected=int(rd + wr),
                     ratio_to_algorithmic=round((rd + wr) / algo, 3), l2_hit_rate=round(hit, 4))
out = {"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum, separate passes, MI355X (tools/pmc_svd_traffic.sh {tag})",
       "correction": "gfx950: read bytes = 2 x FETCH_SIZE x 1024 for 16-B-per-lane reads; WRITE_SIZE x 1024 as is (MI355X_MICROARCH.md, HBM)",
       "kernels": res}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "svd_traffic.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(res, indent=1))
