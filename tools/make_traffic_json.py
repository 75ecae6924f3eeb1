"""profiles/raster_traffic.json from the PMC passes of tools/profile_raster.sh.
Usage: python tools/make_traffic_json.py gpurun_out/<tag> <tag>   (reads pmc_fetch.txt / pmc_write.txt / pmc_sq.txt / pmc_clk.txt)
The file records the digest of the rasterizer sources it was measured on (multiview_inpaint_amd/_lib.py:
raster_source_digest); bench.py prints the counters only while that digest matches the build it runs.
Correction (MI355X_MICROARCH.md, HBM / rocprofv3): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts
a 128-B request as 64 B for 16-B/lane reads -> read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 as is."""
import json
import os
import re
import sys

d, tag = sys.argv[1], sys.argv[2]
STAGE = {"preprocess_forward_kernel": "preprocess_forward", "total_block_sums_kernel": "scan_block_sums",
         "scan_block_sums_kernel": "duplicate_keys", "depth_keys_kernel": "radix_sort", "radix_count_kernel": "radix_sort", "radix_scan_rows_kernel": "radix_sort",
         "radix_scatter_kernel": "radix_sort", "perm_block_sums_kernel": "duplicate_keys", "emit_pairs_kernel": "duplicate_keys",
         "tile_ranges_kernel": "tile_ranges", "render_forward_kernel": "render_forward",
         "render_backward_kernel": "render_backward", "preprocess_backward_kernel": "preprocess_backward",
         # binning version 2 (csrc/raster_binning2.hip) and the sparse backward, under the stage names of the bench line
         "totals2_kernel": "scan_block_sums", "depth_count_kernel": "radix_sort", "depth_scatter_kernel": "radix_sort",
         "column_count_kernel": "duplicate_keys", "columns_scan_kernel": "duplicate_keys",
         "expand_scatter_kernel<1, 128>": "duplicate_keys", "expand_scatter_kernel<1, 256>": "duplicate_keys",
         "row_count_kernel": "radix_sort", "row_scan_kernel": "radix_sort",
         "expand_scatter_kernel<2, 128>": "radix_sort", "expand_scatter_kernel<2, 256>": "radix_sort",
         "compact_touched_kernel": "preprocess_backward", "preprocess_backward_sparse_kernel": "preprocess_backward",
         "zero_regions_kernel": "render_backward",
         # deferred SH colours (round 4): evaluated ahead of / inside the render kernel, part of its stage
         "mark_front_kernel": "render_forward", "resolve_marked_kernel": "render_forward", "resolve_colors_kernel": "render_forward"}
STEPS = None       # launches of a once-per-step kernel in the profiled run (bench.py --steps 5 --warmup 1: 1 warm-up + 5
                   # instrumented + 5 timed = 11), read from the render_backward_kernel entry


def parse(path):
    out, k = {}, None
    for line in open(path):
        if line.startswith(("mvi::", "void mvi::")):
            k = line.strip().split("::")[1]
            out[k] = {}
        else:
            m = re.match(r"\s+(\S+)\s+([\d.]+)\s+\(n=(\d+)\)", line)
            if m and k:
                out[k][m.group(1)] = (float(m.group(2)), int(m.group(3)))
    return out


f, w = parse(os.path.join(d, "pmc_fetch.txt")), parse(os.path.join(d, "pmc_write.txt"))
STEPS = f["render_backward_kernel"]["FETCH_SIZE"][1]
per_kernel, per_stage = {}, {}
for k in sorted(f):
    if k not in STAGE:
        # templated kernels appear as name<...>
        base = k.split("<")[0]
        STAGE[k] = STAGE.get(base, "other")
    fv, n = f[k]["FETCH_SIZE"]
    wv = w[k]["WRITE_SIZE"][0]
    lps = n / STEPS
    fb, wb = fv * 1024 * lps, wv * 1024 * lps
    corr = int(2 * fb + wb)
    per_kernel[k] = dict(launches_per_step=lps, FETCH_SIZE_bytes=int(fb), WRITE_SIZE_bytes=int(wb), hbm_bytes_corrected=corr)
    per_stage[STAGE[k]] = per_stage.get(STAGE[k], 0) + corr
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from multiview_inpaint_amd import _lib as _mvi_lib  # noqa: E402

# VALU issue: SQ_INSTS_VALU wave-instructions x 4 cycles each against SIMDs x cycles (GRBM_GUI_ACTIVE summed over the 8 XCDs
# / 8 = the dispatch's cycles), per launch — the roof of the two render kernels, whose HBM traffic is far below the
# algorithmic bytes
issue = {}
sq_path, clk_path = os.path.join(d, "pmc_sq.txt"), os.path.join(d, "pmc_clk.txt")
if os.path.exists(sq_path) and os.path.exists(clk_path):
    sq, clk = parse(sq_path), parse(clk_path)
    for k in sorted(sq):
        if k in clk and "SQ_INSTS_VALU" in sq[k] and "GRBM_GUI_ACTIVE" in clk[k]:
            cycles = clk[k]["GRBM_GUI_ACTIVE"][0] / 8.0
            valu = sq[k]["SQ_INSTS_VALU"][0]
            issue[k] = dict(SQ_INSTS_VALU=valu, cycles=cycles, SQ_WAVE_CYCLES=sq[k].get("SQ_WAVE_CYCLES", (0, 0))[0],
                            valu_issue_frac=round(valu * 4.0 / (1024.0 * cycles), 4))

out = {"build": _mvi_lib.raster_source_digest(),
       "issue_per_kernel": issue,
       "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum (separate passes), python3 bench.py --steps 5 "
                 f"--warmup 1 --path raster --no-cpu-baseline, MI355X (tools/profile_raster.sh {tag})",
       "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B for 16-B/lane reads -> read bytes = 2 x FETCH_SIZE x 1024 "
                     "(MI355X_MICROARCH.md, HBM); WRITE_SIZE x 1024 as is (includes the 64-B float-atomic requests)",
       "per_step_bytes_by_stage": per_stage, "per_kernel": per_kernel,
       "render_backward_atomic_requests_64B": w["render_backward_kernel"]["TCC_EA0_ATOMIC_sum"][0]}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "raster_traffic.json"), "w"), indent=1)
print(json.dumps(per_stage, indent=1))
