// Per-tile alpha compositing with median depth, forward and backward, for gfx950.
// Replaces the plug-in's render stages behind gs-simp/gaussian_renderer/__init__.py:85-93
// (forward: colour [3,H,W] + depth [1,H,W]) and loss.backward() (gs-simp/train.py:93).
//
// One block per 16x16-pixel tile. Forward: 256 threads = 4 wave64, each wave an 8x8-pixel quadrant, one pixel per
// lane. Backward: 128 threads = 2 wave64, each wave a 16x8 half tile, two pixels per lane on packed fp32 math.
// The tile's depth-sorted list is staged through LDS one entry per thread and round by the whole block. Each wave
// then tests 64 staged entries at a time against its own pixel box (one lane per entry): an entry is kept iff its
// "active ellipse" {alpha >= 1/255} = {d^T conic d <= 2 ln(255 o)} can reach the box (exact minimum of the
// quadratic form over the box, quad_overlap). Only kept entries are evaluated per pixel. The test is conservative
// (margin on 2 ln(255 o)), so results are identical to evaluating every entry.
#include <cstdlib>

#include "raster_common.h"

namespace mvi {

// ballot of a predicate that already lives in a lane mask: no 0/1 materialisation (__ballot(int) costs two VALU issues)
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }


// Per staged entry: t2 = 2 ln(255 o) (+ margin), the largest Mahalanobis distance^2 d^T conic d at which
// alpha = o exp(-d^T conic d / 2) still reaches 1/255; negative = can never be active.
__device__ __forceinline__ float active_t2(float4 co) {
    float t2 = 2.0f * __logf(255.0f * co.w);
    return t2 > 0.0f ? t2 * 1.001f + 0.01f : -1.0f;
}

// Exact conservative test "can any point of the pixel quadrant [qx0,qx0+7] x [qy0,qy0+7] be active?":
// the minimum of the convex form f(d) = A dx^2 + 2 B dx dy + C dy^2 over the box is 0 if the centre
// is inside; otherwise it lies on one of the (at most two) box edges facing the centre, where the
// 1-D minimiser is a clamped closed form. Continuous minimum <= minimum over pixel centres, and t2
// carries a margin that dwarfs the rcp / rounding error, so no active pair is ever dropped.
__device__ __forceinline__ bool quad_overlap(float2 c, float4 co, float t2, float qx0, float qy0, float bw = 7.0f,
                                             float bh = 7.0f) {
    const float dxl = qx0 - c.x, dxr = dxl + bw, dyl = qy0 - c.y, dyr = dyl + bh;
    const float px = fminf(fmaxf(0.0f, dxl), dxr), py = fminf(fmaxf(0.0f, dyl), dyr);   // box point nearest the centre, per axis
    // edge x = px: dy* = clamp(-B px / C); edge y = py: dx* = clamp(-B py / A)
    const float dy1 = fminf(fmaxf(-co.y * px * __builtin_amdgcn_rcpf(co.z), dyl), dyr);
    const float dx2 = fminf(fmaxf(-co.y * py * __builtin_amdgcn_rcpf(co.x), dxl), dxr);
    const float f1 = co.x * px * px + 2.0f * co.y * px * dy1 + co.z * dy1 * dy1;
    const float f2 = co.x * dx2 * dx2 + 2.0f * co.y * dx2 * py + co.z * py * py;
    // px == 0: only the y-facing edge matters (f2); py == 0: only f1; both 0: centre inside, f = 0
    const float f = px == 0.0f ? (py == 0.0f ? 0.0f : f2) : (py == 0.0f ? f1 : fminf(f1, f2));
    return f <= t2;
}

// Workgroups are handed to the 8 XCDs round-robin by linear id, and every XCD has its own L2. Tile t of a launch of
// `tiles` workgroups is chosen so that XCD x works through ONE contiguous band of tiles (rows of the image): the ~11
// tiles a Gaussian touches then gather its 40 bytes through the same L2 (and their gradient atomics meet there)
// instead of through up to 8 of them.
__device__ __forceinline__ int xcd_band_tile(int bid, int tiles) {
    const int q = tiles >> 3, r = tiles & 7, x = bid & 7, i = bid >> 3;
    return x * q + (x < r ? x : r) + i;
}

typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat(float v) { return f2{v, v}; }
constexpr int kB2 = 128;                // threads per tile in the backward kernel: 2 wave64, two pixels per lane

// power = -0.5 (A dx^2 + C dy^2) - B dx dy for the lane's two pixels (same x, adjacent y), ONE rounding recipe shared by
// the forward and the backward kernel so the backward replays the forward's alpha / threshold decisions exactly:
// products rounded, fma(dx, A dx, (C dy) dy), then fma(., -0.5, -(B dx) dy).
__device__ __forceinline__ f2 gauss_power(float4 co, float dx, f2 dy) {
#pragma clang fp contract(off)
    const float ax = co.x * dx, bx = co.y * dx;
    const f2 cyy = (co.z * dy) * dy;
    const f2 sq = pk_fma(splat(dx), splat(ax), cyy);
    const f2 bxy = bx * dy;
    return pk_fma(sq, splat(-0.5f), -bxy);
}

__device__ __forceinline__ float gauss_power1(float4 co, float dx, float dy) {     // the same recipe, one pixel
#pragma clang fp contract(off)
    const float ax = co.x * dx, bx = co.y * dx;
    const float cyy = (co.z * dy) * dy;
    const float sq = __builtin_fmaf(dx, ax, cyy);
    const float bxy = bx * dy;
    return __builtin_fmaf(sq, -0.5f, -bxy);
}

// The CU's scalar unit issues one instruction per cycle for all four SIMDs, so the inner loop must
// stay light on scalar work: the kept entries of a 64-entry chunk are compacted into a per-wave LDS
// strip (lane-parallel copies at popcount positions) and walked by a plain counted loop, and a
// pixel that has saturated is represented by T = 0 (its committed transmittance lives in T_out), so
// the body needs no per-lane "done" mask and no divergent branch.
__global__ __launch_bounds__(kBlock) void render_forward_kernel(
    Frame f, const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float2* __restrict__ xy, float4* rgbd, const float4* __restrict__ conic_opacity,
    float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, float* __restrict__ out_color,
    float* __restrict__ out_depth, ZeroRegions zero, const ColorSource* __restrict__ color_src, uint8_t* clamped) {
    __shared__ float2 s_xy[kBlock];
    __shared__ float s_t2[kBlock];
    __shared__ float4 s_co[kBlock];
    __shared__ float4 s_cd[kBlock];                 // r, g, b, depth
    // per-wave compacted strip: [0] (x, y, list position, -), [1] conic + opacity, [2] colour + depth — ONE array, so that the
    // three reads of an entry share one address register (constant offsets) instead of costing an address add each
    __shared__ float4 w_s[4][3][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // the backward's accumulation rows (and the touched flags) are zeroed HERE: this kernel is issue-bound with the memory
    // pipes nearly idle, so the 65 bytes per Gaussian ride along for free instead of costing a 96 MB fill pass in front of
    // the render backward
    zero_share(zero, tid, kBlock);
    const int tile = xcd_band_tile(blockIdx.x, f.gx * f.gy);
    const int tile_y = tile / f.gx, tile_x = tile - tile_y * f.gx;
    const int qx0 = tile_x * kTile + 8 * (wave & 1), qy0 = tile_y * kTile + 8 * (wave >> 1);
    const int pxi = qx0 + (lane & 7), pyi = qy0 + (lane >> 3);
    const bool inside = pxi < f.W && pyi < f.H;
    const float pfx = (float)pxi, pfy = (float)pyi;
    const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    int todo = (int)(r1 - r0);
    const int rounds = (todo + kBlock - 1) / kBlock;

    float T = inside ? 1.0f : 0.0f;                 // 0 = this pixel takes no further contributions
    float T_out = 1.0f, C2 = 0.f, Dp = kDepthSentinel, Tc = 1.0f;
    f2 C01 = {0.f, 0.f};                            // red, green: one packed fma on the (aligned) first two registers of the colour
    uint32_t last = 0;

    // Deferred SH colours (raster_common.h, ColorSource): rgbd holds (-1, -1, -1, depth) until somebody needs the colour.
    // The lane that stages an entry evaluates it (the same arithmetic as the eager preprocess), writes it back for the other
    // tiles and for the backward, and stages its OWN value: whatever another block wrote meanwhile is either complete
    // (every channel >= 0) or treated as pending.
    const ColorSource cs = *color_src;              // uniform: scalar loads
    // software pipeline: the gathers of batch r+1 are issued before batch r is composited
    float2 n_xy = make_float2(0.f, 0.f);
    float4 n_co = make_float4(0.f, 0.f, 0.f, 0.f), n_cd = n_co;
    uint32_t n_id = 0;
    auto fetch = [&](int r) {
        const uint32_t q = r0 + (uint32_t)(r * kBlock + tid);
        if (q < r1) {
            n_id = point_list[q];
            n_xy = xy[n_id]; n_co = conic_opacity[n_id]; n_cd = rgbd[n_id];
        }
    };
    if (rounds > 0) fetch(0);
    for (int r = 0; r < rounds; ++r, todo -= kBlock) {
        if (__syncthreads_count(T <= 0.0f) == kBlock) break;
        if (cs.deferred && tid < todo && color_pending(n_cd)) n_cd = resolve_color_small(cs, f.campos, n_id, n_cd.w, rgbd, clamped);
        s_xy[tid] = n_xy;
        s_co[tid] = n_co;
        s_t2[tid] = active_t2(n_co);
        s_cd[tid] = n_cd;
        __syncthreads();
        if (r + 1 < rounds) fetch(r + 1);
        const int n = todo < kBlock ? todo : kBlock;
        for (int c = 0; c < n; c += 64) {
            if (ballot64(T > 0.0f) == 0ull) break;
            const int e = c + lane;
            const bool keep = e < n && quad_overlap(s_xy[e], s_co[e], s_t2[e], (float)qx0, (float)qy0);
            const unsigned long long mask = ballot64(keep);
            if (mask == 0ull) continue;
            if (keep) {
                const int pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                const float2 q = s_xy[e];
                w_s[wave][0][pos] = make_float4(q.x, q.y, __uint_as_float((uint32_t)(r * kBlock + e + 1)), 0.f);
                w_s[wave][1][pos] = s_co[e];
                w_s[wave][2][pos] = s_cd[e];
            }
            const int cnt = __popcll(mask);
            float4 a = w_s[wave][0][0], co = w_s[wave][1][0], cd = w_s[wave][2][0];
            for (int k = 0; k < cnt; ++k) {
                const int kn = k + 1 < 64 ? k + 1 : 63;
                const float4* nx = &w_s[wave][0][kn];
                const float4 an = nx[0], con = nx[64], cdn = nx[128];   // next entry's reads overlap the math
                float dx = a.x - pfx, dy = a.y - pfy;
                float power = gauss_power1(co, dx, dy);
                float alpha = fminf(kAlphaMax, co.w * __expf(power));
                alpha = power <= 0.0f ? alpha : 0.0f;
                const float test_T = T * (1.0f - alpha);                 // 0 for a saturated pixel
                const bool valid = alpha >= kAlphaMin;
                const bool contrib = valid && test_T >= kTEps;
                const float w = contrib ? alpha * T : 0.0f;
                C01 = pk_fma(f2{cd.x, cd.y}, splat(w), C01);
                C2 = __builtin_fmaf(cd.z, w, C2);
                // median depth of the w-depth fork = depth of the entry that takes T from above 0.5 to below it (T > 0.5 and
                // T (1 - alpha) < 0.5, both strict: oracle/raster_oracle.c orc_render_forward). Every contributing entry met
                // while T > 0.5 overwrites the candidate and the T it leaves behind (two selects on one condition): the last
                // of them is the crossing entry iff it left T below 0.5 — checked once, after the loop. An entry that leaves
                // T at exactly 0.5 is not a crossing, and nothing behind it starts from above 0.5: the sentinel stays.
                const bool above = contrib && T > 0.5f;
                Dp = above ? cd.w : Dp;
                Tc = above ? test_T : Tc;
                T_out = contrib ? test_T : T_out;
                last = contrib ? __float_as_uint(a.z) : last;           // 1-based position in the tile's list
                T = contrib ? test_T : (valid ? 0.0f : T);              // valid but below 1e-4: saturated from here on
                a = an; co = con; cd = cdn;
            }
        }
    }
    if (inside) {
        size_t pix = (size_t)pyi * f.W + pxi, hw = (size_t)f.H * f.W;
        final_T[pix] = T_out;
        n_contrib[pix] = last;
        out_color[pix] = C01.x + T_out * f.bg[0];
        out_color[hw + pix] = C01.y + T_out * f.bg[1];
        out_color[2 * hw + pix] = C2 + T_out * f.bg[2];
        out_depth[pix] = Tc < 0.5f ? Dp : kDepthSentinel;             // never crossed 0.5: the sentinel (gs-simp/gen_seq.py:50)
    }
}

// Deferred SH colours, ahead of the render kernel: the leading `n_front` entries of every tile's list are entries the render
// kernel is going to stage for certain (its first round), and neighbouring tiles share most of them (2.1 M leading entries of
// the bench view are 45 k distinct Gaussians): left to the render kernel, every one of those tiles finds the colour pending at
// the same moment and evaluates it itself (measured: +140 us). Two small kernels instead: mark_front sets a byte per Gaussian
// named by a leading entry (plain idempotent stores: no atomics, no reads), resolve_marked walks the P flags, collects the
// marked Gaussians of 1024 in LDS and evaluates them with full waves. (Claiming with a device-scope compare-and-swap in one
// kernel was measured at 53 us: ~2 M memory-side atomics, the claims invisible through the other XCDs' L2.) What the render
// kernel meets later is either a colour or, beyond the leading entries, the rare Gaussian nobody needed yet.
constexpr int kFrontTiles = 4;
constexpr int kFrontMax = 1024;
__global__ __launch_bounds__(kBlock) void mark_front_kernel(int tiles, const uint32_t* __restrict__ ranges,
                                                            const uint32_t* __restrict__ point_list, uint8_t* __restrict__ front,
                                                            int n_front, const ColorSource* __restrict__ color_src, int gx,
                                                            int every) {
    if (!color_src->deferred) return;                    // precomputed colours: nothing is pending (the host does not know)
    const int tid = threadIdx.x;
    // every == 2: only the tiles of even column AND even row name Gaussians (a Gaussian in the leading entries of a tile
    // covers ~46 tiles of the bench view: it leads a sampled one too; whoever is missed is evaluated by the render kernel)
    const int sx = (gx + every - 1) / every;
#pragma unroll
    for (int t = 0; t < kFrontTiles; ++t) {
        const int j = blockIdx.x * kFrontTiles + t;
        const int tile = every == 1 ? j : (j / sx) * every * gx + (j % sx) * every;
        if (tile >= tiles) break;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        const uint32_t n = min(r1 - r0, (uint32_t)n_front);
        for (uint32_t e = tid; e < n; e += kBlock) front[point_list[r0 + e]] = 1;
    }
}
constexpr int kMarkedPer = 4;                            // flags per thread: one 4-byte load
__global__ __launch_bounds__(kBlock) void resolve_marked_kernel(Frame f, GeomView g) {
    const ColorSource cs = *g.color_src;
    if (!cs.deferred) return;                            // precomputed colours (or an eager forward): the flags were never zeroed
    __shared__ uint32_t s_id[kBlock * kMarkedPer];
    __shared__ uint32_t s_n;
    const int tid = threadIdx.x;
    if (tid == 0) s_n = 0;
    __syncthreads();
    const int i0 = (blockIdx.x * kBlock + tid) * kMarkedPer;
    if (i0 < f.P) {                                      // the 4-byte load may reach into the segment's padding: bytes >= P are ignored
        const uint32_t w = *reinterpret_cast<const uint32_t*>(g.front + i0);
#pragma unroll
        for (int k = 0; k < kMarkedPer; ++k)
            if (((w >> (8 * k)) & 0xFFu) && i0 + k < f.P) s_id[atomicAdd(&s_n, 1u)] = (uint32_t)(i0 + k);
    }
    __syncthreads();
    const uint32_t total = s_n;
    for (uint32_t i = tid; i < total; i += kBlock) {
        const uint32_t id = s_id[i];
        const float4 cd = g.rgbd[id];
        if (color_pending(cd)) resolve_color(cs, f.campos, id, cd.w, g.rgbd, g.clamped);
    }
}

int launch_zero_fill(void* p, size_t bytes, hipStream_t st);
int launch_render_forward(const Frame& f, GeomView g, BinningView b, ImageView im, int64_t D,
                          float* out_color, float* out_depth, hipStream_t st, float* zero_rows) {
    ZeroRegions z;
    if (zero_rows) {
        z.add(zero_rows, (size_t)f.P * kGradRow);
        z.add(g.touched, ((size_t)f.P + 15) / 16 * 4);           // bytes, rounded up to the 16 the compaction reads at once
                                                                  // (inside the scratch segment's 256-byte padding)
        z.add(g.touched_count, 1);
    }
    if (f.W <= 0 || f.H <= 0) return launch_zero_regions(z, st);
    const uint32_t* plist = b.vals[b.passes & 1];
    const unsigned blocks = (unsigned)(f.gx * f.gy);
    if (f.defer_colors && D > 0) {
        // MVI_RASTER_FRONT_ENTRIES: leading entries per tile evaluated ahead of the render kernel (0: none, A/B runs)
        static const int n_front = [] { const char* e = getenv("MVI_RASTER_FRONT_ENTRIES"); const int v = e ? atoi(e) : kBlock;
                                        return v < 0 ? 0 : (v > kFrontMax ? kFrontMax : v); }();
        static const int every = [] { const char* e = getenv("MVI_RASTER_FRONT_EVERY"); const int v = e ? atoi(e) : 2; return v < 1 ? 1 : v; }();
        if (n_front > 0) {
            const unsigned sampled = (unsigned)(((f.gx + every - 1) / every) * ((f.gy + every - 1) / every));
            hipLaunchKernelGGL(mark_front_kernel, dim3((sampled + kFrontTiles - 1) / kFrontTiles), dim3(kBlock), 0, st, (int)blocks,
                               im.ranges, plist, g.front, n_front, g.color_src, f.gx, every);
            hipLaunchKernelGGL(resolve_marked_kernel, dim3((f.P + kBlock * kMarkedPer - 1) / (kBlock * kMarkedPer)), dim3(kBlock), 0, st,
                               f, g);
        }
    }
    hipLaunchKernelGGL(render_forward_kernel, dim3(blocks), dim3(kBlock), 0, st, f, im.ranges, plist, g.xy,
                       g.rgbd, g.conic_opacity, im.final_T, im.n_contrib, out_color, out_depth, z, g.color_src, g.clamped);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

__global__ __launch_bounds__(256) void zero_regions_kernel(ZeroRegions z) { zero_share(z, threadIdx.x, 256); }
int launch_zero_regions(const ZeroRegions& z, hipStream_t st) {
    if (z.n == 0) return 0;
    hipLaunchKernelGGL(zero_regions_kernel, dim3(2048), dim3(256), 0, st, z);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

// ------------------------------------------------------------------------------------ backward
// Back-to-front replay over the entries [0, max n_contrib of the tile) only.
// Per evaluated (half tile, Gaussian) pair every pixel produces 9 RAW moments
//     W = o G dL/dalpha,  W dx,  W dy,  W dx^2,  W dx dy,  W dy^2,  alpha T dL/dC_{r,g,b}
// (the per-Gaussian factors — conic coefficients, 1/o, the 0.5 W / 0.5 H screen scale — are applied
// once per (tile, Gaussian) after the sums). A lane adds its two pixels, two DPP steps give quad totals,
// then the 16 quad partials of up to kSlabG Gaussians x 9 moments are parked in a per-wave LDS slab
// (rows of 16, stride 20) and summed by ONE lane per row, which adds the row total into the block's
// per-entry accumulator. Once per staged batch the block turns the raw sums into gradients and flushes
// them with float atomics shaped as whole 64-byte rows (16 lanes per Gaussian): MI355X float atomics
// run at the 64-B-request rate. Row layout of grad_rows [P][16]:
//   0 mean2D.x  1 mean2D.y  2 conic A  3 conic B  4 conic C  5 opacity  6 r  7 g  8 b  9..15 unused
constexpr int kRow = 16;
#ifndef MVI_RB_SLAB
#define MVI_RB_SLAB 3
#endif
constexpr int kSlabG = MVI_RB_SLAB;   // Gaussians parked per wave before a row-sum pass (3 x 9 = 27 rows; <= 7: one lane per row)
constexpr int kSlabStride = 20;    // floats per slab row (16 used; 80-byte rows keep b128 reads conflict-free)


// Quad totals of nine values in 18 instructions: v_add_f32_dpp reads the neighbour lane and adds in ONE instruction.
// Written as asm because the compiler lowers update_dpp + add to v_mov_b32_dpp + v_add (two issues each; the kernel is
// VALU-issue-bound, SQ_INSTS_VALU x 4 cycles ~ its whole duration). The leading s_nop covers the VALU-write -> DPP-read
// hazard for whatever instruction precedes the block (the assembler does not pad inline asm); inside the block every
// value is re-read nine instructions after it was written.
#define MVI_DPP1 " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define MVI_DPP2 " quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
__device__ __forceinline__ void quad_sum9(float& a, float& b, float& c, float& d, float& e, float& f, float& g, float& h,
                                          float& i) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0" MVI_DPP1 "v_add_f32_dpp %1, %1, %1" MVI_DPP1 "v_add_f32_dpp %2, %2, %2" MVI_DPP1
        "v_add_f32_dpp %3, %3, %3" MVI_DPP1 "v_add_f32_dpp %4, %4, %4" MVI_DPP1 "v_add_f32_dpp %5, %5, %5" MVI_DPP1
        "v_add_f32_dpp %6, %6, %6" MVI_DPP1 "v_add_f32_dpp %7, %7, %7" MVI_DPP1 "v_add_f32_dpp %8, %8, %8" MVI_DPP1
        "v_add_f32_dpp %0, %0, %0" MVI_DPP2 "v_add_f32_dpp %1, %1, %1" MVI_DPP2 "v_add_f32_dpp %2, %2, %2" MVI_DPP2
        "v_add_f32_dpp %3, %3, %3" MVI_DPP2 "v_add_f32_dpp %4, %4, %4" MVI_DPP2 "v_add_f32_dpp %5, %5, %5" MVI_DPP2
        "v_add_f32_dpp %6, %6, %6" MVI_DPP2 "v_add_f32_dpp %7, %7, %7" MVI_DPP2 "v_add_f32_dpp %8, %8, %8" MVI_DPP2
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(i));
}
#undef MVI_DPP1
#undef MVI_DPP2

// ---- backward, two pixels per lane --------------------------------------------------------------------------------
// Back-to-front replay of the tile's list (positions < max n_contrib of the tile only) with the same exact box
// culling as the forward. Every pixel produces 9 raw moments per evaluated Gaussian (W, W dx, W dy, W dx^2, W dx dy,
// W dy^2, alpha T dL/dC); quad totals by DPP, the 16 quad partials of up to kSlabG Gaussians are parked in a per-wave
// LDS slab and summed by one lane per row; per (tile, Gaussian) the block converts raw sums into gradients once and
// flushes them with float atomics shaped as whole 64-byte rows (grad_rows [P][16]).
// The kernel sits at the fp32 VALU issue roof. With one pixel per lane it needed ~90 vector
// instructions per (wave, Gaussian), ~33 of them (DPP quad sums, slab, drain) independent of how many pixels the wave
// covers. Here a wave covers a 16x8 half tile, lane l owning the vertically adjacent pixels (x = l & 15, y = 2 (l >> 4) + {0, 1}):
// the per-pixel math runs on packed fp32 (v_pk_mul/add/fma_f32: two pixels per instruction at full rate), the two
// pixels are summed in the lane before the quad reduction, and the reduction/slab cost is paid once per 128 pixels.
// Block = 128 threads = one tile; 128 list entries are staged per round. Decisions (alpha, active) use the same
// operation order as the forward kernel (products rounded, one fma per sum), element-wise.
#ifndef MVI_RB_WAVES
#define MVI_RB_WAVES 5             // waves per SIMD the register allocation aims at (A/B builds: -DMVI_RB_WAVES=4; HISTORY.md round 5)
#endif
__global__ __launch_bounds__(kB2) __attribute__((amdgpu_waves_per_eu(MVI_RB_WAVES, MVI_RB_WAVES))) void render_backward_kernel(
    Frame f, const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float2* __restrict__ xy, const float4* __restrict__ rgbd, const float4* __restrict__ conic_opacity,
    const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dL_dpix, float* __restrict__ grad_rows, uint8_t* __restrict__ touched, ZeroRegions zero) {
#pragma clang fp contract(off)
    // the dense gradient outputs of the per-Gaussian chain rule are zeroed HERE (this kernel is bound by vector issue, its
    // memory pipes are idle): preprocess_backward then only writes the few per cent of rows that received a gradient
    zero_share(zero, threadIdx.x, kB2);
    __shared__ uint32_t s_id[kB2];
    __shared__ float2 s_xy[kB2];
    __shared__ float s_t2[kB2];
    __shared__ float4 s_co[kB2];
    __shared__ float4 s_rgb[kB2];
    __shared__ float s_acc[kB2][9];                                   // raw moment sums per staged entry
    __shared__ __attribute__((aligned(16))) float s_slab[2][kSlabG * 9][kSlabStride];
    __shared__ int s_slot[2][8];
    __shared__ uint32_t s_blast[2];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tile = xcd_band_tile(blockIdx.x, f.gx * f.gy);
    const int tile_y = tile / f.gx, tile_x = tile - tile_y * f.gx;
    const int hx0 = tile_x * kTile, hy0 = tile_y * kTile + 8 * wave;                  // this wave's 16 x 8 half tile
    const int pxi = hx0 + (lane & 15), py0 = hy0 + 2 * (lane >> 4), py1 = py0 + 1;
    const bool in0 = pxi < f.W && py0 < f.H, in1 = pxi < f.W && py1 < f.H;
    const float pfx = (float)pxi;
    const f2 pfy = {(float)py0, (float)py1};
    const uint32_t r0 = ranges[2 * tile];
    const size_t hw = (size_t)f.H * f.W, pix0 = (size_t)py0 * f.W + pxi, pix1 = pix0 + f.W;

    const f2 T_final = {in0 ? final_T[pix0] : 0.0f, in1 ? final_T[pix1] : 0.0f};
    f2 T = T_final;
    const uint32_t last0 = in0 ? n_contrib[pix0] : 0u, last1 = in1 ? n_contrib[pix1] : 0u;
    // The reference's recurrence carries (accum_rec, last_color, last_alpha) per pixel and forms
    //     a_dot = last_alpha * last_color + (1 - last_alpha) * accum_rec
    // when the next contributing entry arrives. Here ONE value per pixel is carried: that a_dot for the next entry, updated by
    // the entry that changes it — A <- alpha * c_dot + (1 - alpha) * A, the same expression with the same operands, evaluated
    // one entry earlier (same bits). An entry that is not active for a pixel enters with alpha = 0 and o G = 0: A, T and every
    // moment are then unchanged by plain arithmetic (1 - 0 = 1, rcp(1) = 1, 0 * x = 0), so the eight selects that used to keep
    // six state values apart for the two pixels are two selects on alpha and two on o G.
    f2 A_dot = {0.f, 0.f};
    f2 dp0 = {0.f, 0.f}, dp1 = {0.f, 0.f}, dp2 = {0.f, 0.f};
    if (in0) { dp0.x = dL_dpix[pix0]; dp1.x = dL_dpix[hw + pix0]; dp2.x = dL_dpix[2 * hw + pix0]; }
    if (in1) { dp0.y = dL_dpix[pix1]; dp1.y = dL_dpix[hw + pix1]; dp2.y = dL_dpix[2 * hw + pix1]; }
    const f2 bg_dot = f.bg[0] * dp0 + f.bg[1] * dp1 + f.bg[2] * dp2;
    const f2 bg_term = -T_final * bg_dot;                   // dL/dalpha's background part is bg_term / (1 - alpha)
    uint32_t wave_last = max(last0, last1);
    for (int o = 32; o > 0; o >>= 1) wave_last = max(wave_last, (uint32_t)__shfl_xor((int)wave_last, o));
    if (lane == 0) s_blast[wave] = wave_last;
    __syncthreads();
    const int total = (int)max(s_blast[0], s_blast[1]);
    const int rounds = (total + kB2 - 1) / kB2;
    const int my_slot = (lane * 57) >> 9, my_mom = lane - 9 * my_slot;      // lane / 9, lane % 9 for lane < 64
    float (*slab)[kSlabStride] = s_slab[wave];
    auto drain = [&](int parked) {
        if (lane < 9 * parked) {
            const float4* row = reinterpret_cast<const float4*>(slab[lane]);
            float4 a = row[0], b = row[1], c = row[2], d = row[3];
            float sum = ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w)) + ((c.x + c.y) + (c.z + c.w)) +
                        ((d.x + d.y) + (d.z + d.w));
            atomicAdd(&s_acc[0][0] + (__mul24(s_slot[wave][my_slot], 9) + my_mom), sum);
        }
    };

    uint32_t n_id = 0;
    float2 n_xy = make_float2(0.f, 0.f);
    float4 n_co = make_float4(0.f, 0.f, 0.f, 0.f), n_cd = n_co;
    auto fetch = [&](int r) {
        const int hi_ = total - r * kB2;
        if (tid < hi_) {                                        // staged slot tid <-> list position hi_-1-tid
            n_id = point_list[r0 + (uint32_t)(hi_ - 1 - tid)];
            n_xy = xy[n_id]; n_co = conic_opacity[n_id]; n_cd = rgbd[n_id];
        }
    };
    if (rounds > 0) fetch(0);
    for (int r = 0; r < rounds; ++r) {
        const int hi = total - r * kB2;
        const int n = hi < kB2 ? hi : kB2;
        __syncthreads();
        s_id[tid] = n_id;
        s_xy[tid] = n_xy;
        s_co[tid] = n_co;
        s_t2[tid] = active_t2(n_co);
        s_rgb[tid] = n_cd;
#pragma unroll
        for (int c = 0; c < 9; ++c) s_acc[tid][c] = 0.0f;
        __syncthreads();
        if (r + 1 < rounds) fetch(r + 1);
        int parked = 0;                                         // wave-uniform
        for (int c = 0; c < n; c += 64) {
            const int e = c + lane;
            const bool keep = e < n && (uint32_t)(hi - 1 - e) < wave_last &&
                              quad_overlap(s_xy[e], s_co[e], s_t2[e], (float)hx0, (float)hy0, 15.0f, 7.0f);
            unsigned long long mask = ballot64(keep);
            if (mask == 0ull) continue;
            int j = c + __builtin_ctzll(mask);                  // wave-uniform
            auto entry = [&](const int jc, const float2 p, const float4 co, const float4 col, float2& nx_p, float4& nx_co,
                             float4& nx_col, int& jn) -> bool {
                mask &= mask - 1;
                const bool more = mask != 0ull;
                jn = more ? c + __builtin_ctzll(mask) : jc;
                nx_p = s_xy[jn];
                nx_co = s_co[jn];
                nx_col = s_rgb[jn];
                const uint32_t pos = (uint32_t)(hi - 1 - jc);
                const float dx = p.x - pfx;
                const f2 dy = splat(p.y) - pfy;
                const f2 power = gauss_power(co, dx, dy);          // the forward's rounding recipe
                const f2 G = {__expf(power.x), __expf(power.y)};    // power > 0 (inf, NaN) is never active: selected away
                const f2 oG = co.w * G;
                const f2 alpha = __builtin_elementwise_min(splat(kAlphaMax), oG);
                const bool act0 = pos < last0 && power.x <= 0.0f && alpha.x >= kAlphaMin;
                const bool act1 = pos < last1 && power.y <= 0.0f && alpha.y >= kAlphaMin;
                const f2 am = {act0 ? alpha.x : 0.0f, act1 ? alpha.y : 0.0f};          // alpha, o G of the ACTIVE pixels, else 0
                const f2 gm = {act0 ? oG.x : 0.0f, act1 ? oG.y : 0.0f};
                // (an active alpha is >= 1/255: nonzero bits. Asking the two lane predicates themselves made the compiler
                // materialise them as 0 / 1 and compare again: four vector instructions per entry)
                const unsigned long long any_active = ballot64((__float_as_uint(am.x) | __float_as_uint(am.y)) != 0u);
                const f2 om = splat(1.0f) - am;                      // in [0.01, 1]; exactly 1 for an inactive pixel
                const f2 inv = {__builtin_amdgcn_rcpf(om.x), __builtin_amdgcn_rcpf(om.y)};
                const f2 Tn = T * inv;                               // (= T for an inactive pixel)
                const f2 c_dot = pk_fma(splat(col.z), dp2, pk_fma(splat(col.y), dp1, col.x * dp0));
                const f2 dL_dalpha = pk_fma(c_dot - A_dot, Tn, bg_term * inv);
                const f2 dch = am * Tn, mw = gm * dL_dalpha;
                T = Tn;
                A_dot = pk_fma(am, c_dot, om * A_dot);
                if (any_active != 0ull) {                       // wave-uniform
                    const f2 vr = dch * dp0, vg = dch * dp1, vb = dch * dp2;
                    const f2 vx = mw * dx, vy = mw * dy;
                    const f2 vxx = vx * dx, vxy = vx * dy, vyy = vy * dy;
                    float m_w = mw.x + mw.y, m_x = vx.x + vx.y, m_y = vy.x + vy.y, m_xx = vxx.x + vxx.y,
                          m_xy = vxy.x + vxy.y, m_yy = vyy.x + vyy.y, m_r = vr.x + vr.y, m_g = vg.x + vg.y, m_b = vb.x + vb.y;
                    quad_sum9(m_w, m_x, m_y, m_xx, m_xy, m_yy, m_r, m_g, m_b);
                    const int pk = __builtin_amdgcn_readfirstlane(parked);
                    if ((lane & 3) == 3) {
                        float* colp = &slab[0][0] + (pk * (9 * kSlabStride) + (lane >> 2));
                        colp[0 * kSlabStride] = m_w;  colp[1 * kSlabStride] = m_x;  colp[2 * kSlabStride] = m_y;
                        colp[3 * kSlabStride] = m_xx; colp[4 * kSlabStride] = m_xy; colp[5 * kSlabStride] = m_yy;
                        colp[6 * kSlabStride] = m_r;  colp[7 * kSlabStride] = m_g;  colp[8 * kSlabStride] = m_b;
                    }
                    if (lane == 0) s_slot[wave][pk] = jc;
                    parked = pk + 1;
                    if (parked == kSlabG) { drain(kSlabG); parked = 0; }
                }
                return more;
            };
            float2 pA = s_xy[j], pB;
            float4 coA = s_co[j], coB, colA = s_rgb[j], colB;
            int jB = j;
            while (true) {
                if (!entry(j, pA, coA, colA, pB, coB, colB, jB)) break;
                if (!entry(jB, pB, coB, colB, pA, coA, colA, j)) break;
            }
        }
        if (parked) drain(parked);
        __syncthreads();
        // flush, in two steps per wave (a wave handles the 64 staged entries with its own index range, so only the wave has to
        // agree on LDS): (1) ONE lane per entry turns the raw sums into the nine gradients in place and marks entries that
        // nobody evaluated; (2) 16 lanes per entry add the row with float atomics, one 64-byte request per Gaussian.
        // (Before: every one of the 16 lanes of an entry read all nine sums, tested them and ran a switch over its component.)
        {
            const int e1 = wave * 64 + lane;
            if (e1 < n) {
                float* a = s_acc[e1];
                const float a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4], a5 = a[5], a6 = a[6], a7 = a[7], a8 = a[8];
                // any sum != 0 (as floats: +0 and -0 both count as zero), on the bit patterns: nine compares and their 0 / 1
                // bookkeeping became five bitwise instructions and one compare
                const bool any = (((__float_as_uint(a0) | __float_as_uint(a1) | __float_as_uint(a2)) |
                                   (__float_as_uint(a3) | __float_as_uint(a4) | __float_as_uint(a5)) |
                                   (__float_as_uint(a6) | __float_as_uint(a7) | __float_as_uint(a8))) & 0x7FFFFFFFu) != 0u;
                if (any) {
                    const float4 co = s_co[e1];
                    a[0] = -0.5f * (float)f.W * (co.x * a1 + co.y * a2);       // dL/dmean2D.x (NDC-scaled)
                    a[1] = -0.5f * (float)f.H * (co.z * a2 + co.y * a1);       // dL/dmean2D.y
                    a[2] = -0.5f * a3;                                         // dL/dA
                    a[3] = -a4;                                                // dL/dB
                    a[4] = -0.5f * a5;                                         // dL/dC
                    a[5] = a0 / co.w;                                          // dL/dopacity = sum G dL/dalpha
                } else {
                    s_id[e1] = 0xFFFFFFFFu;                                    // never evaluated by either half tile
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const int comp = lane & 15;
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int e = wave * 64 + it * 4 + (lane >> 4);
                const uint32_t id = e < n ? s_id[e] : 0xFFFFFFFFu;
                if (id == 0xFFFFFFFFu || comp >= 9) continue;
                atomicAdd(&grad_rows[(size_t)id * kRow + comp], s_acc[e][comp]);
                if (comp == 0) touched[id] = 1;                // gradient support (idempotent store)
            }
        }
    }
}

int launch_render_backward(const Frame& f, GeomView g, BinningView b, ImageView im, int64_t D,
                           const float* dL_dpix, float* grad_rows, hipStream_t st, const ZeroRegions* zero) {
    ZeroRegions z;
    if (zero) z = *zero;
    if (f.W <= 0 || f.H <= 0 || D <= 0) return launch_zero_regions(z, st);
    const uint32_t* plist = b.vals[b.passes & 1];
    hipLaunchKernelGGL(render_backward_kernel, dim3(f.gx * f.gy), dim3(kB2), 0, st, f, im.ranges, plist, g.xy,
                       g.rgbd, g.conic_opacity, im.final_T, im.n_contrib, dL_dpix, grad_rows, g.touched, z);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

}  // namespace mvi
