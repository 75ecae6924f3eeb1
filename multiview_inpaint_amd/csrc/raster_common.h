// Shared declarations of the gfx950 rasterizer kernels (device helpers + scratch layouts).
// Plug-in being replaced: diff_gaussian_rasterization, called at
// gs-simp/gaussian_renderer/__init__.py:85-93. Written for CDNA4 only (wave64, 256-thread blocks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <type_traits>

#include "../../include/mvi_raster.h"

namespace mvi {

constexpr int kTile = MVI_TILE;            // 16x16 pixels per tile, one 256-thread block (4 wave64)
constexpr int kBlock = 256;
constexpr int kPB = 64;                    // threads (= Gaussians) per block of the per-Gaussian preprocess kernels: their LDS
                                           // rows (192 B of SH per Gaussian) cap residency at ~9 waves per CU, and one-wave
                                           // blocks overlap their load / compute / store phases best (measured, backward
                                           // kernel: 256 threads 0.203 ms, 128: 0.202, 64: 0.184; forward 0.134 / 0.122 / 0.125)
constexpr float kNearZ = 0.2f;             // view-space z cull
constexpr float kLowpass = 0.3f;           // cov2D diagonal dilation
constexpr float kFrustumClamp = 1.3f;
constexpr float kLambdaFloor = 0.1f;
constexpr float kAlphaMax = 0.99f;
constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kTEps = 0.0001f;
constexpr float kDepthSentinel = 15.0f;    // gs-simp/gen_seq.py:50

constexpr int kSortItems = 8;              // pairs per thread of a radix scatter block
constexpr int kSortWavesP = 4;             // waves per scatter block: 2048-pair tiles for the P-sized level-1 passes,
constexpr int kSortWavesD = 8;             //                          4096-pair tiles for the D-sized level-2 passes
constexpr int kCountTiles = 4;             // sort tiles per block of the count kernel: one 16-byte run per histogram row
__host__ __device__ inline int sort_blocks(int64_t n, int waves) {   // radix-pass blocks, padded so the count kernel writes whole runs
    const int tile = 512 * waves;
    int b = (int)((n + tile - 1) / tile);
    return (b + kCountTiles - 1) / kCountTiles * kCountTiles;
}

// Per-call constants every kernel needs, passed by value (lives in SGPRs).
struct Frame {
    int P, M, deg, W, H, gx, gy;
    float tanfovx, tanfovy, fx, fy, scale_modifier;
    const float* view;   // [16] device
    const float* proj;   // [16] device
    const float* campos; // [3]  device
    const float* bg;     // [3]  device
    // raw-parameter mode (mvi_raster_*_raw): `shs` is features_dc [P,1,3], shs_rest is features_rest [P,M-1,3], and
    // opacities / scales / rotations are the un-activated parameters (sigmoid / exp / normalize happen in the kernels)
    const float* shs_rest = nullptr;
    int raw = 0;
    int bin_v2 = 0;      // binning version 2 (rectangle-expanding partition, raster_binning2.hip): set by make_frame
    int defer_colors = 0;    // SH input: colours evaluated by the render kernel on first use (ColorSource below): set by make_frame
};

// ---- binning version 2 (raster_binning2.hip): tile grids of at most 256 x 256 tiles (images up to 4096 x 4096) ------------
constexpr int kDsThreads = 256;                     // depth sort: 256 threads x 8 keys = 2048-key tiles
constexpr int kDsItems = 8;
constexpr int kDsTile = kDsThreads * kDsItems;
constexpr int kDsSuper = 16;                        // count-table rows per super row (two-level predecessor sums)
constexpr int kExThreads = 512;                     // expanding partition passes: 512 threads x 4 items
constexpr int kExItems = 4;
constexpr int kExChunk = kExThreads * kExItems;     // items (Gaussians in pass 1, column segments in pass 2) per block
bool binning_v2_enabled();                          // false when MVI_BINNING_LEGACY=1 or after set_binning_version(1)
void set_dev_stamps(int pass, void* buf);            // diagnostics: shader-clock stamps of the partition kernel's phases
int set_binning_version(int v);                     // 1 | 2 (anything else: query only); returns the previous version
inline bool binning_v2_ok(int gx, int gy) { return gx <= 256 && gy <= 256 && binning_v2_enabled(); }

// Deferred SH colours. With SH input the forward preprocess does NOT evaluate SH -> RGB for every visible Gaussian (192 bytes
// of coefficients each at degree 3, 63 % of everything that kernel reads, for colours of which ~3 % are ever composited: a
// tile's pixels saturate after the first few hundred entries of its list). It writes rgbd = (-1, -1, -1, depth) instead and
// leaves this record in the geom scratch; render_forward evaluates a colour the first time it STAGES the Gaussian (same
// arithmetic, shared function sh_rgb below), writes it back for the other tiles and for the backward, and uses its own value.
// Every channel of an evaluated colour is >= 0 (clamped), so "any channel < 0" means "not evaluated": a torn or stale read
// of another block's write-back just evaluates again (same bits). Positions the backward replays were all staged by the
// forward of the same tile. mvi_raster_resolve_colors evaluates what is left (tests / introspection).
struct ColorSource {
    const float* means3D;     // [P,3] the caller's array (must stay valid and unchanged until the backward, as before)
    const float* shs;         // [P,M,3], or features_dc [P,1,3] in raw mode
    const float* shs_rest;    // raw mode: features_rest [P,M-1,3]
    int32_t M, deg, raw;
    int32_t deferred;         // 0: every colour was evaluated by the preprocess kernel (or colours were given precomputed)
    int32_t vec16;            // rows of shs may be read as 16-byte vectors (M % 4 == 0, base 16-byte aligned, not raw)
};

// ---- scratch layouts (all offsets 256-B aligned) ---------------------------------------------
struct GeomView {
    float* depths;            // [P]
    float2* xy;               // [P]
    float4* cov_a;            // [P] cov3D xx, xy, xz, yy   (16-byte and 8-byte records: every lane's
    float2* cov_b;            // [P] cov3D yz, zz           store/load is one whole aligned access)
    float4* conic_opacity;    // [P]
    float4* rgbd;             // [P] r, g, b, view-space depth (one gather per staged list entry)
    uint32_t* tiles_touched;  // [P]
    uint2* rect;              // [P] tile rectangle (x0 | y0 << 16, w | h << 16); w * h = tiles touched, 0 when culled:
                              //     pair emission gathers this one 8-byte record per depth-ordered Gaussian
    uint8_t* clamped;         // [P] bit c set = colour channel c was clamped at 0
    uint8_t* front;           // [P] deferred colours: 1 = named by a leading entry of some tile's list (zeroed by the preprocess)
    struct ColorSource* color_src;   // [1] where the SH colours of this forward come from (deferred evaluation, see ColorSource)
    uint32_t* touched_list;   // [P] the touched Gaussians, compacted (any order); touched_count[0] of them
    uint32_t* touched_count;  // [1]
    uint8_t* touched;         // [P] 1 = the render backward added something to this Gaussian's accumulation row (gradient
                              //     support: ~3 % of the Gaussians of the bench scene; the others are occluded). Zeroed with the rows.
    uint32_t* block_sums;     // [npre]   tiles touched per preprocess block (kPB Gaussians)
    uint32_t* block_offsets;  // [npre+1] exclusive scan of block_sums; [npre] = D
    // depth sort of the Gaussians (binning level 1): ping-pong (depth bits, index) pairs
    uint32_t* dkeys[2];       // [P]
    uint32_t* dvals[2];       // [P]; after the 4 passes dvals[0] = Gaussian indices in depth order
    uint32_t* dhist;          // [256 * nsortP]
    uint32_t* dtot;           // [256]
    uint32_t* perm_sums;      // [nblk]   tiles touched per 256 depth-ordered Gaussians
    uint32_t* perm_offsets;   // [nblk+1]
    // binning version 2
    uint2* rect_sorted;       // [P] the tile rectangles in depth order (written by the column count, read by pass 1)
    uint32_t* col_table;      // [256][nblk1] column-major: column segments per (column, block of kExChunk Gaussians)
    uint32_t* col_tot;        // [256] column segments per tile column
    uint32_t* seg_sums;       // [npre] column segments (sum of rectangle widths) per preprocess block
    uint32_t* chunk_first;    // [257] first pass-2 chunk of tile column x; [gx] = number of chunks
    uint32_t* col_start;      // [257] first column segment of tile column x; [gx] = number of segments
    uint32_t* arrivals;       // [1] blocks of the column scan that have finished (the last one builds the chunk table)
    uint32_t* ds_table;       // [nds][256] depth-sort digit counts per 4096-key tile (aliases dhist)
    uint32_t* ds_super[2];    // [nsuper][256] sums over kDsSuper tiles, one region per pass parity (inside dhist)
    int nds, nsuper;
    int nblk1;                // blocks of pass 1
    int nsortP;
    size_t bytes;
};
struct ImageView {
    uint32_t* ranges;     // [2*tiles]
    float* final_T;       // [H*W]
    uint32_t* n_contrib;  // [H*W]
    size_t bytes;
};
struct BinningView {
    // Version 2 (v2 != 0, grids up to 256 x 256 tiles): keys[0] = (first row | rows - 1 << 8) of each column segment,
    // vals[0] = its Gaussian (pass 1 output, in column order); keys[1] = tile id of each sorted pair, vals[1] = the sorted
    // point list (pass 2 output); passes = 1 so that keys / vals[passes & 1] is the result in both versions.
    // block_hist = [256][nsort] rows-per-chunk table, digit_tot = pairs per tile row, col_rel = [gy][gx + 1].
    void* keys[2];        // ping-pong [D] tile ids, key_bytes each (the 64-bit (tile|depth) key is implicit: pairs are
                          // emitted in depth order and stably partitioned by tile)
    int key_bytes;        // 2 while the image has at most 65536 tiles, else 4
    uint32_t* vals[2];    // ping-pong [D] Gaussian indices
    uint32_t* block_hist; // [256 * nsort]  digit-major per-block digit counts / offsets
    uint32_t* digit_tot;  // [256]
    int nsort;            // blocks per radix pass
    int passes;           // 8-bit passes over the tile bits; sorted pairs end in keys/vals[passes & 1]
    int key_bits;         // tile bits
    int v2;               // binning version 2 layout
    uint32_t* col_rel;    // v2: [gy][gx + 1] pairs of tile row y in front of column x (row-relative)
    size_t bytes;
};

inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

inline GeomView carve_geom(void* base, int P) {
    GeomView g;
    char* p = (char*)base;
    size_t o = 0;
    size_t n = (size_t)(P > 0 ? P : 1);
    int nblk = (int)((n + kBlock - 1) / kBlock);
    auto take = [&](size_t b) { char* r = p ? p + o : nullptr; o += align256(b); return r; };
    g.depths = (float*)take(4 * n);
    g.xy = (float2*)take(8 * n);
    g.cov_a = (float4*)take(16 * n);
    g.cov_b = (float2*)take(8 * n);
    g.conic_opacity = (float4*)take(16 * n);
    g.rgbd = (float4*)take(16 * n);
    g.tiles_touched = (uint32_t*)take(4 * n);
    g.rect = (uint2*)take(8 * n);
    g.clamped = (uint8_t*)take(n);
    g.touched = (uint8_t*)take(n);
    g.touched_list = (uint32_t*)take(4 * n);
    g.touched_count = (uint32_t*)take(4);
    g.front = (uint8_t*)take(n);
    g.color_src = (ColorSource*)take(sizeof(ColorSource));
    const size_t npre = (n + kPB - 1) / kPB;
    g.block_sums = (uint32_t*)take(4 * npre);
    g.block_offsets = (uint32_t*)take(4 * (npre + 1));
    g.nsortP = sort_blocks(n, kSortWavesP);
    for (int i = 0; i < 2; ++i) { g.dkeys[i] = (uint32_t*)take(4 * n); g.dvals[i] = (uint32_t*)take(4 * n); }
    g.dhist = (uint32_t*)take(4 * 256 * (size_t)g.nsortP);
    g.dtot = (uint32_t*)take(4 * 256);
    g.perm_sums = (uint32_t*)take(4 * (size_t)nblk);
    g.perm_offsets = (uint32_t*)take(4 * (size_t)(nblk + 1));
    g.rect_sorted = (uint2*)take(8 * n);
    g.nblk1 = (int)((n + kExChunk - 1) / kExChunk);
    g.col_table = (uint32_t*)take(4 * 256 * (size_t)g.nblk1);
    g.col_tot = (uint32_t*)take(4 * 256);
    g.seg_sums = (uint32_t*)take(4 * npre);
    g.chunk_first = (uint32_t*)take(4 * 257);
    g.col_start = (uint32_t*)take(4 * 257);
    g.arrivals = (uint32_t*)take(4);
    g.nds = (int)((n + kDsTile - 1) / kDsTile);
    g.nsuper = (g.nds + kDsSuper - 1) / kDsSuper;
    g.ds_table = (uint32_t*)take(4 * 256 * (size_t)g.nds);
    g.ds_super[0] = (uint32_t*)take(4 * 256 * (size_t)g.nsuper);
    g.ds_super[1] = (uint32_t*)take(4 * 256 * (size_t)g.nsuper);
    g.bytes = o;
    return g;
}
inline ImageView carve_image(void* base, int W, int H) {
    ImageView v;
    char* p = (char*)base;
    size_t o = 0;
    size_t tiles = (size_t)((W + kTile - 1) / kTile) * ((H + kTile - 1) / kTile);
    size_t px = (size_t)W * H;
    auto take = [&](size_t b) { char* r = p ? p + o : nullptr; o += align256(b); return r; };
    v.ranges = (uint32_t*)take(8 * (tiles ? tiles : 1));
    v.final_T = (float*)take(4 * (px ? px : 1));
    v.n_contrib = (uint32_t*)take(4 * (px ? px : 1));
    v.bytes = o;
    return v;
}
inline int tile_bits(int W, int H) {
    unsigned tiles = (unsigned)(((W + kTile - 1) / kTile) * ((H + kTile - 1) / kTile));
    int b = 0;
    while ((1u << b) < tiles && b < 31) ++b;
    return b > 0 ? b : 1;
}
inline BinningView carve_binning(void* base, int64_t D, int W, int H) {
    BinningView v;
    char* p = (char*)base;
    size_t o = 0;
    size_t n = (size_t)(D > 0 ? D : 1);
    v.nsort = sort_blocks(n, kSortWavesD);
    v.key_bits = tile_bits(W, H);
    v.passes = (v.key_bits + 7) / 8;
    auto take = [&](size_t b) { char* r = p ? p + o : nullptr; o += align256(b); return r; };
    const unsigned tiles = (unsigned)(((W + kTile - 1) / kTile) * ((H + kTile - 1) / kTile));
    v.key_bytes = tiles <= 65536u ? 2 : 4;
    const int gx = (W + kTile - 1) / kTile, gy = (H + kTile - 1) / kTile;
    v.v2 = binning_v2_ok(gx, gy) ? 1 : 0;
    v.col_rel = nullptr;
    if (v.v2) {
        // chunks of pass 2: every tile column owns at least one, so at most D / kExChunk + gx (column segments <= pairs);
        // the launches use the exact segment count when the forward knows it
        v.nsort = (int)((n / kExChunk + (size_t)gx + 1 + 3) / 4 * 4);
        v.passes = 1;
        v.key_bytes = 2;
    }
    v.keys[0] = take((size_t)v.key_bytes * n);
    v.keys[1] = take((size_t)v.key_bytes * n);
    v.vals[0] = (uint32_t*)take(4 * n);
    v.vals[1] = (uint32_t*)take(4 * n);
    v.block_hist = (uint32_t*)take(4 * 256 * (size_t)v.nsort);
    v.digit_tot = (uint32_t*)take(4 * 256);
    if (v.v2) v.col_rel = (uint32_t*)take(4 * (size_t)gy * (size_t)(gx + 1));
    v.bytes = o;
    return v;
}

// ---- device helpers ---------------------------------------------------------------------------
// Arithmetic contract shared with oracle/raster_oracle.c: explicit fma chains, no other fusion in
// the functions that feed integer outputs (those are compiled under `#pragma clang fp contract(off)`).
__device__ __forceinline__ float affine3(float m0, float m1, float m2, float m3, float x, float y, float z) {
    return __builtin_fmaf(m2, z, __builtin_fmaf(m1, y, __builtin_fmaf(m0, x, m3)));
}
__device__ __forceinline__ float dot3(float a0, float a1, float a2, float b0, float b1, float b2) {
#pragma clang fp contract(off)
    return __builtin_fmaf(a2, b2, __builtin_fmaf(a1, b1, a0 * b0));
}

__device__ __forceinline__ void tile_rect(float px, float py, int radius, int gx, int gy, int& x0,
                                          int& y0, int& x1, int& y1) {
#pragma clang fp contract(off)
    x0 = min(gx, max(0, (int)((px - (float)radius) / (float)kTile)));
    y0 = min(gy, max(0, (int)((py - (float)radius) / (float)kTile)));
    x1 = min(gx, max(0, (int)((px + (float)radius + (float)(kTile - 1)) / (float)kTile)));
    y1 = min(gy, max(0, (int)((py + (float)radius + (float)(kTile - 1)) / (float)kTile)));
}

// ---- SH -> RGB (gs-simp/utils/sh_utils.py:57-112 basis and signs; + 0.5 and clamp at 0 as gaussian_renderer/__init__.py:77-78)
__device__ constexpr float SH_C0 = 0.28209479177387814f;
__device__ constexpr float SH_C1 = 0.4886025119029199f;
__device__ constexpr float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                       -1.0925484305920792f, 0.5462742152960396f};
__device__ constexpr float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                       0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                       -0.5900435899266435f};

__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float* b) {
#pragma clang fp contract(off)
    b[0] = SH_C0;
    if (deg > 0) {
        b[1] = -SH_C1 * y;
        b[2] = SH_C1 * z;
        b[3] = -SH_C1 * x;
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = SH_C2[0] * xy;
            b[5] = SH_C2[1] * yz;
            b[6] = SH_C2[2] * (2.0f * zz - xx - yy);
            b[7] = SH_C2[3] * xz;
            b[8] = SH_C2[4] * (xx - yy);
            if (deg > 2) {
                b[9] = SH_C3[0] * y * (3.0f * xx - yy);
                b[10] = SH_C3[1] * xy * z;
                b[11] = SH_C3[2] * y * (4.0f * zz - xx - yy);
                b[12] = SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                b[13] = SH_C3[4] * x * (4.0f * zz - xx - yy);
                b[14] = SH_C3[5] * z * (xx - yy);
                b[15] = SH_C3[6] * x * (xx - 3.0f * yy);
            }
        }
    }
}

// Colour of one Gaussian seen from campos, in four steps that are the ONE piece of arithmetic behind both the eager evaluation
// (preprocess kernel, coefficients from LDS: sh_rgb) and the deferred one (render / resolve kernels, coefficients loaded in
// chunks: resolve_color_deg). Explicit fma chain in coefficient order, nothing else fused: shared with oracle/raster_oracle.c.
__device__ __forceinline__ void sh_direction(float px, float py, float pz, const float* __restrict__ campos, float& dx, float& dy,
                                             float& dz) {
#pragma clang fp contract(off)
    dx = px - campos[0]; dy = py - campos[1]; dz = pz - campos[2];
    const float len = sqrtf(dot3(dx, dy, dz, dx, dy, dz));
    dx = dx / len; dy = dy / len; dz = dz / len;
}
__device__ __forceinline__ void sh_first(float b0, float c0, float c1, float c2, float& r0, float& r1, float& r2) {
#pragma clang fp contract(off)
    r0 = b0 * c0; r1 = b0 * c1; r2 = b0 * c2;
}
__device__ __forceinline__ void sh_next(float bk, float c0, float c1, float c2, float& r0, float& r1, float& r2) {
    r0 = __builtin_fmaf(bk, c0, r0); r1 = __builtin_fmaf(bk, c1, r1); r2 = __builtin_fmaf(bk, c2, r2);
}
__device__ __forceinline__ void sh_finish(float& r0, float& r1, float& r2, uint32_t& clamp_bits) {
#pragma clang fp contract(off)
    r0 += 0.5f; r1 += 0.5f; r2 += 0.5f;
    clamp_bits = (r0 < 0.0f ? 1u : 0u) | (r1 < 0.0f ? 2u : 0u) | (r2 < 0.0f ? 4u : 0u);
    r0 = fmaxf(r0, 0.0f); r1 = fmaxf(r1, 0.0f); r2 = fmaxf(r2, 0.0f);
}
// coef(k, c) returns coefficient k < (deg + 1)^2 of channel c
template <typename Coef>
__device__ __forceinline__ void sh_rgb(int deg, float px, float py, float pz, const float* __restrict__ campos, Coef coef,
                                       float& r0, float& r1, float& r2, uint32_t& clamp_bits) {
    float dx, dy, dz, bs[16];
    sh_direction(px, py, pz, campos, dx, dy, dz);
    sh_basis(deg, dx, dy, dz, bs);
    const int nb = (deg + 1) * (deg + 1);
    sh_first(bs[0], coef(0, 0), coef(0, 1), coef(0, 2), r0, r1, r2);
    for (int k = 1; k < nb; ++k) sh_next(bs[k], coef(k, 0), coef(k, 1), coef(k, 2), r0, r1, r2);
    sh_finish(r0, r1, r2, clamp_bits);
}

// Deferred evaluation of Gaussian `id` (see ColorSource): loads its active coefficients, evaluates, writes rgbd / clamped
// back. DEG is a template parameter so that everything is indexed by constants (registers, no scratch). The coefficients come
// in chunks of 8 (six 16-byte loads), each consumed before the next is requested: the render kernel runs this in a cold
// branch beside ~25 live registers of compositing state and must stay within 80 (6 waves per SIMD); the latency of the
// second chunk is covered by the CU's other tiles.
template <int DEG>
__device__ __forceinline__ float4 resolve_color_deg(const ColorSource& cs, const float* __restrict__ campos, uint32_t id, float depth,
                                                    float4* rgbd, uint8_t* clamped) {
    constexpr int NB = (DEG + 1) * (DEG + 1), KC = 8;
    const size_t si = (size_t)id;
    const float* row = cs.raw ? cs.shs_rest + si * (size_t)(3 * cs.M - 3) - 3 : cs.shs + si * (size_t)(3 * cs.M);   // + 3 k (raw: k >= 1)
    float dx, dy, dz, bs[NB > 1 ? NB : 1], r0 = 0.f, r1 = 0.f, r2 = 0.f;
    sh_direction(cs.means3D[3 * si], cs.means3D[3 * si + 1], cs.means3D[3 * si + 2], campos, dx, dy, dz);
    {
        float b16[16];
        sh_basis(DEG, dx, dy, dz, b16);
#pragma unroll
        for (int k = 0; k < NB; ++k) bs[k] = b16[k];
    }
#pragma unroll
    for (int k0 = 0; k0 < NB; k0 += KC) {
        constexpr int kMaxF = 3 * KC;
        const int nk = NB - k0 < KC ? NB - k0 : KC;          // compile-time after unrolling
        float v[kMaxF];
        if (!cs.raw && cs.vec16) {
#pragma unroll
            for (int q = 0; q < kMaxF / 4; ++q) {
                if (4 * q < 3 * nk) {                         // whole vectors; a row holds 3 M >= the padded count (M % 4 == 0)
                    const float4 t = reinterpret_cast<const float4*>(row + 3 * k0)[q];
                    v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < kMaxF; ++j)
                if (j < 3 * nk) v[j] = (cs.raw && k0 == 0 && j < 3) ? cs.shs[3 * si + j] : row[3 * k0 + j];
        }
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            if (k < nk) {
                if (k0 + k == 0) sh_first(bs[0], v[0], v[1], v[2], r0, r1, r2);
                else sh_next(bs[k0 + k], v[3 * k], v[3 * k + 1], v[3 * k + 2], r0, r1, r2);
            }
        }
        __builtin_amdgcn_sched_barrier(0);                    // the next chunk's loads stay behind this chunk's arithmetic
    }
    uint32_t bits;
    sh_finish(r0, r1, r2, bits);
    const float4 out = make_float4(r0, r1, r2, depth);
    rgbd[si] = out;
    clamped[si] = (uint8_t)bits;
    return out;
}
__device__ __forceinline__ float4 resolve_color(const ColorSource& cs, const float* __restrict__ campos, uint32_t id, float depth,
                                             float4* rgbd, uint8_t* clamped) {
    switch (cs.deg) {                 // uniform
        case 0: return resolve_color_deg<0>(cs, campos, id, depth, rgbd, clamped);
        case 1: return resolve_color_deg<1>(cs, campos, id, depth, rgbd, clamped);
        case 2: return resolve_color_deg<2>(cs, campos, id, depth, rgbd, clamped);
        default: return resolve_color_deg<3>(cs, campos, id, depth, rgbd, clamped);
    }
}
// The same evaluation for the render kernel's cold branch, written for a SMALL register footprint instead of loads in
// flight: one code path for every input form (a coefficient = one 12-byte load from wherever it lives), the coefficients of
// at most four coefficients requested together and consumed before the next ones (uniform branches / scheduling barriers
// keep the scheduler from hoisting all 16 loads, which cost the compositing loop its registers: 143 VGPRs / 3 waves per SIMD when
// the fast form was inlined there). Identical arithmetic: sh_direction, sh_basis, sh_first / sh_next in coefficient
// order, sh_finish.
__device__ __forceinline__ float4 resolve_color_small(const ColorSource& cs, const float* __restrict__ campos, uint32_t id,
                                                      float depth, float4* rgbd, uint8_t* clamped) {
    const size_t si = (size_t)id;
    const float* c0p = cs.shs + (cs.raw ? 3 * si : si * (size_t)(3 * cs.M));                                            // k = 0
    const float* row = cs.raw ? cs.shs_rest + si * (size_t)(3 * cs.M - 3) - 3 : cs.shs + si * (size_t)(3 * cs.M);      // + 3 k, k >= 1
    float dx, dy, dz, bs[16], r0, r1, r2;
    sh_direction(cs.means3D[3 * si], cs.means3D[3 * si + 1], cs.means3D[3 * si + 2], campos, dx, dy, dz);
    sh_basis(cs.deg, dx, dy, dz, bs);
    struct F3 { float a, b, c; };
    {
        const F3 t = *reinterpret_cast<const F3*>(c0p);
        sh_first(bs[0], t.a, t.b, t.c, r0, r1, r2);
    }
    auto band = [&](auto first_c, auto last_c) __attribute__((always_inline)) {
        constexpr int kFirst = decltype(first_c)::value, kLast = decltype(last_c)::value;
        F3 t[kLast - kFirst + 1];
#pragma unroll
        for (int k = kFirst; k <= kLast; ++k) t[k - kFirst] = *reinterpret_cast<const F3*>(row + 3 * k);
#pragma unroll
        for (int k = kFirst; k <= kLast; ++k) sh_next(bs[k], t[k - kFirst].a, t[k - kFirst].b, t[k - kFirst].c, r0, r1, r2);
    };
    using std::integral_constant;
    if (cs.deg > 0) {
        band(integral_constant<int, 1>{}, integral_constant<int, 3>{});
        if (cs.deg > 1) {
            band(integral_constant<int, 4>{}, integral_constant<int, 6>{});
            __builtin_amdgcn_sched_barrier(0);
            band(integral_constant<int, 7>{}, integral_constant<int, 8>{});
            if (cs.deg > 2) {
                band(integral_constant<int, 9>{}, integral_constant<int, 12>{});
                __builtin_amdgcn_sched_barrier(0);
                band(integral_constant<int, 13>{}, integral_constant<int, 15>{});
            }
        }
    }
    uint32_t bits;
    sh_finish(r0, r1, r2, bits);
    const float4 out = make_float4(r0, r1, r2, depth);
    rgbd[si] = out;
    clamped[si] = (uint8_t)bits;
    return out;
}
__device__ __forceinline__ bool color_pending(float4 cd) { return cd.x < 0.0f || cd.y < 0.0f || cd.z < 0.0f; }

// Sum over the 64 lanes of a wave with DPP moves; the total lands in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true);
    return v + __int_as_float(t);
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v = dpp_add<0xb1, 0xf>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4e, 0xf>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x114, 0xf>(v);   // row_shr:4
    v = dpp_add<0x118, 0xf>(v);   // row_shr:8
    v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1,3
    v = dpp_add<0x143, 0xc>(v);   // row_bcast:31 into rows 2,3
    return v;
}
// Sum over each 16-lane DPP row; the row total lands in lanes 15, 31, 47, 63.
__device__ __forceinline__ float row_sum_to_lane15(float v) {
    v = dpp_add<0xb1, 0xf>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4e, 0xf>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x114, 0xf>(v);   // row_shr:4
    v = dpp_add<0x118, 0xf>(v);   // row_shr:8
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Stage timing (include/mvi_raster.h: mvi_raster_timing_*). A StageTimer brackets the launches
// enqueued during its lifetime with two hipEvents when timing is enabled; otherwise it is free.
enum Stage { kStPreFwd = 0, kStScan, kStDup, kStSort, kStRanges, kStRenderFwd, kStRenderBwd, kStPreBwd };
void stage_begin(int stage, hipStream_t st);
void stage_end(int stage, hipStream_t st);
struct StageTimer {
    int stage; hipStream_t st;
    StageTimer(int s, hipStream_t q) : stage(s), st(q) { stage_begin(stage, st); }
    ~StageTimer() { stage_end(stage, st); }
};

// Memory a kernel zeroes on the side while it runs (the render kernels are issue-bound with the memory pipes idle): up to
// kMaxZero arrays of fp32 words, each cut into an unaligned head (< 4 words), a 16-byte-aligned body and a tail (< 4 words);
// the bodies are shared out over the blocks of the launch, block 0 also writes heads and tails.
constexpr int kMaxZero = 10;
struct ZeroRegions {
    float* head[kMaxZero];      // first word of the array
    uint4* body[kMaxZero];
    float* tail[kMaxZero];
    uint32_t n_head[kMaxZero], n_body[kMaxZero], n_tail[kMaxZero];   // words, 16-byte units, words
    int n = 0;
    bool overflow = false;      // an array was offered when all kMaxZero slots were taken: the caller must fail, not run
    void add(void* ptr, size_t words) {
        if (!ptr || !words) return;
        if (n >= kMaxZero) { overflow = true; return; }
        float* p = (float*)ptr;
        const size_t mis = ((uintptr_t)p & 15u) / 4u;
        size_t h = mis ? 4 - mis : 0;
        if (h > words) h = words;
        const size_t b = (words - h) / 4, t = words - h - 4 * b;
        head[n] = p; n_head[n] = (uint32_t)h;
        body[n] = (uint4*)(p + h); n_body[n] = (uint32_t)b;
        tail[n] = p + h + 4 * b; n_tail[n] = (uint32_t)t;
        ++n;
    }
};
__device__ __forceinline__ void zero_share(const ZeroRegions& z, int tid, int nthreads) {
    for (int r = 0; r < z.n; ++r) {
        const uint32_t per = (z.n_body[r] + gridDim.x - 1) / gridDim.x;
        const size_t z0 = (size_t)blockIdx.x * per;
        // write-through stores that do not stay in the XCD's L2 (MI355X_MICROARCH.md, stores of each flavour): the zeros are
        // not read again by this kernel, and the lines they would occupy hold the tile lists the kernel gathers from
        typedef uint32_t zero_u32x4 __attribute__((ext_vector_type(4)));
        const zero_u32x4 zz = {0u, 0u, 0u, 0u};
        for (uint32_t i = tid; i < per && z0 + i < z.n_body[r]; i += nthreads)
            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(z.body[r] + z0 + i), "v"(zz) : "memory");
        if (blockIdx.x == 0) {
            if (tid < (int)z.n_head[r]) z.head[r][tid] = 0.f;
            if (tid < (int)z.n_tail[r]) z.tail[r][tid] = 0.f;
        }
    }
}
int launch_zero_regions(const ZeroRegions& z, hipStream_t st);

// launchers (defined in the .hip files; all enqueue on `st`, none synchronise)
int launch_preprocess_forward(const Frame& f, const float* means3D, const float* shs,
                              const float* colors_precomp, const float* opacities, const float* scales,
                              const float* rotations, const float* cov3D_precomp, GeomView g,
                              int32_t* radii, hipStream_t st);
// raw mode only: raw_opacity [P] (chain rule of the sigmoid), dL_dshs_rest [P,M-1,3]
struct RawBackwardExtra { const float* raw_opacity = nullptr; float* dL_dshs_rest = nullptr; int rows_prezeroed = 0; };
// sparse form: the gradient outputs are already zero (the render backward zeroed them on the side) and only the Gaussians
// whose accumulation row was touched are read, computed and written (per-lane accesses: ~3 % of the rows)
int launch_preprocess_backward_sparse(const Frame& f, const float* means3D, const float* shs, const float* scales,
                                      const float* rotations, const float* cov3D_precomp, GeomView g, const float* grad_rows,
                                      float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dopacity, float* dL_dcolors,
                                      float* dL_dshs, float* dL_dcov3D, float* dL_dscales, float* dL_drots, hipStream_t st,
                                      RawBackwardExtra raw);
int launch_preprocess_backward(const Frame& f, const float* means3D, const float* shs,
                               const float* scales, const float* rotations, const float* cov3D_precomp,
                               const int32_t* radii, GeomView g, const float* grad_rows, float* dL_dmeans3D,
                               float* dL_dmeans2D, float* dL_dopacity, float* dL_dcolors, float* dL_dshs,
                               float* dL_dcov3D, float* dL_dscales, float* dL_drots, hipStream_t st,
                               RawBackwardExtra raw = RawBackwardExtra());
int launch_binning(const Frame& f, GeomView g, const int32_t* radii, BinningView b, ImageView im,
                   int64_t D, hipStream_t st);
int launch_zero_fill(void* p, size_t bytes, hipStream_t st);
int launch_binning2_totals(GeomView g, int P, unsigned long long* totals_host_devptr, hipStream_t st);   // [0] = D, [1] = segments
int launch_binning2_level1(const Frame& f, GeomView g, hipStream_t st);
// segments: the exact number of column segments if known (> 0), else the bound D is used to size the pass-2 grids
int launch_binning2(const Frame& f, GeomView g, BinningView b, ImageView im, int64_t D, int64_t segments, hipStream_t st);
int launch_render_forward(const Frame& f, GeomView g, BinningView b, ImageView im, int64_t D,
                          float* out_color, float* out_depth, hipStream_t st, float* zero_rows = nullptr);
int launch_resolve_colors(const Frame& f, GeomView g, hipStream_t st);      // evaluates every colour still pending
int launch_render_backward(const Frame& f, GeomView g, BinningView b, ImageView im, int64_t D,
                           const float* dL_dpix, float* grad_rows, hipStream_t st, const ZeroRegions* zero = nullptr);
constexpr int kGradRow = 16;   // floats per Gaussian in the backward accumulation rows (64 B)
int launch_sh_backward_views(int P, int M, int deg, int n_views, const float* means3D, const float* campos,
                             int64_t campos_stride, const float* gcol, int64_t gcol_stride, float* dL_dshs, hipStream_t st);
int launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* visible, hipStream_t st);
int launch_color_factors(int P, const int32_t* radii, const float* grad_rows, const uint8_t* clamped, float* out, hipStream_t st);

}  // namespace mvi
