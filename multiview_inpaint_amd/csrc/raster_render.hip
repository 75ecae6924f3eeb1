// Per-tile alpha compositing with median depth, forward and backward, for gfx950.
// Replaces the plug-in's render stages behind gs-simp/gaussian_renderer/__init__.py:85-93
// (forward: colour [3,H,W] + depth [1,H,W]) and loss.backward() (gs-simp/train.py:93).
//
// One 256-thread block per 16x16-pixel tile = 4 wave64, each wave an 8x8-pixel quadrant. The tile's
// depth-sorted list is staged through LDS 256 entries at a time by the whole block; every staged
// entry also gets the axis-aligned half-extents of its "active ellipse" {alpha >= 1/255}
// (power >= -ln(255 o)  <=>  d^T conic d <= 2 ln(255 o); half-extents sqrt(2 ln(255 o) * cov_xx|yy)).
// Each wave then tests 64 staged entries at a time against its own quadrant (one lane per entry),
// takes the __ballot mask and walks only the set bits with scalar s_ff1: a Gaussian that cannot
// reach alpha >= 1/255 anywhere in the quadrant is never evaluated. The test is conservative
// (extents inflated by 0.1 % + 0.05 px), so results are identical to evaluating every entry.
#include "raster_common.h"

namespace mvi {

struct StagedExt {
    // s_ext[e] = (hx, hy); negative hx = can never be active (opacity * 255 <= 1)
    __device__ static float2 compute(float4 co) {
        float t2 = 2.0f * __logf(255.0f * co.w);
        if (!(t2 > 0.0f)) return make_float2(-1.0f, -1.0f);
        float idet = 1.0f / (co.x * co.z - co.y * co.y);
        float hx = sqrtf(t2 * co.z * idet), hy = sqrtf(t2 * co.x * idet);
        return make_float2(hx * 1.001f + 0.05f, hy * 1.001f + 0.05f);
    }
};

__device__ __forceinline__ bool quad_overlap(float2 c, float2 ext, float qx0, float qy0) {
    // pixel centres of the quadrant span [qx0, qx0+7] x [qy0, qy0+7]
    return ext.x >= 0.0f && (c.x + ext.x >= qx0) && (c.x - ext.x <= qx0 + 7.0f) && (c.y + ext.y >= qy0) &&
           (c.y - ext.y <= qy0 + 7.0f);
}

__global__ __launch_bounds__(kBlock) void render_forward_kernel(
    Frame f, const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float2* __restrict__ xy, const float* __restrict__ rgb, const float4* __restrict__ conic_opacity,
    const float* __restrict__ depths, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
    float* __restrict__ out_color, float* __restrict__ out_depth) {
    __shared__ float2 s_xy[kBlock];
    __shared__ float2 s_ext[kBlock];
    __shared__ float4 s_co[kBlock];
    __shared__ float4 s_cd[kBlock];   // r, g, b, depth
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tile = blockIdx.y * f.gx + blockIdx.x;
    const int qx0 = blockIdx.x * kTile + 8 * (wave & 1), qy0 = blockIdx.y * kTile + 8 * (wave >> 1);
    const int pxi = qx0 + (lane & 7), pyi = qy0 + (lane >> 3);
    const bool inside = pxi < f.W && pyi < f.H;
    const float pfx = (float)pxi, pfy = (float)pyi;
    const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    int todo = (int)(r1 - r0);
    const int rounds = (todo + kBlock - 1) / kBlock;

    bool done = !inside;
    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = kDepthSentinel;
    uint32_t last = 0;

    for (int r = 0; r < rounds; ++r, todo -= kBlock) {
        if (__syncthreads_count(done) == kBlock) break;
        int progress = r * kBlock + tid;
        if (r0 + progress < r1) {
            uint32_t id = point_list[r0 + progress];
            float4 co = conic_opacity[id];
            s_xy[tid] = xy[id];
            s_co[tid] = co;
            s_ext[tid] = StagedExt::compute(co);
            s_cd[tid] = make_float4(rgb[3 * (size_t)id], rgb[3 * (size_t)id + 1], rgb[3 * (size_t)id + 2], depths[id]);
        }
        __syncthreads();
        const int n = todo < kBlock ? todo : kBlock;
        for (int c = 0; c < n; c += 64) {
            if (__ballot(!done) == 0ull) break;
            const int e = c + lane;
            const bool keep = e < n && quad_overlap(s_xy[e], s_ext[e], (float)qx0, (float)qy0);
            unsigned long long mask = __ballot(keep);
            while (mask) {
                const int j = c + __builtin_ctzll(mask);        // wave-uniform
                mask &= mask - 1;
                if (done) continue;
                float2 p = s_xy[j];
                float4 co = s_co[j];
                float dx = p.x - pfx, dy = p.y - pfy;
                float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
                if (power > 0.0f) continue;
                float alpha = fminf(kAlphaMax, co.w * __expf(power));
                if (alpha < kAlphaMin) continue;
                float test_T = T * (1.0f - alpha);
                if (test_T < kTEps) { done = true; continue; }
                float4 cd = s_cd[j];
                float w = alpha * T;
                C0 += cd.x * w; C1 += cd.y * w; C2 += cd.z * w;
                if (T > 0.5f && test_T < 0.5f) Dp = cd.w;       // median depth of the w-depth fork
                T = test_T;
                last = (uint32_t)(r * kBlock + j + 1);           // 1-based position in the tile's list
            }
        }
    }
    if (inside) {
        size_t pix = (size_t)pyi * f.W + pxi, hw = (size_t)f.H * f.W;
        final_T[pix] = T;
        n_contrib[pix] = last;
        out_color[pix] = C0 + T * f.bg[0];
        out_color[hw + pix] = C1 + T * f.bg[1];
        out_color[2 * hw + pix] = C2 + T * f.bg[2];
        out_depth[pix] = Dp;
    }
}

int launch_render_forward(const Frame& f, GeomView g, BinningView b, ImageView im, int64_t D,
                          float* out_color, float* out_depth, hipStream_t st) {
    if (f.W <= 0 || f.H <= 0) return 0;
    const uint32_t* plist = b.vals[b.passes & 1];
    hipLaunchKernelGGL(render_forward_kernel, dim3(f.gx, f.gy), dim3(kBlock), 0, st, f, im.ranges, plist, g.xy,
                       g.rgb, g.conic_opacity, g.depths, im.final_T, im.n_contrib, out_color, out_depth);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

// ------------------------------------------------------------------------------------ backward
// Back-to-front replay over the entries [0, max n_contrib of the tile) only. Each lane owns a
// pixel; per evaluated Gaussian the 9 partial gradients are summed over the wave with DPP,
// accumulated over the block's 4 waves in LDS rows of 16 floats, and flushed once per 256-entry
// batch with float atomics shaped as whole 64-byte rows (16 lanes per Gaussian, 4 Gaussians per
// wave-instruction): MI355X float atomics run at the 64-B-request rate, so one dword per row would
// be 16x slower for the same sums. Row layout of grad_rows [P][16]:
//   0 mean2D.x  1 mean2D.y  2 conic A  3 conic B  4 conic C  5 opacity  6 r  7 g  8 b  9..15 unused
constexpr int kRow = 16;

__global__ __launch_bounds__(kBlock) void render_backward_kernel(
    Frame f, const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float2* __restrict__ xy, const float* __restrict__ rgb, const float4* __restrict__ conic_opacity,
    const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dL_dpix, float* __restrict__ grad_rows) {
    __shared__ uint32_t s_id[kBlock];
    __shared__ float2 s_xy[kBlock];
    __shared__ float2 s_ext[kBlock];
    __shared__ float4 s_co[kBlock];
    __shared__ float4 s_rgb[kBlock];
    __shared__ float s_acc[kBlock][kRow];
    __shared__ uint32_t s_touched[kBlock / 32];
    __shared__ uint32_t s_blast[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tile = blockIdx.y * f.gx + blockIdx.x;
    const int qx0 = blockIdx.x * kTile + 8 * (wave & 1), qy0 = blockIdx.y * kTile + 8 * (wave >> 1);
    const int pxi = qx0 + (lane & 7), pyi = qy0 + (lane >> 3);
    const bool inside = pxi < f.W && pyi < f.H;
    const float pfx = (float)pxi, pfy = (float)pyi;
    const uint32_t r0 = ranges[2 * tile];
    const size_t pix = (size_t)pyi * f.W + pxi, hw = (size_t)f.H * f.W;

    const float T_final = inside ? final_T[pix] : 0.0f;
    float T = T_final;
    const uint32_t last = inside ? n_contrib[pix] : 0u;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, last_alpha = 0.f;
    float dp0 = 0.f, dp1 = 0.f, dp2 = 0.f;
    if (inside) { dp0 = dL_dpix[pix]; dp1 = dL_dpix[hw + pix]; dp2 = dL_dpix[2 * hw + pix]; }
    const float bg_dot = f.bg[0] * dp0 + f.bg[1] * dp1 + f.bg[2] * dp2;
    const float ddelx_dx = 0.5f * (float)f.W, ddely_dy = 0.5f * (float)f.H;
    // deepest list position composited by any pixel of this wave / of the block
    uint32_t wave_last = last;
    for (int o = 32; o > 0; o >>= 1) wave_last = max(wave_last, (uint32_t)__shfl_xor((int)wave_last, o));
    if (lane == 0) s_blast[wave] = wave_last;
    __syncthreads();
    const int total = (int)max(max(s_blast[0], s_blast[1]), max(s_blast[2], s_blast[3]));
    const int rounds = (total + kBlock - 1) / kBlock;

    for (int r = 0; r < rounds; ++r) {
        // batch r holds list positions hi-1 ... lo (descending); staged slot s <-> position hi-1-s
        const int hi = total - r * kBlock;
        const int n = hi < kBlock ? hi : kBlock;
        __syncthreads();
        if (tid < n) {
            uint32_t id = point_list[r0 + (uint32_t)(hi - 1 - tid)];
            float4 co = conic_opacity[id];
            s_id[tid] = id;
            s_xy[tid] = xy[id];
            s_co[tid] = co;
            s_ext[tid] = StagedExt::compute(co);
            s_rgb[tid] = make_float4(rgb[3 * (size_t)id], rgb[3 * (size_t)id + 1], rgb[3 * (size_t)id + 2], 0.f);
        }
#pragma unroll
        for (int c = 0; c < kRow; ++c) s_acc[tid][c] = 0.0f;
        if (tid < kBlock / 32) s_touched[tid] = 0u;
        __syncthreads();
        for (int c = 0; c < n; c += 64) {
            const int e = c + lane;
            // position of staged slot e is hi-1-e; this wave composited positions < wave_last only
            const bool keep = e < n && (uint32_t)(hi - 1 - e) < wave_last &&
                              quad_overlap(s_xy[e], s_ext[e], (float)qx0, (float)qy0);
            unsigned long long mask = __ballot(keep);
            while (mask) {
                const int j = c + __builtin_ctzll(mask);        // wave-uniform
                mask &= mask - 1;
                const uint32_t pos = (uint32_t)(hi - 1 - j);
                float g_mx = 0.f, g_my = 0.f, g_a = 0.f, g_b = 0.f, g_c = 0.f, g_o = 0.f, g_r = 0.f, g_g = 0.f, g_bl = 0.f;
                bool active = false;
                if (pos < last) {
                    float2 p = s_xy[j];
                    float4 co = s_co[j];
                    float dx = p.x - pfx, dy = p.y - pfy;
                    float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
                    if (power <= 0.0f) {
                        float G = __expf(power);
                        float alpha = fminf(kAlphaMax, co.w * G);
                        if (alpha >= kAlphaMin) {
                            active = true;
                            const float inv_1ma = __builtin_amdgcn_rcpf(1.0f - alpha);   // 1-alpha in [0.01, 1]
                            T = T * inv_1ma;
                            float dchannel = alpha * T;
                            float4 col = s_rgb[j];
                            acc0 = last_alpha * lc0 + (1.0f - last_alpha) * acc0;
                            acc1 = last_alpha * lc1 + (1.0f - last_alpha) * acc1;
                            acc2 = last_alpha * lc2 + (1.0f - last_alpha) * acc2;
                            lc0 = col.x; lc1 = col.y; lc2 = col.z;
                            float dL_dalpha = (col.x - acc0) * dp0 + (col.y - acc1) * dp1 + (col.z - acc2) * dp2;
                            g_r = dchannel * dp0; g_g = dchannel * dp1; g_bl = dchannel * dp2;
                            dL_dalpha *= T;
                            last_alpha = alpha;
                            dL_dalpha += (-T_final * inv_1ma) * bg_dot;
                            float dL_dG = co.w * dL_dalpha;
                            float gdx = G * dx, gdy = G * dy;
                            float dG_ddelx = -gdx * co.x - gdy * co.y;
                            float dG_ddely = -gdy * co.z - gdx * co.y;
                            g_mx = dL_dG * dG_ddelx * ddelx_dx;
                            g_my = dL_dG * dG_ddely * ddely_dy;
                            g_a = -0.5f * gdx * dx * dL_dG;
                            g_b = -gdx * dy * dL_dG;
                            g_c = -0.5f * gdy * dy * dL_dG;
                            g_o = G * dL_dalpha;
                        }
                    }
                }
                if (__ballot(active) == 0ull) continue;         // wave-uniform
                g_mx = wave_sum_to_lane63(g_mx); g_my = wave_sum_to_lane63(g_my);
                g_a = wave_sum_to_lane63(g_a);   g_b = wave_sum_to_lane63(g_b);   g_c = wave_sum_to_lane63(g_c);
                g_o = wave_sum_to_lane63(g_o);
                g_r = wave_sum_to_lane63(g_r);   g_g = wave_sum_to_lane63(g_g);   g_bl = wave_sum_to_lane63(g_bl);
                if (lane == 63) {
                    float* a = s_acc[j];
                    atomicAdd(a + 0, g_mx); atomicAdd(a + 1, g_my);
                    atomicAdd(a + 2, g_a);  atomicAdd(a + 3, g_b);  atomicAdd(a + 4, g_c);
                    atomicAdd(a + 5, g_o);
                    atomicAdd(a + 6, g_r);  atomicAdd(a + 7, g_g);  atomicAdd(a + 8, g_bl);
                    atomicOr(&s_touched[j >> 5], 1u << (j & 31));
                }
            }
        }
        __syncthreads();
        // flush: 16 lanes per staged Gaussian, one whole 64-byte row per atomic request
        const int comp = tid & 15;
#pragma unroll 4
        for (int it = 0; it < kBlock / 16; ++it) {
            const int e = it * 16 + (tid >> 4);
            if (e < n && ((s_touched[e >> 5] >> (e & 31)) & 1u) && comp < 9)
                atomicAdd(&grad_rows[(size_t)s_id[e] * kRow + comp], s_acc[e][comp]);
        }
    }
}

int launch_render_backward(const Frame& f, GeomView g, BinningView b, ImageView im, int64_t D,
                           const float* dL_dpix, float* grad_rows, hipStream_t st) {
    if (f.W <= 0 || f.H <= 0 || D <= 0) return 0;
    const uint32_t* plist = b.vals[b.passes & 1];
    hipLaunchKernelGGL(render_backward_kernel, dim3(f.gx, f.gy), dim3(kBlock), 0, st, f, im.ranges, plist, g.xy,
                       g.rgb, g.conic_opacity, im.final_T, im.n_contrib, dL_dpix, grad_rows);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

}  // namespace mvi
