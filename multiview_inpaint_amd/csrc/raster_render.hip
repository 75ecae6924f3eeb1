// Per-tile alpha compositing with median depth, forward and backward, for gfx950.
// Replaces the plug-in's render stages behind gs-simp/gaussian_renderer/__init__.py:85-93
// (forward: colour [3,H,W] + depth [1,H,W]) and loss.backward() (gs-simp/train.py:93).
// One 256-thread block (4 wave64) per 16x16-pixel tile; the tile's depth-sorted Gaussian list is
// staged through LDS 256 entries at a time and read back as wave-uniform broadcasts.
#include "raster_common.h"

namespace mvi {

__global__ __launch_bounds__(kBlock) void render_forward_kernel(
    Frame f, const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float2* __restrict__ xy, const float* __restrict__ rgb, const float4* __restrict__ conic_opacity,
    const float* __restrict__ depths, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
    float* __restrict__ out_color, float* __restrict__ out_depth) {
    __shared__ float2 s_xy[kBlock];
    __shared__ float4 s_co[kBlock];
    __shared__ float4 s_cd[kBlock];   // r, g, b, depth
    const int tid = threadIdx.x;
    const int tile = blockIdx.y * f.gx + blockIdx.x;
    const int pxi = blockIdx.x * kTile + (tid & 15), pyi = blockIdx.y * kTile + (tid >> 4);
    const bool inside = pxi < f.W && pyi < f.H;
    const float pfx = (float)pxi, pfy = (float)pyi;
    const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    int todo = (int)(r1 - r0);
    const int rounds = (todo + kBlock - 1) / kBlock;

    bool done = !inside;
    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = kDepthSentinel;
    uint32_t contributor = 0, last = 0;

    for (int r = 0; r < rounds; ++r, todo -= kBlock) {
        if (__syncthreads_count(done) == kBlock) break;
        int progress = r * kBlock + tid;
        if (r0 + progress < r1) {
            uint32_t id = point_list[r0 + progress];
            s_xy[tid] = xy[id];
            s_co[tid] = conic_opacity[id];
            s_cd[tid] = make_float4(rgb[3 * (size_t)id], rgb[3 * (size_t)id + 1], rgb[3 * (size_t)id + 2], depths[id]);
        }
        __syncthreads();
        const int n = todo < kBlock ? todo : kBlock;
        for (int j = 0; !done && j < n; ++j) {
            ++contributor;
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
            if (T > 0.5f && test_T < 0.5f) Dp = cd.w;   // median depth of the w-depth fork
            T = test_T;
            last = contributor;
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
// Back-to-front replay. Each lane owns a pixel; per staged Gaussian the 9 partial gradients are
// summed over the wave with DPP, accumulated across the block's 4 waves in LDS, and flushed with
// one global atomic per (tile, Gaussian, component) at the end of each 256-entry batch.
constexpr int kAcc = 10;  // mean2D.x, mean2D.y, conic A, B, C, opacity, r, g, b, (pad)

__global__ __launch_bounds__(kBlock) void render_backward_kernel(
    Frame f, const uint32_t* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float2* __restrict__ xy, const float* __restrict__ rgb, const float4* __restrict__ conic_opacity,
    const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dL_dpix, float* __restrict__ dL_dmean2D, float4* __restrict__ dL_dconic_op,
    float* __restrict__ dL_dcolor) {
    __shared__ uint32_t s_id[kBlock];
    __shared__ float2 s_xy[kBlock];
    __shared__ float4 s_co[kBlock];
    __shared__ float s_rgb[3][kBlock];
    __shared__ float s_acc[kBlock][kAcc + 1];   // +1 pad: thread t flushes row t
    const int tid = threadIdx.x, lane = tid & 63;
    const int tile = blockIdx.y * f.gx + blockIdx.x;
    const int pxi = blockIdx.x * kTile + (tid & 15), pyi = blockIdx.y * kTile + (tid >> 4);
    const bool inside = pxi < f.W && pyi < f.H;
    const float pfx = (float)pxi, pfy = (float)pyi;
    const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    int todo = (int)(r1 - r0);
    const int rounds = (todo + kBlock - 1) / kBlock;
    const size_t pix = (size_t)pyi * f.W + pxi, hw = (size_t)f.H * f.W;

    const float T_final = inside ? final_T[pix] : 0.0f;
    float T = T_final;
    uint32_t contributor = (uint32_t)todo;
    const uint32_t last = inside ? n_contrib[pix] : 0u;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, last_alpha = 0.f;
    float dp0 = 0.f, dp1 = 0.f, dp2 = 0.f;
    if (inside) { dp0 = dL_dpix[pix]; dp1 = dL_dpix[hw + pix]; dp2 = dL_dpix[2 * hw + pix]; }
    const float bg_dot = f.bg[0] * dp0 + f.bg[1] * dp1 + f.bg[2] * dp2;
    const float ddelx_dx = 0.5f * (float)f.W, ddely_dy = 0.5f * (float)f.H;
    // highest list position any pixel of this wave composited: entries behind it are skipped by the
    // whole wave without touching the reduction
    uint32_t wave_last = last;
    for (int o = 32; o > 0; o >>= 1) wave_last = max(wave_last, (uint32_t)__shfl_xor((int)wave_last, o));

    for (int r = 0; r < rounds; ++r, todo -= kBlock) {
        __syncthreads();
        int progress = r * kBlock + tid;
        if (r0 + progress < r1) {
            uint32_t id = point_list[r1 - progress - 1];
            s_id[tid] = id;
            s_xy[tid] = xy[id];
            s_co[tid] = conic_opacity[id];
            s_rgb[0][tid] = rgb[3 * (size_t)id];
            s_rgb[1][tid] = rgb[3 * (size_t)id + 1];
            s_rgb[2][tid] = rgb[3 * (size_t)id + 2];
        }
#pragma unroll
        for (int c = 0; c < kAcc; ++c) s_acc[tid][c] = 0.0f;
        __syncthreads();
        const int n = todo < kBlock ? todo : kBlock;
        for (int j = 0; j < n; ++j) {
            --contributor;
            if (contributor >= wave_last) continue;       // wave-uniform
            float g_mx = 0.f, g_my = 0.f, g_a = 0.f, g_b = 0.f, g_c = 0.f, g_o = 0.f, g_r = 0.f, g_g = 0.f, g_bl = 0.f;
            bool active = false;
            if (contributor < last) {
                float2 p = s_xy[j];
                float4 co = s_co[j];
                float dx = p.x - pfx, dy = p.y - pfy;
                float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
                if (power <= 0.0f) {
                    float G = __expf(power);
                    float alpha = fminf(kAlphaMax, co.w * G);
                    if (alpha >= kAlphaMin) {
                        active = true;
                        T = T / (1.0f - alpha);
                        float dchannel = alpha * T;
                        float c0 = s_rgb[0][j], c1 = s_rgb[1][j], c2 = s_rgb[2][j];
                        acc0 = last_alpha * lc0 + (1.0f - last_alpha) * acc0;
                        acc1 = last_alpha * lc1 + (1.0f - last_alpha) * acc1;
                        acc2 = last_alpha * lc2 + (1.0f - last_alpha) * acc2;
                        lc0 = c0; lc1 = c1; lc2 = c2;
                        float dL_dalpha = (c0 - acc0) * dp0 + (c1 - acc1) * dp1 + (c2 - acc2) * dp2;
                        g_r = dchannel * dp0; g_g = dchannel * dp1; g_bl = dchannel * dp2;
                        dL_dalpha *= T;
                        last_alpha = alpha;
                        dL_dalpha += (-T_final / (1.0f - alpha)) * bg_dot;
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
            if (__ballot(active) == 0ull) continue;       // wave-uniform
            g_mx = wave_sum_to_lane63(g_mx); g_my = wave_sum_to_lane63(g_my);
            g_a = wave_sum_to_lane63(g_a);   g_b = wave_sum_to_lane63(g_b);   g_c = wave_sum_to_lane63(g_c);
            g_o = wave_sum_to_lane63(g_o);
            g_r = wave_sum_to_lane63(g_r);   g_g = wave_sum_to_lane63(g_g);   g_bl = wave_sum_to_lane63(g_bl);
            if (lane == 63) {
                atomicAdd(&s_acc[j][0], g_mx); atomicAdd(&s_acc[j][1], g_my);
                atomicAdd(&s_acc[j][2], g_a);  atomicAdd(&s_acc[j][3], g_b);  atomicAdd(&s_acc[j][4], g_c);
                atomicAdd(&s_acc[j][5], g_o);
                atomicAdd(&s_acc[j][6], g_r);  atomicAdd(&s_acc[j][7], g_g);  atomicAdd(&s_acc[j][8], g_bl);
            }
        }
        __syncthreads();
        if (tid < n) {
            const uint32_t id = s_id[tid];
            const float* a = s_acc[tid];
            bool any = false;
#pragma unroll
            for (int c = 0; c < 9; ++c) any |= (a[c] != 0.0f);
            if (any) {
                atomicAdd(&dL_dmean2D[3 * (size_t)id], a[0]);
                atomicAdd(&dL_dmean2D[3 * (size_t)id + 1], a[1]);
                float* co = reinterpret_cast<float*>(dL_dconic_op + id);
                atomicAdd(co, a[2]); atomicAdd(co + 1, a[3]); atomicAdd(co + 2, a[4]); atomicAdd(co + 3, a[5]);
                atomicAdd(&dL_dcolor[3 * (size_t)id], a[6]);
                atomicAdd(&dL_dcolor[3 * (size_t)id + 1], a[7]);
                atomicAdd(&dL_dcolor[3 * (size_t)id + 2], a[8]);
            }
        }
    }
}

int launch_render_backward(const Frame& f, GeomView g, BinningView b, ImageView im, int64_t D,
                           const float* dL_dpix, float* dL_dmean2D, float4* dL_dconic_op,
                           float* dL_dcolor, hipStream_t st) {
    if (f.W <= 0 || f.H <= 0 || D <= 0) return 0;
    const uint32_t* plist = b.vals[b.passes & 1];
    hipLaunchKernelGGL(render_backward_kernel, dim3(f.gx, f.gy), dim3(kBlock), 0, st, f, im.ranges, plist, g.xy,
                       g.rgb, g.conic_opacity, im.final_T, im.n_contrib, dL_dpix, dL_dmean2D, dL_dconic_op,
                       dL_dcolor);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

}  // namespace mvi
