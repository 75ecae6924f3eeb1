// Fused photometric loss of the 3DGS training loop for gfx950:
//     loss = (1 - lambda) * mean|x - y| + lambda * (1 - mean(SSIM_11x11(x, y))),   x = image * w, y = gt * w
// (gs-simp/train.py:90-93, gs-simp/inpaint_rec.py:112-125 with w = 1 - gt_mask; l1_loss / ssim / _ssim:
// gs-simp/utils/loss_utils.py:17-18, :33-62; Gaussian window sigma 1.5, zero padding 5, C1 = 0.01^2, C2 = 0.03^2)
// and its gradient with respect to `image`, in two HBM-bound launches instead of PyTorch's 5 depthwise 11x11
// convolutions + ~20 elementwise passes forward and as many backward.
//
//   stats   : 32x16 output tile per block and channel; the 42x26 halo of x and y goes to LDS once, a separable
//             11-tap pass gives mu1, mu2, E[x^2], E[y^2], E[xy]; the block adds sum(ssim_map) and sum|x - y| to its
//             partial and writes the three maps D1 = d map/d mu1, D2 = d map/d E[x^2], D3 = d map/d E[xy];
//   gradient: dL/dx[q] = (1-lambda)/N sign(x - y) - lambda/N (conv(D1) + 2 x conv(D2) + y conv(D3))[q]  (the window is
//             symmetric and the padding zero, so the adjoint of the convolution is the same convolution), times w.
// The per-block partial sums are reduced in a fixed order by one small block: the loss value is deterministic.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mvi_raster.h"
#include "../../include/mvi_train_ops.h"

namespace mvi {

int train_fail(int code, const char* msg);

constexpr int kLX = 32, kLY = 16;      // output tile (x, y): 512 pixels per 256-thread block, two per thread
constexpr int kLR = 5;                  // window radius
constexpr int kHX = kLX + 2 * kLR, kHY = kLY + 2 * kLR;      // 42 x 26: tile + halo
constexpr float kC1 = 0.01f * 0.01f, kC2 = 0.03f * 0.03f;

struct Win { float w[11]; };

// x/y value at (py, px) of channel plane `c`, zero outside the image (F.conv2d zero padding)
__device__ __forceinline__ float ld_img(const float* __restrict__ img, const float* __restrict__ wmap, int c, int py, int px,
                                        int H, int W) {
    if (py < 0 || py >= H || px < 0 || px >= W) return 0.0f;
    const size_t o = (size_t)py * W + px;
    const float v = img[(size_t)c * H * W + o];
    return wmap ? v * wmap[o] : v;
}

// Separable 11-tap passes with register sliding windows (LDS and the vector ALU, not HBM, bound these kernels): in the
// horizontal pass a thread produces 4 adjacent outputs of one halo row from 14 loaded values, in the vertical pass 2
// vertically adjacent outputs from 12 loaded rows. The five moment maps travel as packed pairs (x, y), (x^2, y^2)
// and xy alone: three fma issues per tap instead of five (v_pk_fma_f32 is full rate).
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void loss_stats_kernel(const float* __restrict__ img, const float* __restrict__ gt,
                                                         const float* __restrict__ wmap, int H, int W, Win win,
                                                         float* __restrict__ dmaps, double* __restrict__ partials) {
    __shared__ f2 s_xy[kHY][kHX + 1];                      // (x, y) of the halo
    __shared__ f2 s_h1[kHY][kLX + 1], s_h2[kHY][kLX + 1];  // row sums of (x, y) and (x^2, y^2)
    __shared__ float s_h3[kHY][kLX + 1];                   // row sums of x y
    __shared__ float s_red[2][4];
    const int tid = threadIdx.x, c = blockIdx.z;
    const int x0 = blockIdx.x * kLX, y0 = blockIdx.y * kLY;
    // halo rows: wave w takes rows w, w + 4, ...; lanes 0..41 one column each (no div/mod; block-uniform fast path
    // without bounds checks for tiles whose halo lies inside the image)
    const int lane = tid & 63, wv = tid >> 6;
    const bool interior = x0 >= kLR && x0 + kLX + kLR <= W && y0 >= kLR && y0 + kLY + kLR <= H;
    if (lane < kHX) {
        const int px = x0 + lane - kLR;
        if (interior) {
            const size_t base = ((size_t)c * H + (y0 - kLR)) * W + px, mbase = (size_t)(y0 - kLR) * W + px;
#pragma unroll
            for (int r = wv; r < kHY; r += 4) {
                const float m = wmap ? wmap[mbase + (size_t)r * W] : 1.0f;
                s_xy[r][lane] = f2{img[base + (size_t)r * W] * m, gt[base + (size_t)r * W] * m};
            }
        } else {
            for (int r = wv; r < kHY; r += 4)
                s_xy[r][lane] = f2{ld_img(img, wmap, c, y0 + r - kLR, px, H, W), ld_img(gt, wmap, c, y0 + r - kLR, px, H, W)};
        }
    }
    __syncthreads();
    if (tid < kHY * (kLX / 4)) {                           // 26 rows x 8 groups of 4 output columns
        const int r = tid / (kLX / 4), q0 = (tid % (kLX / 4)) * 4;
        f2 v[14], vv[14];
        float xy[14];
#pragma unroll
        for (int k = 0; k < 14; ++k) { v[k] = s_xy[r][q0 + k]; vv[k] = v[k] * v[k]; xy[k] = v[k].x * v[k].y; }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            f2 a = {0.f, 0.f}, aa = {0.f, 0.f};
            float ab = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const float w = win.w[k];
                a += w * v[o + k]; aa += w * vv[o + k]; ab += w * xy[o + k];
            }
            s_h1[r][q0 + o] = a; s_h2[r][q0 + o] = aa; s_h3[r][q0 + o] = ab;
        }
    }
    __syncthreads();
    const int tx = tid & 31, ty = 2 * (tid >> 5);          // outputs (ty, tx) and (ty + 1, tx)
    f2 m1[2] = {{0.f, 0.f}, {0.f, 0.f}}, m2[2] = {{0.f, 0.f}, {0.f, 0.f}};
    float m3[2] = {0.f, 0.f};
    {
        f2 c1[12], c2[12];
        float c3[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) { c1[k] = s_h1[ty + k][tx]; c2[k] = s_h2[ty + k][tx]; c3[k] = s_h3[ty + k][tx]; }
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float w = win.w[k];
            m1[0] += w * c1[k]; m2[0] += w * c2[k]; m3[0] += w * c3[k];
            m1[1] += w * c1[k + 1]; m2[1] += w * c2[k + 1]; m3[1] += w * c3[k + 1];
        }
    }
    float map_sum = 0.f, l1_sum = 0.f;
    const int px = x0 + tx;
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const int py = y0 + ty + o;
        if (px < W && py < H) {
            const float mu1 = m1[o].x, mu2 = m1[o].y, e11 = m2[o].x, e22 = m2[o].y, e12 = m3[o];
            const float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
            const float s1 = e11 - mu1s, s2 = e22 - mu2s, s12 = e12 - mu12;
            const float A1 = 2.0f * mu12 + kC1, A2 = 2.0f * s12 + kC2, B1 = mu1s + mu2s + kC1, B2 = s1 + s2 + kC2;
            const float inv = 1.0f / (B1 * B2);
            const float map = A1 * A2 * inv;
            const size_t idx = ((size_t)c * H + py) * W + px, plane = (size_t)3 * H * W;
            dmaps[idx] = (2.0f * mu2 * (A2 - A1) - 2.0f * mu1 * map * (B2 - B1)) * inv;     // d map / d mu1
            dmaps[plane + idx] = -map / B2;                                                 // d map / d E[x^2]
            dmaps[2 * plane + idx] = 2.0f * A1 * inv;                                       // d map / d E[xy]
            map_sum += map;
            const f2 ctr = s_xy[ty + o + kLR][tx + kLR];
            l1_sum += fabsf(ctr.x - ctr.y);
        }
    }
    for (int o = 32; o > 0; o >>= 1) { map_sum += __shfl_xor(map_sum, o); l1_sum += __shfl_xor(l1_sum, o); }
    if ((tid & 63) == 0) { s_red[0][tid >> 6] = map_sum; s_red[1][tid >> 6] = l1_sum; }
    __syncthreads();
    if (tid == 0) {
        const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partials[2 * b] = (double)s_red[0][0] + (double)s_red[0][1] + (double)s_red[0][2] + (double)s_red[0][3];
        partials[2 * b + 1] = (double)s_red[1][0] + (double)s_red[1][1] + (double)s_red[1][2] + (double)s_red[1][3];
    }
}

// out[0] = loss, out[1] = mean |x - y|, out[2] = mean ssim; fixed summation order
__global__ __launch_bounds__(1024) void loss_reduce_kernel(const double* __restrict__ partials, int nblocks, double n_elem,
                                                           float lambda, float* __restrict__ out) {
    __shared__ double s_a[1024], s_b[1024];
    const double2* p2 = reinterpret_cast<const double2*>(partials);
    double a = 0.0, b = 0.0;
    for (int i0 = threadIdx.x; i0 < nblocks; i0 += 8 * 1024) {          // 8 independent loads in flight per lane
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * 1024;
            v[u] = i < nblocks ? p2[i] : double2{0.0, 0.0};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { a += v[u].x; b += v[u].y; }
    }
    s_a[threadIdx.x] = a; s_b[threadIdx.x] = b;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { s_a[threadIdx.x] += s_a[threadIdx.x + o]; s_b[threadIdx.x] += s_b[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double ssim = s_a[0] / n_elem, l1 = s_b[0] / n_elem;
        out[0] = (float)((1.0 - (double)lambda) * l1 + (double)lambda * (1.0 - ssim));
        out[1] = (float)l1;
        out[2] = (float)ssim;
    }
}

// kTwo = false: dL/dimage of upstream * loss(lambda) (host scalars). kTwo = true: w2[0] * d mean|x - y| + w2[1] * d mean SSIM with the
// two weights read from DEVICE memory — the backward of a caller that combines the two means itself (train.py:91-92 as written:
// the upstream gradients of autograd are device scalars; reading them back would stall the loop).
template <bool kTwo>
__global__ __launch_bounds__(256) void loss_grad_kernel(const float* __restrict__ img, const float* __restrict__ gt,
                                                        const float* __restrict__ wmap, const float* __restrict__ dmaps,
                                                        int H, int W, Win win, float lambda, float upstream,
                                                        const float* __restrict__ w2, float* __restrict__ dL_dimg) {
    __shared__ f2 s_d12[kHY][kHX + 1];                     // (D1, D2) of the halo
    __shared__ float s_d3[kHY][kHX + 1];
    __shared__ f2 s_h12[kHY][kLX + 1];
    __shared__ float s_h3[kHY][kLX + 1];
    const int tid = threadIdx.x, c = blockIdx.z;
    const int x0 = blockIdx.x * kLX, y0 = blockIdx.y * kLY;
    const size_t plane = (size_t)3 * H * W;
    const int lane = tid & 63, wv = tid >> 6;
    const bool interior = x0 >= kLR && x0 + kLX + kLR <= W && y0 >= kLR && y0 + kLY + kLR <= H;
    if (lane < kHX) {
        const int px = x0 + lane - kLR;
        if (interior) {
            const size_t base = ((size_t)c * H + (y0 - kLR)) * W + px;
#pragma unroll
            for (int r = wv; r < kHY; r += 4) {
                const size_t o = base + (size_t)r * W;
                s_d12[r][lane] = f2{dmaps[o], dmaps[plane + o]};
                s_d3[r][lane] = dmaps[2 * plane + o];
            }
        } else {
            for (int r = wv; r < kHY; r += 4) {
                const int py = y0 + r - kLR;
                const bool in = py >= 0 && py < H && px >= 0 && px < W;
                const size_t o = ((size_t)c * H + (in ? py : 0)) * W + (in ? px : 0);
                s_d12[r][lane] = in ? f2{dmaps[o], dmaps[plane + o]} : f2{0.0f, 0.0f};
                s_d3[r][lane] = in ? dmaps[2 * plane + o] : 0.0f;
            }
        }
    }
    __syncthreads();
    if (tid < kHY * (kLX / 4)) {
        const int r = tid / (kLX / 4), q0 = (tid % (kLX / 4)) * 4;
        f2 v[14];
        float v3[14];
#pragma unroll
        for (int k = 0; k < 14; ++k) { v[k] = s_d12[r][q0 + k]; v3[k] = s_d3[r][q0 + k]; }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            f2 a = {0.f, 0.f};
            float a3 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) { a += win.w[k] * v[o + k]; a3 += win.w[k] * v3[o + k]; }
            s_h12[r][q0 + o] = a; s_h3[r][q0 + o] = a3;
        }
    }
    __syncthreads();
    const int tx = tid & 31, ty = 2 * (tid >> 5);
    f2 g12[2] = {{0.f, 0.f}, {0.f, 0.f}};
    float g3[2] = {0.f, 0.f};
    {
        f2 c12[12];
        float c3[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) { c12[k] = s_h12[ty + k][tx]; c3[k] = s_h3[ty + k][tx]; }
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float w = win.w[k];
            g12[0] += w * c12[k]; g3[0] += w * c3[k];
            g12[1] += w * c12[k + 1]; g3[1] += w * c3[k + 1];
        }
    }
    const int px = x0 + tx;
    const float inv_n = 1.0f / (3.0f * (float)H * (float)W);
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const int py = y0 + ty + o;
        if (px >= W || py >= H) continue;
        const size_t o2 = (size_t)py * W + px, idx = (size_t)c * H * W + o2;
        const float wv = wmap ? wmap[o2] : 1.0f;
        const float x = img[idx] * wv, y = gt[idx] * wv;
        const float sgn = x > y ? 1.0f : (x < y ? -1.0f : 0.0f);
        const float conv = g12[o].x + 2.0f * x * g12[o].y + y * g3[o];
        if (kTwo) {
            dL_dimg[idx] = (w2[0] * inv_n * sgn + w2[1] * inv_n * conv) * wv;
        } else {
            const float g = (1.0f - lambda) * inv_n * sgn - lambda * inv_n * conv;
            dL_dimg[idx] = upstream * g * wv;
        }
    }
}

}  // namespace mvi

using namespace mvi;

static Win make_window() {
    Win w;
    double s = 0.0, g[11];
    for (int i = 0; i < 11; ++i) { g[i] = exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); s += g[i]; }
    // loss_utils.py:23-25 builds the window in fp32 (torch.Tensor of Python floats, divided by its fp32 sum)
    float gf[11], sf = 0.0f;
    for (int i = 0; i < 11; ++i) { gf[i] = (float)g[i]; }
    for (int i = 0; i < 11; ++i) sf += gf[i];
    for (int i = 0; i < 11; ++i) w.w[i] = gf[i] / sf;
    (void)s;
    return w;
}

extern "C" size_t mvi_photometric_loss_workspace_bytes(int32_t H, int32_t W) {
    if (H <= 0 || W <= 0) return 0;
    const size_t blocks = (size_t)((W + kLX - 1) / kLX) * ((H + kLY - 1) / kLY) * 3;
    return (size_t)9 * H * W * sizeof(float) + blocks * 2 * sizeof(double) + 256;
}

// what: 1 = the two forward launches (statistics + reduction), 2 = the gradient launch, 3 = both
static int photometric_impl(int what, const float* image, const float* gt, const float* weight, int32_t H, int32_t W, float lambda_dssim,
                            float upstream, const float* w2, float* out3, float* dL_dimage, void* workspace, size_t workspace_bytes,
                            void* stream) {
    if (H <= 0 || W <= 0) return train_fail(MVI_EINVAL, "photometric_loss: empty image");
    if (!image || !gt || !workspace || ((what & 1) && !out3)) return train_fail(MVI_EINVAL, "photometric_loss: NULL pointer");
    if (workspace_bytes < mvi_photometric_loss_workspace_bytes(H, W))
        return train_fail(MVI_ENOMEM, "photometric_loss: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const Win win = make_window();
    const dim3 grid((W + kLX - 1) / kLX, (H + kLY - 1) / kLY, 3);
    const int nblocks = (int)(grid.x * grid.y * grid.z);
    // workspace: [partials: nblocks x 2 double, padded to 256 B][dmaps: 9 H W float]
    double* partials = (double*)workspace;
    size_t off = ((size_t)nblocks * 2 * sizeof(double) + 255) / 256 * 256;
    float* dmaps = (float*)((char*)workspace + off);
    if (what & 1) {
        hipLaunchKernelGGL(loss_stats_kernel, grid, dim3(256), 0, st, image, gt, weight, H, W, win, dmaps, partials);
        hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(1024), 0, st, partials, nblocks, 3.0 * (double)H * (double)W,
                           lambda_dssim, out3);
    }
    if ((what & 2) && dL_dimage) {
        if (w2)
            hipLaunchKernelGGL(loss_grad_kernel<true>, grid, dim3(256), 0, st, image, gt, weight, dmaps, H, W, win, 0.0f, 1.0f, w2, dL_dimage);
        else
            hipLaunchKernelGGL(loss_grad_kernel<false>, grid, dim3(256), 0, st, image, gt, weight, dmaps, H, W, win, lambda_dssim,
                               upstream, (const float*)nullptr, dL_dimage);
    }
    return hipGetLastError() == hipSuccess ? MVI_OK : train_fail(MVI_EHIP, "photometric_loss: kernel launch failed");
}

extern "C" int mvi_photometric_loss(const float* image, const float* gt, const float* weight, int32_t H, int32_t W,
                                    float lambda_dssim, float upstream, float* out3, float* dL_dimage, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    return photometric_impl(3, image, gt, weight, H, W, lambda_dssim, upstream, nullptr, out3, dL_dimage, workspace, workspace_bytes, stream);
}

extern "C" int mvi_photometric_loss_stats(const float* image, const float* gt, const float* weight, int32_t H, int32_t W, float* out3,
                                          void* workspace, size_t workspace_bytes, void* stream) {
    return photometric_impl(1, image, gt, weight, H, W, 0.0f, 1.0f, nullptr, out3, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int mvi_photometric_loss_grad2(const float* image, const float* gt, const float* weight, int32_t H, int32_t W,
                                          const float* weights2, float* dL_dimage, void* workspace, size_t workspace_bytes, void* stream) {
    if (!weights2 || !dL_dimage) return train_fail(MVI_EINVAL, "photometric_loss_grad2: NULL pointer");
    return photometric_impl(2, image, gt, weight, H, W, 0.0f, 1.0f, weights2, nullptr, dL_dimage, workspace, workspace_bytes, stream);
}

static thread_local char g_terr[256] = "";
extern "C" const char* mvi_train_last_error(void) { return g_terr; }
namespace mvi {
int train_fail(int code, const char* msg) { snprintf(g_terr, sizeof(g_terr), "%s", msg); return code; }
}
