// Elementwise ops of the 3DGS training step for gfx950 — all HBM-bound, one pass each:
//
//   adam_step_kernel      : torch.optim.Adam(lr, betas, eps, weight_decay=0, amsgrad=False) over up to 8 parameter
//                           tensors in ONE launch (gs-simp/scene/gaussian_model.py:154-163: six groups, lr per group,
//                           eps 1e-15; stepped at gs-simp/train.py:126-128). 28 bytes of traffic per parameter
//                           (read p, g, m, v; write p, m, v) instead of the ~10 passes of the unfused formula.
//   gaussian_activations  : scales = exp(_scaling), rotations = normalize(_rotation), opacities = sigmoid(_opacity),
//                           shs = cat(_features_dc, _features_rest, dim=1) — the activated views the renderer reads
//                           (gaussian_model.py:95-115, gaussian_renderer/__init__.py:56-79) in one launch, and their
//                           chain rule in one launch on the way back (PyTorch: 5 forward + ~12 backward kernels, the
//                           cat alone moves 2 x 192 B per Gaussian).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mvi_raster.h"
#include "../../include/mvi_train_ops.h"

namespace mvi {

int train_fail(int code, const char* msg);

struct AdamTable {
    mvi_adam_group g[MVI_ADAM_MAX_GROUPS];
    int64_t first_block[MVI_ADAM_MAX_GROUPS + 1];          // prefix of per-group block counts
    int n;
};

constexpr int kAdamPerThread = 4;                          // elements per thread (one float4 when aligned)
constexpr int kAdamBlock = 256;

// Same operation order as torch.optim.adam._single_tensor_adam: exp_avg.lerp_(grad, 1 - beta1);
// exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2); denom = sqrt(exp_avg_sq) / sqrt(bc2) + eps;
// param.addcdiv_(exp_avg, denom, value = -lr / bc1).
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float lr_over_bc1, float one_m_b1, float b2,
                                         float one_m_b2, float inv_bc2_sqrt, float eps) {
    m = m + one_m_b1 * (g - m);
    v = v * b2 + one_m_b2 * (g * g);
    const float denom = sqrtf(v) * inv_bc2_sqrt + eps;
    p = p - lr_over_bc1 * (m / denom);
}

__global__ __launch_bounds__(kAdamBlock) void adam_step_kernel(AdamTable t, float one_m_b1, float b2, float one_m_b2,
                                                               float inv_bc1, float inv_bc2_sqrt, float eps) {
    int gi = 0;
#pragma unroll
    for (int k = 1; k < MVI_ADAM_MAX_GROUPS; ++k) gi += (k < t.n && (int64_t)blockIdx.x >= t.first_block[k]) ? 1 : 0;
    const mvi_adam_group G = t.g[gi];
    const int64_t e0 = (((int64_t)blockIdx.x - t.first_block[gi]) * kAdamBlock + threadIdx.x) * kAdamPerThread;
    if (e0 >= G.n) return;
    const float lr1 = G.lr * inv_bc1;
    const bool vec = e0 + kAdamPerThread <= G.n &&
                     ((((uintptr_t)G.param | (uintptr_t)G.grad | (uintptr_t)G.exp_avg | (uintptr_t)G.exp_avg_sq) & 15) == 0);
    if (vec) {
        float4 p = *reinterpret_cast<float4*>(G.param + e0), m = *reinterpret_cast<float4*>(G.exp_avg + e0),
               v = *reinterpret_cast<float4*>(G.exp_avg_sq + e0);
        const float4 g = *reinterpret_cast<const float4*>(G.grad + e0);
        adam_one(p.x, g.x, m.x, v.x, lr1, one_m_b1, b2, one_m_b2, inv_bc2_sqrt, eps);
        adam_one(p.y, g.y, m.y, v.y, lr1, one_m_b1, b2, one_m_b2, inv_bc2_sqrt, eps);
        adam_one(p.z, g.z, m.z, v.z, lr1, one_m_b1, b2, one_m_b2, inv_bc2_sqrt, eps);
        adam_one(p.w, g.w, m.w, v.w, lr1, one_m_b1, b2, one_m_b2, inv_bc2_sqrt, eps);
        *reinterpret_cast<float4*>(G.param + e0) = p;
        *reinterpret_cast<float4*>(G.exp_avg + e0) = m;
        *reinterpret_cast<float4*>(G.exp_avg_sq + e0) = v;
    } else {
        for (int64_t e = e0; e < e0 + kAdamPerThread && e < G.n; ++e) {
            float p = G.param[e], m = G.exp_avg[e], v = G.exp_avg_sq[e];
            adam_one(p, G.grad[e], m, v, lr1, one_m_b1, b2, one_m_b2, inv_bc2_sqrt, eps);
            G.param[e] = p; G.exp_avg[e] = m; G.exp_avg_sq[e] = v;
        }
    }
}

// ---- activations -----------------------------------------------------------------------------------------------
// One thread per Gaussian for the small attributes; the SH concat is a flat float copy done by all threads.
template <int ROW>                       // ROW = 3 M floats per Gaussian (0 = run-time value): constant divisions
__global__ __launch_bounds__(256) void gaussian_activations_kernel(int P, int M, const float* __restrict__ raw_scale,
                                                                   const float* __restrict__ raw_rot,
                                                                   const float* __restrict__ raw_opacity,
                                                                   const float* __restrict__ f_dc,
                                                                   const float* __restrict__ f_rest, float* __restrict__ scales,
                                                                   float* __restrict__ rots, float* __restrict__ opac,
                                                                   float* __restrict__ shs) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < P) {
#pragma unroll
        for (int k = 0; k < 3; ++k) scales[3 * i + k] = expf(raw_scale[3 * i + k]);
        const float4 q = *reinterpret_cast<const float4*>(raw_rot + 4 * i);
        const float n = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);     // F.normalize eps
        *reinterpret_cast<float4*>(rots + 4 * i) = make_float4(q.x / n, q.y / n, q.z / n, q.w / n);
        opac[i] = 1.0f / (1.0f + expf(-raw_opacity[i]));
    }
    // shs[g][0][:] = f_dc[g][0][:], shs[g][1..M-1][:] = f_rest[g][:][:]: consecutive lanes write consecutive floats
    // (32-bit index math: P * 3M < 2^31 is checked on the host)
    const uint32_t row = ROW ? ROW : 3u * (uint32_t)M;
    const uint32_t total = (uint32_t)P * row, stride = gridDim.x * 256u;
    for (uint32_t e = (uint32_t)i; e < total; e += stride) {
        const uint32_t g = e / row, k = e - g * row;
        shs[e] = k < 3 ? f_dc[3 * g + k] : f_rest[g * (row - 3) + (k - 3)];
    }
}

template <int ROW>
__global__ __launch_bounds__(256) void gaussian_activations_backward_kernel(
    int P, int M, const float* __restrict__ raw_rot, const float* __restrict__ scales, const float* __restrict__ opac,
    const float* __restrict__ d_scales, const float* __restrict__ d_rots, const float* __restrict__ d_opac,
    const float* __restrict__ d_shs, float* __restrict__ d_raw_scale, float* __restrict__ d_raw_rot,
    float* __restrict__ d_raw_opacity, float* __restrict__ d_f_dc, float* __restrict__ d_f_rest) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < P) {
#pragma unroll
        for (int k = 0; k < 3; ++k) d_raw_scale[3 * i + k] = d_scales[3 * i + k] * scales[3 * i + k];      // d exp = exp
        const float4 q = *reinterpret_cast<const float4*>(raw_rot + 4 * i);
        const float4 g = *reinterpret_cast<const float4*>(d_rots + 4 * i);
        const float nrm = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
        float4 o;
        if (nrm > 1e-12f) {                                 // y = q / |q|: dq = (g - y (y . g)) / |q|
            const float inv = 1.0f / nrm;
            const float yx = q.x * inv, yy = q.y * inv, yz = q.z * inv, yw = q.w * inv;
            const float dot = yx * g.x + yy * g.y + yz * g.z + yw * g.w;
            o = make_float4((g.x - yx * dot) * inv, (g.y - yy * dot) * inv, (g.z - yz * dot) * inv, (g.w - yw * dot) * inv);
        } else {                                            // clamped denominator: y = q / eps
            o = make_float4(g.x * 1e12f, g.y * 1e12f, g.z * 1e12f, g.w * 1e12f);
        }
        *reinterpret_cast<float4*>(d_raw_rot + 4 * i) = o;
        const float s = opac[i];
        d_raw_opacity[i] = d_opac[i] * s * (1.0f - s);
    }
    const uint32_t row = ROW ? ROW : 3u * (uint32_t)M;
    const uint32_t total = (uint32_t)P * row, stride = gridDim.x * 256u;
    for (uint32_t e = (uint32_t)i; e < total; e += stride) {
        const uint32_t g = e / row, k = e - g * row;
        const float v = d_shs[e];
        if (k < 3) d_f_dc[3 * g + k] = v; else d_f_rest[g * (row - 3) + (k - 3)] = v;
    }
}

}  // namespace mvi

using namespace mvi;

extern "C" int mvi_adam_step(const mvi_adam_group* groups_host, int32_t n_groups, double beta1, double beta2, double eps,
                             int32_t step, void* stream) {
    if (n_groups < 0 || n_groups > MVI_ADAM_MAX_GROUPS) return train_fail(MVI_EINVAL, "adam_step: 0..8 groups per call");
    if (step < 1) return train_fail(MVI_EINVAL, "adam_step: step counts from 1");
    if (n_groups == 0) return MVI_OK;
    if (!groups_host) return train_fail(MVI_EINVAL, "adam_step: NULL group table");
    AdamTable t;
    t.n = 0;
    int64_t blocks = 0;
    for (int i = 0; i < n_groups; ++i) {
        const mvi_adam_group& g = groups_host[i];
        if (g.n < 0) return train_fail(MVI_EINVAL, "adam_step: negative element count");
        if (g.n == 0) continue;
        if (!g.param || !g.grad || !g.exp_avg || !g.exp_avg_sq) return train_fail(MVI_EINVAL, "adam_step: NULL tensor pointer");
        t.g[t.n] = g;
        t.first_block[t.n] = blocks;
        blocks += (g.n + (int64_t)kAdamBlock * kAdamPerThread - 1) / ((int64_t)kAdamBlock * kAdamPerThread);
        ++t.n;
    }
    for (int i = t.n; i <= MVI_ADAM_MAX_GROUPS; ++i) t.first_block[i] = blocks;
    if (t.n == 0) return MVI_OK;
    if (blocks > 0x7FFFFFFFll) return train_fail(MVI_EINVAL, "adam_step: too many elements for one launch");
    // bias corrections in double like Python's float arithmetic in torch.optim.Adam, then rounded once
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)blocks), dim3(kAdamBlock), 0, (hipStream_t)stream, t,
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)),
                       (float)eps);
    return hipGetLastError() == hipSuccess ? MVI_OK : train_fail(MVI_EHIP, "adam_step: kernel launch failed");
}

extern "C" int mvi_gaussian_activations(int32_t P, int32_t M, const float* raw_scaling, const float* raw_rotation,
                                        const float* raw_opacity, const float* features_dc, const float* features_rest,
                                        float* scales, float* rotations, float* opacities, float* shs, void* stream) {
    if (P < 0 || M < 1) return train_fail(MVI_EINVAL, "gaussian_activations: bad shape");
    if (P == 0) return MVI_OK;
    if (!raw_scaling || !raw_rotation || !raw_opacity || !features_dc || (M > 1 && !features_rest) || !scales || !rotations ||
        !opacities || !shs)
        return train_fail(MVI_EINVAL, "gaussian_activations: NULL pointer");
    if ((int64_t)P * 3 * M >= 0x7FFFFFFFll) return train_fail(MVI_EINVAL, "gaussian_activations: P * 3M must fit 31 bits");
    const dim3 grid((unsigned)((P + 255) / 256) * 4), blk(256);
#define MVI_ACT(R)                                                                                                      \
    hipLaunchKernelGGL((gaussian_activations_kernel<R>), grid, blk, 0, (hipStream_t)stream, P, M, raw_scaling, raw_rotation, \
                       raw_opacity, features_dc, features_rest, scales, rotations, opacities, shs)
    switch (M) {
        case 1: MVI_ACT(3); break;
        case 4: MVI_ACT(12); break;
        case 9: MVI_ACT(27); break;
        case 16: MVI_ACT(48); break;
        default: MVI_ACT(0); break;
    }
#undef MVI_ACT
    return hipGetLastError() == hipSuccess ? MVI_OK : train_fail(MVI_EHIP, "gaussian_activations: kernel launch failed");
}

extern "C" int mvi_gaussian_activations_backward(int32_t P, int32_t M, const float* raw_rotation, const float* scales,
                                                 const float* opacities, const float* dL_dscales,
                                                 const float* dL_drotations, const float* dL_dopacities,
                                                 const float* dL_dshs, float* dL_draw_scaling, float* dL_draw_rotation,
                                                 float* dL_draw_opacity, float* dL_dfeatures_dc, float* dL_dfeatures_rest,
                                                 void* stream) {
    if (P < 0 || M < 1) return train_fail(MVI_EINVAL, "gaussian_activations_backward: bad shape");
    if (P == 0) return MVI_OK;
    if (!raw_rotation || !scales || !opacities || !dL_dscales || !dL_drotations || !dL_dopacities || !dL_dshs ||
        !dL_draw_scaling || !dL_draw_rotation || !dL_draw_opacity || !dL_dfeatures_dc || (M > 1 && !dL_dfeatures_rest))
        return train_fail(MVI_EINVAL, "gaussian_activations_backward: NULL pointer");
    if ((int64_t)P * 3 * M >= 0x7FFFFFFFll) return train_fail(MVI_EINVAL, "gaussian_activations_backward: P * 3M must fit 31 bits");
    const dim3 grid((unsigned)((P + 255) / 256) * 4), blk(256);
#define MVI_ACTB(R)                                                                                                          \
    hipLaunchKernelGGL((gaussian_activations_backward_kernel<R>), grid, blk, 0, (hipStream_t)stream, P, M, raw_rotation, scales, \
                       opacities, dL_dscales, dL_drotations, dL_dopacities, dL_dshs, dL_draw_scaling, dL_draw_rotation,           \
                       dL_draw_opacity, dL_dfeatures_dc, dL_dfeatures_rest)
    switch (M) {
        case 1: MVI_ACTB(3); break;
        case 4: MVI_ACTB(12); break;
        case 9: MVI_ACTB(27); break;
        case 16: MVI_ACTB(48); break;
        default: MVI_ACTB(0); break;
    }
#undef MVI_ACTB
    return hipGetLastError() == hipSuccess ? MVI_OK : train_fail(MVI_EHIP, "gaussian_activations_backward: kernel launch failed");
}
