// Per-Gaussian preprocess, forward and backward, for gfx950.
// Replaces the plug-in's preprocess stages behind gs-simp/gaussian_renderer/__init__.py:85-93:
// frustum cull, cov3D from scale/quaternion (gs-simp/utils/general_utils.py:66-112 conventions),
// EWA cov2D, conic, 3-sigma radius, tile rectangle, SH colour (gs-simp/utils/sh_utils.py:57-112).
// One thread per Gaussian, kPB-thread blocks; each block also emits the sum of tiles touched so
// the binning stage only has to scan ceil(P/kPB) block sums.
//
// HBM access: the caller's AoS arrays with a record that is not a power of two (shs [P,M,3] = 192 B
// per Gaussian at degree 3, means3D/scales [P,3], the gradient outputs of the same shapes) are moved
// between HBM and the block through LDS with fully coalesced 16-byte (or 4-byte) lane accesses — a
// block's kPB records are one contiguous span — and each thread then reads / writes its own record
// from LDS with an odd row stride (conflict-free). Per-lane strided global accesses of those arrays
// cost 2.4x the bytes on reads and 3.3x on writes (measured: FETCH_SIZE / WRITE_SIZE).
#include "raster_common.h"

namespace mvi {

// Copies the block's `n_rec` records of `rec` floats each from `src` (contiguous) into LDS rows of
// `stride` floats. All threads of the block must call it; caller synchronises afterwards.
__device__ __forceinline__ void stage_in(const float* __restrict__ src, float* lds, int n_rec, int rec, int stride) {
    const int total = n_rec * rec;
    if ((rec & 3) == 0) {
        // float4 per lane, kStageUnroll loads issued back to back before the first LDS write: with one load in
        // flight per lane (the rolled loop waits vmcnt(0) every trip) these HBM-bound kernels sat at ~3 TB/s
        constexpr int kStageUnroll = 6;
        const int rv = rec >> 2;                           // float4 per record: a vector never straddles records
        const int nvec = total >> 2;
        const float4* s4 = reinterpret_cast<const float4*>(src);
        for (int v0 = threadIdx.x; v0 < nvec; v0 += kPB * kStageUnroll) {
            float4 x[kStageUnroll];
#pragma unroll
            for (int u = 0; u < kStageUnroll; ++u) {
                const int v = v0 + u * kPB;
                if (v < nvec) x[u] = s4[v];
            }
#pragma unroll
            for (int u = 0; u < kStageUnroll; ++u) {
                const int v = v0 + u * kPB;
                if (v < nvec) {
                    float* d = lds + (v / rv) * stride + (v % rv) * 4;
                    d[0] = x[u].x; d[1] = x[u].y; d[2] = x[u].z; d[3] = x[u].w;
                }
            }
        }
    } else {
        for (int e = threadIdx.x; e < total; e += kPB) lds[(e / rec) * stride + (e % rec)] = src[e];
    }
}
// Inverse: LDS rows -> contiguous global span.
__device__ __forceinline__ void stage_out(float* __restrict__ dst, const float* lds, int n_rec, int rec, int stride) {
    const int total = n_rec * rec;
    if ((rec & 3) == 0) {
        const int rv = rec >> 2;
        float4* d4 = reinterpret_cast<float4*>(dst);
        for (int v = threadIdx.x; v < (total >> 2); v += kPB) {
            const float* s_ = lds + (v / rv) * stride + (v % rv) * 4;
            d4[v] = make_float4(s_[0], s_[1], s_[2], s_[3]);
        }
    } else {
        for (int e = threadIdx.x; e < total; e += kPB) dst[e] = lds[(e / rec) * stride + (e % rec)];
    }
}
// The same for records whose length is not a multiple of 4 floats (features_rest: 3 (M - 1) = 45 / 24 / 9): float4
// loads over the contiguous span, every component placed by (record, offset) — the record index comes from one fp32
// multiply (exact for spans < 2^22 floats), not from an integer division. `lds` points at column 0 of the target.
__device__ __forceinline__ void stage_in_split(const float* __restrict__ src, float* lds, int n_rec, int rec, int stride) {
    const int total = n_rec * rec, nvec = total >> 2;
    const float inv = 1.0f / (float)rec;
    auto put = [&](int e, float v) {
        const int r = (int)(((float)e + 0.5f) * inv);
        lds[r * stride + (e - r * rec)] = v;
    };
    constexpr int kU = 6;
    const float4* s4 = reinterpret_cast<const float4*>(src);
    for (int v0 = threadIdx.x; v0 < nvec; v0 += kPB * kU) {
        float4 x[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) { const int v = v0 + u * kPB; if (v < nvec) x[u] = s4[v]; }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int v = v0 + u * kPB;
            if (v < nvec) { put(4 * v, x[u].x); put(4 * v + 1, x[u].y); put(4 * v + 2, x[u].z); put(4 * v + 3, x[u].w); }
        }
    }
    for (int e = 4 * nvec + threadIdx.x; e < total; e += kPB) put(e, src[e]);
}
__device__ __forceinline__ void stage_out_split(float* __restrict__ dst, const float* lds, int n_rec, int rec, int stride) {
    const int total = n_rec * rec, nvec = total >> 2;
    const float inv = 1.0f / (float)rec;
    auto get = [&](int e) {
        const int r = (int)(((float)e + 0.5f) * inv);
        return lds[r * stride + (e - r * rec)];
    };
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int v = threadIdx.x; v < nvec; v += kPB) d4[v] = make_float4(get(4 * v), get(4 * v + 1), get(4 * v + 2), get(4 * v + 3));
    for (int e = 4 * nvec + threadIdx.x; e < total; e += kPB) dst[e] = get(e);
}
// SH rows of a block into / out of LDS rows of `stride` floats: one [P,M,3] array, or (raw mode) features_dc [P,1,3] +
// features_rest [P,M-1,3] landing in the same rows, so the rest of the kernel does not care
__device__ __forceinline__ void stage_sh_in(const Frame& f, const float* shs, float* s_sh, int blk0, int n_rec, int stride) {
    if (!f.raw) { stage_in(shs + (size_t)blk0 * f.M * 3, s_sh, n_rec, 3 * f.M, stride); return; }
    stage_in(shs + (size_t)blk0 * 3, s_sh, n_rec, 3, stride);
    if (f.M > 1) stage_in_split(f.shs_rest + (size_t)blk0 * (3 * f.M - 3), s_sh + 3, n_rec, 3 * f.M - 3, stride);
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__host__ __device__ inline int sh_stride(int M) { return 3 * M + 1 - ((3 * M) & 1); }   // odd row stride

// SH constants, sh_basis and the colour evaluation sh_rgb live in raster_common.h (shared with the render kernel's deferred
// evaluation).

__device__ __forceinline__ void quat_to_rot(float r, float x, float y, float z, float R[3][3]) {
#pragma clang fp contract(off)
    R[0][0] = __builtin_fmaf(-2.0f, __builtin_fmaf(y, y, z * z), 1.0f);
    R[0][1] = 2.0f * __builtin_fmaf(x, y, -(r * z));
    R[0][2] = 2.0f * __builtin_fmaf(x, z, r * y);
    R[1][0] = 2.0f * __builtin_fmaf(x, y, r * z);
    R[1][1] = __builtin_fmaf(-2.0f, __builtin_fmaf(x, x, z * z), 1.0f);
    R[1][2] = 2.0f * __builtin_fmaf(y, z, -(r * x));
    R[2][0] = 2.0f * __builtin_fmaf(x, z, -(r * y));
    R[2][1] = 2.0f * __builtin_fmaf(y, z, r * x);
    R[2][2] = __builtin_fmaf(-2.0f, __builtin_fmaf(x, x, y * y), 1.0f);
}

struct Ewa {
    float Tm[2][3];
    float tx, ty, tz, xmul, ymul;
};

__device__ __forceinline__ void ewa_setup(const Frame& f, const float* V, float vx, float vy, float vz, Ewa& e) {
#pragma clang fp contract(off)
    float limx = kFrustumClamp * f.tanfovx, limy = kFrustumClamp * f.tanfovy;
    float txtz = vx / vz, tytz = vy / vz;
    e.xmul = (txtz < -limx || txtz > limx) ? 0.0f : 1.0f;
    e.ymul = (tytz < -limy || tytz > limy) ? 0.0f : 1.0f;
    e.tx = fminf(limx, fmaxf(-limx, txtz)) * vz;
    e.ty = fminf(limy, fmaxf(-limy, tytz)) * vz;
    e.tz = vz;
    float J00 = f.fx / vz, J02 = -(f.fx * e.tx) / (vz * vz);
    float J11 = f.fy / vz, J12 = -(f.fy * e.ty) / (vz * vz);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        e.Tm[0][j] = __builtin_fmaf(J02, V[2 + 4 * j], J00 * V[0 + 4 * j]);
        e.Tm[1][j] = __builtin_fmaf(J12, V[2 + 4 * j], J11 * V[1 + 4 * j]);
    }
}

__device__ __forceinline__ void cov2d_from_cov3d(const Ewa& e, const float* c6, float& a, float& b, float& c) {
#pragma clang fp contract(off)
    float S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
    float t0[3], t1[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        t0[j] = dot3(e.Tm[0][0], e.Tm[0][1], e.Tm[0][2], S[0][j], S[1][j], S[2][j]);
        t1[j] = dot3(e.Tm[1][0], e.Tm[1][1], e.Tm[1][2], S[0][j], S[1][j], S[2][j]);
    }
    a = dot3(t0[0], t0[1], t0[2], e.Tm[0][0], e.Tm[0][1], e.Tm[0][2]) + kLowpass;
    b = dot3(t0[0], t0[1], t0[2], e.Tm[1][0], e.Tm[1][1], e.Tm[1][2]);
    c = dot3(t1[0], t1[1], t1[2], e.Tm[1][0], e.Tm[1][1], e.Tm[1][2]) + kLowpass;
}

__global__ __launch_bounds__(kPB) void preprocess_forward_kernel(
    Frame f, const float* __restrict__ means3D, const float* __restrict__ shs,
    const float* __restrict__ colors_precomp, const float* __restrict__ opacities,
    const float* __restrict__ scales, const float* __restrict__ rotations,
    const float* __restrict__ cov3D_precomp, GeomView g, int32_t* __restrict__ radii, int defer_colors, int sh_vec16) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];   // [kPB][sh_stride(M)] SH rows (eager colours only), then [kPB][3] x 2
    __shared__ uint32_t s_sum;
    const int tid = threadIdx.x;
    const int blk0 = blockIdx.x * kPB;
    const int i = blk0 + tid;
    const int n_rec = min(kPB, f.P - blk0);
    const bool eager_sh = shs && !defer_colors;          // deferred (raster_common.h, ColorSource): the SH rows are not even read here
    const int shs_w = eager_sh ? sh_stride(f.M) : 0;
    float* s_sh = s_dyn;
    float* s_mean = s_dyn + (size_t)kPB * shs_w;        // [kPB][3]
    float* s_scale = s_mean + 3 * kPB;                  // [kPB][3]
    if (tid == 0) s_sum = 0;
    if (blockIdx.x == 0 && tid == 0) {                   // where this forward's colours come from, for the render kernel
        ColorSource cs;
        cs.means3D = means3D; cs.shs = shs; cs.shs_rest = f.shs_rest; cs.M = f.M; cs.deg = f.deg; cs.raw = f.raw;
        cs.deferred = (shs && defer_colors) ? 1 : 0; cs.vec16 = sh_vec16;
        *g.color_src = cs;
    }
    if (eager_sh) stage_sh_in(f, shs, s_sh, blk0, n_rec, shs_w);
    stage_in(means3D + (size_t)blk0 * 3, s_mean, n_rec, 3, 3);
    if (scales) stage_in(scales + (size_t)blk0 * 3, s_scale, n_rec, 3, 3);
    // per-lane operands requested before the barrier so they travel with the staged rows
    float4 q = make_float4(1.f, 0.f, 0.f, 0.f);
    float opac = 0.f;
    if (i < f.P) {
        if (!cov3D_precomp) q = *reinterpret_cast<const float4*>(rotations + 4 * (size_t)i);
        opac = opacities[i];
        if (f.raw) {                                        // gaussian_model.py:44-59: sigmoid, F.normalize (eps 1e-12)
            const float nq = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
            q = make_float4(q.x / nq, q.y / nq, q.z / nq, q.w / nq);
            opac = sigmoidf_(opac);
        }
    }
    __syncthreads();
    float V[16], PM[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { V[k] = f.view[k]; PM[k] = f.proj[k]; }

    uint32_t touched = 0;
    uint2 rect = make_uint2(0u, 0u);
    float depth_key = 0.0f;
    int rad_out = 0;
    if (i < f.P) {
        do {
            float px = s_mean[3 * tid], py = s_mean[3 * tid + 1], pz = s_mean[3 * tid + 2];
            float vx = affine3(V[0], V[4], V[8], V[12], px, py, pz);
            float vy = affine3(V[1], V[5], V[9], V[13], px, py, pz);
            float vz = affine3(V[2], V[6], V[10], V[14], px, py, pz);
            if (vz <= kNearZ) break;
            float hx = affine3(PM[0], PM[4], PM[8], PM[12], px, py, pz);
            float hy = affine3(PM[1], PM[5], PM[9], PM[13], px, py, pz);
            float hw = affine3(PM[3], PM[7], PM[11], PM[15], px, py, pz);
            float pw = 1.0f / (hw + 0.0000001f);
            float ndx = hx * pw, ndy = hy * pw;

            float c6[6];
            if (cov3D_precomp) {
#pragma unroll
                for (int k = 0; k < 6; ++k) c6[k] = cov3D_precomp[6 * (size_t)i + k];
            } else {
                float R[3][3], Mx[3][3];
                quat_to_rot(q.x, q.y, q.z, q.w, R);
                float sa[3] = {s_scale[3 * tid], s_scale[3 * tid + 1], s_scale[3 * tid + 2]};
                if (f.raw) { sa[0] = expf(sa[0]); sa[1] = expf(sa[1]); sa[2] = expf(sa[2]); }
                float s[3] = {f.scale_modifier * sa[0], f.scale_modifier * sa[1], f.scale_modifier * sa[2]};
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int j = 0; j < 3; ++j) Mx[r][j] = R[r][j] * s[j];
                c6[0] = dot3(Mx[0][0], Mx[0][1], Mx[0][2], Mx[0][0], Mx[0][1], Mx[0][2]);
                c6[1] = dot3(Mx[0][0], Mx[0][1], Mx[0][2], Mx[1][0], Mx[1][1], Mx[1][2]);
                c6[2] = dot3(Mx[0][0], Mx[0][1], Mx[0][2], Mx[2][0], Mx[2][1], Mx[2][2]);
                c6[3] = dot3(Mx[1][0], Mx[1][1], Mx[1][2], Mx[1][0], Mx[1][1], Mx[1][2]);
                c6[4] = dot3(Mx[1][0], Mx[1][1], Mx[1][2], Mx[2][0], Mx[2][1], Mx[2][2]);
                c6[5] = dot3(Mx[2][0], Mx[2][1], Mx[2][2], Mx[2][0], Mx[2][1], Mx[2][2]);
            }
            g.cov_a[i] = make_float4(c6[0], c6[1], c6[2], c6[3]);
            g.cov_b[i] = make_float2(c6[4], c6[5]);

            Ewa e;
            ewa_setup(f, V, vx, vy, vz, e);
            float a, b, c;
            cov2d_from_cov3d(e, c6, a, b, c);
            float det = __builtin_fmaf(a, c, -(b * b));
            if (det == 0.0f) break;
            float det_inv = 1.0f / det;
            float mid = 0.5f * (a + c);
            float disc = sqrtf(fmaxf(kLambdaFloor, __builtin_fmaf(mid, mid, -det)));
            float lam1 = mid + disc, lam2 = mid - disc;
            int rad = (int)ceilf(3.0f * sqrtf(fmaxf(lam1, lam2)));     // v_cvt_i32_f32: saturating, NaN -> 0
            // A covariance that is not positive definite (possible through cov3D_precomp) makes max(lambda) negative and
            // the radius NaN -> 0. The plug-in then still counts the tile under the centre in num_rendered but emits no key
            // for it (its duplicateWithKeys tests radii > 0): a slot of UNINITIALISED memory in its sort buffers. There is
            // no behaviour to match, so such a Gaussian is culled here like any other invisible one.
            if (rad <= 0) break;
            float pix_x = ((ndx + 1.0f) * (float)f.W - 1.0f) * 0.5f;
            float pix_y = ((ndy + 1.0f) * (float)f.H - 1.0f) * 0.5f;
            int x0, y0, x1, y1;
            tile_rect(pix_x, pix_y, rad, f.gx, f.gy, x0, y0, x1, y1);
            if ((x1 - x0) * (y1 - y0) == 0) break;

            float r0, r1, r2;
            uint32_t clamp_bits = 0;
            if (colors_precomp) {
                r0 = colors_precomp[3 * (size_t)i]; r1 = colors_precomp[3 * (size_t)i + 1]; r2 = colors_precomp[3 * (size_t)i + 2];
            } else if (eager_sh) {
                const float* sh = s_sh + (size_t)tid * shs_w;
                sh_rgb(f.deg, px, py, pz, f.campos, [&](int k, int c) { return sh[3 * k + c]; }, r0, r1, r2, clamp_bits);
            } else {
                r0 = r1 = r2 = -1.0f;                    // pending: evaluated by the render kernel when it first stages this Gaussian
            }
            g.clamped[i] = (uint8_t)clamp_bits;
            g.rgbd[i] = make_float4(r0, r1, r2, vz);
            g.depths[i] = vz;
            depth_key = vz;
            g.xy[i] = make_float2(pix_x, pix_y);
            g.conic_opacity[i] = make_float4(c * det_inv, -b * det_inv, a * det_inv, opac);
            rad_out = rad;
            touched = (uint32_t)((x1 - x0) * (y1 - y0));
            rect = make_uint2((uint32_t)x0 | ((uint32_t)y0 << 16), (uint32_t)(x1 - x0) | ((uint32_t)(y1 - y0) << 16));
        } while (false);
        radii[i] = rad_out;
        if (defer_colors) g.front[i] = 0;
        g.tiles_touched[i] = touched;
        g.rect[i] = rect;
        // level-1 sort input (binning): depth bits (monotonic for depth > 0.2), culled Gaussians last
        g.dkeys[0][i] = touched ? __float_as_uint(depth_key) : 0xFFFFFFFFu;
        if (!f.bin_v2) g.dvals[0][i] = (uint32_t)i;          // version 2's first sort pass takes the index itself
    }
    uint32_t wsum = wave_sum_u32(touched);
    if ((tid & 63) == 0 && wsum) atomicAdd(&s_sum, wsum);
    if (f.bin_v2) {   // column segments (rectangle widths) of the block: sizes the grids of binning version 2's second pass
        const uint32_t segs = wave_sum_u32(touched ? (rect.y & 0xFFFFu) : 0u);
        static_assert(kPB == 64, "one wave per block");
        if (tid == 0) g.seg_sums[blockIdx.x] = segs;
    }
    __syncthreads();
    if (tid == 0) g.block_sums[blockIdx.x] = s_sum;
}

int launch_preprocess_forward(const Frame& f, const float* means3D, const float* shs,
                              const float* colors_precomp, const float* opacities, const float* scales,
                              const float* rotations, const float* cov3D_precomp, GeomView g,
                              int32_t* radii, hipStream_t st) {
    if (f.P <= 0) return 0;
    int nblk = (f.P + kPB - 1) / kPB;
    const int defer = (shs && f.defer_colors) ? 1 : 0;
    const int vec16 = (shs && !f.raw && (f.M & 3) == 0 && ((uintptr_t)shs & 15u) == 0) ? 1 : 0;
    size_t lds = sizeof(float) * ((size_t)kPB * ((shs && !defer) ? sh_stride(f.M) : 0) + 6 * kPB);
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)preprocess_forward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return MVI_EHIP;
    hipLaunchKernelGGL(preprocess_forward_kernel, dim3(nblk), dim3(kPB), lds, st, f, means3D, shs,
                       colors_precomp, opacities, scales, rotations, cov3D_precomp, g, radii, defer, vec16);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

// ------------------------------------------------------------------------------------ backward
__device__ __forceinline__ void sh_basis_grad(int deg, float x, float y, float z, float* dx, float* dy, float* dz) {
#pragma unroll
    for (int k = 0; k < 16; ++k) dx[k] = dy[k] = dz[k] = 0.0f;
    if (deg > 0) {
        dy[1] = -SH_C1; dz[2] = SH_C1; dx[3] = -SH_C1;
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z;
            dx[4] = SH_C2[0] * y; dy[4] = SH_C2[0] * x;
            dy[5] = SH_C2[1] * z; dz[5] = SH_C2[1] * y;
            dx[6] = SH_C2[2] * (-2.0f * x); dy[6] = SH_C2[2] * (-2.0f * y); dz[6] = SH_C2[2] * (4.0f * z);
            dx[7] = SH_C2[3] * z; dz[7] = SH_C2[3] * x;
            dx[8] = SH_C2[4] * (2.0f * x); dy[8] = SH_C2[4] * (-2.0f * y);
            if (deg > 2) {
                dx[9] = SH_C3[0] * (6.0f * x * y); dy[9] = SH_C3[0] * (3.0f * xx - 3.0f * yy);
                dx[10] = SH_C3[1] * (y * z); dy[10] = SH_C3[1] * (x * z); dz[10] = SH_C3[1] * (x * y);
                dx[11] = SH_C3[2] * (-2.0f * x * y); dy[11] = SH_C3[2] * (4.0f * zz - xx - 3.0f * yy);
                dz[11] = SH_C3[2] * (8.0f * y * z);
                dx[12] = SH_C3[3] * (-6.0f * x * z); dy[12] = SH_C3[3] * (-6.0f * y * z);
                dz[12] = SH_C3[3] * (6.0f * zz - 3.0f * xx - 3.0f * yy);
                dx[13] = SH_C3[4] * (4.0f * zz - 3.0f * xx - yy); dy[13] = SH_C3[4] * (-2.0f * x * y);
                dz[13] = SH_C3[4] * (8.0f * x * z);
                dx[14] = SH_C3[5] * (2.0f * x * z); dy[14] = SH_C3[5] * (-2.0f * y * z);
                dz[14] = SH_C3[5] * (xx - yy);
                dx[15] = SH_C3[6] * (3.0f * xx - 3.0f * yy); dy[15] = SH_C3[6] * (-6.0f * x * y);
            }
        }
    }
}

// One thread per Gaussian. Input: grad_rows [P][16] from render_backward (mean2D.xy NDC-scaled,
// dL/dA, dL/dB, dL/dC of power = -0.5(A dx^2 + C dy^2) - B dx dy, dL/dopacity, dL/drgb).
// Outputs are written for every Gaussian (zeros where radii == 0).
// Chain rule from the 2-D gradients of one Gaussian (gr: dL/dmean2D (NDC-scaled) 2, dL/dconic 3, ...) to dL/dcov3D (g6)
// and the covariance / projection part of dL/dmean3D (dm). Shared by the dense and the sparse backward kernel.
__device__ __forceinline__ void backward_cov_and_mean(const Frame& f, float px, float py, float pz, const float* gr,
                                                  float4 ca, float2 cb, float* dm, float* g6) {
    float V[16], PM[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { V[k] = f.view[k]; PM[k] = f.proj[k]; }
    float vx = affine3(V[0], V[4], V[8], V[12], px, py, pz);
    float vy = affine3(V[1], V[5], V[9], V[13], px, py, pz);
    float vz = affine3(V[2], V[6], V[10], V[14], px, py, pz);
    float c6[6] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y};
    Ewa e;
    ewa_setup(f, V, vx, vy, vz, e);
    float a, b, c;
    cov2d_from_cov3d(e, c6, a, b, c);
    float denom = a * c - b * b;
    float d2inv = 1.0f / (denom * denom + 0.0000001f);
    const float4 gc = make_float4(gr[2], gr[3], gr[4], gr[5]);
    float da = 0, db = 0, dc = 0;
    if (d2inv != 0.0f) {
        da = d2inv * (-c * c * gc.x + b * c * gc.y + (denom - a * c) * gc.z);
        dc = d2inv * (-a * a * gc.z + a * b * gc.y + (denom - a * c) * gc.x);
        db = d2inv * (2.0f * b * c * gc.x - (denom + 2.0f * b * b) * gc.y + 2.0f * a * b * gc.z);
        const float(*Tm)[3] = e.Tm;
        g6[0] = Tm[0][0] * Tm[0][0] * da + Tm[0][0] * Tm[1][0] * db + Tm[1][0] * Tm[1][0] * dc;
        g6[3] = Tm[0][1] * Tm[0][1] * da + Tm[0][1] * Tm[1][1] * db + Tm[1][1] * Tm[1][1] * dc;
        g6[5] = Tm[0][2] * Tm[0][2] * da + Tm[0][2] * Tm[1][2] * db + Tm[1][2] * Tm[1][2] * dc;
        g6[1] = 2 * Tm[0][0] * Tm[0][1] * da + (Tm[0][0] * Tm[1][1] + Tm[0][1] * Tm[1][0]) * db +
                2 * Tm[1][0] * Tm[1][1] * dc;
        g6[2] = 2 * Tm[0][0] * Tm[0][2] * da + (Tm[0][0] * Tm[1][2] + Tm[0][2] * Tm[1][0]) * db +
                2 * Tm[1][0] * Tm[1][2] * dc;
        g6[4] = 2 * Tm[0][2] * Tm[0][1] * da + (Tm[0][1] * Tm[1][2] + Tm[0][2] * Tm[1][1]) * db +
                2 * Tm[1][1] * Tm[1][2] * dc;
    }
    float S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
    float dJ00 = 0, dJ02 = 0, dJ11 = 0, dJ12 = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float st0 = S[k][0] * e.Tm[0][0] + S[k][1] * e.Tm[0][1] + S[k][2] * e.Tm[0][2];
        float st1 = S[k][0] * e.Tm[1][0] + S[k][1] * e.Tm[1][1] + S[k][2] * e.Tm[1][2];
        float dT0 = 2.0f * da * st0 + db * st1;
        float dT1 = 2.0f * dc * st1 + db * st0;
        dJ00 += dT0 * V[0 + 4 * k];
        dJ02 += dT0 * V[2 + 4 * k];
        dJ11 += dT1 * V[1 + 4 * k];
        dJ12 += dT1 * V[2 + 4 * k];
    }
    float tz = 1.0f / e.tz, tz2 = tz * tz, tz3 = tz2 * tz;
    float dtx = e.xmul * -f.fx * tz2 * dJ02;
    float dty = e.ymul * -f.fy * tz2 * dJ12;
    float dtz = -f.fx * tz2 * dJ00 - f.fy * tz2 * dJ11 + (2 * f.fx * e.tx) * tz3 * dJ02 +
                (2 * f.fy * e.ty) * tz3 * dJ12;
    dm[0] = V[0] * dtx + V[1] * dty + V[2] * dtz;
    dm[1] = V[4] * dtx + V[5] * dty + V[6] * dtz;
    dm[2] = V[8] * dtx + V[9] * dty + V[10] * dtz;

    float hx = affine3(PM[0], PM[4], PM[8], PM[12], px, py, pz);
    float hy = affine3(PM[1], PM[5], PM[9], PM[13], px, py, pz);
    float hw = affine3(PM[3], PM[7], PM[11], PM[15], px, py, pz);
    float mw = 1.0f / (hw + 0.0000001f);
    float mul1 = hx * mw * mw, mul2 = hy * mw * mw;
    float g2x = gr[0], g2y = gr[1];
    dm[0] += (PM[0] * mw - PM[3] * mul1) * g2x + (PM[1] * mw - PM[3] * mul2) * g2y;
    dm[1] += (PM[4] * mw - PM[7] * mul1) * g2x + (PM[5] * mw - PM[7] * mul2) * g2y;
    dm[2] += (PM[8] * mw - PM[11] * mul1) * g2x + (PM[9] * mw - PM[11] * mul2) * g2y;

}

// dL/dcov3D (g6) -> dL/dscale (ds), dL/dquaternion (dq); raw mode: through exp and the normalisation as well
__device__ __forceinline__ void backward_scale_rot(const Frame& f, const float* g6, float4 qrot, const float* sa_in, float* ds,
                                                   float* dq) {
    float4 q = qrot;
    float nq = 1.0f;
    if (f.raw) {
        nq = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
        q = make_float4(q.x / nq, q.y / nq, q.z / nq, q.w / nq);
    }
    float R[3][3];
    quat_to_rot(q.x, q.y, q.z, q.w, R);
    float sa[3] = {sa_in[0], sa_in[1], sa_in[2]};
    if (f.raw) { sa[0] = expf(sa[0]); sa[1] = expf(sa[1]); sa[2] = expf(sa[2]); }
    float s[3] = {f.scale_modifier * sa[0], f.scale_modifier * sa[1], f.scale_modifier * sa[2]};
    float Gf[3][3] = {{g6[0], 0.5f * g6[1], 0.5f * g6[2]},
                      {0.5f * g6[1], g6[3], 0.5f * g6[4]},
                      {0.5f * g6[2], 0.5f * g6[4], g6[5]}};
    float dR[3][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float acc_s = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float dMrj = 2.0f * (Gf[r][0] * R[0][j] + Gf[r][1] * R[1][j] + Gf[r][2] * R[2][j]) * s[j];
            acc_s += dMrj * R[r][j];
            dR[r][j] = dMrj * s[j];
        }
        ds[j] = acc_s * f.scale_modifier;
    }
    float qr = q.x, qx = q.y, qy = q.z, qz = q.w;
    dq[0] = 2.0f * (-qz * dR[0][1] + qy * dR[0][2] + qz * dR[1][0] - qx * dR[1][2] - qy * dR[2][0] + qx * dR[2][1]);
    dq[1] = 2.0f * (qy * dR[0][1] + qz * dR[0][2] + qy * dR[1][0] - 2 * qx * dR[1][1] - qr * dR[1][2] +
                    qz * dR[2][0] + qr * dR[2][1] - 2 * qx * dR[2][2]);
    dq[2] = 2.0f * (-2 * qy * dR[0][0] + qx * dR[0][1] + qr * dR[0][2] + qx * dR[1][0] + qz * dR[1][2] -
                    qr * dR[2][0] + qz * dR[2][1] - 2 * qy * dR[2][2]);
    dq[3] = 2.0f * (-2 * qz * dR[0][0] - qr * dR[0][1] + qx * dR[0][2] + qr * dR[1][0] - 2 * qz * dR[1][1] +
                    qy * dR[1][2] + qx * dR[2][0] + qy * dR[2][1]);
    if (f.raw) {                                   // chain rule of exp and of y = q / |q|: (g - y (y . g)) / |q|
        ds[0] *= sa[0]; ds[1] *= sa[1]; ds[2] *= sa[2];
        const float dot = q.x * dq[0] + q.y * dq[1] + q.z * dq[2] + q.w * dq[3];
        const float inq = 1.0f / nq;
        dq[0] = (dq[0] - q.x * dot) * inq; dq[1] = (dq[1] - q.y * dot) * inq;
        dq[2] = (dq[2] - q.z * dot) * inq; dq[3] = (dq[3] - q.w * dot) * inq;
    }
}

__global__ __launch_bounds__(kPB) void preprocess_backward_kernel(
    Frame f, const float* __restrict__ means3D, const float* __restrict__ shs,
    const float* __restrict__ scales, const float* __restrict__ rotations,
    const float* __restrict__ cov3D_precomp, const int32_t* __restrict__ radii, GeomView g,
    const float* __restrict__ grad_rows, float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dmeans2D,
    float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolors, float* __restrict__ dL_dshs,
    float* __restrict__ dL_dcov3D, float* __restrict__ dL_dscales, float* __restrict__ dL_drots, RawBackwardExtra rawx) {
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];   // [kPB][sh_stride(M)] + 5 x [kPB][3]
    const int tid = threadIdx.x;
    const int blk0 = blockIdx.x * kPB;
    const int i = blk0 + tid;
    const int n_rec = min(kPB, f.P - blk0);
    const int shs_w = shs ? sh_stride(f.M) : 0;
    float* s_sh = s_dyn;                                   // SH rows in, dL/dSH rows out (same thread, same row)
    float* s_mean = s_dyn + (size_t)kPB * shs_w;
    float* s_scale = s_mean + 3 * kPB;
    float* s_dmean = s_scale + 3 * kPB;
    float* s_dm2d = s_dmean + 3 * kPB;
    float* s_dscale = s_dm2d + 3 * kPB;
    if (shs) stage_sh_in(f, shs, s_sh, blk0, n_rec, shs_w);
    stage_in(means3D + (size_t)blk0 * 3, s_mean, n_rec, 3, 3);
    if (scales) stage_in(scales + (size_t)blk0 * 3, s_scale, n_rec, 3, 3);
    // per-lane operands requested before the barrier so they travel with the staged rows
    const bool in_range = i < f.P;
    const float raw_o = (f.raw && in_range) ? rawx.raw_opacity[i] : 0.0f;
    const bool live = in_range && radii[i] > 0;
    float gr[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    float4 ca = make_float4(0.f, 0.f, 0.f, 0.f), qrot = make_float4(1.f, 0.f, 0.f, 0.f);
    float2 cb = make_float2(0.f, 0.f);
    uint32_t cl_bits = 0;
    if (live) {
        const float4* row = reinterpret_cast<const float4*>(grad_rows + (size_t)i * kGradRow);
        float4 a = row[0], b4 = row[1];
        gr[0] = a.x; gr[1] = a.y; gr[2] = a.z; gr[3] = a.w; gr[4] = b4.x; gr[5] = b4.y; gr[6] = b4.z; gr[7] = b4.w;
        gr[8] = grad_rows[(size_t)i * kGradRow + 8];
        ca = g.cov_a[i];
        cb = g.cov_b[i];
        if (!cov3D_precomp) qrot = *reinterpret_cast<const float4*>(rotations + 4 * (size_t)i);
        if (shs) cl_bits = g.clamped[i];
    }
    __syncthreads();
    s_dm2d[3 * tid] = gr[0]; s_dm2d[3 * tid + 1] = gr[1]; s_dm2d[3 * tid + 2] = 0.0f;
    if (in_range) {
        float go = gr[5];
        if (f.raw) { const float o = sigmoidf_(raw_o); go *= o * (1.0f - o); }    // d sigmoid
        dL_dopacity[i] = go;
        if (dL_dcolors) {
            // colours-precomp input: the colour gradient itself. SH input: the colour factor of the rank-1 SH gradient
            // dL/dSH[k][c] = Y_k(dir) * (clamped_c ? 0 : dL/dcolour_c) (mvi_raster_sh_backward_views rebuilds the rest)
            const uint32_t cl = cl_bits;
            dL_dcolors[3 * (size_t)i] = (cl & 1u) ? 0.0f : gr[6];
            dL_dcolors[3 * (size_t)i + 1] = (cl & 2u) ? 0.0f : gr[7];
            dL_dcolors[3 * (size_t)i + 2] = (cl & 4u) ? 0.0f : gr[8];
        }
    }
    float dm[3] = {0, 0, 0};
    float g6[6] = {0, 0, 0, 0, 0, 0};
    const int nb = (f.deg + 1) * (f.deg + 1);
    if (live) {
        const float px = s_mean[3 * tid], py = s_mean[3 * tid + 1], pz = s_mean[3 * tid + 2];
        backward_cov_and_mean(f, px, py, pz, gr, ca, cb, dm, g6);

        if (shs) {
            float ox = px - f.campos[0], oy = py - f.campos[1], oz = pz - f.campos[2];
            float len = sqrtf(ox * ox + oy * oy + oz * oz);
            float dx = ox / len, dy = oy / len, dz = oz / len;
            float bs[16], bx[16], by[16], bz[16];
            sh_basis(f.deg, dx, dy, dz, bs);
            sh_basis_grad(f.deg, dx, dy, dz, bx, by, bz);
            float* sh = s_sh + (size_t)tid * shs_w;        // this thread's row: read SH, then overwrite with dL/dSH
            const uint32_t cl = cl_bits;
            float gcol[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) gcol[ch] = ((cl >> ch) & 1u) ? 0.0f : gr[6 + ch];
            float ddx = 0, ddy = 0, ddz = 0;
            for (int k = 0; k < nb; ++k) {
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    float w = sh[3 * k + ch] * gcol[ch];
                    sh[3 * k + ch] = bs[k] * gcol[ch];
                    ddx += bx[k] * w; ddy += by[k] * w; ddz += bz[k] * w;
                }
            }
            for (int k = 3 * nb; k < 3 * f.M; ++k) sh[k] = 0.0f;
            float dotp = dx * ddx + dy * ddy + dz * ddz;
            dm[0] += (ddx - dx * dotp) / len;
            dm[1] += (ddy - dy * dotp) / len;
            dm[2] += (ddz - dz * dotp) / len;
        }
    } else if (shs) {
        float* dsh = s_sh + (size_t)tid * shs_w;
        for (int k = 0; k < 3 * f.M; ++k) dsh[k] = 0.0f;
    }
    s_dmean[3 * tid] = dm[0]; s_dmean[3 * tid + 1] = dm[1]; s_dmean[3 * tid + 2] = dm[2];

    if (cov3D_precomp) {
        if (in_range) {
#pragma unroll
            for (int k = 0; k < 6; ++k) dL_dcov3D[6 * (size_t)i + k] = g6[k];
        }
    } else {
        float ds[3] = {0, 0, 0}, dq[4] = {0, 0, 0, 0};
        if (live) {
            const float sa_in[3] = {s_scale[3 * tid], s_scale[3 * tid + 1], s_scale[3 * tid + 2]};
            backward_scale_rot(f, g6, qrot, sa_in, ds, dq);
        }
        s_dscale[3 * tid] = ds[0]; s_dscale[3 * tid + 1] = ds[1]; s_dscale[3 * tid + 2] = ds[2];
        if (in_range) *reinterpret_cast<float4*>(dL_drots + 4 * (size_t)i) = make_float4(dq[0], dq[1], dq[2], dq[3]);
    }
    __syncthreads();
    stage_out(dL_dmeans3D + (size_t)blk0 * 3, s_dmean, n_rec, 3, 3);
    stage_out(dL_dmeans2D + (size_t)blk0 * 3, s_dm2d, n_rec, 3, 3);
    if (!cov3D_precomp) stage_out(dL_dscales + (size_t)blk0 * 3, s_dscale, n_rec, 3, 3);
    if (shs && dL_dshs) {
        if (!f.raw) {
            stage_out(dL_dshs + (size_t)blk0 * f.M * 3, s_sh, n_rec, 3 * f.M, shs_w);
        } else {                                           // dL/dfeatures_dc [P,1,3] and dL/dfeatures_rest [P,M-1,3]
            stage_out(dL_dshs + (size_t)blk0 * 3, s_sh, n_rec, 3, shs_w);
            if (f.M > 1) stage_out_split(rawx.dL_dshs_rest + (size_t)blk0 * (3 * f.M - 3), s_sh + 3, n_rec, 3 * f.M - 3, shs_w);
        }
    }
}

// The per-Gaussian chain rule for the Gaussians whose accumulation row the render backward TOUCHED (g.touched), one lane per
// Gaussian, no LDS: in the bench scene 3 % of the Gaussians receive a gradient (the rest are occluded), and an untouched row
// is exactly zero, so its outputs are exactly zero — the caller zeroed every output array beforehand (the render backward does
// it on the side) and this kernel reads and writes the touched rows only, with per-lane accesses. Same expressions as
// preprocess_backward_kernel (shared helpers), which stays the form for dense outputs (split / ranged backward).
constexpr int kSparseBlock = 64;          // one-wave blocks: ~750 of them for the 48 k touched Gaussians of the bench scene, spread over
                                          // every CU (256-thread blocks: 188 blocks, most CUs idle behind a 4-round-trip latency chain)
__global__ __launch_bounds__(kSparseBlock) void preprocess_backward_sparse_kernel(
    Frame f, const float* __restrict__ means3D, const float* __restrict__ shs, const float* __restrict__ scales,
    const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp, GeomView g,
    const float* __restrict__ grad_rows, float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dmeans2D,
    float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolors, float* __restrict__ dL_dshs,
    float* __restrict__ dL_dcov3D, float* __restrict__ dL_dscales, float* __restrict__ dL_drots, RawBackwardExtra rawx,
    int vec_ok) {   // bit 0: the shs / dL_dshs rows are 16-byte aligned, bit 1: the rotations are (else: the same widths at dword alignment / per word)
    // a fixed, moderate grid walks the list with a grid stride: the list's length is only known on the device, and a grid
    // sized for P would be 23 k blocks of which a few hundred find work
    const uint32_t n_touched = *g.touched_count;
    for (uint32_t slot = blockIdx.x * (uint32_t)kSparseBlock + threadIdx.x; slot < n_touched; slot += gridDim.x * (uint32_t)kSparseBlock) {
    const int i = (int)g.touched_list[slot];
    const size_t si = (size_t)i;
    float gr[9];
    {
        const float4* row = reinterpret_cast<const float4*>(grad_rows + si * kGradRow);
        const float4 a = row[0], b4 = row[1];
        gr[0] = a.x; gr[1] = a.y; gr[2] = a.z; gr[3] = a.w; gr[4] = b4.x; gr[5] = b4.y; gr[6] = b4.z; gr[7] = b4.w;
        gr[8] = grad_rows[si * kGradRow + 8];
    }
    const float4 ca = g.cov_a[i];
    const float2 cb = g.cov_b[i];
    const uint32_t cl = shs ? g.clamped[i] : 0u;
    const float px = means3D[3 * si], py = means3D[3 * si + 1], pz = means3D[3 * si + 2];
    dL_dmeans2D[3 * si] = gr[0];
    dL_dmeans2D[3 * si + 1] = gr[1];
    {
        float go = gr[5];
        if (f.raw) { const float o = sigmoidf_(rawx.raw_opacity[i]); go *= o * (1.0f - o); }    // d sigmoid
        dL_dopacity[i] = go;
    }
    if (dL_dcolors) {
        dL_dcolors[3 * si] = (cl & 1u) ? 0.0f : gr[6];
        dL_dcolors[3 * si + 1] = (cl & 2u) ? 0.0f : gr[7];
        dL_dcolors[3 * si + 2] = (cl & 4u) ? 0.0f : gr[8];
    }
    float dm[3] = {0, 0, 0};
    float g6[6] = {0, 0, 0, 0, 0, 0};
    backward_cov_and_mean(f, px, py, pz, gr, ca, cb, dm, g6);
    if (shs) {
        const int M = f.M, nb = (f.deg + 1) * (f.deg + 1);
        float ox = px - f.campos[0], oy = py - f.campos[1], oz = pz - f.campos[2];
        float len = sqrtf(ox * ox + oy * oy + oz * oz);
        float dx = ox / len, dy = oy / len, dz = oz / len;
        float bs[16], bx[16], by[16], bz[16];
        sh_basis(f.deg, dx, dy, dz, bs);
        sh_basis_grad(f.deg, dx, dy, dz, bx, by, bz);
        float gcol[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) gcol[ch] = ((cl >> ch) & 1u) ? 0.0f : gr[6 + ch];
        // coefficient k of this Gaussian: one [P,M,3] array, or (raw mode) features_dc [P,1,3] + features_rest [P,M-1,3]
        const float* in0 = f.raw ? shs + 3 * si : shs + si * M * 3;
        const float* in_rest = f.raw ? f.shs_rest + si * (3 * M - 3) - 3 : in0;          // + 3 k for k >= 1
        float* out0 = dL_dshs ? (f.raw ? dL_dshs + 3 * si : dL_dshs + si * M * 3) : nullptr;
        float* out_rest = dL_dshs ? (f.raw ? rawx.dL_dshs_rest + si * (3 * M - 3) - 3 : out0) : nullptr;
        float ddx = 0, ddy = 0, ddz = 0;
        // a whole 16-coefficient row held in registers: w = coefficient * dL/dcolour feeds the direction gradient, the row becomes dL/dSH
        auto row16 = [&](float (&rowv)[48]) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const float w = k < nb ? rowv[3 * k + ch] * gcol[ch] : 0.f;
                    rowv[3 * k + ch] = k < nb ? bs[k] * gcol[ch] : 0.f;
                    ddx += bx[k] * w; ddy += by[k] * w; ddz += bz[k] * w;      // bx / by / bz are zero above nb
                }
            }
        };
        if (!f.raw && M == 16 && (vec_ok & 1)) {
            // the common case (sh_degree <= 3 stored with 16 coefficients): the 192-byte row moves as twelve 16-byte
            // accesses per lane instead of 48 + 48 scattered words
            float rowv[48];
            const float4* in4 = reinterpret_cast<const float4*>(in0);
#pragma unroll
            for (int v = 0; v < 12; ++v) { const float4 t = in4[v]; rowv[4 * v] = t.x; rowv[4 * v + 1] = t.y; rowv[4 * v + 2] = t.z; rowv[4 * v + 3] = t.w; }
            row16(rowv);
            if (out0) {
                float4* out4 = reinterpret_cast<float4*>(out0);
#pragma unroll
                for (int v = 0; v < 12; ++v) out4[v] = make_float4(rowv[4 * v], rowv[4 * v + 1], rowv[4 * v + 2], rowv[4 * v + 3]);
            }
        } else if (!f.raw && M == 16) {
            // the same row at 4-byte alignment only (the caller's arrays inside a larger buffer, e.g. dist.GradBucket): 16-byte accesses
            // at dword alignment
            struct W4 { float a, b, c, d; };
            float rowv[48];
#pragma unroll
            for (int v = 0; v < 12; ++v) {
                const W4 t = *reinterpret_cast<const W4*>(in0 + 4 * v);
                rowv[4 * v] = t.a; rowv[4 * v + 1] = t.b; rowv[4 * v + 2] = t.c; rowv[4 * v + 3] = t.d;
            }
            row16(rowv);
            if (out0) {
#pragma unroll
                for (int v = 0; v < 12; ++v) *reinterpret_cast<W4*>(out0 + 4 * v) = W4{rowv[4 * v], rowv[4 * v + 1], rowv[4 * v + 2], rowv[4 * v + 3]};
            }
        } else if (f.raw && M == 16) {
            // raw mode (features_dc [P,1,3] + features_rest [P,15,3]: 12- and 180-byte rows at 4-byte alignment): one 12-byte access,
            // eleven 16-byte accesses at dword alignment and one word — the per-word form cost the training iteration 94 us here
            // against 28 us for the same Gaussians through the [P,16,3] entry. (Its own branch, not a variant of the one above: with
            // both load sequences feeding one array the compiler merged them into 48 single-word loads from a selected address.)
            struct W3 { float a, b, c; };
            struct W4 { float a, b, c, d; };
            float rowv[48];
            const float* rp = f.shs_rest + si * 45;
            const W3 t0 = *reinterpret_cast<const W3*>(in0);
            rowv[0] = t0.a; rowv[1] = t0.b; rowv[2] = t0.c;
#pragma unroll
            for (int v = 0; v < 11; ++v) {
                const W4 t = *reinterpret_cast<const W4*>(rp + 4 * v);
                rowv[3 + 4 * v] = t.a; rowv[4 + 4 * v] = t.b; rowv[5 + 4 * v] = t.c; rowv[6 + 4 * v] = t.d;
            }
            rowv[47] = rp[44];
            row16(rowv);
            if (out0) {
                float* op = rawx.dL_dshs_rest + si * 45;
                *reinterpret_cast<W3*>(out0) = W3{rowv[0], rowv[1], rowv[2]};
#pragma unroll
                for (int v = 0; v < 11; ++v)
                    *reinterpret_cast<W4*>(op + 4 * v) = W4{rowv[3 + 4 * v], rowv[4 + 4 * v], rowv[5 + 4 * v], rowv[6 + 4 * v]};
                op[44] = rowv[47];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (k < nb) {
                    const float* in = k == 0 ? in0 : in_rest + 3 * k;
                    float* out = k == 0 ? out0 : out_rest + 3 * k;
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        const float w = in[ch] * gcol[ch];
                        if (out0) out[ch] = bs[k] * gcol[ch];
                        ddx += bx[k] * w; ddy += by[k] * w; ddz += bz[k] * w;
                    }
                }
            }
        }
        float dotp = dx * ddx + dy * ddy + dz * ddz;
        dm[0] += (ddx - dx * dotp) / len;
        dm[1] += (ddy - dy * dotp) / len;
        dm[2] += (ddz - dz * dotp) / len;
    }
    dL_dmeans3D[3 * si] = dm[0]; dL_dmeans3D[3 * si + 1] = dm[1]; dL_dmeans3D[3 * si + 2] = dm[2];
    if (cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; ++k) dL_dcov3D[6 * si + k] = g6[k];
    } else {
        const float4 qrot = (vec_ok & 2) ? *reinterpret_cast<const float4*>(rotations + 4 * si)
                                         : make_float4(rotations[4 * si], rotations[4 * si + 1], rotations[4 * si + 2], rotations[4 * si + 3]);
        const float sa_in[3] = {scales[3 * si], scales[3 * si + 1], scales[3 * si + 2]};
        float ds[3] = {0, 0, 0}, dq[4] = {0, 0, 0, 0};
        backward_scale_rot(f, g6, qrot, sa_in, ds, dq);
        dL_dscales[3 * si] = ds[0]; dL_dscales[3 * si + 1] = ds[1]; dL_dscales[3 * si + 2] = ds[2];
        dL_drots[4 * si] = dq[0]; dL_drots[4 * si + 1] = dq[1]; dL_drots[4 * si + 2] = dq[2]; dL_drots[4 * si + 3] = dq[3];
    }
    }   // grid-stride loop over the touched list
}

// touched flags -> compact list of Gaussian indices (any order) + their number: the chain rule then runs full waves.
// 16 flags per thread, 16384 per block, ONE returning atomic per block: a returning atomic on a single word completes at
// ~88 per microsecond on this chip (MI355X_MICROARCH.md, dequeue), so one per wave of a 1.5 M-flag pass cost 67 us.
constexpr int kCompactThreads = 1024, kCompactPer = 16;
__global__ __launch_bounds__(kCompactThreads) void compact_touched_kernel(int P, const uint8_t* __restrict__ touched,
                                                                         uint32_t* __restrict__ list, uint32_t* __restrict__ count) {
    __shared__ uint32_t s_wave[kCompactThreads / 64];
    __shared__ uint32_t s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = (blockIdx.x * kCompactThreads + tid) * kCompactPer;
    uint4 w = make_uint4(0u, 0u, 0u, 0u);
    if (i0 < P) w = *reinterpret_cast<const uint4*>(touched + i0);      // bytes past P are padding of the segment, never set
    const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
    uint32_t mine = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        mine += ((ww[q] & 0xFFu) != 0) + ((ww[q] & 0xFF00u) != 0) + ((ww[q] & 0xFF0000u) != 0) + ((ww[q] >> 24) != 0);
    uint32_t inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t off = 0, total = 0;
#pragma unroll
    for (int k = 0; k < kCompactThreads / 64; ++k) { const uint32_t c = s_wave[k]; if (k < wave) off += c; total += c; }
    if (total == 0) return;
    if (tid == 0) s_base = atomicAdd(count, total);
    __syncthreads();
    uint32_t dst = s_base + off + inc - mine;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (((ww[q] >> (8 * k)) & 0xFFu) && i0 + 4 * q + k < P) list[dst++] = (uint32_t)(i0 + 4 * q + k);
}

int launch_preprocess_backward_sparse(const Frame& f, const float* means3D, const float* shs, const float* scales,
                                      const float* rotations, const float* cov3D_precomp, GeomView g, const float* grad_rows,
                                      float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dopacity, float* dL_dcolors,
                                      float* dL_dshs, float* dL_dcov3D, float* dL_dscales, float* dL_drots, hipStream_t st,
                                      RawBackwardExtra rawx) {
    if (f.P <= 0) return 0;
    constexpr int kPerBlock = kCompactThreads * kCompactPer;
    hipLaunchKernelGGL(compact_touched_kernel, dim3((f.P + kPerBlock - 1) / kPerBlock), dim3(kCompactThreads), 0, st, f.P, g.touched,
                       g.touched_list, g.touched_count);
    const int sparse_blocks = min((f.P + kSparseBlock - 1) / kSparseBlock, 2048);
    // the caller's arrays may sit at any 4-byte offset (e.g. inside dist.GradBucket, where dL_dshs starts 12 P bytes in):
    // the 16-byte row accesses are taken only when every row is 16-byte aligned, else the per-word form
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15u) == 0; };
    const int vec_ok = ((al16(shs) && al16(dL_dshs)) ? 1 : 0) | (al16(rotations) ? 2 : 0);
    hipLaunchKernelGGL(preprocess_backward_sparse_kernel, dim3(sparse_blocks), dim3(kSparseBlock), 0, st, f, means3D, shs, scales,
                       rotations, cov3D_precomp, g, grad_rows, dL_dmeans3D, dL_dmeans2D, dL_dopacity, dL_dcolors, dL_dshs,
                       dL_dcov3D, dL_dscales, dL_drots, rawx, vec_ok);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

int launch_preprocess_backward(const Frame& f, const float* means3D, const float* shs,
                               const float* scales, const float* rotations, const float* cov3D_precomp,
                               const int32_t* radii, GeomView g, const float* grad_rows, float* dL_dmeans3D,
                               float* dL_dmeans2D, float* dL_dopacity, float* dL_dcolors, float* dL_dshs,
                               float* dL_dcov3D, float* dL_dscales, float* dL_drots, hipStream_t st, RawBackwardExtra rawx) {
    if (f.P <= 0) return 0;
    int nblk = (f.P + kPB - 1) / kPB;
    size_t lds = sizeof(float) * ((size_t)kPB * (shs ? sh_stride(f.M) : 0) + 15 * kPB);
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)preprocess_backward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return MVI_EHIP;
    hipLaunchKernelGGL(preprocess_backward_kernel, dim3(nblk), dim3(kPB), lds, st, f, means3D, shs, scales,
                       rotations, cov3D_precomp, radii, g, grad_rows, dL_dmeans3D, dL_dmeans2D, dL_dopacity,
                       dL_dcolors, dL_dshs, dL_dcov3D, dL_dscales, dL_drots, rawx);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

// View-parallel training: the SH gradient of one view is rank-1 per Gaussian,
// dL/dSH[k][c] = Y_k(dir_view) * gcol_view[c], so ranks exchange gcol (3 floats per Gaussian) instead of the
// 3M-float gradient and every rank rebuilds the SUM over views here from the views' camera centres:
// dL_dshs[g][k][c] = sum_v Y_k(normalize(mean_g - campos_v)) * gcol[v][g][c]   (k < (deg+1)^2, zero above).
// One thread per Gaussian; rows leave through LDS like the single-view backward (coalesced AoS writes).
__global__ __launch_bounds__(kPB) void sh_backward_views_kernel(int P, int M, int deg, int n_views,
                                                                    const float* __restrict__ means3D,
                                                                    const float* __restrict__ campos,
                                                                    int64_t campos_stride,
                                                                    const float* __restrict__ gcol,
                                                                    int64_t gcol_stride,
                                                                    float* __restrict__ dL_dshs) {
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];   // [kPB][sh_stride(M)]
    const int tid = threadIdx.x;
    const int blk0 = blockIdx.x * kPB;
    const int i = blk0 + tid;
    const int n_rec = min(kPB, P - blk0);
    const int w = sh_stride(M);
    const int nb = (deg + 1) * (deg + 1);
    float acc[16][3];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k][0] = acc[k][1] = acc[k][2] = 0.0f;
    if (i < P) {
        const float px = means3D[3 * (size_t)i], py = means3D[3 * (size_t)i + 1], pz = means3D[3 * (size_t)i + 2];
        for (int v = 0; v < n_views; ++v) {
            const float* gc = gcol + (size_t)v * gcol_stride + (size_t)i * 3;
            const float* cp = campos + (size_t)v * campos_stride;
            const float c0 = gc[0], c1 = gc[1], c2 = gc[2];
            if (c0 == 0.0f && c1 == 0.0f && c2 == 0.0f) continue;          // not seen (or fully clamped) in this view
            const float ox = px - cp[0], oy = py - cp[1], oz = pz - cp[2];
            const float len = sqrtf(ox * ox + oy * oy + oz * oz);
            float bs[16];
            sh_basis(deg, ox / len, oy / len, oz / len, bs);
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k < nb) { acc[k][0] += bs[k] * c0; acc[k][1] += bs[k] * c1; acc[k][2] += bs[k] * c2; }
        }
    }
    float* row = s_dyn + (size_t)tid * w;
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (k < M) { row[3 * k] = acc[k][0]; row[3 * k + 1] = acc[k][1]; row[3 * k + 2] = acc[k][2]; }
    __syncthreads();
    stage_out(dL_dshs + (size_t)blk0 * M * 3, s_dyn, n_rec, 3 * M, w);
}

int launch_sh_backward_views(int P, int M, int deg, int n_views, const float* means3D, const float* campos,
                             int64_t campos_stride, const float* gcol, int64_t gcol_stride, float* dL_dshs, hipStream_t st) {
    if (P <= 0) return 0;
    const size_t lds = sizeof(float) * (size_t)kPB * sh_stride(M);
    hipLaunchKernelGGL(sh_backward_views_kernel, dim3((P + kPB - 1) / kPB), dim3(kPB), lds, st, P, M, deg, n_views,
                       means3D, campos, campos_stride, gcol, gcol_stride, dL_dshs);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

__global__ void mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ view,
                                    uint8_t* __restrict__ visible) {
    int i = blockIdx.x * kPB + threadIdx.x;
    if (i >= P) return;
    float vz = affine3(view[2], view[6], view[10], view[14], means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]);
    visible[i] = vz > kNearZ;
}
// Colour factor of the rank-1 SH gradient (or the colour gradient for precomputed colours) straight from the render
// backward's accumulation rows: what preprocess_backward writes into dL_dcolors, available one kernel earlier so that a
// view-parallel trainer can start exchanging it while preprocess_backward runs (dist.FactoredGradExchange).
__global__ __launch_bounds__(256) void color_factor_kernel(int P, const int32_t* __restrict__ radii,
                                                           const float* __restrict__ grad_rows,
                                                           const uint8_t* __restrict__ clamped, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float r = 0.f, g_ = 0.f, b = 0.f;
    if (radii[i] > 0) {
        const float* row = grad_rows + (size_t)i * kGradRow;
        const uint32_t cl = clamped ? clamped[i] : 0u;
        r = (cl & 1u) ? 0.0f : row[6];
        g_ = (cl & 2u) ? 0.0f : row[7];
        b = (cl & 4u) ? 0.0f : row[8];
    }
    out[3 * (size_t)i] = r; out[3 * (size_t)i + 1] = g_; out[3 * (size_t)i + 2] = b;
}
// Evaluates every colour the render kernel has not needed (ColorSource): for the parity tests and for callers that read
// mvi_raster_views.rgbd / clamped of Gaussians that were never staged.
__global__ __launch_bounds__(256) void resolve_colors_kernel(Frame f, GeomView g) {
    const ColorSource cs = *g.color_src;
    if (!cs.deferred) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= f.P || g.tiles_touched[i] == 0) return;
    const float4 cd = g.rgbd[i];
    if (color_pending(cd)) resolve_color(cs, f.campos, (uint32_t)i, cd.w, g.rgbd, g.clamped);
}
int launch_resolve_colors(const Frame& f, GeomView g, hipStream_t st) {
    if (f.P <= 0) return 0;
    hipLaunchKernelGGL(resolve_colors_kernel, dim3((f.P + 255) / 256), dim3(256), 0, st, f, g);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

int launch_color_factors(int P, const int32_t* radii, const float* grad_rows, const uint8_t* clamped, float* out, hipStream_t st) {
    if (P <= 0) return 0;
    hipLaunchKernelGGL(color_factor_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, radii, grad_rows, clamped, out);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

int launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* visible, hipStream_t st) {
    if (P <= 0) return 0;
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + kPB - 1) / kPB), dim3(kPB), 0, st, P, means3D, view, visible);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

}  // namespace mvi
