// Temporal self-attention of the video transformer for gfx950 in bf16 / f16, head dim 64, T <= 16 frames: one softmax problem per
// (video, spatial token, head) with S_q = S_k = T — 92 160 problems of 14 x 14 x 64 at level 0 of a 14-frame 576x1024 step
// (svd_inpaint1/sgm/modules/video_attention.py:115, :136-140 around attention.py:281-300). HBM-bound: a problem
// reads 3 T rows of 128 bytes and writes T (7 KiB at T = 14).
//
// Why it exists: csrc/attn_rowtile.hip (fp32 math, K and V staged in LDS as fp32, every query row reading every K and V row back
// as ds_read_b128) spends ~900 LDS cycles and ~600 VALU cycles per problem and CU against ~550 cycles of HBM time: 0.33 of the HBM
// roofline. Here the two products are MFMAs (v_mfma_f32_16x16x32: 6 per problem) whose operands come straight from global memory
// in the instruction's own lane layout; only V passes through LDS (to be read key-major):
//   S^T[key][query] = K Q^T      A = K rows, B = Q rows: lane (row l & 15, g = l >> 4) loads 16 bytes at d = 32 step + 8 g — no LDS.
//                                C layout: lane (query l & 15, g) holds keys 4 g + r, r < 4.
//   softmax over keys            = over the lane's 4 registers and the 4 lane groups (two xor-shuffles); scale applied in fp32.
//   O^T[d][query] = V^T P^T      B = P^T: contraction index k = 8 g + j stands for key 4 g + j (j < 4) and for nothing (j >= 4), so the
//                                lane's own four probabilities ARE its B fragment (+ four zeros): no lane movement.
//                                A = V^T under the same index map: four 2-byte LDS reads per 16-channel tile from the row-major V
//                                tile (136-byte rows: the four lane groups fall on disjoint banks).
//                                C layout: lane (query l & 15, g) holds channels 16 tile + 4 g + r: one 8-byte store per tile.
// A wave walks problems grid-stride (neighbouring waves work on neighbouring 128-byte segments of every frame) and loads the next
// problem's fragments before it computes the current one.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <cstdint>

#include "../../include/mvi_raster.h"
#include "../../include/mvi_unet_ops.h"

namespace mvi {
namespace at16 {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

constexpr int kD = 64;
constexpr int kWaves = 4;                            // waves per block, one problem each at a time
constexpr int kVRow = 136;                           // bytes per V row in LDS (128 + 8: lane group g lands 8 banks after g - 1)
constexpr int kVTile = 16 * kVRow;

template <typename T> struct Mma;
template <> struct Mma<__hip_bfloat16> {
    using frag = bf16x8;
    __device__ static f32x4 mfma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    __device__ static uint32_t pack2(float lo, float hi) {
        f32x2 f = {lo, hi};
        bf16x2 r = __builtin_convertvector(f, bf16x2);
        return *reinterpret_cast<uint32_t*>(&r);
    }
};
template <> struct Mma<__half> {
    using frag = f16x8;
    __device__ static f32x4 mfma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    __device__ static uint32_t pack2(float lo, float hi) {
        f32x2 f = {lo, hi};
        f16x2 r = __builtin_convertvector(f, f16x2);
        return *reinterpret_cast<uint32_t*>(&r);
    }
};
template <typename F> __device__ __forceinline__ F as_frag(u32x4 v) { return *reinterpret_cast<F*>(&v); }

struct Operands {                                    // one problem's global loads, in flight while the previous problem computes
    u32x4 kq[4];                                     // K step 0, 1, Q step 0, 1 (frame T - 1 again for the rows >= T)
    u32x4 v[2];                                      // V pieces lane and lane + 64 of the T x 8 pieces of 16 bytes
};

template <typename T>
__global__ __launch_bounds__(64 * kWaves) void attn_temporal16_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                                      T* __restrict__ out, int Tn, int S, int H, float scale,
                                                                      int64_t n_problems, int64_t qkv_ts, int64_t o_ts) {
    using M = Mma<T>;
    using frag = typename M::frag;
    __shared__ __attribute__((aligned(16))) char s_v[kWaves][kVTile];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r16 = lane & 15, g = lane >> 4;
    char* const vt = s_v[wave];
    const int64_t n_waves = (int64_t)gridDim.x * kWaves;
    const bool frame_ok = r16 < Tn;                  // this lane's K / Q row (and output row) exists
    const int64_t q_frame = (int64_t)S * qkv_ts, o_frame = (int64_t)S * o_ts;      // elements between frames of one token
    const int rclamp = frame_ok ? r16 : Tn - 1;
    const int vclamp[2] = {min(lane >> 3, Tn - 1), min((lane + 64) >> 3, Tn - 1)};   // frame of V piece lane + 64 i (piece lane % 8 of its row)

    // problem p = (bo S + s) H + h: element offset of (frame 0, token s, head h) in q / k / v and in out
    auto load = [&](int64_t p, Operands& o) __attribute__((always_inline)) {
        const int64_t bs = p / H;
        const int h = (int)(p - bs * H);
        const int64_t bo = bs / S, s = bs - bo * S;
        const int64_t base = (bo * Tn * S + s) * qkv_ts + (int64_t)h * kD;
        // no load sits in a branch: the compiler counts vmcnt only through straight-line code, and a conditional load made it wait
        // for EVERYTHING (the next problem's loads included) before the current problem's first use. Lanes whose frame does not
        // exist read frame T - 1 instead: as K rows they end in masked scores, as Q rows in columns that are never stored.
        const int64_t row = base + rclamp * q_frame + 8 * g;
        o.kq[0] = *reinterpret_cast<const u32x4*>(k + row);
        o.kq[1] = *reinterpret_cast<const u32x4*>(k + row + 32);
        o.kq[2] = *reinterpret_cast<const u32x4*>(q + row);
        o.kq[3] = *reinterpret_cast<const u32x4*>(q + row + 32);
#pragma unroll
        for (int i = 0; i < 2; ++i) o.v[i] = *reinterpret_cast<const u32x4*>(v + base + vclamp[i] * q_frame + 8 * (lane & 7));
    };

    int64_t p = (int64_t)blockIdx.x * kWaves + wave;
    if (p >= n_problems) return;                     // (whole wave)
    Operands cur, nxt;
    load(p, cur);
    // rows T .. 15 of the V tile are read (times probability 0) by lane groups whose keys do not exist: keep them finite
    for (int e = lane; e < (16 - Tn) * 17; e += 64) *reinterpret_cast<u32x2*>(vt + Tn * kVRow + 8 * e) = u32x2{0u, 0u};

    for (; p < n_problems; p += n_waves) {
        const int64_t pn = p + n_waves;
        load(pn < n_problems ? pn : n_problems - 1, nxt);            // (unconditional, like the loads inside: the last round's is unused)

        // ---- V -> LDS, row-major (two 8-byte stores per piece: the rows are 8-byte aligned only)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = lane + 64 * i, t = e >> 3;
            if (t < Tn) {
                char* const dst = vt + t * kVRow + 16 * (e & 7);
                *reinterpret_cast<u32x2*>(dst) = u32x2{cur.v[i][0], cur.v[i][1]};
                *reinterpret_cast<u32x2*>(dst + 8) = u32x2{cur.v[i][2], cur.v[i][3]};
            }
        }
        __builtin_amdgcn_wave_barrier();

        // ---- scores, transposed: lane (query r16, g) gets keys 4 g + r
        f32x4 sc = {0.f, 0.f, 0.f, 0.f};
        sc = M::mfma(as_frag<frag>(cur.kq[0]), as_frag<frag>(cur.kq[2]), sc);
        sc = M::mfma(as_frag<frag>(cur.kq[1]), as_frag<frag>(cur.kq[3]), sc);
        float m = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sc[r] = 4 * g + r < Tn ? sc[r] * scale : -INFINITY;
            m = fmaxf(m, sc[r]);
        }
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));             // (key 0 exists: finite for every query row that does)
        float pr[4], l = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pr[r] = __expf(sc[r] - m);
            l += pr[r];
        }
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const u32x4 pb = {M::pack2(pr[0], pr[1]), M::pack2(pr[2], pr[3]), 0u, 0u};
        const float inv = 1.0f / l;

        // ---- O^T tile by tile: A = V^T[d = 16 tile + r16][keys 4 g .. 4 g + 3] from LDS
        const int64_t bs = p / H;
        const int h = (int)(p - bs * H);
        const int64_t bo = bs / S, s = bs - bo * S;
        T* const orow = out + (bo * Tn * S + s) * o_ts + (int64_t)h * kD + r16 * o_frame + 4 * g;
#pragma unroll
        for (int tile = 0; tile < 4; ++tile) {
            const char* const src = vt + (4 * g) * kVRow + 2 * (16 * tile + r16);
            const uint32_t v0 = *reinterpret_cast<const uint16_t*>(src), v1 = *reinterpret_cast<const uint16_t*>(src + kVRow);
            const uint32_t v2 = *reinterpret_cast<const uint16_t*>(src + 2 * kVRow), v3 = *reinterpret_cast<const uint16_t*>(src + 3 * kVRow);
            const u32x4 va = {v0 | (v1 << 16), v2 | (v3 << 16), 0u, 0u};
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            o = M::mfma(as_frag<frag>(va), as_frag<frag>(pb), o);
            if (frame_ok) {
                const u32x2 w = {M::pack2(o[0] * inv, o[1] * inv), M::pack2(o[2] * inv, o[3] * inv)};
                *reinterpret_cast<u32x2*>(orow + 16 * tile) = w;
            }
        }
        cur = nxt;
    }
}

}  // namespace at16

// Covered: bf16 / f16, D = 64, T <= 16, strides that keep every row 16-byte aligned (q, k, v) / 8-byte aligned (out).
bool attn_temporal16_ok(int T, int D, int dtype, int64_t hd, int64_t qkv_ts, int64_t o_ts, const void* q, const void* k, const void* v,
                        const void* out) {
    if (!(dtype == MVI_DT_BF16 || dtype == MVI_DT_F16) || D != at16::kD || T > 16) return false;
    const int64_t qs = qkv_ts ? qkv_ts : hd, os = o_ts ? o_ts : hd;
    return qs % 8 == 0 && os % 4 == 0 && ((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) % 16 == 0 && (uintptr_t)out % 8 == 0;
}

template <typename T>
int attn_temporal16_launch(const void* q, const void* k, const void* v, void* out, int Bo, int Tn, int S, int H, float scale, hipStream_t st,
                           int64_t qkv_ts, int64_t o_ts) {
    using namespace at16;
    const int64_t hd = (int64_t)H * kD;
    if (qkv_ts == 0) qkv_ts = hd;
    if (o_ts == 0) o_ts = hd;
    const int64_t n = (int64_t)Bo * S * H;
    if (n == 0) return 0;
    // as many waves as fit (88 registers: 5 per SIMD, 20 per CU on 256 CUs), each walking ~n / waves problems with one problem in flight
    int64_t blocks = (n + kWaves - 1) / kWaves;
    if (blocks > 256 * 5) blocks = 256 * 5;
    hipLaunchKernelGGL((attn_temporal16_kernel<T>), dim3((unsigned)blocks), dim3(64 * kWaves), 0, st, (const T*)q, (const T*)k, (const T*)v,
                       (T*)out, Tn, S, H, scale, n, qkv_ts, o_ts);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

template int attn_temporal16_launch<__hip_bfloat16>(const void*, const void*, const void*, void*, int, int, int, int, float, hipStream_t, int64_t, int64_t);
template int attn_temporal16_launch<__half>(const void*, const void*, const void*, void*, int, int, int, int, float, hipStream_t, int64_t, int64_t);

}  // namespace mvi
