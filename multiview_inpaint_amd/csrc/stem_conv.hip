// 3x3 convolution (stride 1, padding 1) with 16 output channels and at most 16 input channels, + bias + SiLU, NCHW bf16 / f16, for
// gfx950 — the first two layers of ControlNet.input_hint_block (svd_inpaint1/models/csvd.py:234-250: conv(7 -> 16), SiLU,
// conv(16 -> 16), SiLU at the 576 x 1024 hint resolution, once per ControlNet call). The library runs them at 1.16 + 1.09 ms for
// 28 frames (its kernels are built for wide channels) plus a pass for bias + SiLU over each 528 MB output; they move 0.76 / 1.06 GB,
// i.e. ~0.2 ms of HBM time each.
//
// Implicit GEMM on v_mfma_f32_16x16x32: M = 16 pixels of an output row, N = the 16 output channels, K = (tap, input channel) —
// 9 x 16 = 144 padded to 160 (five k-steps; CINP = 8 for the 7-channel hint: 9 x 8 = 72 padded to 96, three k-steps).
//   * block = 4 waves = a tile of 4 output rows x 64 pixels; its input patch (6 x 66 pixels, all input channels) is brought from
//     the NCHW planes with 16-byte loads and laid out in LDS pixel-major ([row][pixel][channel], 16 or 32 bytes per pixel), so the
//     A fragment of a k-step is ONE ds_read_b128 per lane: 8 consecutive channels of pixel (p + dx, row + dy);
//   * the weights are the B operand, built once per wave in registers (20 or 12 registers);
//   * the accumulator starts from the bias; a lane ends with 4 consecutive pixels of one output channel: SiLU, pack, one 8-byte store.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <cstdint>

#include "../../include/mvi_raster.h"
#include "../../include/mvi_unet_ops.h"

namespace mvi {
int unet_fail(int code, const char* msg);
namespace sc {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

constexpr int kTW = 64;                // tile width in pixels
constexpr int kTHMax = 8;              // tile rows = waves: 8 for the 16-output forms, 4 for the 32-output form (registers, LDS)

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

template <typename T, int CINP, int COUT, int kTH, int STRIDE>
__global__ __launch_bounds__(64 * kTH) void stem_conv3x3_kernel(const T* __restrict__ x, const T* __restrict__ wp, const float* __restrict__ bias,
                                                                T* __restrict__ y, int Cin, int H, int W, int Ho, int Wo, int silu) {
    using M = Mma<T>;
    using frag = typename M::frag;
    constexpr int kTapsPerStep = 32 / CINP;                       // 1 (32 channels), 2 (16) or 4 (8)
    constexpr int kSteps = (9 + kTapsPerStep - 1) / kTapsPerStep; // 9, 5 or 3
    constexpr int NT = COUT / 16;                                 // output-channel tiles of 16
    constexpr int kPH = STRIDE * (kTH - 1) + 3;                   // input rows / columns a tile of kTH x kTW outputs reads
    constexpr int kPW = STRIDE * kTW + 2;                         // (stride 2: 129 used, the interior is filled in whole 8-pixel pieces)
    constexpr int kInt = STRIDE * kTW / 8;                        // 8-pixel pieces of the patch interior per row
    // LDS patch, pixel-major with 16 bytes of padding after every 8 pixels (keeps the 16-byte alignment of the fragment reads and
    // spreads the 8-pixel pieces a wave writes at once over the banks)
    constexpr int kPixB = CINP * 2;                                             // bytes per pixel
    constexpr int kRowB = kPW * kPixB + 16 * ((kPW + 7) / 8);                   // bytes per patch row
    __shared__ __attribute__((aligned(16))) char s_in[kPH * kRowB];
    auto pix_off = [](int q) { return q * kPixB + (q >> 3) * 16; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x0 = blockIdx.x * kTW, y0 = blockIdx.y * kTH;      // output coordinates of the tile
    const int ix0 = STRIDE * x0, iy0 = STRIDE * y0 - 1;           // input column of patch column 1, input row of patch row 0
    const int64_t img = blockIdx.z;
    const uint16_t* xin = reinterpret_cast<const uint16_t*>(x) + img * Cin * (int64_t)H * W;

    // ---- input patch -> LDS. An item = 8 pixels of a channel PAIR (two 16-byte loads from two planes, eight 4-byte LDS writes);
    // consecutive lanes take consecutive pairs of the same pixels, i.e. consecutive LDS words. The two halo columns one element at a
    // time; everything outside the image, and the padding channels, is zero
    constexpr int kPairs = CINP / 2;
    for (int i = tid; i < kPairs * kInt * kPH; i += 64 * kTH) {
        const int cp = i % kPairs, ch8 = (i / kPairs) % kInt, r = i / (kPairs * kInt);
        const int gy = iy0 + r, gx = ix0 + 8 * ch8;
        u32x4 v0 = {0, 0, 0, 0}, v1 = {0, 0, 0, 0};
        if (gy >= 0 && gy < H && gx < W) {
            if (2 * cp < Cin) v0 = *reinterpret_cast<const u32x4*>(xin + ((int64_t)(2 * cp) * H + gy) * W + gx);
            if (2 * cp + 1 < Cin) v1 = *reinterpret_cast<const u32x4*>(xin + ((int64_t)(2 * cp + 1) * H + gy) * W + gx);
        }
        char* const dst = s_in + r * kRowB + 4 * cp;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            *reinterpret_cast<uint32_t*>(dst + pix_off(1 + 8 * ch8 + 2 * k)) = (v0[k] & 0xFFFFu) | (v1[k] << 16);
            *reinterpret_cast<uint32_t*>(dst + pix_off(1 + 8 * ch8 + 2 * k + 1)) = (v0[k] >> 16) | (v1[k] & 0xFFFF0000u);
        }
    }
    for (int i = tid; i < CINP * kPH * 2; i += 64 * kTH) {
        const int c = i % CINP, side = (i / CINP) & 1, r = i / (2 * CINP);
        const int gy = iy0 + r, gx = side ? ix0 + STRIDE * kTW : ix0 - 1;
        uint16_t v = 0;
        if (c < Cin && gy >= 0 && gy < H && gx >= 0 && gx < W) v = xin[((int64_t)c * H + gy) * W + gx];
        *reinterpret_cast<uint16_t*>(s_in + r * kRowB + pix_off(side ? STRIDE * kTW + 1 : 0) + 2 * c) = v;
    }

    // ---- weights: B operand. Lane (n = lane & 15, g = lane >> 4), k-step s, element j: tap = kTapsPerStep s + g / (CINP / 8),
    // channel = 8 (g % (CINP / 8)) + j; zero past tap 8 and past the real input channels. They arrive PACKED in that order
    // (stem_pack_weights_kernel, once per parameter version: [C_out / 16][k-steps][64 lanes][8]), one 16-byte load per fragment:
    // picked out of the raw [C_out][C_in][3][3] tensor they were 24 - 144 two-byte reads + as many pack operations per lane and
    // block — for the 32 -> 32 layer (18 fragments, 4-row tiles) more work than the convolution itself (0.50 ms for 528 MB).
    const int n = lane & 15, g = lane >> 4;
    __syncthreads();
    const int row = wave;                                         // tile row; input rows STRIDE row .. + 2 of the patch
    const int gy = y0 + row;
    const int m = lane & 15;
    // The output channels in chunks of NTC tiles of 16 (all of them at once for 16 / 32 outputs; four at a time for the 320 outputs of
    // the UNet / ControlNet input convolution, whose fragments would not fit the registers): per chunk the weight fragments are picked
    // from LDS, then the wave's row is walked in four M-tiles of 16 pixels (the A fragments are re-read from LDS per chunk)
    constexpr int NTC = NT < 4 ? NT : 4;
    static_assert(NT % NTC == 0, "output tiles must divide into chunks");
#pragma unroll 1
    for (int nc = 0; nc < NT; nc += NTC) {
        float b[NTC];
        u32x4 wf[NTC][kSteps];
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) {
            b[nt] = bias ? bias[16 * (nc + nt) + n] : 0.f;
#pragma unroll
            for (int s = 0; s < kSteps; ++s) wf[nt][s] = reinterpret_cast<const u32x4*>(wp)[((nc + nt) * kSteps + s) * 64 + lane];
        }
#pragma unroll
        for (int mt = 0; mt < kTW / 16; ++mt) {
            f32x4 acc[NTC];
#pragma unroll
            for (int nt = 0; nt < NTC; ++nt) acc[nt] = f32x4{b[nt], b[nt], b[nt], b[nt]};
#pragma unroll
            for (int s = 0; s < kSteps; ++s) {
                int tap = kTapsPerStep * s + g / (CINP / 8);
                tap = tap < 9 ? tap : 0;                          // (its weights are zero: any finite operand will do)
                const int dy = tap / 3, dx = tap % 3;
                const u32x4 a = *reinterpret_cast<const u32x4*>(s_in + (STRIDE * row + dy) * kRowB + pix_off(STRIDE * (16 * mt + m) + dx) + 16 * (g % (CINP / 8)));
#pragma unroll
                for (int nt = 0; nt < NTC; ++nt) acc[nt] = M::mfma(as_frag<frag>(a), as_frag<frag>(wf[nt][s]), acc[nt]);
            }
            // lane: output channel 16 (nc + nt) + n, pixels 16 mt + 4 g + 0..3 of the row
            const int gx = x0 + 16 * mt + 4 * g;
            if (gy < Ho && gx < Wo) {
#pragma unroll
                for (int nt = 0; nt < NTC; ++nt) {
                    float o[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float t = acc[nt][i];
                        o[i] = silu ? t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t * -1.4426950408889634f)) : t;
                    }
                    const u32x2 pk = {M::pack2(o[0], o[1]), M::pack2(o[2], o[3])};
                    *reinterpret_cast<u32x2*>(y + ((img * COUT + 16 * (nc + nt) + n) * Ho + gy) * (int64_t)Wo + gx) = pk;
                }
            }
        }
    }
}

// weight [C_out][C_in][3][3] (16-bit elements) -> [C_out / 16][k-steps][64 lanes][8]: lane (n, g) of tile nt, k-step s, element j
// = w[16 nt + n][8 (g % (CINP / 8)) + j][tap kTapsPerStep s + g / (CINP / 8)], zero past tap 8 / past C_in
__global__ __launch_bounds__(256) void stem_pack_weights_kernel(const uint16_t* __restrict__ w, uint16_t* __restrict__ packed, int Cin, int cinp,
                                                                int n_tiles, int k_steps) {
    const int i = blockIdx.x * 256 + threadIdx.x;                 // (tile, step, lane)
    if (i >= n_tiles * k_steps * 64) return;
    const int lane = i & 63, s = (i >> 6) % k_steps, nt = (i >> 6) / k_steps;
    const int n = lane & 15, g = lane >> 4, taps_per_step = 32 / cinp;
    const int tap = taps_per_step * s + g / (cinp / 8), c0 = 8 * (g % (cinp / 8));
    for (int j = 0; j < 8; ++j) packed[(int64_t)i * 8 + j] = (tap < 9 && c0 + j < Cin) ? w[((16 * nt + n) * Cin + c0 + j) * 9 + tap] : (uint16_t)0;
}

// the kernel form that serves (C_in, C_out, stride): padded input channels of its k-steps
static int stem_cinp(int Cin, int Cout, int stride) { return Cout == 320 ? 8 : stride == 2 ? 16 : Cout == 32 ? 32 : Cin <= 8 ? 8 : 16; }
static int stem_ksteps(int cinp) { const int tps = 32 / cinp; return (9 + tps - 1) / tps; }

}  // namespace sc
}  // namespace mvi

extern "C" int mvi_stem_conv3x3_supported(int32_t Cin, int32_t Cout, int32_t W, int32_t stride, int32_t dtype) {
    if (!(dtype == MVI_DT_BF16 || dtype == MVI_DT_F16) || Cin < 1) return 0;
    if (stride == 1) return ((Cout == 16 && Cin <= 16) || (Cout == 32 && Cin <= 32) || (Cout == 320 && Cin <= 8)) && W % 8 == 0;
    if (stride == 2) return Cout == 32 && Cin <= 16 && W % 16 == 0;             // (the output rows are stored in 8-byte pieces too)
    return 0;
}

template <typename T>
static void stem_conv_launch(const void* x, const void* w, const float* bias, void* y, int64_t N, int Cin, int Cout, int H, int W, int stride,
                             int silu, hipStream_t st) {
    using namespace mvi::sc;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const unsigned gx = (unsigned)((Wo + kTW - 1) / kTW);
    if (Cout == 320)
        hipLaunchKernelGGL((stem_conv3x3_kernel<T, 8, 320, 8, 1>), dim3(gx, (unsigned)((Ho + 7) / 8), (unsigned)N), dim3(512), 0, st, (const T*)x, (const T*)w,
                           bias, (T*)y, Cin, H, W, Ho, Wo, silu);
    else if (stride == 2)
        hipLaunchKernelGGL((stem_conv3x3_kernel<T, 16, 32, 4, 2>), dim3(gx, (unsigned)((Ho + 3) / 4), (unsigned)N), dim3(256), 0, st, (const T*)x, (const T*)w,
                           bias, (T*)y, Cin, H, W, Ho, Wo, silu);
    else if (Cout == 32)
        hipLaunchKernelGGL((stem_conv3x3_kernel<T, 32, 32, 4, 1>), dim3(gx, (unsigned)((Ho + 3) / 4), (unsigned)N), dim3(256), 0, st, (const T*)x, (const T*)w,
                           bias, (T*)y, Cin, H, W, Ho, Wo, silu);
    else if (Cin <= 8)
        hipLaunchKernelGGL((stem_conv3x3_kernel<T, 8, 16, 8, 1>), dim3(gx, (unsigned)((Ho + 7) / 8), (unsigned)N), dim3(512), 0, st, (const T*)x, (const T*)w,
                           bias, (T*)y, Cin, H, W, Ho, Wo, silu);
    else
        hipLaunchKernelGGL((stem_conv3x3_kernel<T, 16, 16, 8, 1>), dim3(gx, (unsigned)((Ho + 7) / 8), (unsigned)N), dim3(512), 0, st, (const T*)x, (const T*)w,
                           bias, (T*)y, Cin, H, W, Ho, Wo, silu);
}

extern "C" size_t mvi_stem_conv3x3_packed_bytes(int32_t Cin, int32_t Cout, int32_t stride) {
    if (Cin < 1 || Cout < 16 || Cout % 16) return 0;
    return (size_t)(Cout / 16) * mvi::sc::stem_ksteps(mvi::sc::stem_cinp(Cin, Cout, stride)) * 64 * 8 * 2;
}

extern "C" int mvi_stem_conv3x3_pack(const void* weight, void* packed, int32_t Cin, int32_t Cout, int32_t W, int32_t stride, int32_t dtype,
                                     void* stream) {
    if (!mvi_stem_conv3x3_supported(Cin, Cout, W, stride, dtype)) return mvi::unet_fail(MVI_EINVAL, "stem_conv3x3_pack: unsupported shape");
    if (!weight || !packed || (uintptr_t)packed % 16) return mvi::unet_fail(MVI_EINVAL, "stem_conv3x3_pack: NULL / unaligned pointer");
    const int cinp = mvi::sc::stem_cinp(Cin, Cout, stride), ks = mvi::sc::stem_ksteps(cinp), nt = Cout / 16;
    hipLaunchKernelGGL(mvi::sc::stem_pack_weights_kernel, dim3((unsigned)((nt * ks * 64 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)weight, (uint16_t*)packed, Cin, cinp, nt, ks);
    return hipGetLastError() == hipSuccess ? MVI_OK : mvi::unet_fail(MVI_EHIP, "stem_conv3x3_pack: kernel launch failed");
}

extern "C" int mvi_stem_conv3x3_silu(const void* x, const void* weight, const float* bias, void* y, int64_t N, int32_t Cin, int32_t Cout,
                                     int32_t H, int32_t W, int32_t stride, int32_t fuse_silu, int32_t dtype, void* stream) {
    if (N < 0 || H <= 0 || W <= 0 || !mvi_stem_conv3x3_supported(Cin, Cout, W, stride, dtype))
        return mvi::unet_fail(MVI_EINVAL, "stem_conv3x3: needs stride 1 with (C_out 16, C_in <= 16), (C_out 32, C_in <= 32) or (C_out 320, C_in <= 8) and W % 8 == 0, or stride 2 with "
                                          "C_out 32, C_in <= 16 and W % 16 == 0; bf16 or f16");
    if (N == 0) return MVI_OK;
    if (!x || !weight || !y) return mvi::unet_fail(MVI_EINVAL, "stem_conv3x3: NULL pointer");
    if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)weight) % 16 || N > 65535 || (H + 3) / 4 > 65535)
        return mvi::unet_fail(MVI_EINVAL, "stem_conv3x3: x / y / weight must be 16-byte aligned, N and H / 4 at most 65535");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MVI_DT_BF16) stem_conv_launch<__hip_bfloat16>(x, weight, bias, y, N, Cin, Cout, H, W, stride, fuse_silu, st);
    else stem_conv_launch<__half>(x, weight, bias, y, N, Cin, Cout, H, W, stride, fuse_silu, st);
    return hipGetLastError() == hipSuccess ? MVI_OK : mvi::unet_fail(MVI_EHIP, "stem_conv3x3: kernel launch failed");
}
