// Row compaction by one keep-mask over many tensors at once, for gfx950 — the tensor surgery of prune_points /
// _prune_optimizer (gs-simp/scene/gaussian_model.py:351-382): the reference evaluates `t[mask]` separately for the 6
// parameter tensors, their 12 Adam moment tensors and 3 per-Gaussian statistics, i.e. 21 x (nonzero + gather), each with
// its own host synchronisation. Here the mask is scanned ONCE into a source-row list (two small kernels + one 4-byte
// read-back of the kept count, needed to size the outputs) and ONE launch gathers the rows of every tensor.
// Integer / copy work: results are bit-identical to boolean indexing (tested). HBM-bound: every kept row is read and
// written once; the row list (4 B per kept row) is read once per tensor and stays in L2.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mvi_raster.h"
#include "../../include/mvi_train_ops.h"

namespace mvi {

int train_fail(int code, const char* msg);

constexpr int kCpBlock = 1024;      // mask entries per block of the count / fill kernels

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}

__global__ __launch_bounds__(kCpBlock) void compact_count_kernel(const uint8_t* __restrict__ mask, int P,
                                                                 uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t s_w[kCpBlock / 64];
    const int i = blockIdx.x * kCpBlock + threadIdx.x;
    const uint64_t b = __ballot(i < P && mask[i] != 0);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = (uint32_t)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t s = 0;
#pragma unroll
        for (int w = 0; w < kCpBlock / 64; ++w) s += s_w[w];
        block_sums[blockIdx.x] = s;
    }
}

// one block: exclusive scan of the block sums in place; offsets[n] = total
__global__ __launch_bounds__(1024) void compact_scan_kernel(uint32_t* __restrict__ sums, int n) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const uint32_t v = i < n ? sums[i] : 0u;
        const uint32_t inc = wave_incl_scan(v, lane);
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        uint32_t wave_off = 0;
        for (int w = 0; w < wave; ++w) wave_off += s_wave[w];
        const uint32_t carry = s_carry;
        if (i < n) sums[i] = carry + wave_off + inc - v;
        __syncthreads();
        if (tid == 1023) s_carry = carry + wave_off + inc;
        __syncthreads();
    }
    if (tid == 0) sums[n] = s_carry;
}

__global__ __launch_bounds__(kCpBlock) void compact_fill_kernel(const uint8_t* __restrict__ mask, int P,
                                                                const uint32_t* __restrict__ block_offsets,
                                                                uint32_t* __restrict__ src_rows) {
    __shared__ uint32_t s_w[kCpBlock / 64];
    const int i = blockIdx.x * kCpBlock + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool keep = i < P && mask[i] != 0;
    const uint64_t b = __ballot(keep);
    if (lane == 0) s_w[wave] = (uint32_t)__popcll(b);
    __syncthreads();
    uint32_t off = block_offsets[blockIdx.x];
    for (int w = 0; w < wave; ++w) off += s_w[w];
    if (keep) src_rows[off + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))] = (uint32_t)i;
}

// Gradient supports as BIT masks (view-parallel training: what the ranks tell each other about their views before the compacted
// exchange — P / 8 bytes per rank in one all-gather instead of a P-byte all-reduce(MAX)).
// pack: bits[w] bit b = flags[32 w + b] != 0, one word per thread from two 16-byte loads; union: mask[i] = 1 if any of the W
// gathered bit arrays has bit i set, one word (32 flags, two 16-byte stores) per thread.
__global__ __launch_bounds__(256) void support_pack_kernel(const uint8_t* __restrict__ flags, int P, uint32_t* __restrict__ bits, int words) {
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w >= words) return;
    const int i0 = 32 * w;
    uint32_t r = 0;
    if (i0 + 32 <= P && ((uintptr_t)flags & 15u) == 0) {
        const uint4 a = *reinterpret_cast<const uint4*>(flags + i0), b = *reinterpret_cast<const uint4*>(flags + i0 + 16);
        const uint32_t v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) r |= (((v[q] >> (8 * k)) & 0xFFu) ? 1u : 0u) << (4 * q + k);
    } else {
        for (int k = 0; k < 32 && i0 + k < P; ++k) r |= (flags[i0 + k] ? 1u : 0u) << k;
    }
    bits[w] = r;
}

__global__ __launch_bounds__(256) void support_union_kernel(const uint32_t* __restrict__ bits_all, int W, int words, int P,
                                                            uint8_t* __restrict__ mask) {
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w >= words) return;
    uint32_t r = 0;
    for (int k = 0; k < W; ++k) r |= bits_all[(size_t)k * words + w];
    const int i0 = 32 * w;
    if (i0 + 32 <= P && ((uintptr_t)mask & 15u) == 0) {
        uint32_t v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q)
            v[q] = ((r >> (4 * q)) & 1u) | (((r >> (4 * q + 1)) & 1u) << 8) | (((r >> (4 * q + 2)) & 1u) << 16) | (((r >> (4 * q + 3)) & 1u) << 24);
        *reinterpret_cast<uint4*>(mask + i0) = make_uint4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<uint4*>(mask + i0 + 16) = make_uint4(v[4], v[5], v[6], v[7]);
    } else {
        for (int k = 0; k < 32 && i0 + k < P; ++k) mask[i0 + k] = (uint8_t)((r >> k) & 1u);
    }
}

struct GatherTable {
    mvi_compact_tensor t[MVI_COMPACT_MAX_TENSORS];
};

// blockIdx.y = tensor; a block takes 256 kept rows at a time (their source-row ids staged in LDS) and copies their
// 256 * width 4-byte words with consecutive lanes on consecutive words of the OUTPUT (coalesced writes; reads are
// coalesced within a source row). The row of local word e comes from one fp32 multiply — exact while
// 256 * width < 2^22 (checked on the host) — not from an integer division.
__global__ __launch_bounds__(256) void compact_gather_kernel(GatherTable tab, const uint32_t* __restrict__ src_rows,
                                                             uint32_t n_keep) {
    __shared__ uint32_t s_src[256];
    const mvi_compact_tensor t = tab.t[blockIdx.y];
    const uint32_t w = (uint32_t)t.width;
    const uint32_t* __restrict__ in = (const uint32_t*)t.in;
    uint32_t* __restrict__ out = (uint32_t*)t.out;
    const float inv = 1.0f / (float)w;
    for (uint32_t row0 = blockIdx.x * 256u; row0 < n_keep; row0 += gridDim.x * 256u) {
        const uint32_t rows = min(256u, n_keep - row0);
        __syncthreads();
        if (threadIdx.x < rows) s_src[threadIdx.x] = src_rows[row0 + threadIdx.x];
        __syncthreads();
        const uint32_t words = rows * w;
        uint32_t* __restrict__ o = out + (uint64_t)row0 * w;
        for (uint32_t e = threadIdx.x; e < words; e += 256u) {
            const uint32_t j = (uint32_t)(((float)e + 0.5f) * inv);
            o[e] = in[(uint64_t)s_src[j] * w + (e - j * w)];
        }
    }
}

// The same gather — and its inverse — for a window [first, first + capacity) of the plan's row list whose length is only known
// on the device (*n_keep): view-parallel training exchanges the rows of the union of the ranks' gradient supports, sized by a
// capacity the host chose from the previous step, without waiting for this step's count (multiview_inpaint_amd/dist.py).
// kScatter = false: out[j, :] = in[src_row(first + j), :];  kScatter = true: out[src_row(first + j), :] = in[j, :] (in == NULL:
// zeros), for j < min(capacity, n_keep - first). Rows of `out` outside the window are not touched.
template <bool kScatter>
__global__ __launch_bounds__(256) void compact_window_kernel(GatherTable tab, const uint32_t* __restrict__ src_rows,
                                                             const uint32_t* __restrict__ n_keep_dev, uint32_t first,
                                                             uint32_t capacity) {
    __shared__ uint32_t s_src[256];
    const uint32_t n_all = *n_keep_dev;
    const uint32_t n = n_all > first ? min(capacity, n_all - first) : 0u;
    const mvi_compact_tensor t = tab.t[blockIdx.y];
    const uint32_t w = (uint32_t)t.width;
    const uint32_t ps = t.packed_stride > 0 ? (uint32_t)t.packed_stride : w;       // words between rows of the compact side
    const uint32_t* __restrict__ in = (const uint32_t*)t.in;
    uint32_t* __restrict__ out = (uint32_t*)t.out;
    const float inv = 1.0f / (float)w;
    for (uint32_t row0 = blockIdx.x * 256u; row0 < n; row0 += gridDim.x * 256u) {
        const uint32_t rows = min(256u, n - row0);
        __syncthreads();
        if (threadIdx.x < rows) s_src[threadIdx.x] = src_rows[first + row0 + threadIdx.x];
        __syncthreads();
        const uint32_t words = rows * w;
        for (uint32_t e = threadIdx.x; e < words; e += 256u) {
            const uint32_t j = (uint32_t)(((float)e + 0.5f) * inv);
            const uint64_t full = (uint64_t)s_src[j] * w + (e - j * w), packed = (uint64_t)(row0 + j) * ps + (e - j * w);
            if (kScatter) out[full] = in ? in[packed] : 0u;
            else out[packed] = in[full];
        }
    }
}

}  // namespace mvi

// workspace: [nblk + 1] block offsets (256-byte aligned segment) | [P] source rows of the kept entries
static inline size_t cp_offsets_bytes(int32_t P) {
    const size_t nblk = ((size_t)P + mvi::kCpBlock - 1) / mvi::kCpBlock;
    return (4 * (nblk + 1) + 255) & ~(size_t)255;
}
extern "C" size_t mvi_compact_workspace_bytes(int32_t P) {
    if (P < 0) return 0;
    return cp_offsets_bytes(P) + 4 * (size_t)(P > 0 ? P : 1);
}

static inline uint32_t* cp_offsets(void* ws) { return (uint32_t*)ws; }
static inline uint32_t* cp_rows(void* ws, int32_t P) { return (uint32_t*)((char*)ws + cp_offsets_bytes(P)); }

extern "C" int mvi_compact_plan(const uint8_t* keep_mask, int32_t P, void* workspace, size_t workspace_bytes,
                                uint32_t* n_keep_device, void* stream) {
    using namespace mvi;
    if (P < 0) return train_fail(MVI_EINVAL, "compact_plan: P < 0");
    if (!workspace || workspace_bytes < mvi_compact_workspace_bytes(P)) return train_fail(MVI_ENOMEM, "compact_plan: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    uint32_t* offs = cp_offsets(workspace);
    if (P == 0) {
        if (hipMemsetAsync(offs, 0, 4, st) != hipSuccess) return train_fail(MVI_EHIP, "compact_plan: memset failed");
        if (n_keep_device && hipMemsetAsync(n_keep_device, 0, 4, st) != hipSuccess) return train_fail(MVI_EHIP, "compact_plan: memset failed");
        return MVI_OK;
    }
    if (!keep_mask) return train_fail(MVI_EINVAL, "compact_plan: NULL mask");
    const int nblk = (P + kCpBlock - 1) / kCpBlock;
    hipLaunchKernelGGL(compact_count_kernel, dim3(nblk), dim3(kCpBlock), 0, st, keep_mask, P, offs);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, st, offs, nblk);
    hipLaunchKernelGGL(compact_fill_kernel, dim3(nblk), dim3(kCpBlock), 0, st, keep_mask, P, offs, cp_rows(workspace, P));
    if (hipGetLastError() != hipSuccess) return train_fail(MVI_EHIP, "compact_plan: kernel launch failed");
    if (n_keep_device &&
        hipMemcpyAsync(n_keep_device, offs + nblk, 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return train_fail(MVI_EHIP, "compact_plan: copy of the kept count failed");
    return MVI_OK;
}

extern "C" int mvi_support_pack_bits(const uint8_t* flags, int32_t P, uint32_t* bits, void* stream) {
    if (P < 0) return mvi::train_fail(MVI_EINVAL, "support_pack_bits: P < 0");
    if (P == 0) return MVI_OK;
    if (!flags || !bits) return mvi::train_fail(MVI_EINVAL, "support_pack_bits: NULL pointer");
    const int words = (P + 31) / 32;
    hipLaunchKernelGGL(mvi::support_pack_kernel, dim3((words + 255) / 256), dim3(256), 0, (hipStream_t)stream, flags, P, bits, words);
    return hipGetLastError() == hipSuccess ? MVI_OK : mvi::train_fail(MVI_EHIP, "support_pack_bits: kernel launch failed");
}

extern "C" int mvi_support_union_bits(const uint32_t* bits_all, int32_t n_ranks, int32_t P, uint8_t* mask, void* stream) {
    if (P < 0 || n_ranks < 1) return mvi::train_fail(MVI_EINVAL, "support_union_bits: bad argument");
    if (P == 0) return MVI_OK;
    if (!bits_all || !mask) return mvi::train_fail(MVI_EINVAL, "support_union_bits: NULL pointer");
    const int words = (P + 31) / 32;
    hipLaunchKernelGGL(mvi::support_union_kernel, dim3((words + 255) / 256), dim3(256), 0, (hipStream_t)stream, bits_all, n_ranks, words, P, mask);
    return hipGetLastError() == hipSuccess ? MVI_OK : mvi::train_fail(MVI_EHIP, "support_union_bits: kernel launch failed");
}

extern "C" int mvi_compact_gather(const mvi_compact_tensor* tensors_host, int32_t n_tensors, int32_t P, uint32_t n_keep,
                                  const void* workspace, void* stream) {
    using namespace mvi;
    if (n_tensors < 0 || n_tensors > MVI_COMPACT_MAX_TENSORS) return train_fail(MVI_EINVAL, "compact_gather: too many tensors");
    if (P < 0 || n_keep > (uint32_t)P) return train_fail(MVI_EINVAL, "compact_gather: n_keep exceeds P");
    if (n_tensors == 0 || n_keep == 0) return MVI_OK;
    if (!tensors_host || !workspace) return train_fail(MVI_EINVAL, "compact_gather: NULL pointer");
    GatherTable tab;
    for (int i = 0; i < n_tensors; ++i) {
        tab.t[i] = tensors_host[i];
        if (!tab.t[i].in || !tab.t[i].out || tab.t[i].width <= 0 || tab.t[i].width > 8192)
            return train_fail(MVI_EINVAL, "compact_gather: bad tensor entry (NULL pointer, or width outside 1..8192 words)");
    }
    uint32_t blocks = (n_keep + 255u) / 256u;
    if (blocks > 4096u) blocks = 4096u;                   // grid-stride beyond: 16 blocks per CU and tensor
    const uint32_t* rows = cp_rows(const_cast<void*>(workspace), P);
    hipLaunchKernelGGL(compact_gather_kernel, dim3(blocks, (unsigned)n_tensors), dim3(256), 0, (hipStream_t)stream, tab, rows, n_keep);
    return hipGetLastError() == hipSuccess ? MVI_OK : train_fail(MVI_EHIP, "compact_gather: kernel launch failed");
}

static int compact_window(bool scatter, const mvi_compact_tensor* tensors_host, int32_t n_tensors, int32_t P,
                          const uint32_t* n_keep_device, uint32_t first, uint32_t capacity, const void* workspace, void* stream) {
    using namespace mvi;
    const char* who = scatter ? "compact_scatter_window: bad argument" : "compact_gather_window: bad argument";
    if (n_tensors < 0 || n_tensors > MVI_COMPACT_MAX_TENSORS || P < 0 || capacity > (uint32_t)P || first > (uint32_t)P) return train_fail(MVI_EINVAL, who);
    if (n_tensors == 0 || capacity == 0) return MVI_OK;
    if (!tensors_host || !workspace || !n_keep_device) return train_fail(MVI_EINVAL, who);
    GatherTable tab;
    for (int i = 0; i < n_tensors; ++i) {
        tab.t[i] = tensors_host[i];
        // a scatter may have no input (zero fill); every other pointer is needed; 256 * width must stay below 2^22 (fp32 row index)
        if ((!tab.t[i].in && !scatter) || !tab.t[i].out || tab.t[i].width <= 0 || tab.t[i].width > 8192 ||
            (tab.t[i].packed_stride != 0 && tab.t[i].packed_stride < tab.t[i].width)) return train_fail(MVI_EINVAL, who);
    }
    uint32_t blocks = (capacity + 255u) / 256u;
    if (blocks > 4096u) blocks = 4096u;
    const uint32_t* rows = cp_rows(const_cast<void*>(workspace), P);
    if (scatter) hipLaunchKernelGGL(compact_window_kernel<true>, dim3(blocks, (unsigned)n_tensors), dim3(256), 0, (hipStream_t)stream, tab, rows, n_keep_device, first, capacity);
    else hipLaunchKernelGGL(compact_window_kernel<false>, dim3(blocks, (unsigned)n_tensors), dim3(256), 0, (hipStream_t)stream, tab, rows, n_keep_device, first, capacity);
    return hipGetLastError() == hipSuccess ? MVI_OK : train_fail(MVI_EHIP, "compact window: kernel launch failed");
}
extern "C" int mvi_compact_gather_window(const mvi_compact_tensor* tensors_host, int32_t n_tensors, int32_t P,
                                         const uint32_t* n_keep_device, uint32_t first, uint32_t capacity, const void* workspace,
                                         void* stream) {
    return compact_window(false, tensors_host, n_tensors, P, n_keep_device, first, capacity, workspace, stream);
}
extern "C" int mvi_compact_scatter_window(const mvi_compact_tensor* tensors_host, int32_t n_tensors, int32_t P,
                                          const uint32_t* n_keep_device, uint32_t first, uint32_t capacity, const void* workspace,
                                          void* stream) {
    return compact_window(true, tensors_host, n_tensors, P, n_keep_device, first, capacity, workspace, stream);
}
