// Tile binning for gfx950. Replaces the plug-in's InclusiveSum / duplicateWithKeys /
// DeviceRadixSort::SortPairs / identifyTileRanges stages behind
// gs-simp/gaussian_renderer/__init__.py:85-93 and produces the SAME sorted pair order (stable sort
// of key = tile << 32 | depth bits, ties in Gaussian-index order) — but never materialises or moves
// 64-bit keys:
//   level 1  sort the P Gaussians by depth bits (32-bit keys written by the preprocess kernel, 4 stable 8-bit
//            passes over P items);
//   emit     walk the Gaussians in that order and write one (tile id, Gaussian index) pair per
//            covered tile, coalesced per 256-Gaussian block;
//   level 2  stable partition of the D pairs by tile id (ceil(log2 T / 8) = 2 passes).
// HBM traffic per pair drops from 6 passes x 32 B to 2 passes x 20 B; the order is bit-identical
// because both levels are stable. Integer data: the parity tests compare it bit for bit.
#include "raster_common.h"

namespace mvi {

// In-stream zero fill (16 B per lane). hipMemsetAsync goes through the runtime's blit path and leaves ~6 us of idle GPU
// on either side of it (tools/experiments/raster_gaps.sh); an ordinary kernel does not. `bytes` is a multiple of 16.
__global__ __launch_bounds__(256) void zero_fill_kernel(uint4* __restrict__ p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
int launch_zero_fill(void* p, size_t bytes, hipStream_t st) {
    if (bytes == 0) return 0;
    const size_t n16 = (bytes + 15) / 16;
    size_t blocks = (n16 + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (uint4*)p, n16);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

// ---- exclusive scan of per-block sums (one 1024-thread block) ---------------------------------
__global__ __launch_bounds__(1024) void scan_block_sums_kernel(const uint32_t* __restrict__ sums,
                                                               uint32_t* __restrict__ offsets, int n) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        int i = base + tid;
        uint32_t v = i < n ? sums[i] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        uint32_t wave_off = 0;
        for (int w = 0; w < wave; ++w) wave_off += s_wave[w];
        uint32_t carry = s_carry;
        if (i < n) offsets[i] = carry + wave_off + inc - v;
        __syncthreads();
        if (tid == 1023) s_carry = carry + wave_off + inc;
        __syncthreads();
    }
    if (tid == 0) offsets[n] = s_carry;
}

// tiles touched per 256 depth-ordered Gaussians
__global__ __launch_bounds__(kBlock) void perm_block_sums_kernel(int P, const uint32_t* __restrict__ order,
                                                                 const uint32_t* __restrict__ tiles_touched,
                                                                 uint32_t* __restrict__ sums) {
    __shared__ uint32_t s_sum;
    const int tid = threadIdx.x;
    if (tid == 0) s_sum = 0;
    __syncthreads();
    int i = blockIdx.x * kBlock + tid;
    uint32_t t = i < P ? tiles_touched[order[i]] : 0u;
    uint32_t w = wave_sum_u32(t);
    if ((tid & 63) == 0 && w) atomicAdd(&s_sum, w);
    __syncthreads();
    if (tid == 0) sums[blockIdx.x] = s_sum;
}

// ---- pair emission in depth order: block = 256 consecutive entries of `order`; the block's output
// slots are written coalesced (a thread finds the Gaussian of its first slot by binary search over the in-block scan and
// walks from there; a mark + max-scan ownership pass per 256 slots measured slower: three barriers per round)
constexpr int kEmitRun = 8;                    // consecutive slots a thread produces per chunk
constexpr int kEmitChunk = kBlock * kEmitRun;  // slots staged in LDS per round

template <typename KeyT>
__global__ __launch_bounds__(kBlock) void emit_pairs_kernel(Frame f, GeomView g, const uint32_t* __restrict__ order,
                                                            const uint32_t* __restrict__ block_offsets,
                                                            KeyT* __restrict__ tile_keys,
                                                            uint32_t* __restrict__ vals) {
    __shared__ uint32_t s_off[kBlock];
    __shared__ uint32_t s_wave[4];
    __shared__ int s_x0[kBlock], s_y0[kBlock], s_w[kBlock];
    __shared__ uint32_t s_gi[kBlock], s_t[kBlock];
    __shared__ KeyT s_outk[kEmitChunk + kEmitChunk / kEmitRun];
    __shared__ uint32_t s_outv[kEmitChunk + kEmitChunk / kEmitRun];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pos = blockIdx.x * kBlock + tid;
    uint32_t t = 0, gi = 0;
    int x0 = 0, y0 = 0, x1 = 0;
    if (pos < f.P) {
        gi = order[pos];
        const uint2 rc = g.rect[gi];                          // the preprocess kernel's tile_rect result, one gather
        x0 = (int)(rc.x & 0xFFFFu); y0 = (int)(rc.x >> 16);
        const int w = (int)(rc.y & 0xFFFFu), h = (int)(rc.y >> 16);
        x1 = x0 + w;
        t = (uint32_t)(w * h);
    }
    s_x0[tid] = x0; s_y0[tid] = y0; s_w[tid] = x1 - x0; s_gi[tid] = gi; s_t[tid] = t;
    uint32_t inc = t;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t wave_off = 0;
    for (int w = 0; w < wave; ++w) wave_off += s_wave[w];
    s_off[tid] = wave_off + inc - t;
    const uint32_t total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
    const uint32_t base = block_offsets[blockIdx.x];
    // Chunks of kEmitChunk slots: every thread produces kEmitRun CONSECUTIVE slots — one binary search for the first,
    // then it walks (x, y) through the rectangle and on to the following Gaussians — into a padded LDS image, and the
    // block writes the chunk out with consecutive lanes on consecutive slots. One search + one division per kEmitRun
    // slots instead of per slot.
    for (uint32_t c0 = 0; c0 < total; c0 += kEmitChunk) {
        const uint32_t j0 = c0 + (uint32_t)tid * kEmitRun;
        if (j0 < total) {
            int lo = 0, hi = kBlock - 1;                      // largest entry with s_off <= j0 (it has t > 0)
            while (lo < hi) {
                int mid = (lo + hi + 1) >> 1;
                if (s_off[mid] <= j0) lo = mid; else hi = mid - 1;
            }
            uint32_t k = j0 - s_off[lo];
            int w = s_w[lo], x = (int)(k % (uint32_t)w), y = (int)(k / (uint32_t)w);
            uint32_t left = s_t[lo] - k;                      // slots of this Gaussian from here on
            int rowbase = (s_y0[lo] + y) * f.gx + s_x0[lo];
            uint32_t gcur = s_gi[lo];
            const int n = (int)min((uint32_t)kEmitRun, total - j0);
            const int p0 = tid * (kEmitRun + 1);              // stride kEmitRun + 1: conflict-free writes
            for (int i = 0; i < n; ++i) {
                s_outk[p0 + i] = (KeyT)(rowbase + x);
                s_outv[p0 + i] = gcur;
                if (--left == 0) {                            // next Gaussian that covers any tile
                    while (lo < kBlock - 1) { ++lo; if (s_t[lo]) break; }   // (past the last one only when no slot is left)
                    w = s_w[lo]; x = 0; left = s_t[lo]; gcur = s_gi[lo];
                    rowbase = s_y0[lo] * f.gx + s_x0[lo];
                } else if (++x == w) {
                    x = 0;
                    rowbase += f.gx;
                }
            }
        }
        __syncthreads();
        const uint32_t cn = min((uint32_t)kEmitChunk, total - c0);
        for (uint32_t q = tid; q < cn; q += kBlock) {
            const uint32_t pq = q + q / kEmitRun;             // padded position
            tile_keys[base + c0 + q] = s_outk[pq];
            vals[base + c0 + q] = s_outv[pq];
        }
        __syncthreads();
    }
}

// ---- stable 8-bit LSD radix pass over (u32 key, u32 value) pairs: count / scan / scatter -------
// Sort tile = 512 * kW pairs (kW waves per scatter block). The P-sized level-1 passes use kW = 4 (2048-pair tiles: they
// are short and need many blocks); the D-sized level-2 passes use kW = 8 (4096-pair tiles): a block's pairs leave in
// runs of tile / 2^bits pairs per digit, and longer runs mean fewer partially written cache lines.
// Count: one 1024-thread block = kCountTiles (4) consecutive sort tiles, one per 256-thread group. Row d of block_hist
// then receives 4 consecutive counts as one 16-byte store instead of four 4-byte words at a stride of nblk words
// (measured before: 23 MB written per launch for 3 MB of counts).
template <int kW, typename KeyT>
__global__ __launch_bounds__(1024) void radix_count_kernel(const KeyT* __restrict__ keys, int64_t D, int shift,
                                                           uint32_t mask, uint32_t* __restrict__ block_hist,
                                                           int nblk) {
    constexpr int kItems = 2 * kW;                       // per thread of a 256-thread group
    constexpr int kTileW = 512 * kW;
    __shared__ uint32_t s_hist[kCountTiles][256];
    const int tid = threadIdx.x, grp = tid >> 8, t = tid & 255;
    s_hist[grp][t] = 0;
    __syncthreads();
    const int64_t base = ((int64_t)blockIdx.x * kCountTiles + grp) * kTileW;
    uint32_t key[kItems];
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
        const int64_t idx = base + it * 256 + t;
        key[it] = idx < D ? (uint32_t)keys[idx] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int it = 0; it < kItems; ++it)
        if (base + it * 256 + t < D) atomicAdd(&s_hist[grp][(key[it] >> shift) & mask], 1u);
    __syncthreads();
    if (tid < 256) {
        uint4 v = make_uint4(s_hist[0][tid], s_hist[1][tid], s_hist[2][tid], s_hist[3][tid]);
        *reinterpret_cast<uint4*>(block_hist + (size_t)tid * nblk + (size_t)blockIdx.x * kCountTiles) = v;
    }
}

// block d turns row d of block_hist into exclusive offsets inside digit d; digit_tot[d] = row sum
__global__ __launch_bounds__(1024) void radix_scan_rows_kernel(uint32_t* __restrict__ block_hist, int nblk,
                                                               uint32_t* __restrict__ digit_tot) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    uint32_t* row = block_hist + (size_t)blockIdx.x * nblk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < nblk; base += 1024) {
        int i = base + tid;
        uint32_t v = i < nblk ? row[i] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        uint32_t wave_off = 0;
        for (int w = 0; w < wave; ++w) wave_off += s_wave[w];
        uint32_t carry = s_carry;
        if (i < nblk) row[i] = carry + wave_off + inc - v;
        __syncthreads();
        if (tid == 1023) s_carry = carry + wave_off + inc;
        __syncthreads();
    }
    if (tid == 0) digit_tot[blockIdx.x] = s_carry;
}

// Stable scatter. Wave w of a block owns 512 consecutive pairs, item `it` of lane l is pair
// w*512 + it*64 + l, so ranking items in `it` order inside a wave (64-bit __ballot peer masks) and
// waves in order keeps the input order among equal digits. The block's pairs are first put in
// digit order in LDS, then written out by consecutive lanes: every digit's run leaves as one
// contiguous, coalesced segment instead of 64 scattered dwords per wave-instruction.
// The per-digit bookkeeping (256 digits) is done by the first 256 threads of the block.
// kBits = digit width of the pass, known at compile time: the per-wave ranking costs one ballot (two VALU issues) per
// digit bit and item, and the D-sized passes (6 / 7-bit digits) sit at 71 % VALU issue (profiles/r02d): no ballots for
// bits the digit does not have.
template <int kW, typename KeyT, int kBits>
__global__ __launch_bounds__(64 * kW) void radix_scatter_kernel(
    const KeyT* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, KeyT* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out, int64_t D, int shift, uint32_t mask, const uint32_t* __restrict__ block_hist,
    int nblk, const uint32_t* __restrict__ digit_tot) {
    constexpr int kNT = 64 * kW, kTileW = 512 * kW;
    static_assert(kW >= 4, "the 256 digits are kept by the first 256 threads of the block");
    __shared__ uint32_t s_wave_hist[kW][256];
    __shared__ uint32_t s_digit_base[256];     // global position of this block's first pair of digit d
    __shared__ uint32_t s_local_start[256];    // position of digit d's run inside the block-sorted tile
    __shared__ uint32_t s_w4[4];
    __shared__ KeyT s_key[kTileW];
    __shared__ uint32_t s_val[kTileW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool digit_thread = tid < 256;
#pragma unroll
    for (int i = tid; i < kW * 256; i += kNT) (&s_wave_hist[0][0])[i] = 0;
    uint32_t dv = 0, dinc = 0;
    if (digit_thread) {   // exclusive scan of the 256 digit totals -> start of each digit in the output
        dv = digit_tot[tid];
        dinc = dv;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t t = __shfl_up(dinc, o);
            if (lane >= o) dinc += t;
        }
        if (lane == 63) s_w4[wave] = dinc;
    }
    __syncthreads();
    if (digit_thread) {
        uint32_t wave_off = 0;
        for (int w = 0; w < wave; ++w) wave_off += s_w4[w];
        s_digit_base[tid] = wave_off + dinc - dv + block_hist[(size_t)tid * nblk + blockIdx.x];
    }
    __syncthreads();

    const int64_t tile0 = (int64_t)blockIdx.x * kTileW;
    const int64_t base = tile0 + wave * (kSortItems * 64);
    uint32_t key[kSortItems], val[kSortItems], rank[kSortItems], dig[kSortItems];
    const uint64_t lanemask_lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int it = 0; it < kSortItems; ++it) {
        int64_t idx = base + it * 64 + lane;
        bool valid = idx < D;
        key[it] = valid ? (uint32_t)keys_in[idx] : 0u;
        val[it] = valid ? vals_in[idx] : 0u;
        uint32_t d = (key[it] >> shift) & mask;
        dig[it] = valid ? d : 0xFFFFFFFFu;
        uint64_t peers = __ballot(valid);                // lanes of this wave holding the same digit
#pragma unroll
        for (int b = 0; b < kBits; ++b) {
            uint64_t bit = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? bit : ~bit;
        }
        uint32_t before = (uint32_t)__popcll(peers & lanemask_lt);
        uint32_t prev = 0;
        if (valid) {
            prev = s_wave_hist[wave][d];
            // one writer per digit: the highest peer lane stores the new running count
            if ((peers >> lane) == 1ull) s_wave_hist[wave][d] = prev + (uint32_t)__popcll(peers);
        }
        rank[it] = prev + before;
    }
    __syncthreads();
    // per digit: offset of each wave inside the block's run, and the run's start in the sorted tile
    uint32_t tot = 0, inc = 0;
    if (digit_thread) {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < kW; ++w) {
            const uint32_t c = s_wave_hist[w][tid];
            s_wave_hist[w][tid] = run;                    // same thread reads and rewrites column tid
            run += c;
        }
        tot = run;
        inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (lane == 63) s_w4[wave] = inc;                 // last read of s_w4 was two barriers ago
    }
    __syncthreads();
    if (digit_thread) {
        uint32_t wave_off = 0;
        for (int w = 0; w < wave; ++w) wave_off += s_w4[w];
        s_local_start[tid] = wave_off + inc - tot;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kSortItems; ++it) {
        if (dig[it] != 0xFFFFFFFFu) {
            uint32_t d = dig[it];
            uint32_t lp = s_local_start[d] + s_wave_hist[wave][d] + rank[it];
            s_key[lp] = (KeyT)key[it];
            s_val[lp] = val[it];
        }
    }
    __syncthreads();
    const int count = (int)((D - tile0) < kTileW ? (D - tile0) : kTileW);
#pragma unroll
    for (int it = 0; it < kSortItems; ++it) {
        int lp = it * kNT + tid;
        if (lp < count) {
            const KeyT kk = s_key[lp];
            uint32_t k = (uint32_t)kk;
            uint32_t d = (k >> shift) & mask;
            uint32_t dst = s_digit_base[d] + ((uint32_t)lp - s_local_start[d]);
            keys_out[dst] = kk;
            vals_out[dst] = s_val[lp];
        }
    }
}

template <typename KeyT>
__global__ __launch_bounds__(kBlock) void tile_ranges_kernel(const KeyT* __restrict__ tile_keys, int64_t D,
                                                             uint32_t* __restrict__ ranges) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= D) return;
    uint32_t t = tile_keys[i];
    if (i == 0) ranges[2 * t] = 0;
    else {
        uint32_t tp = tile_keys[i - 1];
        if (t != tp) { ranges[2 * tp + 1] = (uint32_t)i; ranges[2 * t] = (uint32_t)i; }
    }
    if (i == D - 1) ranges[2 * t + 1] = (uint32_t)D;
}

// The per-block sums of the preprocess kernel are only needed as their TOTAL (num_rendered; pair emission works from
// the depth-ordered sums): one block adds them up with no barrier inside the loop and leaves the total where the
// scan used to put it (offsets[n]).
__global__ __launch_bounds__(1024) void total_block_sums_kernel(const uint32_t* __restrict__ sums,
                                                                uint32_t* __restrict__ offsets, int n,
                                                                unsigned long long* __restrict__ total_host) {
    __shared__ unsigned long long s_wave[16];
    unsigned long long acc = 0;                       // 64-bit: the host rejects totals the 32-bit pair offsets cannot hold
    for (int i = threadIdx.x; i < n; i += 1024) acc += sums[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += s_wave[w];
        offsets[n] = (uint32_t)t;
        // num_rendered goes straight into the caller's pinned, device-mapped word: no D2H copy launch (its blit path idles
        // the queue like the memset's); the event recorded behind this kernel makes the store visible to the host
        if (total_host) __hip_atomic_store(total_host, t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int launch_scan_block_sums(GeomView g, int P, unsigned long long* total_host_devptr, hipStream_t st) {
    int nblk = (P + kPB - 1) / kPB;
    hipLaunchKernelGGL(total_block_sums_kernel, dim3(1), dim3(1024), 0, st, g.block_sums, g.block_offsets, nblk,
                       total_host_devptr);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

// one stable pass over n pairs in tiles of 512 * kW; nsort = sort_blocks(n, kW); returns the index (0/1) of the buffer
// holding the result
template <int kW, typename KeyT>
static int radix_pass(KeyT* const keys[2], uint32_t* const vals[2], int cur, int64_t n, int shift, int bits,
                      uint32_t* hist, uint32_t* tot, int nsort, hipStream_t st) {
    uint32_t mask = (1u << bits) - 1u;
    hipLaunchKernelGGL((radix_count_kernel<kW, KeyT>), dim3(nsort / kCountTiles), dim3(1024), 0, st, keys[cur], n, shift, mask, hist, nsort);
    hipLaunchKernelGGL(radix_scan_rows_kernel, dim3(256), dim3(1024), 0, st, hist, nsort, tot);
#define MVI_SCATTER(B)                                                                                                       \
    case B:                                                                                                                  \
        hipLaunchKernelGGL((radix_scatter_kernel<kW, KeyT, B>), dim3(nsort), dim3(64 * kW), 0, st, keys[cur], vals[cur],      \
                           keys[cur ^ 1], vals[cur ^ 1], n, shift, mask, hist, nsort, tot);                                   \
        break;
    switch (bits) {
        MVI_SCATTER(1) MVI_SCATTER(2) MVI_SCATTER(3) MVI_SCATTER(4) MVI_SCATTER(5) MVI_SCATTER(6) MVI_SCATTER(7)
        default:
            hipLaunchKernelGGL((radix_scatter_kernel<kW, KeyT, 8>), dim3(nsort), dim3(64 * kW), 0, st, keys[cur], vals[cur],
                               keys[cur ^ 1], vals[cur ^ 1], n, shift, mask, hist, nsort, tot);
            break;
    }
#undef MVI_SCATTER
    return cur ^ 1;
}

// Level 1 + the scan of tiles touched in depth order. Needs nothing that depends on num_rendered, so
// the forward enqueues it BEFORE the host reads num_rendered back: the device stays busy while the
// host allocates the per-pair scratch.
int launch_binning_level1(const Frame& f, GeomView g, hipStream_t st) {
    if (f.P <= 0) return 0;
    const int nblk = (f.P + kBlock - 1) / kBlock;
    {   // Gaussians by depth (4 passes over P items; even count -> result back in buffer 0)
        StageTimer tm(kStSort, st);
        // the keys (depth bits, 0xFFFFFFFF for culled) and indices were written by the preprocess kernel
        int cur = 0;
        for (int p = 0; p < 4; ++p)
            cur = radix_pass<kSortWavesP, uint32_t>(g.dkeys, g.dvals, cur, f.P, 8 * p, 8, g.dhist, g.dtot, g.nsortP, st);
    }
    StageTimer tm(kStDup, st);
    hipLaunchKernelGGL(perm_block_sums_kernel, dim3(nblk), dim3(kBlock), 0, st, f.P, g.dvals[0], g.tiles_touched,
                       g.perm_sums);
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(1024), 0, st, g.perm_sums, g.perm_offsets, nblk);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

// Emission in depth order + level 2. After this call the sorted pairs are in b.keys[b.passes & 1]
// (tile ids), b.vals[b.passes & 1].
// Emission in depth order + level 2 + tile ranges for one tile-id type. After this call the sorted pairs are in
// b.keys[b.passes & 1] (tile ids), b.vals[b.passes & 1].
template <typename KeyT>
static int binning_typed(const Frame& f, GeomView g, BinningView b, ImageView im, int64_t D, hipStream_t st) {
    KeyT* const keys[2] = {(KeyT*)b.keys[0], (KeyT*)b.keys[1]};
    const int nblk = (f.P + kBlock - 1) / kBlock;
    {
        StageTimer tm(kStDup, st);
        hipLaunchKernelGGL((emit_pairs_kernel<KeyT>), dim3(nblk), dim3(kBlock), 0, st, f, g, g.dvals[0], g.perm_offsets,
                           keys[0], b.vals[0]);
    }
    int cur = 0;
    {   // level 2: stable partition by tile id
        StageTimer tm(kStSort, st);
        // the tile bits are split evenly over the passes (13 bits: 7 + 6, not 8 + 5): a block's 4096 pairs leave in
        // runs of 4096 / 2^bits pairs per digit, and the first pass's 32-byte runs were the least coalesced stores
        int shift = 0;
        for (int p = 0; p < b.passes; ++p) {
            const int left = b.passes - p;
            const int bits = (b.key_bits - shift + left - 1) / left;
            cur = radix_pass<kSortWavesD, KeyT>(keys, b.vals, cur, D, shift, bits, b.block_hist, b.digit_tot, b.nsort, st);
            shift += bits;
        }
    }
    int nrb = (int)((D + kBlock - 1) / kBlock);
    StageTimer tm(kStRanges, st);
    hipLaunchKernelGGL((tile_ranges_kernel<KeyT>), dim3(nrb), dim3(kBlock), 0, st, keys[cur], D, im.ranges);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

// Tile ids travel as 16-bit words whenever the image has at most 65536 tiles (any image up to 4096 x 4096): a pair is then
// 6 bytes instead of 8 in every pass of emission, partition and range finding (36 instead of 52 bytes per pair in all).
int launch_binning(const Frame& f, GeomView g, const int32_t* radii, BinningView b, ImageView im,
                   int64_t D, hipStream_t st) {
    size_t tiles = (size_t)f.gx * f.gy;
    if (launch_zero_fill(im.ranges, 8 * tiles, st)) return MVI_EHIP;
    if (D <= 0 || f.P <= 0) return 0;
    return b.key_bytes == 2 ? binning_typed<uint16_t>(f, g, b, im, D, st) : binning_typed<uint32_t>(f, g, b, im, D, st);
}

}  // namespace mvi
