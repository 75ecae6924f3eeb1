// Tile binning, version 2, for gfx950: tile grids of at most 256 x 256 tiles (images up to 4096 x 4096; larger ones keep
// raster_binning.hip). Same result as the plug-in's InclusiveSum / duplicateWithKeys / DeviceRadixSort::SortPairs /
// identifyTileRanges behind gs-simp/gaussian_renderer/__init__.py:85-93 — the stable order of key = tile << 32 | depth bits,
// ties in Gaussian-index order, and the per-tile ranges — bit for bit, but no (tile, Gaussian) pair is written before it is
// written to its final place:
//   depth sort   the P Gaussians by depth bits: 4 stable 8-bit passes of TWO launches each (count, scatter). The count kernel
//                leaves a row of digit counts per 4096-key tile and adds it into a row per 16 tiles; the scatter block sums the
//                few rows in front of it itself: no scan launch, no spin-wait;
//   pass 1       a Gaussian's tile rectangle [x0, x0 + w) x [y0, y0 + h) is w COLUMN SEGMENTS (Gaussian, first row, rows).
//                The Gaussians are walked in depth order and their segments are partitioned by tile column (stable): the
//                output holds, column by column, the segments in depth order — 6 bytes per segment, 3.3 times fewer items
//                than pairs, and the input is the 8-byte rectangle, not a list of pairs;
//   pass 2       every column is cut into chunks of 2048 segments; a chunk's segments are expanded along y and partitioned by
//                tile row (stable). (row, column, depth order) IS the tile-major order, so the output is the sorted point
//                list; the row-scan of the chunk table knows where every (row, column) starts, so the tile ranges fall out of
//                it and the first chunk of each column writes them (every tile: empty ones get (0, 0) like the plug-in's
//                zero-initialised ranges).
// Both passes are the same kernel (expand_scatter_kernel): items with an extent [a, a + n) over at most 256 bins are expanded
// into (item, bin) entries and partitioned by bin, without keys and without ballots: every item ORs its lane bit into the LDS
// mask of each bin it covers (one 64-bit mask per wave round and bin); the thread that OWNS a bin then walks the set bits of
// the bin's masks in round order — that is the bin's entries in item order — and writes them into the block's LDS image, which
// leaves in bin runs.
// Integer data: the parity tests compare point list, tile ids and ranges bit for bit with the oracle and with version 1.
#include <cstdlib>

#include "raster_common.h"

namespace mvi {

static int g_binning_version = [] { const char* e = getenv("MVI_BINNING_LEGACY"); return (e && e[0] == '1') ? 1 : 2; }();
bool binning_v2_enabled() { return g_binning_version == 2; }
int set_binning_version(int v) {
    const int old = g_binning_version;
    if (v == 1 || v == 2) g_binning_version = v;
    return old;
}

// exclusive scan over the values of the first 256 threads of a block (the others pass 0); every thread must call it
__device__ __forceinline__ uint32_t block_excl_scan256(uint32_t v, int tid, uint32_t* s_w4, uint32_t* total) {
    const int lane = tid & 63, wave = tid >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (wave < 4 && lane == 63) s_w4[wave] = inc;
    __syncthreads();
    uint32_t off = 0;
#pragma unroll
    for (int w = 0; w < 3; ++w) if (w < wave) off += s_w4[w];
    if (total) *total = s_w4[0] + s_w4[1] + s_w4[2] + s_w4[3];
    __syncthreads();
    return off + inc - v;
}

// =================================================================================================== totals
// num_rendered (pairs) and the number of column segments from the preprocess kernel's per-block sums, straight into the
// caller's pinned, device-mapped words; also zeroes the super rows the first depth-sort count adds into.
__global__ __launch_bounds__(1024) void totals2_kernel(const uint32_t* __restrict__ pair_sums,
                                                       const uint32_t* __restrict__ seg_sums, int n,
                                                       uint32_t* __restrict__ super0, int super_words,
                                                       uint32_t* __restrict__ arrivals,
                                                       unsigned long long* __restrict__ totals_host) {
    __shared__ unsigned long long s_wave[2][16];
    const int tid = threadIdx.x;
    for (int i = tid; i < super_words; i += 1024) super0[i] = 0u;
    if (tid == 0) *arrivals = 0u;                 // scratch is uninitialised memory: the column scan counts its blocks here
    unsigned long long a = 0, b = 0;
    const int n4 = n >> 2;
    const uint4* p4 = reinterpret_cast<const uint4*>(pair_sums);
    const uint4* s4 = reinterpret_cast<const uint4*>(seg_sums);
#pragma unroll 4
    for (int i = tid; i < n4; i += 1024) {     // independent 16-byte loads
        const uint4 u = p4[i], v = s4[i];
        a += (unsigned long long)u.x + u.y + u.z + u.w;
        b += (unsigned long long)v.x + v.y + v.z + v.w;
    }
    for (int i = 4 * n4 + tid; i < n; i += 1024) { a += pair_sums[i]; b += seg_sums[i]; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d); b += __shfl_xor(b, d); }
    if ((tid & 63) == 0) { s_wave[0][tid >> 6] = a; s_wave[1][tid >> 6] = b; }
    __syncthreads();
    if (tid < 2) {
        unsigned long long t = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += s_wave[tid][w];
        if (totals_host) __hip_atomic_store(totals_host + tid, t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
int launch_binning2_totals(GeomView g, int P, unsigned long long* totals_host_devptr, hipStream_t st) {
    const int npre = (P + kPB - 1) / kPB;
    hipLaunchKernelGGL(totals2_kernel, dim3(1), dim3(1024), 0, st, g.block_sums, g.seg_sums, npre, g.ds_super[0],
                       256 * g.nsuper, g.arrivals, totals_host_devptr);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

// =================================================================================================== depth sort
__device__ __forceinline__ void lds_wave_sync() {
    // LDS operations of one wave complete in order; behind the wait every lane sees every other lane's LDS writes / atomics
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// count: table[tile][digit] for the tile's 2048 keys (one coalesced 1 KB row per tile), added into super[tile / 16][digit]
__global__ __launch_bounds__(kDsThreads) void depth_count_kernel(const uint32_t* __restrict__ keys, int P, int shift,
                                                                 uint32_t* __restrict__ table,
                                                                 uint32_t* __restrict__ super) {
    static_assert(kDsThreads == 256, "thread = digit");
    __shared__ uint32_t s_h[256];
    const int tid = threadIdx.x;
    s_h[tid] = 0;
    __syncthreads();
    const int base = blockIdx.x * kDsTile;
    uint4 k4[kDsItems / 4];
#pragma unroll
    for (int it = 0; it < kDsItems / 4; ++it) {
        const int i0 = base + (it * kDsThreads + tid) * 4;
        k4[it] = i0 < P ? *reinterpret_cast<const uint4*>(keys + i0) : make_uint4(0u, 0u, 0u, 0u);   // buffer padded to 256 B
    }
#pragma unroll
    for (int it = 0; it < kDsItems / 4; ++it) {
        const int i0 = base + (it * kDsThreads + tid) * 4;
        if (i0 < P) {
            const uint32_t k[4] = {k4[it].x, k4[it].y, k4[it].z, k4[it].w};
            // the high digits of depth bits are nearly constant: merge equal digits of a thread before the LDS atomic
            uint32_t d_prev = (k[0] >> shift) & 255u, run = 1;
#pragma unroll
            for (int j = 1; j < 4; ++j) {
                const uint32_t d = (k[j] >> shift) & 255u;
                if (i0 + j < P) {
                    if (d == d_prev) ++run;
                    else { atomicAdd(&s_h[d_prev], run); d_prev = d; run = 1; }
                }
            }
            atomicAdd(&s_h[d_prev], run);
        }
    }
    __syncthreads();
    const uint32_t c = s_h[tid];
    table[(size_t)blockIdx.x * 256 + tid] = c;
    if (c) atomicAdd(&super[(size_t)(blockIdx.x / kDsSuper) * 256 + tid], c);
}

// Stable scatter of one 8-bit pass. Wave w of a block owns 512 consecutive keys, item `it` of lane l is key
// w * 512 + it * 64 + l; items are ranked in `it` order inside a wave and waves in order, which keeps the input order among
// equal digits. The rank of a key among the 64 keys of its item row: every lane ORs its bit into the LDS mask of its digit
// (ds_or_b64), reads the mask back, and counts the lower lanes — two population counts instead of eight ballots and their
// selects (119 -> ~20 vector instructions per key); the highest lane of a digit advances the wave's running count and clears
// the mask. Where each digit starts in the output and how many keys of it the earlier tiles hold: every super row gives the
// digit totals, the super rows in front of this tile's group plus the table rows of the group's earlier tiles give the
// predecessors — nsuper + 15 coalesced 1 KB rows, read as 16-byte pieces by four row groups. Block 0 also zeroes the super
// rows the NEXT pass's count adds into.
// PASS 0: the value of a key is its index (the preprocess kernel writes no index array); PASS 3: only the values leave, and
// the tile rectangle of every Gaussian travels to its depth-ordered place (rect_sorted): the one random 8-byte gather per
// Gaussian the binning needs is in flight underneath the ranking, and the later kernels read rectangles coalesced (the top
// digit of depth bits takes 2 - 3 values, so these writes are nearly contiguous).
template <int PASS>
__global__ __launch_bounds__(kDsThreads) void depth_scatter_kernel(
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out, int P, const uint32_t* __restrict__ table, const uint32_t* __restrict__ super,
    int nsuper, uint32_t* __restrict__ super_next, const uint2* __restrict__ rect, uint2* __restrict__ rect_sorted) {
    constexpr int kW = kDsThreads / 64;       // 4
    constexpr int shift = 8 * PASS;
    __shared__ uint32_t s_wave_hist[kW][256];
    __shared__ unsigned long long s_match[kW][256];
    __shared__ uint32_t s_digit_base[256];
    __shared__ uint32_t s_local_start[256];
    __shared__ uint32_t s_red[4][2][256];     // [row group][all / before][digit]
    __shared__ uint32_t s_w4[4];
    __shared__ uint32_t s_key[kDsTile];
    __shared__ uint32_t s_val[kDsTile];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int i = tid; i < kW * 256; i += kDsThreads) { (&s_wave_hist[0][0])[i] = 0; (&s_match[0][0])[i] = 0ull; }

    const int tile0 = blockIdx.x * kDsTile;
    const int base = tile0 + wave * (kDsItems * 64);
    uint32_t key[kDsItems], val[kDsItems];
#pragma unroll
    for (int it = 0; it < kDsItems; ++it) {
        const int idx = base + it * 64 + lane;
        const bool valid = idx < P;
        key[it] = valid ? keys_in[idx] : 0u;
        val[it] = PASS == 0 ? (uint32_t)idx : (valid ? vals_in[idx] : 0u);
    }
    uint2 rc[PASS == 3 ? kDsItems : 1];
    if (PASS == 3) {
#pragma unroll
        for (int it = 0; it < kDsItems; ++it)
            rc[it] = (base + it * 64 + lane) < P ? rect[val[it]] : make_uint2(0u, 0u);
    }
    {   // row group rg sums rows rg, rg + 4, .. of the digits 4 dq .. 4 dq + 3
        const int dq = tid & 63, rg = tid >> 6;
        const int grp = (int)blockIdx.x / kDsSuper;
        uint4 all = make_uint4(0u, 0u, 0u, 0u), bef = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll 4
        for (int j = rg; j < nsuper; j += 4) {
            const uint4 c = *reinterpret_cast<const uint4*>(super + (size_t)j * 256 + 4 * dq);
            all.x += c.x; all.y += c.y; all.z += c.z; all.w += c.w;
            if (j < grp) { bef.x += c.x; bef.y += c.y; bef.z += c.z; bef.w += c.w; }
        }
#pragma unroll
        for (int j = rg; j < kDsSuper - 1; j += 4) {
            const int t = grp * kDsSuper + j;
            if (t < (int)blockIdx.x) {
                const uint4 c = *reinterpret_cast<const uint4*>(table + (size_t)t * 256 + 4 * dq);
                bef.x += c.x; bef.y += c.y; bef.z += c.z; bef.w += c.w;
            }
        }
        *reinterpret_cast<uint4*>(&s_red[rg][0][4 * dq]) = all;
        *reinterpret_cast<uint4*>(&s_red[rg][1][4 * dq]) = bef;
        if (super_next && blockIdx.x == 0)
            for (int i = tid; i < 256 * nsuper; i += kDsThreads) super_next[i] = 0u;
    }
    __syncthreads();
    const uint32_t all = s_red[0][0][tid] + s_red[1][0][tid] + s_red[2][0][tid] + s_red[3][0][tid];
    const uint32_t before = s_red[0][1][tid] + s_red[1][1][tid] + s_red[2][1][tid] + s_red[3][1][tid];
    uint32_t dinc = all;                      // exclusive scan of the 256 digit totals -> start of each digit in the output
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(dinc, o);
        if (lane >= o) dinc += t;
    }
    if (lane == 63) s_w4[wave] = dinc;

    // ranks inside the wave's 512 keys
    uint32_t rank[kDsItems], dig[kDsItems];
    const unsigned long long bit = 1ull << lane, lt = bit - 1ull;
    unsigned long long* const mk = s_match[wave];
    uint32_t* const cnt = s_wave_hist[wave];
#pragma unroll
    for (int it = 0; it < kDsItems; ++it) {
        const bool valid = (base + it * 64 + lane) < P;
        const uint32_t d = (key[it] >> shift) & 255u;
        dig[it] = valid ? d : 0xFFFFFFFFu;
        if (valid) atomicOr(&mk[d], bit);
        lds_wave_sync();
        const unsigned long long m = valid ? mk[d] : 0ull;
        const uint32_t prev = valid ? cnt[d] : 0u;
        lds_wave_sync();                                   // every lane has read before the digit's highest lane writes
        if (valid && (m >> lane) == 1ull) { cnt[d] = prev + (uint32_t)__popcll(m); mk[d] = 0ull; }
        rank[it] = prev + (uint32_t)__popcll(m & lt);
    }
    __syncthreads();
    {
        uint32_t wave_off = 0;
        for (int w = 0; w < wave; ++w) wave_off += s_w4[w];
        s_digit_base[tid] = wave_off + dinc - all + before;
    }
    // per digit: offset of each wave inside the block's run, and the run's start in the sorted tile
    uint32_t tot = 0;
    {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < kW; ++w) {
            const uint32_t c = s_wave_hist[w][tid];
            s_wave_hist[w][tid] = run;                    // same thread reads and rewrites column tid
            run += c;
        }
        tot = run;
    }
    uint32_t inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    __syncthreads();                                      // s_w4 of the first scan has been read
    if (lane == 63) s_w4[wave] = inc;
    __syncthreads();
    {
        uint32_t wave_off = 0;
        for (int w = 0; w < wave; ++w) wave_off += s_w4[w];
        s_local_start[tid] = wave_off + inc - tot;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kDsItems; ++it) {
        if (dig[it] != 0xFFFFFFFFu) {
            const uint32_t d = dig[it];
            const uint32_t lp = s_local_start[d] + s_wave_hist[wave][d] + rank[it];
            s_key[lp] = key[it];
            s_val[lp] = val[it];
            if (PASS == 3) rect_sorted[s_digit_base[d] + (lp - s_local_start[d])] = rc[it];
        }
    }
    __syncthreads();
    const int count = (P - tile0) < kDsTile ? (P - tile0) : kDsTile;
#pragma unroll
    for (int it = 0; it < kDsItems; ++it) {
        const int lp = it * kDsThreads + tid;
        if (lp < count) {
            const uint32_t k = s_key[lp];
            const uint32_t d = (k >> shift) & 255u;
            const uint32_t dst = s_digit_base[d] + ((uint32_t)lp - s_local_start[d]);
            if (PASS != 3) keys_out[dst] = k;
            vals_out[dst] = s_val[lp];
        }
    }
}

// =================================================================================================== pass 1 count
// Column segments per (tile column, block of kExChunk depth-ordered Gaussians): a difference array (+1 at x0, -1 at x0 + w,
// two LDS atomics per Gaussian) and one prefix sum over the rectangles the last sort pass left in depth order.
__global__ __launch_bounds__(kExThreads) void column_count_kernel(int P, int gx, const uint2* __restrict__ rect_sorted,
                                                                  uint32_t* __restrict__ col_table, int nblk1) {
    __shared__ uint32_t s_d[257];
    __shared__ uint32_t s_w4[4];
    const int tid = threadIdx.x;
    if (tid <= 256) s_d[tid] = 0;
    const int base = blockIdx.x * kExChunk;
    uint2 rc[kExItems];
#pragma unroll
    for (int k = 0; k < kExItems; ++k) {
        const int i = base + k * kExThreads + tid;
        rc[k] = i < P ? rect_sorted[i] : make_uint2(0u, 0u);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kExItems; ++k) {
        const uint32_t x0 = rc[k].x & 0xFFFFu, w = rc[k].y & 0xFFFFu, h = rc[k].y >> 16;
        if (w * h) { atomicAdd(&s_d[x0], 1u); atomicSub(&s_d[x0 + w], 1u); }
    }
    __syncthreads();
    const uint32_t v = tid < 256 ? s_d[tid] : 0u;
    const uint32_t ex = block_excl_scan256(v, tid, s_w4, nullptr);
    if (tid < gx) col_table[(size_t)tid * nblk1 + blockIdx.x] = ex + v;      // inclusive prefix of the differences
}

// =================================================================================================== row scans
// Chunk table of pass 2 from the column totals: chunk_first[x] = first chunk of tile column x, chunk_first[gx] = number of
// chunks; col_start[x] = first segment of column x, col_start[gx] = number of segments. Every column owns at least one chunk
// (an empty one writes the column's empty ranges). Built by the block of columns_scan_kernel whose arrival is last.
__device__ __forceinline__ void build_chunk_table(const uint32_t* col_tot, int gx, int tid, uint32_t* __restrict__ chunk_first,
                                                  uint32_t* __restrict__ col_start, uint32_t* s_w4) {
    // the totals were stored by other blocks with agent-scope stores, each drained before that block's arrival: L2 loads
    const uint32_t v = tid < gx ? __hip_atomic_load(col_tot + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    const uint32_t c = tid < gx ? max(1u, (v + (uint32_t)kExChunk - 1u) / (uint32_t)kExChunk) : 0u;
    uint32_t tv, tc;
    const uint32_t ev = block_excl_scan256(v, tid, s_w4, &tv);
    const uint32_t ec = block_excl_scan256(c, tid, s_w4, &tc);
    if (tid < gx) { col_start[tid] = ev; chunk_first[tid] = ec; }
    if (tid == 0) { col_start[gx] = tv; chunk_first[gx] = tc; }
}
// pass 1 scan: block x turns row x of the column table into exclusive offsets and stores the column total; the block that
// arrives last (a counter, reset by that block for the next call) builds the chunk table — one launch instead of two. No
// block waits for another.
__global__ __launch_bounds__(1024) void columns_scan_kernel(uint32_t* __restrict__ table, int stride, int n,
                                                            uint32_t* __restrict__ tot, int gx, uint32_t* __restrict__ arrivals,
                                                            uint32_t* __restrict__ chunk_first,
                                                            uint32_t* __restrict__ col_start) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry, s_last;
    __shared__ uint32_t s_w4[4];
    uint32_t* row = table + (size_t)blockIdx.x * stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const uint32_t v = i < n ? row[i] : 0u;
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
        const uint32_t carry = s_carry;
        if (i < n) row[i] = carry + wave_off + inc - v;
        __syncthreads();
        if (tid == 1023) s_carry = carry + wave_off + inc;
        __syncthreads();
    }
    if (tid == 0) {
        __hip_atomic_store(tot + blockIdx.x, s_carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // the total has left before the arrival
        const uint32_t prev = __hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = prev == (uint32_t)gridDim.x - 1u;
        if (s_last) __hip_atomic_store(arrivals, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    build_chunk_table(tot, gx, tid, chunk_first, col_start, s_w4);
}
// the block's chunk: false when blockIdx.x is not a chunk. Loads the 2 x 257-entry table into LDS (one round trip).
struct Chunk { int x; uint32_t seg0, seg1; bool first; };
__device__ __forceinline__ bool load_chunk(const uint32_t* __restrict__ chunk_first, const uint32_t* __restrict__ col_start,
                                           int gx, int tid, uint32_t* s_first, uint32_t* s_col, Chunk& ck) {
    if (tid <= gx) { s_first[tid] = chunk_first[tid]; s_col[tid] = col_start[tid]; }
    __syncthreads();
    const uint32_t c = blockIdx.x;
    if (c >= s_first[gx]) return false;
    int lo = 0, hi = gx - 1;                       // largest x with s_first[x] <= c
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (s_first[mid] <= c) lo = mid; else hi = mid - 1;
    }
    ck.x = lo;
    ck.first = c == s_first[lo];
    ck.seg0 = s_col[lo] + (c - s_first[lo]) * (uint32_t)kExChunk;
    ck.seg1 = min(ck.seg0 + (uint32_t)kExChunk, s_col[lo + 1]);
    return true;
}

// pass 2 count: pairs per (tile row, chunk), by difference array
__global__ __launch_bounds__(kExThreads) void row_count_kernel(int gx, int gy, const uint32_t* __restrict__ chunk_first,
                                                               const uint32_t* __restrict__ col_start,
                                                               const uint16_t* __restrict__ seg_yh,
                                                               uint32_t* __restrict__ row_table, int stride) {
    __shared__ uint32_t s_first[257], s_col[257], s_w4[4];
    __shared__ uint32_t s_d[257];
    const int tid = threadIdx.x;
    if (tid <= 256) s_d[tid] = 0;
    Chunk ck;
    if (!load_chunk(chunk_first, col_start, gx, tid, s_first, s_col, ck)) return;
#pragma unroll
    for (int k = 0; k < kExItems; ++k) {
        const uint32_t i = ck.seg0 + (uint32_t)(k * kExThreads + tid);
        if (i < ck.seg1) {
            const uint32_t yh = seg_yh[i];
            const uint32_t y0 = yh & 255u, h = (yh >> 8) + 1u;
            atomicAdd(&s_d[y0], 1u);
            atomicSub(&s_d[y0 + h], 1u);
        }
    }
    __syncthreads();
    const uint32_t v = tid < 256 ? s_d[tid] : 0u;
    const uint32_t ex = block_excl_scan256(v, tid, s_w4, nullptr);
    if (tid < gy) row_table[(size_t)tid * stride + blockIdx.x] = ex + v;
}

// pass 2 row scan: exclusive offsets of the chunks inside tile row blockIdx.x, the row total, and where every tile column
// starts inside the row (col_rel[y][x], col_rel[y][gx] = row total): the tile ranges.
__global__ __launch_bounds__(1024) void row_scan_kernel(int gx, const uint32_t* __restrict__ chunk_first,
                                                        uint32_t* __restrict__ row_table, int stride,
                                                        uint32_t* __restrict__ row_tot, uint32_t* __restrict__ col_rel) {
    __shared__ uint32_t s_first[257];
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid <= gx) s_first[tid] = chunk_first[tid];
    if (tid == 0) s_carry = 0;
    __syncthreads();
    const int n = (int)s_first[gx];
    uint32_t* row = row_table + (size_t)blockIdx.x * stride;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const uint32_t v = i < n ? row[i] : 0u;
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
        const uint32_t carry = s_carry;
        if (i < n) row[i] = carry + wave_off + inc - v;
        __syncthreads();
        if (tid == 1023) s_carry = carry + wave_off + inc;
        __syncthreads();
    }
    // the offsets just written by this block are read back by it, behind the barrier (which drains the stores)
    __syncthreads();
    uint32_t* cr = col_rel + (size_t)blockIdx.x * (gx + 1);
    if (tid < gx) cr[tid] = __hip_atomic_load(row + s_first[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // L2, not L1
    if (tid == 0) { cr[gx] = s_carry; row_tot[blockIdx.x] = s_carry; }
}

// =================================================================================================== expanding partition
// One block = kExChunk items in 32 wave rounds of 64 (wave w loads rounds 4w .. 4w + 3). An item covers the bins
// [a, a + n); its entries (item, bin) leave the block grouped by bin, inside a bin in item order.
//   A  every item ORs its lane bit into mask[round][bin] for each bin it covers (LDS atomics, nothing waits for them).
//   B  thread (part q, bin b) — 512 threads = bins in use x parts of the rounds — counts the set bits of ITS masks; the bins'
//      totals are scanned into the image layout, the parts of a bin follow each other.
//   C  the same thread walks the set bits of its masks in round order = the bin's entries in item order, and writes
//      (item, bin) words into the LDS image at consecutive places: no LDS read, no wait inside the loop.
//   D  the image leaves in bin runs, consecutive lanes on consecutive entries; the item's payload is fetched from LDS here.
// The image holds kCap entries; B to D repeat over groups of consecutive rounds that fit (one group for ordinary scenes; a
// round never exceeds 64 * NB = kCap entries). Skew: a bin that holds most of the block's entries is walked by few threads
// (all Gaussians in one tile column) — slower, never wrong.
struct ExpandArgs {
    int n_items;                        // pass 1: P
    int nbins;                          // pass 1: gx, pass 2: gy
    int gx;
    // pass 1
    const uint32_t* order;              // Gaussian of depth-ordered position i
    const uint2* rect_sorted;           // its tile rectangle
    // pass 2
    const uint32_t* chunk_first;        // chunk table
    const uint32_t* col_start;
    const uint32_t* seg_idx_in;
    const uint16_t* seg_yh_in;
    const uint32_t* col_rel;            // [gy][gx + 1]
    uint32_t* ranges;                   // [tiles][2]
    // both
    const uint32_t* table;              // [bins][stride] exclusive offsets of this block inside each bin
    const uint32_t* tot;                // [bins] totals
    int stride;
    uint32_t* out_idx;                  // pass 1: segment's Gaussian; pass 2: the sorted point list
    uint16_t* out_aux;                  // pass 1: first row | rows - 1 << 8; pass 2: tile id of the pair
    unsigned long long* stamps;         // diagnostics (mvi_raster_dev_stamps): [blocks][8] shader-clock stamps, else null
};
static unsigned long long* g_dev_stamps[2] = {nullptr, nullptr};     // per pass
void set_dev_stamps(int pass, void* buf) { if (pass == 1 || pass == 2) g_dev_stamps[pass - 1] = (unsigned long long*)buf; }
#define MVI_STAMP(i) do { if (A.stamps && tid == 0) A.stamps[(size_t)blk * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)

// two exclusive scans over the 512 threads of a block at once (one barrier): a -> (exclusive, total), b -> exclusive
__device__ __forceinline__ void block_excl_scan512x2(uint32_t a, uint32_t b, int tid, uint32_t (*s_w)[8], uint32_t& ea,
                                                     uint32_t& ta, uint32_t& eb) {
    const int lane = tid & 63, wave = tid >> 6;
    uint32_t ia = a, ib = b;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t x = __shfl_up(ia, o), y = __shfl_up(ib, o);
        if (lane >= o) { ia += x; ib += y; }
    }
    if (lane == 63) { s_w[0][wave] = ia; s_w[1][wave] = ib; }
    __syncthreads();
    uint32_t oa = 0, ob = 0, t = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const uint32_t x = s_w[0][w], y = s_w[1][w];
        if (w < wave) { oa += x; ob += y; }
        t += x;
    }
    ea = oa + ia - a; eb = ob + ib - b; ta = t;
}

template <int PASS, int NB>
__global__ __launch_bounds__(kExThreads) void expand_scatter_kernel(ExpandArgs A) {
    constexpr int kCap = 64 * NB;                    // 8192 / 16384 entries
    constexpr int kRounds = kExChunk / 64;           // 32
    constexpr int kStride = NB + 1;                  // mask row stride (uint64): rows of one bin fall into different banks
    static_assert(kExThreads / 64 * kExItems == kRounds, "wave w loads rounds kExItems * w ..");
    static_assert(kExThreads == 512, "block_excl_scan512x2");
    __shared__ unsigned long long s_mask[kRounds * kStride + 1];
    __shared__ uint32_t s_pay[kExChunk];
    __shared__ uint16_t s_aux[PASS == 1 ? kExChunk : 1];
    __shared__ uint32_t s_img[kCap];                 // item | bin << 16
    __shared__ uint32_t s_lstart[NB + 1], s_delta[NB];
    __shared__ uint32_t s_rtot[kRounds];
    __shared__ uint32_t s_w[2][8];
    __shared__ uint32_t s_first[PASS == 2 ? 257 : 1], s_col[PASS == 2 ? 257 : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t blk = blockIdx.x;

    // ---- which items
    uint32_t item0 = 0, item1 = 0;
    Chunk ck{0, 0u, 0u, false};
    if (PASS == 1) {
        item0 = blk * (uint32_t)kExChunk;
        item1 = min(item0 + (uint32_t)kExChunk, (uint32_t)A.n_items);
    } else {
        if (!load_chunk(A.chunk_first, A.col_start, A.gx, tid, s_first, s_col, ck)) return;
        item0 = ck.seg0; item1 = ck.seg1;
    }
    MVI_STAMP(0);
    // requested first: they travel while the masks are cleared
    uint32_t ia[kExItems], in_[kExItems];
#pragma unroll
    for (int k = 0; k < kExItems; ++k) {
        const int li = (wave * kExItems + k) * 64 + lane;          // item of round wave * kExItems + k, lane
        const uint32_t i = item0 + (uint32_t)li;
        ia[k] = 0; in_[k] = 0;
        uint32_t pay = 0, aux = 0;
        if (i < item1) {
            if (PASS == 1) {
                const uint2 rc = A.rect_sorted[i];
                const uint32_t w = rc.y & 0xFFFFu, h = rc.y >> 16;
                if (w * h) {
                    ia[k] = rc.x & 0xFFFFu; in_[k] = w;
                    pay = A.order[i];
                    aux = (rc.x >> 16) | ((h - 1u) << 8);
                }
            } else {
                const uint32_t yh = A.seg_yh_in[i];
                ia[k] = yh & 255u; in_[k] = (yh >> 8) + 1u;
                pay = A.seg_idx_in[i];
            }
        }
        s_pay[li] = pay;
        if (PASS == 1) s_aux[li] = (uint16_t)aux;
    }
    const int nb = A.nbins;
    // worker thread = (bin b, part q of the rounds), BIN-MAJOR: as many parts as 512 threads allow for the bins in use (68 tile
    // rows -> 7 parts of 4 - 5 rounds, 120 tile columns -> 4 parts of 8). An exclusive scan of the workers' entry counts in
    // thread order is then the image layout itself (bin runs, inside a bin the parts in round order): one barrier.
    const int parts = min(kExThreads / nb, 16);
    const bool worker = tid < parts * nb;
    const int b = worker ? tid / parts : 0, q = worker ? tid % parts : 0;
    const int r_lo = worker ? q * kRounds / parts : 0, r_hi = worker ? (q + 1) * kRounds / parts : 0;
    // total of the bin and this block's offset inside it, held by the bin's first worker: scanned below together with the
    // entry counts (the scan of the totals over the threads in order = over the bins in order)
    const bool bin_head = worker && q == 0;
    const uint32_t bin_tot = bin_head ? A.tot[b] : 0u;
    const uint32_t bin_off = bin_head ? A.table[(size_t)b * A.stride + blk] : 0u;
    {
        uint4* m4 = reinterpret_cast<uint4*>(&s_mask[0]);
        for (int i = tid; i < (kRounds * kStride + 1) / 2; i += kExThreads) m4[i] = make_uint4(0u, 0u, 0u, 0u);
        if (tid < kRounds) s_rtot[tid] = 0;
    }
    __syncthreads();
    MVI_STAMP(1);

    // ---- A: lane bits into the masks of the covered bins
    const unsigned long long bit = 1ull << lane;
#pragma unroll
    for (int k = 0; k < kExItems; ++k) {
        unsigned long long* row = s_mask + (wave * kExItems + k) * kStride;
        for (uint32_t i = 0; i < in_[k]; ++i) atomicOr(&row[ia[k] + i], bit);
    }
    __syncthreads();
    MVI_STAMP(2);

    const unsigned long long* mcol = s_mask + b;
    uint32_t my_base = 0, my_goff = 0;                         // bin heads: global start of the block's run in the bin, entries
                                                               // of the bin placed by earlier groups

    int g0 = 0, g1 = kRounds;
    bool first = true, have_rtot = false;
    for (;;) {
        uint32_t cnt = 0;
        for (int r = max(r_lo, g0); r < min(r_hi, g1); ++r) cnt += (uint32_t)__popcll(mcol[r * kStride]);
        uint32_t pos0, total, bin_start;
        block_excl_scan512x2(cnt, first ? bin_tot : 0u, tid, s_w, pos0, total, bin_start);
        if (first) {
            first = false;
            my_base = bin_start + bin_off;
            if (PASS == 2 && ck.first && bin_head) {
                // tile (row b, column ck.x): its pairs start where this chunk's run of row b starts
                const uint32_t* cr = A.col_rel + (size_t)b * (A.gx + 1);
                const uint32_t c = cr[ck.x + 1] - cr[ck.x];
                uint2* rg = reinterpret_cast<uint2*>(A.ranges) + ((size_t)b * A.gx + ck.x);
                *rg = c ? make_uint2(my_base, my_base + c) : make_uint2(0u, 0u);
            }
            if (total == 0) return;                            // nothing to place (culled tail of the depth order, empty column)
        }
        MVI_STAMP(3);
        if (total > (uint32_t)kCap) {
            // rare: the rounds [g0, g1) do not fit the image -> entries per round, then the longest prefix that fits
            if (!have_rtot) {
                have_rtot = true;
                for (int r = r_lo; r < r_hi; ++r) {
                    const uint32_t c = (uint32_t)__popcll(mcol[r * kStride]);
                    if (c) atomicAdd(&s_rtot[r], c);
                }
            }
            __syncthreads();
            uint32_t acc = 0;
            g1 = g0;
            while (g1 < kRounds && acc + s_rtot[g1] <= (uint32_t)kCap) { acc += s_rtot[g1]; ++g1; }   // >= 1 round: <= 64 * NB each
            continue;
        }
        if (bin_head) {
            s_lstart[b] = pos0;                                // start of bin b's run in the image
            s_delta[b] = my_base + my_goff - pos0;             // + place in the image = place in the output
        }
        if (tid == 0) s_lstart[nb] = total;
        MVI_STAMP(4);
        // C: walk the set bits (32 at a time: one find-first-bit, one clear per entry)
        if (cnt) {
            uint32_t pos = pos0;
            const uint32_t hi = (uint32_t)b << 16;
            for (int r = max(r_lo, g0); r < min(r_hi, g1); ++r) {
                const unsigned long long m = mcol[r * kStride];
                uint32_t m0 = (uint32_t)m, m1 = (uint32_t)(m >> 32);
                const uint32_t item_hi = hi | (uint32_t)(r * 64);
                while (m0) {
                    const uint32_t l = (uint32_t)__builtin_ctz(m0);
                    m0 &= m0 - 1u;
                    s_img[pos++] = item_hi | l;
                }
                while (m1) {
                    const uint32_t l = (uint32_t)__builtin_ctz(m1);
                    m1 &= m1 - 1u;
                    s_img[pos++] = item_hi | 32u | l;
                }
            }
        }
        __syncthreads();
        MVI_STAMP(5);
        // D: the image leaves in bin runs; place in the output = bin's global base + entries of earlier groups + place in the run
        for (uint32_t p = tid; p < total; p += kExThreads) {
            const uint32_t e = s_img[p];
            const uint32_t eb = e >> 16, item = e & 0xFFFFu;
            const uint32_t dst = p + s_delta[eb];
            A.out_idx[dst] = s_pay[item];
            if (PASS == 1) A.out_aux[dst] = s_aux[item];
            else A.out_aux[dst] = (uint16_t)(eb * (uint32_t)A.gx + (uint32_t)ck.x);
        }
        MVI_STAMP(6);
        if (g1 >= kRounds) break;
        if (bin_head) my_goff += s_lstart[b + 1] - pos0;       // written before the barrier in front of D
        __syncthreads();                                       // D has read the image and the bin starts
        g0 = g1; g1 = kRounds;
    }
}

// =================================================================================================== launchers
// Depth sort + column count + column scan + chunk table: needs nothing that depends on num_rendered, so the forward enqueues
// it BEFORE the host reads num_rendered back.
int launch_binning2_level1(const Frame& f, GeomView g, hipStream_t st) {
    if (f.P <= 0) return 0;
    {
        StageTimer tm(kStSort, st);
        // keys (depth bits, 0xFFFFFFFF for culled) come from the preprocess kernel; 0 -> 1 -> 0 -> 1 -> 0.
        // super rows: region p & 1 for pass p; region 0 was zeroed by the totals kernel, scatter p zeroes region (p + 1) & 1
#define MVI_DS_PASS(PASS, SRC, DST)                                                                                              \
        hipLaunchKernelGGL(depth_count_kernel, dim3(g.nds), dim3(kDsThreads), 0, st, g.dkeys[SRC], f.P, 8 * PASS, g.ds_table,     \
                           g.ds_super[PASS & 1]);                                                                                 \
        hipLaunchKernelGGL((depth_scatter_kernel<PASS>), dim3(g.nds), dim3(kDsThreads), 0, st, g.dkeys[SRC], g.dvals[SRC],        \
                           g.dkeys[DST], g.dvals[DST], f.P, g.ds_table, g.ds_super[PASS & 1], g.nsuper,                           \
                           PASS < 3 ? g.ds_super[(PASS + 1) & 1] : (uint32_t*)nullptr, g.rect, g.rect_sorted);
        MVI_DS_PASS(0, 0, 1) MVI_DS_PASS(1, 1, 0) MVI_DS_PASS(2, 0, 1) MVI_DS_PASS(3, 1, 0)
#undef MVI_DS_PASS
    }
    StageTimer tm(kStDup, st);
    hipLaunchKernelGGL(column_count_kernel, dim3(g.nblk1), dim3(kExThreads), 0, st, f.P, f.gx, g.rect_sorted, g.col_table,
                       g.nblk1);
    hipLaunchKernelGGL(columns_scan_kernel, dim3(f.gx), dim3(1024), 0, st, g.col_table, g.nblk1, g.nblk1, g.col_tot, f.gx,
                       g.arrivals, g.chunk_first, g.col_start);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

// pass 1, pass 2 (count, scan, scatter + tile ranges). Result: b.vals[1] = point list, b.keys[1] = tile ids (uint16).
int launch_binning2(const Frame& f, GeomView g, BinningView b, ImageView im, int64_t D, int64_t segments, hipStream_t st) {
    const size_t tiles = (size_t)f.gx * f.gy;
    if (D <= 0 || f.P <= 0) return launch_zero_fill(im.ranges, 8 * tiles, st);
    const bool wide = f.gx > 128 || f.gy > 128;
    ExpandArgs a1{};
    a1.n_items = f.P; a1.nbins = f.gx; a1.gx = f.gx;
    a1.order = g.dvals[0]; a1.rect_sorted = g.rect_sorted;
    a1.table = g.col_table; a1.tot = g.col_tot; a1.stride = g.nblk1;
    a1.out_idx = b.vals[0]; a1.out_aux = (uint16_t*)b.keys[0];
    a1.stamps = g_dev_stamps[0];
    {
        StageTimer tm(kStDup, st);
        if (wide) hipLaunchKernelGGL((expand_scatter_kernel<1, 256>), dim3(g.nblk1), dim3(kExThreads), 0, st, a1);
        else hipLaunchKernelGGL((expand_scatter_kernel<1, 128>), dim3(g.nblk1), dim3(kExThreads), 0, st, a1);
    }
    // chunks: at most segments / kExChunk + gx; the table (stride b.nsort) was sized for segments <= D
    int nchunk = b.nsort;
    if (segments > 0 && segments <= D) nchunk = (int)min((int64_t)b.nsort, segments / kExChunk + f.gx + 1);
    ExpandArgs a2{};
    a2.nbins = f.gy; a2.gx = f.gx;
    a2.chunk_first = g.chunk_first; a2.col_start = g.col_start;
    a2.seg_idx_in = b.vals[0]; a2.seg_yh_in = (const uint16_t*)b.keys[0];
    a2.col_rel = b.col_rel; a2.ranges = im.ranges;
    a2.table = b.block_hist; a2.tot = b.digit_tot; a2.stride = b.nsort;
    a2.out_idx = b.vals[1]; a2.out_aux = (uint16_t*)b.keys[1];
    a2.stamps = g_dev_stamps[1];
    StageTimer tm(kStSort, st);
    hipLaunchKernelGGL(row_count_kernel, dim3(nchunk), dim3(kExThreads), 0, st, f.gx, f.gy, g.chunk_first, g.col_start,
                       (const uint16_t*)b.keys[0], b.block_hist, b.nsort);
    hipLaunchKernelGGL(row_scan_kernel, dim3(f.gy), dim3(1024), 0, st, f.gx, g.chunk_first, b.block_hist, b.nsort,
                       b.digit_tot, b.col_rel);
    if (wide) hipLaunchKernelGGL((expand_scatter_kernel<2, 256>), dim3(nchunk), dim3(kExThreads), 0, st, a2);
    else hipLaunchKernelGGL((expand_scatter_kernel<2, 128>), dim3(nchunk), dim3(kExThreads), 0, st, a2);
    return hipGetLastError() == hipSuccess ? 0 : MVI_EHIP;
}

}  // namespace mvi
