// NOT BUILT INTO libmvi_hip.so. Round-2 experiment, kept for the record: the four 8-bit passes of the depth sort
// of the Gaussians as ONE persistent launch with grid barriers, instead of 12 launches. Correct (sortedness / stability /
// permutation tests at 262144 ... 2.1 M Gaussians), but not faster on MI355X: 110 us against 117 us at 1.5 M pairs. In-kernel
// stamps per pass: load + rank 5.6 us, barrier 6 - 7 us, output bases from the G x 256 table + LDS placement 6.5 us, write-out
// 1.2 us, barrier 7 - 8 us. A device-wide barrier through memory-side atomics (8 XCDs, L2s not coherent with each other) costs
// what a kernel boundary costs, and with one 1024-thread block per CU the phases of a block no longer overlap with another
// block's on the same CU. With agent-scope fences (L2 write-back + invalidate) in the barrier: 9 - 16 us per barrier, 205 us.
// It slots into raster_binning.hip (uses kSortItems and the file's includes); launch: grid G <= min(CUs, P / 2048) blocks of
// 1024 threads, chunk = P / G rounded up to 1024 and <= 8192, sync[0..1] zeroed beforehand, hist = G * 256 words.

// ---- depth sort of the Gaussians as ONE launch -------------------------------------------------------------------------------
// The four 8-bit passes over the P (depth bits, index) pairs were 12 launches of 5 - 15 us each for 12 MB of data that never
// leaves the Infinity Cache: launch-bound. Here one grid of G <= 256 co-resident 1024-thread blocks (one per CU) runs all four
// passes; block b owns the b-th contiguous chunk (<= 8192 pairs, 8 per thread, held in registers) in every pass. Per pass:
// (A) load the chunk, rank every pair among the equal digits of its wave (the ballot scheme of radix_scatter_kernel), digit
// totals of the block -> hist[b][256]; grid barrier; (B) the block's 256 output bases from the G x 256 table (digit totals
// before its digit + the same digit in earlier blocks: the stable order); (C) pairs to their place in the block-sorted LDS
// tile, then out in runs; grid barrier.
// The XCDs' L2s are not coherent with each other, and an agent-scope fence pair (L2 write-back + invalidate) per barrier
// measured 9 - 16 us: everything blocks exchange inside the kernel (pairs, histogram, barrier counter) moves with agent-scope
// relaxed atomic loads / stores instead (sc1: served at the memory side, 12 MB in the 256 MB Infinity Cache), each wave
// drains its stores (vmcnt(0)) before the block arrives at the barrier, and the barrier itself is one atomic add on a
// monotonic counter + a spin on it. The grid is a plain launch: G <= CU count and one block fits every CU, so all blocks are
// resident unless another stream holds CUs, in which case the late blocks start when that work drains; the spin is bounded all
// the same (kSpinLimit: the sort is then wrong and sync[1] says so, but every wave exits). sync[0..1] are zeroed by
// total_block_sums_kernel, which runs before.
constexpr int kPsThreads = 1024, kPsWaves = 16, kPsTile = 512 * kPsWaves;
constexpr int kPsOneLaunchMin = 1 << 18;     // below, the grid would be a few blocks: the 12-launch form keeps the short sorts
constexpr uint32_t kSpinLimit = 1u << 21;

__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void grid_barrier(uint32_t* sync, uint32_t target) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's agent-scope stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t spins = 0;
        while (ld_agent(sync) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kSpinLimit) { st_agent(sync + 1, 1u); break; }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(kPsThreads) void depth_sort_persistent_kernel(uint32_t* k0, uint32_t* v0, uint32_t* k1, uint32_t* v1,
                                                                          int P, int chunk, uint32_t* hist, uint32_t* sync) {
    __shared__ uint32_t s_part[4][2][256];           // phase B: [quarter of the blocks][below me | all][digit]
    __shared__ uint32_t s_wave_hist[kPsWaves][256];
    __shared__ uint32_t s_digit_base[256];           // global position of this block's first pair of digit d
    __shared__ uint32_t s_local_start[256];
    __shared__ uint32_t s_w4[4];
    __shared__ uint32_t s_key[kPsTile];
    __shared__ uint32_t s_val[kPsTile];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x, G = gridDim.x;
    const bool digit_thread = tid < 256;
    const int begin = min(P, b * chunk), end = min(P, begin + chunk), count = end - begin;
    // wave w owns per_wave consecutive pairs of the chunk (a multiple of 64), item `it` of lane l is pair w*per_wave + it*64 + l
    const int per_wave = chunk / kPsWaves;           // the host makes chunk a multiple of 64 * kPsWaves
    const int nit = per_wave >> 6;                   // <= kSortItems
    const uint64_t lanemask_lt = (1ull << lane) - 1ull;
    uint32_t barriers = 0;
#ifdef MVI_SORT_STAMPS
    uint64_t tA = 0, tB1 = 0, tB = 0, tC = 0, tB2 = 0, ts = __builtin_amdgcn_s_memrealtime(), tn;
#define MVI_STAMP(acc) do { tn = __builtin_amdgcn_s_memrealtime(); acc += tn - ts; ts = tn; } while (0)
#else
#define MVI_STAMP(acc) do {} while (0)
#endif
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 8 * pass;
        const uint32_t* kin = (pass & 1) ? k1 : k0;
        const uint32_t* vin = (pass & 1) ? v1 : v0;
        uint32_t* kout = (pass & 1) ? k0 : k1;
        uint32_t* vout = (pass & 1) ? v0 : v1;
        // (A) load, rank inside the wave, digit totals of the block
#pragma unroll
        for (int i = tid; i < kPsWaves * 256; i += kPsThreads) (&s_wave_hist[0][0])[i] = 0;
        uint32_t key[kSortItems], val[kSortItems], rank[kSortItems], dig[kSortItems];
        const int base = begin + wave * per_wave + lane;
#pragma unroll
        for (int it = 0; it < kSortItems; ++it) {
            const int idx = base + it * 64;
            const bool valid = it < nit && idx < end;
            key[it] = valid ? ld_agent(kin + idx) : 0u;
            val[it] = valid ? ld_agent(vin + idx) : 0u;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < kSortItems; ++it) {
            if (it < nit) {                                  // block-uniform
                const bool valid = base + it * 64 < end;
                const uint32_t d = (key[it] >> shift) & 255u;
                dig[it] = valid ? d : 0xFFFFFFFFu;
                uint64_t peers = __ballot(valid);
#pragma unroll
                for (int bit_i = 0; bit_i < 8; ++bit_i) {
                    const uint64_t bit = __ballot((d >> bit_i) & 1u);
                    peers &= ((d >> bit_i) & 1u) ? bit : ~bit;
                }
                const uint32_t before = (uint32_t)__popcll(peers & lanemask_lt);
                uint32_t prev = 0;
                if (valid) {
                    prev = s_wave_hist[wave][d];
                    if ((peers >> lane) == 1ull) s_wave_hist[wave][d] = prev + (uint32_t)__popcll(peers);
                }
                rank[it] = prev + before;
            } else {
                dig[it] = 0xFFFFFFFFu;
                rank[it] = 0;
            }
        }
        __syncthreads();
        uint32_t tot = 0, inc = 0;
        if (digit_thread) {
            uint32_t run = 0;
#pragma unroll
            for (int w = 0; w < kPsWaves; ++w) {
                const uint32_t c = s_wave_hist[w][tid];
                s_wave_hist[w][tid] = run;
                run += c;
            }
            tot = run;
            st_agent(hist + (size_t)b * 256 + tid, tot);
            inc = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                uint32_t t = __shfl_up(inc, o);
                if (lane >= o) inc += t;
            }
            if (lane == 63) s_w4[wave] = inc;
        }
        MVI_STAMP(tA);
        grid_barrier(sync, ++barriers * (uint32_t)G);        // (its block barriers also publish s_w4 / s_wave_hist)
        MVI_STAMP(tB1);
        // (B) output bases: thread (d, q) adds up digit d over the blocks bb = q mod 4, 8 loads in flight
        {
            const int d = tid & 255, q = tid >> 8;
            uint32_t below = 0, all = 0;
            const uint32_t* col = hist + d;
            int bb = q;
            for (; bb + 28 < G; bb += 32) {
                uint32_t c[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) c[u] = ld_agent(col + (size_t)(bb + 4 * u) * 256);
#pragma unroll
                for (int u = 0; u < 8; ++u) { all += c[u]; below += (bb + 4 * u) < b ? c[u] : 0u; }
            }
            for (; bb < G; bb += 4) {
                const uint32_t c = ld_agent(col + (size_t)bb * 256);
                all += c;
                below += bb < b ? c : 0u;
            }
            s_part[q][0][d] = below;
            s_part[q][1][d] = all;
        }
        if (digit_thread) {
            uint32_t wave_off = 0;
            for (int w = 0; w < wave; ++w) wave_off += s_w4[w];
            s_local_start[tid] = wave_off + inc - tot;
        }
        __syncthreads();
        uint32_t dv = 0, dinc = 0, dbelow = 0;
        if (digit_thread) {
            dbelow = s_part[0][0][tid] + s_part[1][0][tid] + s_part[2][0][tid] + s_part[3][0][tid];
            dv = s_part[0][1][tid] + s_part[1][1][tid] + s_part[2][1][tid] + s_part[3][1][tid];
            dinc = dv;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                uint32_t t = __shfl_up(dinc, o);
                if (lane >= o) dinc += t;
            }
            if (lane == 63) s_w4[wave] = dinc;               // last read of s_w4 (local starts) was before the barrier above
        }
        // (C) pairs to their place in the block-sorted tile
#pragma unroll
        for (int it = 0; it < kSortItems; ++it) {
            if (dig[it] != 0xFFFFFFFFu) {
                const uint32_t d = dig[it];
                const uint32_t lp = s_local_start[d] + s_wave_hist[wave][d] + rank[it];
                s_key[lp] = key[it];
                s_val[lp] = val[it];
            }
        }
        __syncthreads();
        if (digit_thread) {
            uint32_t wave_off = 0;
            for (int w = 0; w < wave; ++w) wave_off += s_w4[w];
            s_digit_base[tid] = wave_off + dinc - dv + dbelow;
        }
        __syncthreads();
        MVI_STAMP(tB);
#pragma unroll
        for (int it = 0; it < kSortItems; ++it) {
            const int lp = it * kPsThreads + tid;
            if (lp < count) {
                const uint32_t k = s_key[lp];
                const uint32_t d = (k >> shift) & 255u;
                const uint32_t dst = s_digit_base[d] + ((uint32_t)lp - s_local_start[d]);
                st_agent(kout + dst, k);
                st_agent(vout + dst, s_val[lp]);
            }
        }
        MVI_STAMP(tC);
        if (pass < 3) grid_barrier(sync, ++barriers * (uint32_t)G);
        MVI_STAMP(tB2);
    }
#ifdef MVI_SORT_STAMPS
    if (tid == 0 && (b == 0 || b == G - 1 || b == G / 2))
        printf("block %d of %d: load+rank %.1f us, barrier %.1f, bases+place %.1f, write %.1f, barrier %.1f (sums over 4 passes)\n", b, G,
               tA * 0.01, tB1 * 0.01, tB * 0.01, tC * 0.01, tB2 * 0.01);
#endif
#undef MVI_STAMP
}

