// Where and when do the workgroups of one launch run? (MI355X, 8 XCDs x 32 CUs.) Every block records its XCC_ID / HW_ID and its
// start and end time (s_memrealtime, 10 ns ticks) around a ~20 us busy wait; the host reports how ids map to XCDs, how many
// blocks an XCD / a CU holds at once, and whether an XCD starts its blocks in id order. Registers are padded to ~120 per lane
// so that two 512-thread blocks fit a CU, like gn_cluster_kernel<T, 12>.
// Build: hipcc -O2 --offload-arch=gfx950 tools/experiments/xcd_probe.cpp -o tools/experiments/xcd_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void probe(uint64_t* rec, float* sink, int spin_ticks) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    float acc[96];
#pragma unroll
    for (int i = 0; i < 96; ++i) acc[i] = (float)(threadIdx.x + i);
    while (__builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)spin_ticks) {
#pragma unroll
        for (int i = 0; i < 96; ++i) acc[i] = acc[i] * 1.0001f + 0.5f;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 96; ++i) s += acc[i];
    if (s == 12345.678f) sink[threadIdx.x] = s;
    if (threadIdx.x == 0) {
        uint64_t* r = rec + 4 * (uint64_t)blockIdx.x;
        r[0] = t0;
        r[1] = __builtin_amdgcn_s_memrealtime();
        r[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_ID
        r[3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // XCC_ID
    }
}

int main() {
    const int N = 3392;
    uint64_t* d; float* sink;
    hipMalloc(&d, N * 4 * sizeof(uint64_t)); hipMalloc(&sink, 4096);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(N), dim3(512), 0, 0, d, sink, 2000);
    hipDeviceSynchronize();
    std::vector<uint64_t> r(N * 4);
    hipMemcpy(r.data(), d, r.size() * 8, hipMemcpyDeviceToHost);
    int same = 0; std::map<int, std::vector<int>> by_xcd; std::map<uint32_t, std::vector<int>> by_cu;
    for (int i = 0; i < N; ++i) {
        const int xcc = (int)(r[4 * i + 3] & 15);
        same += xcc == (i & 7);
        by_xcd[xcc].push_back(i);
        by_cu[((uint32_t)xcc << 16) | ((uint32_t)r[4 * i + 2] & 0xff00u)].push_back(i);
    }
    printf("%d blocks of 512 threads: XCC_ID == id %% 8 for %d of them; %zu XCDs, %zu CUs seen\n", N, same, by_xcd.size(), by_cu.size());
    for (auto& kv : by_xcd) {
        auto& v = kv.second;                                  // ids in ascending order already
        int inversions = 0, max_live = 0;
        for (size_t a = 1; a < v.size(); ++a) inversions += r[4 * v[a]] + 50 < r[4 * v[a - 1]];       // started > 0.5 us before its predecessor
        std::vector<std::pair<uint64_t, int>> ev;
        for (int i : v) { ev.push_back({r[4 * i], 1}); ev.push_back({r[4 * i + 1], -1}); }
        std::sort(ev.begin(), ev.end());
        int live = 0;
        for (auto& e : ev) { live += e.second; max_live = std::max(max_live, live); }
        // how far ahead of the oldest unfinished id does the XCD start blocks? (window in ids of its own sequence)
        size_t window = 0;
        for (size_t a = 0; a < v.size(); ++a) {
            size_t bnd = a;
            while (bnd + 1 < v.size() && r[4 * v[bnd + 1]] < r[4 * v[a] + 1]) ++bnd;     // started before block a ended
            window = std::max(window, bnd - a + 1);
        }
        printf("  XCD %d: %zu blocks, at most %d live at once, %d out-of-order starts, widest run of ids live together %zu\n", kv.first,
               v.size(), max_live, inversions, window);
    }
    size_t mx = 0;
    for (auto& kv : by_cu) {
        std::vector<std::pair<uint64_t, int>> ev;
        for (int i : kv.second) { ev.push_back({r[4 * i], 1}); ev.push_back({r[4 * i + 1], -1}); }
        std::sort(ev.begin(), ev.end());
        int live = 0, m = 0;
        for (auto& e : ev) { live += e.second; m = std::max(m, live); }
        mx = std::max(mx, (size_t)m);
    }
    printf("  at most %zu blocks live on one CU\n", mx);
    // first 64 blocks of XCD 0's sequence: start offsets
    auto& v0 = by_xcd.begin()->second;
    printf("  XCD %d, start time of its first 80 ids relative to the first (us):", by_xcd.begin()->first);
    for (size_t a = 0; a < 80 && a < v0.size(); ++a) printf(" %.1f", (double)(r[4 * v0[a]] - r[4 * v0[0]]) * 0.01);
    printf("\n");
    return 0;
}
