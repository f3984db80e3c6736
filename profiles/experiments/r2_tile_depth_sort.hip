// Round-2 experiment (negative, see README.md), standalone: drop the global depth sort of the N Gaussians (8 launches, ~100 us alone / ~146 us in the step at
// 1 M) and order every tile's list by (depth key, id) in ONE launch after the tile sort instead.  Instances are then emitted
// in Gaussian-index order (duplicate_kernel with the identity order), the stable tile radix sort groups them by tile, and this
// kernel sorts each group in LDS: (key << 32 | id) composites, bitonic network for up to TS_CAP entries, longer lists as sorted
// chunks merged through global scratch (merge path).  Ties in depth fall back to the id, which is what the stable global sort
// produces, so the final (tile, depth, id) order is identical.
//
// This file is the kernel + a harness on a synthetic tile-length distribution shaped like the metric workload (8160 tiles, 17 %
// non-empty, 3.0 M instances, longest list 12 077) with the edge lengths added; it checks every list against std::sort.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -w r2_tile_depth_sort.hip -o r2_tile_depth_sort
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <random>
#include <vector>

typedef unsigned long long u64;

// ascending bitonic sort of s[0 .. np), np a power of two >= 2; called by all threads of the workgroup
__device__ __forceinline__ void bitonic_lds(u64* s, int np) {
    for (int k = 2; k <= np; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int idx = threadIdx.x; idx < (np >> 1); idx += (int)blockDim.x) {
                const int i = ((idx & ~(j - 1)) << 1) | (idx & (j - 1));      // element with bit j clear
                const int p = i | j;
                const bool up = (i & k) == 0;
                const u64 a = s[i], b = s[p];
                if ((a > b) == up) { s[i] = b; s[p] = a; }
            }
            __syncthreads();
        }
    }
}

// first index i in A (length la) such that taking i from A and diag - i from B is a valid merge prefix (A wins ties: never
// happens, the composites are unique)
__device__ __forceinline__ uint32_t merge_path(const u64* A, uint32_t la, const u64* B, uint32_t lb, uint32_t diag) {
    uint32_t lo = diag > lb ? diag - lb : 0u, hi = diag < la ? diag : la;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (A[mid] < B[diag - 1 - mid]) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Lists of nmin < n <= nmax entries only (two launches: 256 threads / 4096-entry LDS for the many short lists, 1024 threads /
// 16384-entry LDS for the long ones, which are the critical path); `cap` = entries the workgroup's LDS holds.
template <int TS_THREADS>
__global__ void __launch_bounds__(TS_THREADS) tile_depth_sort_kernel(int T, const uint32_t* __restrict__ order,
                                                                     const uint2* __restrict__ ranges,
                                                                     const uint32_t* __restrict__ depth_key,
                                                                     uint32_t* __restrict__ point_list, u64* scratch_a,
                                                                     u64* scratch_b, uint32_t TS_CAP, uint32_t nmin, uint32_t nmax) {
    extern __shared__ u64 s[];
    if ((int)blockIdx.x >= T) return;
    const int tile = order ? (int)order[blockIdx.x] : (int)blockIdx.x;
    const uint32_t beg = ranges[tile].x, n = ranges[tile].y - beg;
    if (n <= nmin || n > nmax) return;
    const uint32_t nchunks = (n + TS_CAP - 1) / TS_CAP;
    for (uint32_t c = 0; c < nchunks; ++c) {
        const uint32_t cb = beg + c * TS_CAP, cn = min(TS_CAP, n - c * TS_CAP);
        int np = 2;
        while ((uint32_t)np < cn) np <<= 1;
        for (int i = threadIdx.x; i < np; i += TS_THREADS) {
            u64 v = ~0ull;                                                       // padding sorts last
            if ((uint32_t)i < cn) { const uint32_t id = point_list[cb + i]; v = ((u64)depth_key[id] << 32) | id; }
            s[i] = v;
        }
        __syncthreads();
        bitonic_lds(s, np);
        if (nchunks == 1) { for (uint32_t i = threadIdx.x; i < cn; i += TS_THREADS) point_list[cb + i] = (uint32_t)s[i]; }
        else { for (uint32_t i = threadIdx.x; i < cn; i += TS_THREADS) scratch_a[cb + i] = s[i]; }
        __syncthreads();
    }
    if (nchunks == 1) return;
    // sorted runs of `run` entries -> runs of 2 * run, through the two scratch arrays (this tile's slice [beg, beg + n) only)
    u64* src = scratch_a;
    u64* dst = scratch_b;
    for (uint32_t run = TS_CAP; run < n; run <<= 1) {
        __threadfence_block();
        __syncthreads();
        for (uint32_t p0 = 0; p0 < n; p0 += 2 * run) {
            const uint32_t la = min(run, n - p0), lb = p0 + run < n ? min(run, n - p0 - run) : 0u, len = la + lb;
            const u64* A = src + beg + p0;
            const u64* B = A + la;
            u64* O = dst + beg + p0;
            const uint32_t per = (len + TS_THREADS - 1) / TS_THREADS;
            const uint32_t o0 = min(len, threadIdx.x * per), o1 = min(len, o0 + per);
            if (o0 < o1) {
                uint32_t i = merge_path(A, la, B, lb, o0), j = o0 - i;
                for (uint32_t o = o0; o < o1; ++o) {
                    const bool take_a = j >= lb || (i < la && A[i] < B[j]);
                    O[o] = take_a ? A[i] : B[j];
                    i += take_a ? 1u : 0u; j += take_a ? 0u : 1u;
                }
            }
        }
        u64* t = src; src = dst; dst = t;
    }
    __threadfence_block();
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += TS_THREADS) point_list[beg + i] = (uint32_t)src[beg + i];
}

int main() {
    const int T = 8160, N = 1000000;
    std::mt19937 rng(17);
    std::vector<uint32_t> len(T, 0);
    std::lognormal_distribution<float> ln(7.2f, 0.9f);
    std::vector<int> ne;
    for (int t = 0; t < T; ++t) if (rng() % 100 < 17) ne.push_back(t);
    double sum = 0;
    for (int t : ne) { len[t] = (uint32_t)std::min(12077.f, ln(rng)); sum += len[t]; }
    const double scale = 3.0e6 / sum;
    for (int t : ne) len[t] = (uint32_t)std::min(12077.0, std::max(1.0, len[t] * scale));
    const bool with_huge = getenv("TS_HUGE") != nullptr;      // one list beyond the 16384-entry LDS of the long class (chunks + merge)
    const uint32_t edge[] = {1, 2, 3, 63, 64, 65, 255, 256, 257, 4095, 4096, 4097, 8191, 8192, 8193, 12077, 12000, 11000, with_huge ? 40000u : 10000u};
    for (size_t e = 0; e < sizeof(edge) / sizeof(edge[0]); ++e) len[ne[e]] = edge[e];
    std::vector<uint2> ranges(T);
    uint32_t R = 0, mx = 0;
    for (int t = 0; t < T; ++t) { ranges[t].x = R; R += len[t]; ranges[t].y = R; mx = std::max(mx, len[t]); }
    std::vector<uint32_t> key(N), pl(R);
    for (int i = 0; i < N; ++i) { float z = 0.2f + 20.f * (float)(rng() & 0xFFFFFF) / 16777216.f; if (i % 50 == 0) z = 3.25f; memcpy(&key[i], &z, 4); }
    for (int t = 0; t < T; ++t) {                      // ids of a tile: distinct, ascending (what the stable tile sort of an index-ordered emission gives)
        std::vector<uint32_t> ids(len[t]);
        for (auto& v : ids) v = rng() % N;
        std::sort(ids.begin(), ids.end());
        ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
        while (ids.size() < len[t]) { uint32_t v = rng() % N; auto it = std::lower_bound(ids.begin(), ids.end(), v); if (it == ids.end() || *it != v) ids.insert(it, v); }
        std::copy(ids.begin(), ids.end(), pl.begin() + ranges[t].x);
    }
    std::vector<uint32_t> want(pl);
    for (int t = 0; t < T; ++t)
        std::sort(want.begin() + ranges[t].x, want.begin() + ranges[t].y, [&](uint32_t a, uint32_t b) {
            return key[a] != key[b] ? key[a] < key[b] : a < b; });
    std::vector<uint32_t> order(T);
    for (int t = 0; t < T; ++t) order[t] = t;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return len[a] > len[b]; });

    uint2* d_ranges; uint32_t *d_key, *d_pl, *d_pl0, *d_order; u64 *d_a, *d_b;
    hipMalloc(&d_ranges, T * 8); hipMalloc(&d_key, N * 4); hipMalloc(&d_pl, (size_t)R * 4); hipMalloc(&d_pl0, (size_t)R * 4);
    hipMalloc(&d_order, T * 4); hipMalloc(&d_a, (size_t)R * 8); hipMalloc(&d_b, (size_t)R * 8);
    hipMemcpy(d_ranges, ranges.data(), T * 8, hipMemcpyHostToDevice); hipMemcpy(d_key, key.data(), N * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_pl0, pl.data(), (size_t)R * 4, hipMemcpyHostToDevice); hipMemcpy(d_order, order.data(), T * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1, ej; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&ej);
    hipStream_t s_small, s_large; hipStreamCreate(&s_small); hipStreamCreate(&s_large);
    const uint32_t CAP_S = 4096, CAP_L = 16384;
    hipFuncSetAttribute((const void*)tile_depth_sort_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, CAP_L * 8);
    int n_long = 0;
    for (int t = 0; t < T; ++t) n_long += len[t] > CAP_S;
    int bad_total = 0;
    for (int mode = 0; mode < 3; ++mode) {             // 0: one class (256 threads, chunks + merges), 1: two classes one stream, 2: two streams
        float sum_ms = 0.f; const int reps = 20;
        for (int rep = 0; rep < reps + 3; ++rep) {
            hipMemcpyAsync(d_pl, d_pl0, (size_t)R * 4, hipMemcpyDeviceToDevice, s_small);
            hipEventRecord(e0, s_small);
            if (mode == 0) {
                hipLaunchKernelGGL(tile_depth_sort_kernel<256>, dim3(T), dim3(256), CAP_S * 8, s_small, T, d_order, d_ranges, d_key, d_pl,
                                   d_a, d_b, CAP_S, 1u, 0xFFFFFFFFu);
            } else {
                hipStream_t sl = mode == 2 ? s_large : s_small;
                if (mode == 2) hipStreamWaitEvent(s_large, e0, 0);
                // long lists: the first n_long entries of the longest-first order, one 1024-thread workgroup each
                hipLaunchKernelGGL(tile_depth_sort_kernel<1024>, dim3(n_long), dim3(1024), CAP_L * 8, sl, n_long, d_order, d_ranges, d_key,
                                   d_pl, d_a, d_b, CAP_L, CAP_S, 0xFFFFFFFFu);
                if (mode == 2) hipEventRecord(ej, s_large);
                hipLaunchKernelGGL(tile_depth_sort_kernel<256>, dim3(T), dim3(256), CAP_S * 8, s_small, T, d_order, d_ranges, d_key, d_pl,
                                   d_a, d_b, CAP_S, 1u, CAP_S);
                if (mode == 2) hipStreamWaitEvent(s_small, ej, 0);
            }
            hipEventRecord(e1, s_small);
            if (hipEventSynchronize(e1) != hipSuccess) { printf("device error\n"); return 1; }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 3) sum_ms += ms;
        }
        std::vector<uint32_t> got(R);
        hipMemcpy(got.data(), d_pl, (size_t)R * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int t = 0; t < T; ++t)
            if (memcmp(&got[ranges[t].x], &want[ranges[t].x], (size_t)len[t] * 4) != 0) { if (bad < 5) printf("  tile %d (len %u) wrong\n", t, len[t]); ++bad; }
        const char* names[] = {"256 thr, chunks+merge", "two classes, 1 stream", "two classes, 2 streams"};
        printf("%-24s R=%u non-empty=%zu lists>4096=%d longest=%u  wrong tiles=%d  %.1f us\n", names[mode], R, ne.size(), n_long, mx, bad,
               sum_ms / reps * 1e3f);
        bad_total += bad;
    }
    printf(bad_total ? "FAILED\n" : "all tile lists exact\n");
    return bad_total != 0;
}
