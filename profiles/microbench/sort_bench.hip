// Correctness + timing of csrc/radix_sort.hip against rocPRIM radix_sort_pairs and std::stable_sort, at the two sizes
// the rasterizer sorts (1e6 x 32-bit depth keys, 3e6 x 13-bit tile keys).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../vcr_gaus_amd/csrc sort_bench.hip ../../vcr_gaus_amd/csrc/radix_sort.hip -o sort_bench
#include "vcr_common.h"
#include <rocprim/device/device_radix_sort.hpp>
#include <algorithm>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>
#include <cstring>

void vcr_set_error(const char* fmt, ...) { fprintf(stderr, "error: %s\n", fmt); }
using OneSweep = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;

static void run(const char* name, int64_t n, int bits, bool float_keys, bool iota) {
    std::mt19937 rng(1234 + (unsigned)n);
    std::vector<uint32_t> k(n), v(n);
    for (int64_t i = 0; i < n; ++i) {
        if (float_keys) { float z = 0.2f + 20.f * (float)(rng() & 0xFFFFFF) / 16777216.f; memcpy(&k[i], &z, 4); }
        else k[i] = rng() & ((1u << bits) - 1u);
        v[i] = iota ? (uint32_t)i : rng();
    }
    std::vector<uint32_t> idx(n);
    std::iota(idx.begin(), idx.end(), 0u);
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return k[a] < k[b]; });
    uint32_t *dk, *dv, *ok, *ov, *table, *ticket;   // ticket = digit totals
    uint2 *pa, *pb;                                 // 8-byte (key, value) records of the intermediate passes
    hipMalloc(&dk, n * 4); hipMalloc(&dv, n * 4); hipMalloc(&pa, n * 8); hipMalloc(&pb, n * 8); hipMalloc(&ok, n * 4); hipMalloc(&ov, n * 4);
    hipMalloc(&table, vcr_sort_scratch_bytes(n)); hipMalloc(&ticket, VCR_SORT_TOTALS_WORDS * 4);
    hipMemcpy(dk, k.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dv, v.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms_mine = 0, ms_prim = 0;
    for (int rep = 0; rep < 6; ++rep) {
        hipMemsetAsync(ticket, 0, VCR_SORT_TOTALS_WORDS * 4, 0);
        hipEventRecord(e0);
        vcr_sort_pairs(n, dk, iota ? nullptr : dv, nullptr, pa, pb, ok, ov, 0, bits, table, ticket, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms_mine, e0, e1);
    }
    std::vector<uint32_t> rk(n), rv(n);
    hipMemcpy(rk.data(), ok, n * 4, hipMemcpyDeviceToHost); hipMemcpy(rv.data(), ov, n * 4, hipMemcpyDeviceToHost);
    int64_t bad = 0;
    for (int64_t i = 0; i < n; ++i) bad += (rk[i] != k[idx[i]]) || (rv[i] != v[idx[i]]);
    size_t tb = 0;
    rocprim::radix_sort_pairs<OneSweep>(nullptr, tb, dk, ok, dv, ov, (size_t)n, 0, bits, (hipStream_t)0);
    void* tmp; hipMalloc(&tmp, tb);
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        rocprim::radix_sort_pairs<OneSweep>(tmp, tb, dk, ok, dv, ov, (size_t)n, 0, bits, (hipStream_t)0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms_prim, e0, e1);
    }
    printf("%-28s n=%lld bits=%d  mismatches=%lld  hand-written %.1f us   rocPRIM onesweep %.1f us\n", name, (long long)n, bits,
           (long long)bad, ms_mine * 1e3, ms_prim * 1e3);
    hipFree(dk); hipFree(dv); hipFree(pa); hipFree(pb); hipFree(ok); hipFree(ov); hipFree(table); hipFree(ticket); hipFree(tmp);
}

int main() {
    run("depth keys (float bits)", 1000000, 32, true, true);
    run("depth keys 5M", 5000000, 32, true, true);
    run("tile keys 13 bit", 2954840, 13, false, false);
    run("tile keys 13 bit 12M", 12000000, 13, false, false);
    run("tiny", 777, 32, true, true);
    run("ragged 8193", 8193, 9, false, false);
    // tile order: permutation check
    const int T = 8160;
    std::vector<uint2> rg(T);
    std::mt19937 rng(7);
    for (int i = 0; i < T; ++i) { rg[i].x = 0; rg[i].y = (rng() % 100 < 84) ? 0 : rng() % 3000; }
    uint2* drg; uint32_t *dord, *dmeta; hipMalloc(&drg, T * 8); hipMalloc(&dord, T * 4); hipMalloc(&dmeta, VCR_BIN_META_WORDS * 4);
    hipMemcpy(drg, rg.data(), T * 8, hipMemcpyHostToDevice);
    hipMemset(dord, 0xFF, T * 4);
    vcr_launch_tile_order(T, drg, dord, dmeta, 1000000, true, false, 0);
    std::vector<uint32_t> ord(T);
    hipMemcpy(ord.data(), dord, T * 4, hipMemcpyDeviceToHost);
    std::vector<int> seen(T, 0); int badp = 0, inv = 0;
    for (int i = 0; i < T; ++i) { if (ord[i] >= (uint32_t)T || seen[ord[i]]++) ++badp; }
    for (int i = 1; i < T && !badp; ++i) inv += ((rg[ord[i]].y >> 2) > (rg[ord[i - 1]].y >> 2));      // (empty tiles last)
    vcr_launch_tile_order(T, drg, dord, dmeta, 1000000, true, true, 0);
    hipMemcpy(ord.data(), dord, T * 4, hipMemcpyDeviceToHost);
    std::fill(seen.begin(), seen.end(), 0); int badp2 = 0;
    for (int i = 0; i < T; ++i) { if (ord[i] >= (uint32_t)T || seen[ord[i]]++) ++badp2; }
    printf("tile_order: not-a-permutation=%d class-inversions=%d (snake: not-a-permutation=%d)\n", badp, inv, badp2);
    return 0;
}
