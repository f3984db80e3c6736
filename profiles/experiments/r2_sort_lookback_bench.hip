// Correctness (against std::stable_sort) and timing of csrc/radix_sort.hip in its two forms: histogram + one look-back
// kernel per pass (a zeroed `status` area is passed) against upsweep + downsweep per pass (status = nullptr).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../vcr_gaus_amd/csrc sort_lookback_bench.hip \
//        ../../vcr_gaus_amd/csrc/radix_sort.hip -o sort_lookback_bench
#include "vcr_common.h"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <random>
#include <vector>

void vcr_set_error(const char* fmt, ...) { fprintf(stderr, "error: %s\n", fmt); }

static int run(const char* name, int64_t n, int bits, bool float_keys, bool iota) {
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
    uint32_t *dk, *dv, *tk, *tv, *ok, *ov, *table, *totals;
    unsigned long long* status = nullptr;
    const size_t stb = vcr_sort_status_bytes(n, 0, bits);
    hipMalloc(&dk, n * 4); hipMalloc(&dv, n * 4); hipMalloc(&tk, n * 4); hipMalloc(&tv, n * 4); hipMalloc(&ok, n * 4); hipMalloc(&ov, n * 4);
    hipMalloc(&table, vcr_sort_scratch_bytes(n)); hipMalloc(&totals, VCR_SORT_TOTALS_WORDS * 4);
    if (stb) hipMalloc(&status, stb);
    hipMemcpy(dk, k.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dv, v.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int64_t bad_total = 0;
    float us[2] = {0.f, 0.f};
    for (int mode = 0; mode < 2; ++mode) {                 // 0: upsweep + downsweep, 1: look-back
        if (mode == 1 && !stb) { us[1] = -1.f; continue; }
        const int reps = 20;
        float sum = 0.f;
        for (int rep = 0; rep < reps + 3; ++rep) {
            hipMemsetAsync(ok, 0xFF, n * 4, 0); hipMemsetAsync(ov, 0xFF, n * 4, 0);
            hipMemsetAsync(totals, 0, VCR_SORT_TOTALS_WORDS * 4, 0);
            if (mode == 1) hipMemsetAsync(status, 0, stb, 0);
            hipEventRecord(e0);
            vcr_sort_pairs(n, dk, iota ? nullptr : dv, tk, tv, ok, ov, 0, bits, table, totals, mode ? status : nullptr, 0);
            hipEventRecord(e1);
            if (hipEventSynchronize(e1) != hipSuccess) { printf("%s: device error in mode %d\n", name, mode); return 1; }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 3) sum += ms;
        }
        us[mode] = sum / reps * 1e3f;
        std::vector<uint32_t> rk(n), rv(n);
        hipMemcpy(rk.data(), ok, n * 4, hipMemcpyDeviceToHost); hipMemcpy(rv.data(), ov, n * 4, hipMemcpyDeviceToHost);
        int64_t bad = 0;
        for (int64_t i = 0; i < n; ++i) bad += (rk[i] != k[idx[i]]) || (rv[i] != v[idx[i]]);
        if (bad) printf("%s: mode %d has %lld mismatches\n", name, mode, (long long)bad);
        bad_total += bad;
    }
    printf("%-26s n=%-9lld bits=%-2d mismatches=%lld  upsweep+downsweep %.1f us   look-back %.1f us\n", name, (long long)n, bits,
           (long long)bad_total, us[0], us[1]);
    hipFree(dk); hipFree(dv); hipFree(tk); hipFree(tv); hipFree(ok); hipFree(ov); hipFree(table); hipFree(totals);
    if (status) hipFree(status);
    return bad_total != 0;
}

int main() {
    int bad = 0;
    bad += run("depth keys (float bits)", 1000000, 32, true, true);
    bad += run("depth keys 300k", 300001, 32, true, true);
    bad += run("depth keys 2M", 2000000, 32, true, true);
    bad += run("tile keys 13 bit", 2954840, 13, false, false);
    bad += run("tile keys 512 blocks", 512 * 8192, 13, false, false);
    bad += run("tile keys 513 blocks", 512 * 8192 + 1, 13, false, false);
    bad += run("tiny", 777, 32, true, true);
    bad += run("one", 1, 32, true, true);
    bad += run("ragged 8193", 8193, 9, false, false);
    bad += run("17 bits", 70000, 17, false, false);
    printf(bad ? "FAILED\n" : "all sorts exact\n");
    return bad;
}
