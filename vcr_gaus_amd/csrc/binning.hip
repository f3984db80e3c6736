// K2-K5: depth ordering, tile-instance emission, stable tile sort, per-tile ranges.
//
// MI355X-first ordering scheme (not the 64-bit (tile|depth) key sort of the public rasterizer):
//   1. radix-sort the N Gaussians once by their 32-bit view depth (N*8 B per pass, tiny),
//   2. emit (tile, id) instances in that depth order with a load-balanced wave-cooperative kernel,
//   3. STABLE radix sort of the R instances on the tile bits only (ceil(log2 T) <= 14 bits -> 2 passes
//      over 8-byte pairs instead of 6 passes over 12-byte pairs),
// which yields the same (tile, depth, id) order with ~3.5x less sort traffic.  Both sorts run on the hand-written
// two-kernels-per-pass radix sort of radix_sort.hip up to VCR_SORT_HAND_MAX items (its block-prefix step is quadratic
// in the block count); beyond that, and for the offsets scan, the rocPRIM device primitives are used.
#include "vcr_common.h"
#include <cstring>
#include <cstdlib>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

namespace {

// rocPRIM falls back to a 20-launch merge sort below 2^20 items; the N-Gaussian depth sort is always worth onesweep.
using DepthSortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;

struct GatherTiles {
    const uint32_t* tiles;
    __host__ __device__ uint32_t operator()(uint32_t id) const { return tiles[id]; }
};

__global__ void __launch_bounds__(256) duplicate_kernel(int N, int W, int H, const uint32_t* __restrict__ ids_sorted,
                                                        const uint32_t* __restrict__ offsets,
                                                        const uint2* __restrict__ rect,
                                                        uint32_t* __restrict__ keys_out,
                                                        uint32_t* __restrict__ vals_out, uint2* __restrict__ ranges,
                                                        int num_tiles) {
    __shared__ uint32_t s_end[4][64], s_start[4][64], s_id[4][64];
    __shared__ int s_xmin[4][64], s_ymin[4][64], s_w[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gi = blockIdx.x * 256 + threadIdx.x;
    const int gx = (W + VCR_TILE - 1) / VCR_TILE;
    for (int t = gi; t < num_tiles; t += gridDim.x * 256) ranges[t] = make_uint2(0u, 0u);   // empty tiles (tile_ranges fills the rest)
    uint32_t id = 0, cnt = 0, end;
    int xmin = 0, ymin = 0, w = 1;
    if (gi < N) {
        id = ids_sorted[gi];
        end = offsets[gi];
        const uint2 rc = rect[id];                      // {0, 0} for culled Gaussians
        w = (int)(rc.y & 0xFFFFu);
        cnt = (uint32_t)w * (rc.y >> 16);
        xmin = (int)(rc.x & 0xFFFFu); ymin = (int)(rc.x >> 16);
        if (cnt == 0) w = 1;
    } else {
        end = offsets[N - 1];
    }
    const uint32_t start = end - cnt;
    s_end[wv][lane] = end; s_start[wv][lane] = start; s_id[wv][lane] = id;
    s_xmin[wv][lane] = xmin; s_ymin[wv][lane] = ymin; s_w[wv][lane] = w;
    __builtin_amdgcn_wave_barrier();
    const uint32_t wave_base = s_start[wv][0];
    const uint32_t total = s_end[wv][63] - wave_base;
    for (uint32_t e = lane; e < total; e += 64) {
        const uint32_t target = wave_base + e;
        int lo = 0, hi = 63;                       // first lane whose end > target
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            const int mid = (lo + hi) >> 1;
            if (s_end[wv][mid] > target) hi = mid; else lo = mid + 1;
        }
        const uint32_t local = target - s_start[wv][lo];
        const int ww = s_w[wv][lo];
        const int ty = s_ymin[wv][lo] + (int)(local / (uint32_t)ww);
        const int tx = s_xmin[wv][lo] + (int)(local % (uint32_t)ww);
        keys_out[target] = (uint32_t)(ty * gx + tx);
        vals_out[target] = s_id[wv][lo];
    }
}

__global__ void __launch_bounds__(256) tile_ranges_kernel(int64_t R, const uint32_t* __restrict__ keys,
                                                          uint2* __restrict__ ranges) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= R) return;
    const uint32_t k = keys[i];
    if (i == 0) ranges[k].x = 0;
    else {
        const uint32_t kp = keys[i - 1];
        if (kp != k) { ranges[kp].y = (uint32_t)i; ranges[k].x = (uint32_t)i; }
    }
    if (i == R - 1) ranges[k].y = (uint32_t)R;
}

__global__ void iota_kernel(int n, uint32_t* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (uint32_t)i;
}

}  // namespace

constexpr int64_t VCR_SORT_HAND_MAX = 512ll * 8192ll;       // items; see radix_sort.hip

size_t vcr_binning_temp_bytes(int N, int64_t R, int tile_bits) {
    size_t b0 = 0, b1 = 0, b2 = 0;
    uint32_t* d = nullptr;
    if (N > VCR_SORT_HAND_MAX)
        (void)rocprim::radix_sort_pairs<DepthSortConfig>(nullptr, b0, d, d, d, d, (size_t)N, 0, 32, (hipStream_t)0);
    else b0 = vcr_sort_scratch_bytes(N);
    auto it = rocprim::make_transform_iterator(d, GatherTiles{d});
    (void)rocprim::inclusive_scan(nullptr, b1, it, d, (size_t)N, rocprim::plus<uint32_t>(), (hipStream_t)0);
    if (R > VCR_SORT_HAND_MAX) (void)rocprim::radix_sort_pairs(nullptr, b2, d, d, d, d, (size_t)R, 0, tile_bits, (hipStream_t)0);
    else if (R > 0) b2 = vcr_sort_scratch_bytes(R);
    size_t m = b0 > b1 ? b0 : b1;
    return vcr_align(m > b2 ? m : b2);
}

// depth order of the N Gaussians (ties by index) and the inclusive scan of their tile counts in that order.
// (tmp_k, tmp_v): N words each; `totals`: VCR_SORT_TOTALS_WORDS zeroed words.
int vcr_depth_sort_and_scan(int N, const uint32_t* depth_key, uint32_t* tmp_k, uint32_t* tmp_v, uint32_t* key_sorted,
                            uint32_t* ids_sorted, const uint32_t* tiles, uint32_t* offsets, uint32_t* totals, void* temp,
                            size_t temp_bytes, hipStream_t st) {
    size_t tb = temp_bytes;
    if (N > VCR_SORT_HAND_MAX) {
        hipLaunchKernelGGL(iota_kernel, dim3((N + 255) / 256), dim3(256), 0, st, N, tmp_v);
        VCR_HIP_CHECK(rocprim::radix_sort_pairs<DepthSortConfig>(temp, tb, depth_key, key_sorted, tmp_v, ids_sorted, (size_t)N, 0, 32, st));
    } else if (vcr_sort_pairs(N, depth_key, nullptr, tmp_k, tmp_v, key_sorted, ids_sorted, 0, 32, (uint32_t*)temp, totals, st)) {
        return 1;
    }
    tb = temp_bytes;
    auto it = rocprim::make_transform_iterator(ids_sorted, GatherTiles{tiles});
    VCR_HIP_CHECK(rocprim::inclusive_scan(temp, tb, it, offsets, (size_t)N, rocprim::plus<uint32_t>(), st));
    return 0;
}

// (keys_a, vals_a): instance buffers; (keys_t, vals_t): a second pair; keys_b / point_list: the sorted result.
int vcr_duplicate_and_sort(const VcrRasterArgs& a, GeomState g, const int32_t* radii, const uint32_t* ids_sorted,
                           const uint32_t* offsets, int64_t R, int tile_bits, uint32_t* keys_a, uint32_t* vals_a,
                           uint32_t* keys_t, uint32_t* vals_t, uint32_t* keys_b, uint32_t* point_list, uint2* ranges,
                           uint32_t* tile_order, int num_tiles, uint32_t* totals, void* temp, size_t temp_bytes,
                           hipStream_t st) {
    static const bool no_lpt = getenv("VCR_NO_LPT") != nullptr;          // experiment switches
    static const bool no_snake = getenv("VCR_NO_SNAKE") != nullptr;
    if (R <= 0) {
        VCR_HIP_CHECK(hipMemsetAsync(ranges, 0, sizeof(uint2) * (size_t)num_tiles, st));
        return vcr_launch_tile_order(num_tiles, ranges, tile_order, false, false, st);   // identity order
    }
    const int blocks = (a.N + 255) / 256;
    hipLaunchKernelGGL(duplicate_kernel, dim3(blocks), dim3(256), 0, st, a.N, a.W, a.H, ids_sorted, offsets, g.rect,
                       keys_a, vals_a, ranges, num_tiles);
    VCR_HIP_CHECK(hipGetLastError());
    if (R > VCR_SORT_HAND_MAX) {
        size_t tb = temp_bytes;
        VCR_HIP_CHECK(rocprim::radix_sort_pairs(temp, tb, keys_a, keys_b, vals_a, point_list, (size_t)R, 0, tile_bits, st));
    } else if (vcr_sort_pairs(R, keys_a, vals_a, keys_t, vals_t, keys_b, point_list, 0, tile_bits, (uint32_t*)temp, totals, st)) {
        return 1;
    }
    const int64_t rb = (R + 255) / 256;
    hipLaunchKernelGGL(tile_ranges_kernel, dim3((unsigned)rb), dim3(256), 0, st, R, keys_b, ranges);
    VCR_HIP_CHECK(hipGetLastError());
    return vcr_launch_tile_order(num_tiles, ranges, tile_order, !no_lpt, !no_snake, st);
}
