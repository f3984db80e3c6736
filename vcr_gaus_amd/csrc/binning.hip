// K2-K5: depth ordering, tile-instance emission, stable tile sort, per-tile ranges.
//
// MI355X-first ordering scheme (not the 64-bit (tile|depth) key sort of the public rasterizer):
//   1. radix-sort the N Gaussians once by their 32-bit view depth (8-byte pairs),
//   2. emit (tile, id) instances in that depth order (offsets: a two-level scan of the per-Gaussian instance counts, round 6;
//      rounds 2-5: a decoupled look-back inside the emission kernel).  Which tiles of its 3-sigma rectangle a Gaussian really reaches (alpha >= 1/255 at some pixel centre)
//      was decided EXACTLY by the projection kernel and travels as a bit mask in the 8-byte rectangle record (round 3;
//      rectangles of more than 32 tiles emit every tile),
//   3. STABLE radix sort of the emitted R' instances on the tile bits only (ceil(log2 T) <= 14 bits -> 2 passes of
//      <= 7 bits over 8-byte pairs instead of 6 passes over 12-byte pairs),
// which yields the same (tile, depth, id) order.  The host reads R' back together with V and R (the 3-sigma count the
// reference reports) right after the projection, so buffers and grids are sized for what is really emitted.
#include "vcr_common.h"
#include <cstring>
#include <cstdlib>

namespace {

constexpr int DUP_ROUNDS = 4;                       // Gaussians per thread: 1024 per block keeps the look-back chain short

// Tile-instance emission.  A block takes 1024 Gaussians of the depth order (4 rounds of 256), scans their instance counts (popcount
// of the tile mask, or width x height for the few rectangles without one), adds the instances of the slices before it (see
// duplicate_count_kernel below) and emits:
//   * masked rectangles (<= 32 tiles, nearly all): every lane walks the set bits of ITS mask -- a handful of iterations
//     (round 5 measured the block's piece of the output assembled in LDS and written as full lines: 80.7 against 78.4-80.7 us
//     at 1 M Gaussians -- the scattered 8-byte stores are not what the kernel waits for;
//     profiles/experiments/r5_emission_lds_staging_and_wide_lookback.patch);
//   * unmasked giants: wave-cooperatively and load-balanced -- the wave's giant counts are prefix-summed in LDS and every
//     lane binary-searches the Gaussian its slot belongs to, so a screen-filling Gaussian does not serialise a lane.
// `status`: library-owned words, one 32-bit slice sum per block since round 6 (rounds 2-5: 64-bit look-back words tagged with
// the call number `seq`; `ticket`: their atomic block counter -- both parameters are kept for the callers, unused).
// Round 6: the offsets of the emission come from a TWO-LEVEL scan instead of a ticketed decoupled look-back.  The look-back made
// this kernel the most erratic one of the step (profiles/r5_bench_kernel_stats.csv: 95 us on average, 308 us at worst, sigma 36 us
// for ~28 MB of traffic): a block can only finish when every block before it in ticket order has published, so one late
// workgroup stalls all its successors.  Now `duplicate_count_kernel` sums the instance counts of each 1024-Gaussian slice of the
// depth order (one word per slice, written to the library-owned status buffer), and the emission kernel adds up the words of the
// slices before its own -- at most N / 1024 coalesced 4-byte loads per block -- and never waits for another workgroup.
template <bool QL>
__device__ __forceinline__ uint32_t instance_count(uint2 rc, uint32_t mhi) {
    return (rc.x & VCR_RECT_MASKED) ? (uint32_t)(__popc(rc.y) + (QL ? __popc(mhi) : 0)) : (rc.y & 0xFFFFu) * (rc.y >> 16);
}

template <bool QL>
__global__ void __launch_bounds__(256) duplicate_count_kernel(int N, const uint32_t* __restrict__ ids_sorted, const uint2* __restrict__ rect,
                                                              const uint32_t* __restrict__ rect_hi, uint32_t* __restrict__ slice_sum) {
    // one Gaussian per thread, one word per 256 Gaussians of the depth order (four words per slice of the emission kernel): the
    // kernel is two dependent gathers deep, so it wants many short workgroups, not few long ones (1 024 per workgroup: 33 us at 1 M)
    __shared__ uint32_t s_w[4];
    const int gi = blockIdx.x * 256 + (int)threadIdx.x;
    uint32_t c = 0;
    if (gi < N) {
        const uint32_t id = ids_sorted[gi];
        const uint2 rc = rect[id];
        uint32_t mhi = 0;
        if (QL && (rc.x & (VCR_RECT_MASKED | VCR_RECT_MASK64)) == (VCR_RECT_MASKED | VCR_RECT_MASK64)) mhi = rect_hi[id];
        c = instance_count<QL>(rc, mhi);
    }
    for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) slice_sum[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

template <bool QL>      // QL: 64-bit cell masks (upper word in rect_hi); the per-tile form compiles to its 32-bit walk
__global__ void __launch_bounds__(256) duplicate_kernel(int N, int W, int H, const uint32_t* __restrict__ ids_sorted,
                                                        unsigned long long* __restrict__ status, uint32_t* __restrict__ ticket,
                                                        uint32_t seq, const uint2* __restrict__ rect, const uint32_t* __restrict__ rect_hi,
                                                        uint2* __restrict__ inst_out, uint2* __restrict__ ranges, int num_tiles,
                                                        int gx_keys) {
    __shared__ uint32_t s_gend[4][64], s_start[4][64], s_id[4][64];
    __shared__ int s_xmin[4][64], s_ymin[4][64], s_w[4][64];
    __shared__ uint32_t s_wtot[DUP_ROUNDS][4], s_before[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gx = gx_keys;                             // keys per row: tiles, or 8x8 cells in quad-list mode (rect is in the same unit)
    for (int t = blockIdx.x * 256 + threadIdx.x; t < num_tiles; t += gridDim.x * 256) ranges[t] = make_uint2(0u, 0u);   // empty tiles
    const int bid = (int)blockIdx.x;
    (void)ticket; (void)seq;
    // instances emitted by the slices before this one (duplicate_count_kernel wrote one word per slice)
    const uint32_t* slice_sum = reinterpret_cast<const uint32_t*>(status);
    uint32_t before = 0;
    for (int j = (int)threadIdx.x; j < DUP_ROUNDS * bid; j += 256) before += slice_sum[j];     // (one word per 256 Gaussians)
    for (int o = 32; o > 0; o >>= 1) before += (uint32_t)__shfl_xor((int)before, o);
    if (lane == 0) s_before[wv] = before;
    // round r of this block covers the Gaussians base + r*256 + tid of the depth order
    const int base = bid * (256 * DUP_ROUNDS);
    uint32_t id[DUP_ROUNDS], cnt[DUP_ROUNDS], inc[DUP_ROUNDS], mhi[DUP_ROUNDS];
    uint2 rc[DUP_ROUNDS];
#pragma unroll
    for (int r = 0; r < DUP_ROUNDS; ++r) {
        const int gi = base + r * 256 + (int)threadIdx.x;
        id[r] = 0; cnt[r] = 0; mhi[r] = 0; rc[r] = make_uint2(0u, 0u);
        if (gi < N) {
            id[r] = ids_sorted[gi];
            rc[r] = rect[id[r]];                          // {0, 0} for culled Gaussians
            if (QL && (rc[r].x & (VCR_RECT_MASKED | VCR_RECT_MASK64)) == (VCR_RECT_MASKED | VCR_RECT_MASK64)) mhi[r] = rect_hi[id[r]];
            cnt[r] = (rc[r].x & VCR_RECT_MASKED) ? (uint32_t)(__popc(rc[r].y) + __popc(mhi[r])) : (rc[r].y & 0xFFFFu) * (rc[r].y >> 16);
        }
    }
#pragma unroll
    for (int r = 0; r < DUP_ROUNDS; ++r) {               // inclusive scan inside the wave, wave totals through LDS
        uint32_t v = cnt[r];
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = (uint32_t)__shfl_up((int)v, o);
            if (lane >= o) v += u;
        }
        inc[r] = v;
        if (lane == 63) s_wtot[r][wv] = v;
    }
    __syncthreads();
    uint32_t round_base = s_before[0] + s_before[1] + s_before[2] + s_before[3];

#pragma unroll
    for (int r = 0; r < DUP_ROUNDS; ++r) {
        uint32_t wpre = 0;
        for (int k = 0; k < wv; ++k) wpre += s_wtot[r][k];
        const uint32_t start = round_base + wpre + inc[r] - cnt[r];      // first output slot of this lane's Gaussian
        round_base += s_wtot[r][0] + s_wtot[r][1] + s_wtot[r][2] + s_wtot[r][3];
        const bool masked = (rc[r].x & VCR_RECT_MASKED) != 0;
        const int xmin = (int)(rc[r].x & 0x3FFu), ymin = (int)((rc[r].x >> 10) & 0x3FFu);
        if (masked) {                                      // walk the set bits of the tile mask
            const int w = (int)((rc[r].x >> 20) & 31u) + 1;
            uint32_t at = start;
            if (QL) {
                unsigned long long m = ((unsigned long long)mhi[r] << 32) | rc[r].y;
                while (m) {
                    const int k = __builtin_ctzll(m);
                    m &= m - 1;
                    inst_out[at] = make_uint2((uint32_t)((ymin + k / w) * gx + xmin + k % w), id[r]);      // (cell, Gaussian)
                    ++at;
                }
            } else {
                uint32_t m = rc[r].y;
                while (m) {
                    const int k = __builtin_ctz(m);
                    m &= m - 1;
                    inst_out[at] = make_uint2((uint32_t)((ymin + k / w) * gx + xmin + k % w), id[r]);      // (tile, Gaussian)
                    ++at;
                }
            }
        }
        const bool giant = !masked && cnt[r] != 0;
        if (__builtin_amdgcn_ballot_w64(giant) == 0) continue;            // (wave-uniform)
        uint32_t gc = giant ? cnt[r] : 0u;                 // inclusive scan of the giants' counts inside the wave
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = (uint32_t)__shfl_up((int)gc, o);
            if (lane >= o) gc += u;
        }
        __builtin_amdgcn_wave_barrier();                 // the previous round's reads of this wave's LDS rows are done
        s_gend[wv][lane] = gc; s_start[wv][lane] = start; s_id[wv][lane] = id[r];
        s_xmin[wv][lane] = xmin; s_ymin[wv][lane] = ymin; s_w[wv][lane] = giant ? (int)(rc[r].y & 0xFFFFu) : 1;
        __builtin_amdgcn_wave_barrier();
        const uint32_t total = s_gend[wv][63];
        for (uint32_t e = lane; e < total; e += 64) {
            int lo = 0, hi = 63;                       // first lane whose end > e
#pragma unroll
            for (int it = 0; it < 6; ++it) {
                const int mid = (lo + hi) >> 1;
                if (s_gend[wv][mid] > e) hi = mid; else lo = mid + 1;
            }
            const uint32_t local = e - (lo ? s_gend[wv][lo - 1] : 0u);
            const int ww = s_w[wv][lo];
            const int ty = s_ymin[wv][lo] + (int)(local / (uint32_t)ww);
            const int tx = s_xmin[wv][lo] + (int)(local % (uint32_t)ww);
            const uint32_t target = s_start[wv][lo] + local;
            inst_out[target] = make_uint2((uint32_t)(ty * gx + tx), s_id[wv][lo]);
        }
    }
}

__global__ void __launch_bounds__(256) tile_ranges_kernel(int64_t R, const uint32_t* __restrict__ keys, uint2* __restrict__ ranges) {
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

}  // namespace

size_t vcr_binning_temp_bytes(int N, int64_t R, int tile_bits) {
    const size_t b0 = vcr_sort_scratch_bytes(N), b2 = R > 0 ? vcr_sort_scratch_bytes(R) : 0;
    return vcr_align(b0 > b2 ? b0 : b2);
}

// look-back status words of the emission kernel
// 64-bit words of the library-owned status buffer: one 32-bit slice sum per 256 Gaussians since round 6
size_t vcr_duplicate_status_words(int N) { return (size_t)((N + 511) / 512 + 2); }

// depth order of the N Gaussians (ties by index): three stable passes over the 27 low bits of the depth keys (see vcr_common.h).
// (pair_a, pair_b): N 8-byte records each -- pair_a holds the (key, id) records in that order afterwards; `totals`:
// VCR_SORT_TOTALS_WORDS words.
int vcr_depth_sort(int N, const uint32_t* depth_key, uint2* pair_a, uint2* pair_b, uint32_t* ids_sorted, uint32_t* totals,
                   void* temp, hipStream_t st) {
    if (vcr_sort_passes(VCR_DEPTH_KEY_BITS) != 3)                 // (VCR_SORT_DIGIT_BITS=8: four passes, the third writes pair_a)
        return vcr_sort_pairs(N, depth_key, nullptr, nullptr, pair_a, pair_b, nullptr, ids_sorted, 0, 32, (uint32_t*)temp, totals, st,
                              nullptr);
    return vcr_sort_pairs(N, depth_key, nullptr, nullptr, pair_a, pair_b, nullptr, ids_sorted, 0, VCR_DEPTH_KEY_BITS, (uint32_t*)temp,
                          totals, st, nullptr, pair_a);
}

// the pass over the upper key bits, when the projection reported a visible Gaussian beyond z = 13 107
int vcr_depth_sort_far(int N, uint2* pair_a, uint32_t* ids_sorted, uint32_t* totals, void* temp, hipStream_t st) {
    if (vcr_sort_passes(VCR_DEPTH_KEY_BITS) != 3) return 0;       // (the four-pass plan already covered all 32 bits)
    return vcr_sort_pairs(N, nullptr, nullptr, pair_a, nullptr, nullptr, nullptr, ids_sorted, VCR_DEPTH_KEY_BITS, 32, (uint32_t*)temp,
                          totals, st, nullptr);
}

// inst: the emitted (tile, Gaussian) records; (pair_a, pair_b): buffers of the sort's intermediate passes (pair_b may be
// NULL when the tile bits take at most two passes); keys_b / point_list: the sorted result.
// R: the number of instances the emission kernel writes (the host's read-back of the projection kernel's count).
int vcr_duplicate_and_sort(const VcrRasterArgs& a, GeomState g, const int32_t* radii, const uint32_t* ids_sorted,
                           unsigned long long* status, uint32_t* ticket, uint32_t seq, int64_t R, int tile_bits, uint2* inst,
                           uint2* pair_a, uint2* pair_b,
                           uint32_t* keys_b, uint32_t* point_list, uint2* ranges, uint32_t* tile_order, uint32_t* meta,
                           int num_tiles, uint32_t* totals, void* temp, size_t temp_bytes, hipStream_t st) {
#ifdef VCR_DETERMINISTIC_BWD
    const bool no_lpt = true, no_snake = true;       // test-only build: identity launch order (the counting sort of tile_order
                                                     // places tiles of equal length in the order its LDS atomics resolve)
#else
    const bool no_lpt = false, no_snake = false;     // (longest-first, folded launch order: DESIGN.md section 4 holds the A/B; round 6 with
                                                     //  the two-phase forward: fold on / off 178.7 / 178.3 us forward, 401 / 414 backward at
                                                     //  the metric scene, profiles/r6_persistent_ab.txt rows blk / blkns)
#endif
    const int gx_tiles = (a.W + VCR_TILE - 1) / VCR_TILE;
    const bool ql = a.quad_lists != 0;
    const int num_keys = ql ? 4 * num_tiles : num_tiles, gx_keys = ql ? 2 * gx_tiles : gx_tiles;
    if (R <= 0) {
        VCR_HIP_CHECK(hipMemsetAsync(ranges, 0, sizeof(uint2) * (size_t)num_keys, st));
        return vcr_launch_tile_order(num_tiles, ranges, tile_order, meta, 0, false, false, st, ql ? gx_keys : 0);   // identity order
    }
    const int blocks = (a.N + 256 * DUP_ROUNDS - 1) / (256 * DUP_ROUNDS), cblocks = (a.N + 255) / 256;
    if (ql)
        hipLaunchKernelGGL(duplicate_count_kernel<true>, dim3(cblocks), dim3(256), 0, st, a.N, ids_sorted, g.rect, g.rect_hi, (uint32_t*)status);
    else
        hipLaunchKernelGGL(duplicate_count_kernel<false>, dim3(cblocks), dim3(256), 0, st, a.N, ids_sorted, g.rect, g.rect_hi, (uint32_t*)status);
    if (ql)
        hipLaunchKernelGGL(duplicate_kernel<true>, dim3(blocks), dim3(256), 0, st, a.N, a.W, a.H, ids_sorted, status, ticket, seq, g.rect,
                           g.rect_hi, inst, ranges, num_keys, gx_keys);
    else
        hipLaunchKernelGGL(duplicate_kernel<false>, dim3(blocks), dim3(256), 0, st, a.N, a.W, a.H, ids_sorted, status, ticket, seq, g.rect,
                           g.rect_hi, inst, ranges, num_keys, gx_keys);
    VCR_HIP_CHECK(hipGetLastError());
    if (vcr_sort_pairs(R, nullptr, nullptr, inst, pair_a, pair_b, keys_b, point_list, 0, tile_bits, (uint32_t*)temp, totals, st,
                       nullptr)) {
        return 1;
    }
    const int64_t rb = (R + 255) / 256;
    hipLaunchKernelGGL(tile_ranges_kernel, dim3((unsigned)rb), dim3(256), 0, st, R, keys_b, ranges);
    VCR_HIP_CHECK(hipGetLastError());
    return vcr_launch_tile_order(num_tiles, ranges, tile_order, meta, R, !no_lpt, !no_snake, st, ql ? gx_keys : 0);
}
