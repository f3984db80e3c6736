// Stable LSD radix sort of (u32 key, u32 value) pairs, sized for the rasterizer's two sorts (N ~ 1e6 depth keys,
// R ~ 3e6 tile keys of <= 14 bits) -- problem sizes where a device-wide library sort is launch- and latency-bound
// (rocPRIM onesweep here: 2 memsets + 1 kernel per 8-bit pass, ~40 us per pass at N = 1e6, plus a histogram kernel).
//
// Two kernels per 8-bit pass (three above 4.2 M items, see rs_hist_scan_kernel), no look-back chains, no memsets:
//   upsweep   : every block counts the digits of its slice into LDS, writes one row of hist[block][digit] and adds
//               the row to totals[pass][digit] (256 atomics per block);
//   downsweep : every block sums the rows of the blocks before it (coalesced 1 KB rows out of L2), scans the 256
//               totals, ranks its slice stably (wave w owns consecutive items and walks them in chunks of 64; peers
//               of a digit inside a chunk are found with one ballot per digit bit) and scatters.
// `totals` (VCR_SORT_TOTALS_WORDS words) must be zero on entry; the caller zeroes it together with its other counters.
#include "vcr_common.h"
#include <stdlib.h>

namespace {

constexpr int RS_BLOCK = 1024;                // threads
constexpr int RS_WAVES = RS_BLOCK / 64;
#ifndef VCR_RS_CHUNKS
#define VCR_RS_CHUNKS 8
#endif
constexpr int RS_CHUNKS = VCR_RS_CHUNKS;      // 64-item chunks per wave
constexpr int RS_IPB = RS_BLOCK * RS_CHUNKS;  // items per block
constexpr int RS_RADIX = 256;
#ifndef VCR_RS_INLINE_PREFIX_MAX
#define VCR_RS_INLINE_PREFIX_MAX 512
#endif
constexpr int RS_INLINE_PREFIX_MAX = VCR_RS_INLINE_PREFIX_MAX;   // blocks (x 8192 items) up to which the downsweep sums earlier rows itself

__global__ void __launch_bounds__(RS_BLOCK) rs_upsweep_kernel(int64_t n, const uint32_t* __restrict__ keys, int shift,
                                                             uint32_t mask, uint32_t* __restrict__ hist,
                                                             uint32_t* __restrict__ totals) {
    __shared__ uint32_t cnt[RS_RADIX];
    const int t = threadIdx.x;
    if (t < RS_RADIX) cnt[t] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_IPB;
#pragma unroll
    for (int c = 0; c < RS_CHUNKS; ++c) {
        const int64_t i = base + (int64_t)c * RS_BLOCK + t;   // (any order: counting only)
        if (i < n) atomicAdd(&cnt[(keys[i] >> shift) & mask], 1u);
    }
    __syncthreads();
    if (t < RS_RADIX) {
        const uint32_t c = cnt[t];
        hist[(size_t)blockIdx.x * RS_RADIX + t] = c;
        if (c) atomicAdd(totals + t, c);
    }
}

// Exclusive scan over the blocks of every digit's counts, in place (hist[b][d] -> number of items with digit d in the
// blocks before b): one workgroup per digit.  Used when the block count is large (> RS_INLINE_PREFIX_MAX blocks); below
// that the downsweep sums the rows of the earlier blocks itself, which spares a launch but is quadratic in the block count.
__global__ void __launch_bounds__(256) rs_hist_scan_kernel(int nblk, uint32_t* __restrict__ hist) {
    __shared__ uint32_t s_part[256];
    const int d = blockIdx.x, t = threadIdx.x;
    const int per = (nblk + 255) / 256, b0 = t * per, b1 = min(nblk, b0 + per);
    uint32_t sum = 0;
    for (int b = b0; b < b1; ++b) sum += hist[(size_t)b * RS_RADIX + d];
    s_part[t] = sum;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {                    // Hillis-Steele over the 256 partials
        const uint32_t v = t >= o ? s_part[t - o] : 0u;
        __syncthreads();
        s_part[t] += v;
        __syncthreads();
    }
    uint32_t run = s_part[t] - sum;
    for (int b = b0; b < b1; ++b) {
        const uint32_t c = hist[(size_t)b * RS_RADIX + d];
        hist[(size_t)b * RS_RADIX + d] = run;
        run += c;
    }
}

template <bool IOTA, bool PRESCAN>
__global__ void __launch_bounds__(RS_BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) rs_downsweep_kernel(int64_t n, const uint32_t* __restrict__ keys_in,
                                                               const uint32_t* __restrict__ vals_in, int shift, int nbits,
                                                               const uint32_t* __restrict__ hist,
                                                               const uint32_t* __restrict__ totals,
                                                               uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
    __shared__ uint32_t cnt[RS_WAVES][RS_RADIX];          // running digit counts of each wave's own item stream
    __shared__ uint32_t before_blk[4][RS_RADIX];          // partial sums over the rows of the earlier blocks
    __shared__ uint32_t dig_base[RS_RADIX];               // exclusive scan of the digit totals
    __shared__ uint32_t lstart[RS_RADIX];                 // block-local start of every digit
    // Half of the slice in sorted order (32 KB at 8 chunks): the reorder runs in two rounds so that the workgroup needs
    // 54 KB of LDS, not 86 KB -- with 86 KB it could not become resident on a CU that holds the two persistent 50 KB
    // workgroups of the side stream's SH-update kernel, and every scatter pass waited for that whole kernel
    __shared__ uint2 items[RS_IPB / 2];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const uint32_t mask = (1u << nbits) - 1u;
    for (int i = t; i < RS_WAVES * RS_RADIX; i += RS_BLOCK) (&cnt[0][0])[i] = 0;
    {   // rows of the blocks before this one: 4 thread groups x 256 digits, independent coalesced loads
        const int d = t & (RS_RADIX - 1), grp = t >> 8;
        uint32_t s = 0;
        if (PRESCAN) { if (grp == 0) s = hist[(size_t)blockIdx.x * RS_RADIX + d]; }
        else for (int b = grp; b < (int)blockIdx.x; b += 4) s += hist[(size_t)b * RS_RADIX + d];
        before_blk[grp][d] = s;
        if (t < RS_RADIX) dig_base[t] = totals[t];
    }
    __syncthreads();
    if (t < 64) {                                          // exclusive scan of 256 totals by one wave, 4 per lane
        const uint32_t a0 = dig_base[4 * t], a1 = dig_base[4 * t + 1], a2 = dig_base[4 * t + 2], a3 = dig_base[4 * t + 3];
        const uint32_t sum = a0 + a1 + a2 + a3;
        uint32_t inc = sum;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)inc, o);
            if (t >= o) inc += v;
        }
        const uint32_t ex = inc - sum;
        dig_base[4 * t] = ex; dig_base[4 * t + 1] = ex + a0; dig_base[4 * t + 2] = ex + a0 + a1; dig_base[4 * t + 3] = ex + a0 + a1 + a2;
    }
    const int64_t wbase = (int64_t)blockIdx.x * RS_IPB + (int64_t)w * (64 * RS_CHUNKS);
    uint32_t key[RS_CHUNKS], val[RS_CHUNKS], rank[RS_CHUNKS];
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int c = 0; c < RS_CHUNKS; ++c) {
        const int64_t i = wbase + c * 64 + lane;
        const bool valid = i < n;
        key[c] = valid ? keys_in[i] : 0xFFFFFFFFu;
        val[c] = valid ? (IOTA ? (uint32_t)i : vals_in[i]) : 0u;
    }
#pragma unroll
    for (int c = 0; c < RS_CHUNKS; ++c) {
        const bool valid = wbase + c * 64 + lane < n;
        const uint32_t d = (key[c] >> shift) & mask;
        unsigned long long peers = __builtin_amdgcn_ballot_w64(valid);
        for (int b = 0; b < nbits; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(bit);
            peers &= bit ? bal : ~bal;
        }
        const uint32_t before = cnt[w][d];                 // every peer reads the count before the leader bumps it
        __builtin_amdgcn_wave_barrier();
        if (valid && (peers & lt) == 0) cnt[w][d] = before + (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
        rank[c] = before + (uint32_t)__popcll(peers & lt);
    }
    __syncthreads();
    if (t < 64) {                                          // block-local start of every digit (exclusive scan, 4 per lane)
        uint32_t a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t c = 0;
#pragma unroll
            for (int ww = 0; ww < RS_WAVES; ++ww) c += cnt[ww][4 * t + k];
            a[k] = c;
        }
        const uint32_t sum = a[0] + a[1] + a[2] + a[3];
        uint32_t inc = sum;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)inc, o);
            if (t >= o) inc += v;
        }
        const uint32_t ex = inc - sum;
        lstart[4 * t] = ex; lstart[4 * t + 1] = ex + a[0]; lstart[4 * t + 2] = ex + a[0] + a[1]; lstart[4 * t + 3] = ex + a[0] + a[1] + a[2];
    }
    __syncthreads();
    if (t < RS_RADIX) {                                    // per-wave local bases; global minus local start of the digit
        const uint32_t ls = lstart[t];
        uint32_t run = ls;
#pragma unroll
        for (int ww = 0; ww < RS_WAVES; ++ww) {
            const uint32_t c = cnt[ww][t];
            cnt[ww][t] = run;
            run += c;
        }
        dig_base[t] = dig_base[t] + before_blk[0][t] + before_blk[1][t] + before_blk[2][t] + before_blk[3][t] - ls;
    }
    __syncthreads();
    // reorder through LDS so that the global writes of a wave are runs of consecutive addresses, not 64 scattered words
    const int64_t left = n - (int64_t)blockIdx.x * RS_IPB;
    const int nvalid = left < RS_IPB ? (int)left : RS_IPB;
#pragma unroll
    for (int c = 0; c < RS_CHUNKS; ++c) rank[c] += cnt[w][(key[c] >> shift) & mask];     // position inside the sorted slice
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const uint32_t lo = (uint32_t)half * (RS_IPB / 2);
        if (half) __syncthreads();                          // round 0 has drained `items`
#pragma unroll
        for (int c = 0; c < RS_CHUNKS; ++c) {
            if (wbase + c * 64 + lane < n && rank[c] - lo < (uint32_t)(RS_IPB / 2)) items[rank[c] - lo] = make_uint2(key[c], val[c]);
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < RS_CHUNKS / 2; ++c) {
            const int j = (int)lo + c * RS_BLOCK + t;
            if (j < nvalid) {
                const uint2 kv = items[j - (int)lo];
                const uint32_t dst = (uint32_t)j + dig_base[(kv.x >> shift) & mask];
                keys_out[dst] = kv.x;
                vals_out[dst] = kv.y;
            }
        }
    }
}

// Block-scheduling order of the compositing kernels: tiles by list length, longest first, folded boustrophedon-wise
// with the period of the chip.  The order is a placement policy, so lengths are quantised
// (2048 classes of 4 entries) and ties land in arbitrary order: ONE single-workgroup counting sort instead of a device-wide sort.
// Also decides how many of the heaviest tiles the compositing kernels launch as SPLIT work items (four workgroups of 4x4
// sub-blocks instead of one of 8x8 quads): only as many as there are idle workgroup slots on the chip,
// S = (slots - non-empty tiles) / 3 -- splitting shortens the serial chains of the longest lists (c2, 300 k Gaussians:
// compositing forward 221 -> 160 us, backward 467 -> 352 us) but costs 1.9x their shading work, which only pays while
// the SIMDs are not full (1 M Gaussians / 1080p: none).  meta[0] = S, meta[1] = non-empty tiles, meta[2] = longest list.
__global__ void __launch_bounds__(1024) tile_order_kernel(int T, const uint2* __restrict__ ranges, uint32_t* __restrict__ order,
                                                        uint32_t* __restrict__ meta, int split_slots, int lpt, int snake) {
    constexpr int BINS = 2048, SH = 2;                 // classes of 4 list entries; lists >= 8188 share the first class
    __shared__ uint32_t hist[BINS];
    __shared__ uint32_t wsum[16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (!lpt) {
        for (int i = t; i < T; i += 1024) order[i] = (uint32_t)i;
        if (t < VCR_BIN_META_WORDS) meta[t] = 0u;
        return;
    }
    __shared__ uint32_t s_ne, s_max;
    hist[t] = 0; hist[t + 1024] = 0;
    if (t == 0) { s_ne = 0; s_max = 0; }
    __syncthreads();
    uint32_t ne = 0, mx = 0;
    for (int i = t; i < T; i += 1024) {
        const uint32_t len = ranges[i].y - ranges[i].x;
        ne += len > 0; mx = max(mx, len);
        atomicAdd(&hist[BINS - 1 - min(len >> SH, (uint32_t)(BINS - 1))], 1u);   // bin 0 = longest
    }
    for (int o = 32; o > 0; o >>= 1) { ne += (uint32_t)__shfl_xor((int)ne, o); mx = max(mx, (uint32_t)__shfl_xor((int)mx, o)); }
    if (lane == 0) { atomicAdd(&s_ne, ne); atomicMax(&s_max, mx); }
    __syncthreads();
    if (t == 0) {
        const int idle = split_slots - (int)s_ne;
        int S = idle <= 0 ? 0 : min(min(idle / 3, (int)s_ne), VCR_SPLIT_MAX);
        if (S < 16) S = 0;                               // (a handful of split items only shifts the launch order of the rest)
        meta[0] = (uint32_t)S; meta[1] = s_ne; meta[2] = s_max;
    }
    // exclusive scan of the 2048 classes: 2 per lane, wave scan, 16 wave totals
    const uint32_t h0 = hist[2 * t], h1 = hist[2 * t + 1];
    uint32_t inc = h0 + h1;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)inc, o);
        if (lane >= o) inc += v;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int k = 0; k < w; ++k) base += wsum[k];
    const uint32_t ex = base + inc - (h0 + h1);
    hist[2 * t] = ex; hist[2 * t + 1] = ex + h0;           // exclusive start of every class
    __syncthreads();
    for (int i = t; i < T; i += 1024) {
        const uint32_t r = atomicAdd(&hist[BINS - 1 - min((ranges[i].y - ranges[i].x) >> SH, (uint32_t)(BINS - 1))], 1u);
        uint32_t pos = r;                                   // rank in launch order
        if (snake) {
            const int band = (int)(r >> 8), j = (int)(r & 255);
            const int band_len = min(256, T - (band << 8));
            if ((band & 1) && j < band_len) pos = (uint32_t)((band << 8) + (band_len - 1 - j));
        }
        order[pos] = (uint32_t)i;
    }
}

}  // namespace

size_t vcr_sort_scratch_bytes(int64_t n) {
    const int64_t nblk = (n + RS_IPB - 1) / RS_IPB;
    return vcr_align(sizeof(uint32_t) * (size_t)(RS_RADIX * (nblk > 0 ? nblk : 1)));
}

// Sorts bits [begin_bit, end_bit) of the keys (at most 4 passes of 8 bits).  vals_in == nullptr means vals = 0..n-1.
// The result lands in (keys_out, vals_out); (keys_tmp, vals_tmp) is a second buffer pair of n words each; the inputs are
// left untouched.  `hist` holds vcr_sort_scratch_bytes(n); `totals` is VCR_SORT_TOTALS_WORDS zero-initialised words.
int vcr_sort_pairs(int64_t n, const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_tmp, uint32_t* vals_tmp,
                   uint32_t* keys_out, uint32_t* vals_out, int begin_bit, int end_bit, uint32_t* hist, uint32_t* totals,
                   hipStream_t st) {
    if (n <= 0) return 0;
    const int nblk = (int)((n + RS_IPB - 1) / RS_IPB);
    const int passes = (end_bit - begin_bit + 7) / 8;
    if (passes > VCR_SORT_TOTALS_WORDS / RS_RADIX) { vcr_set_error("vcr_sort_pairs: more than 4 passes"); return 1; }
    const uint32_t* kin = keys_in;
    const uint32_t* vin = vals_in;
    for (int p = 0; p < passes; ++p) {
        const int shift = begin_bit + 8 * p;
        const int nbits = (end_bit - shift) < 8 ? (end_bit - shift) : 8;
        const bool to_out = ((passes - 1 - p) & 1) == 0;             // the last pass writes (keys_out, vals_out)
        uint32_t* kout = to_out ? keys_out : keys_tmp;
        uint32_t* vout = to_out ? vals_out : vals_tmp;
        uint32_t* tot = totals + p * RS_RADIX;
        hipLaunchKernelGGL(rs_upsweep_kernel, dim3(nblk), dim3(RS_BLOCK), 0, st, n, kin, shift, (1u << nbits) - 1u, hist, tot);
        const bool prescan = nblk > RS_INLINE_PREFIX_MAX;
        if (prescan) hipLaunchKernelGGL(rs_hist_scan_kernel, dim3(RS_RADIX), dim3(256), 0, st, nblk, hist);
#define VCR_DOWN(IOTA, PRE) hipLaunchKernelGGL((rs_downsweep_kernel<IOTA, PRE>), dim3(nblk), dim3(RS_BLOCK), 0, st, n, kin, vin, shift, \
                                               nbits, hist, tot, kout, vout)
        if (vin) { if (prescan) VCR_DOWN(false, true); else VCR_DOWN(false, false); }
        else { if (prescan) VCR_DOWN(true, true); else VCR_DOWN(true, false); }
#undef VCR_DOWN
        kin = kout; vin = vout;
    }
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

int vcr_launch_tile_order(int T, const uint2* ranges, uint32_t* order, uint32_t* meta, bool lpt, bool snake, hipStream_t st) {
    // workgroup slots of the compositing kernels on the chip: 5 resident 256-thread workgroups per CU (VCR_SPLIT_SLOTS overrides;
    // 0 disables the split work items)
    static const int slots = [] {
        const char* e = getenv("VCR_SPLIT_SLOTS");
        if (e) return atoi(e);
        int dev = 0, cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
        return 5 * cus;
    }();
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, st, T, ranges, order, meta, slots, lpt ? 1 : 0, snake ? 1 : 0);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}
