// Stable LSD radix sort of (u32 key, u32 value) pairs, sized for the rasterizer's two sorts (N ~ 1e6 depth keys,
// R' ~ 2e6 tile keys of <= 14 bits) -- problem sizes where a device-wide sort is bound by its dependent launches and
// by how much of the chip every launch reaches, not by bytes.
//
// Round-3 form (round 2: 8-bit digits, 8192 items per 1024-thread workgroup = 123 workgroups for 1 M keys on 256 CUs):
//   * WIDE DIGITS: up to 11 bits per pass, so the 32-bit depth keys take 3 passes instead of 4 and the <= 14 tile bits
//     take two passes of <= 7 bits;
//   * 4096 items per workgroup, 4 per lane (16 waves, short dependent chains per wave): 245 workgroups at 1 M keys;
//   * three short kernels per pass, none of them with work quadratic in the block count:
//       upsweep   : digit histogram of every 4096-item slice in LDS -> one row of hist[block][digit];
//       scan      : exclusive scan over the blocks of every digit column, in place, plus the digit totals;
//       downsweep : every block ranks its slice stably (wave w owns 256 consecutive items and walks them in chunks of
//                   64; the peers of a digit inside a chunk come from one ballot per digit bit; per-wave running counts
//                   are 16-bit, 2 x RADIX bytes per wave), reorders it through LDS so that the global writes are runs,
//                   and scatters;
//   * (key, value) pairs travel between the passes as ONE 8-byte record (uint2): the scatter of a pass then writes a
//     run of equal digits as one contiguous piece of 8 x run bytes with one store per item instead of two 4-byte
//     stores into two arrays (at 256 digits and 4096 items per workgroup a run is 16 items = one full 128-byte line
//     instead of two half lines); only the first pass reads and the last pass writes separate arrays;
//   * the element count may also live in DEVICE memory (`n_dev`, <= the host's n): grids are then sized for the host's
//     upper bound and surplus workgroups return at once.
// Nothing has to be zeroed by the caller.
#include "vcr_common.h"
#include <stdlib.h>

namespace {

constexpr int RS_IPB = 4096;                              // items per workgroup (the row granularity of hist[block][digit])
// Workgroup shapes (threads x 64-item chunks per wave = RS_IPB).  The depth sort runs BESIDE the persistent SH-update kernel
// of the second stream (256-thread workgroups, 50 KB of LDS and 124 VGPRs each, up to two per CU), so its
// kernels must fit into what that leaves on a CU -- 60 KB of LDS, 256 VGPRs per SIMD lane -- or every pass waits for the
// whole SH update: the 11-bit downsweep therefore uses 512 threads x 8 chunks (56 KB; 16-bit per-wave counts are what
// make 2048 digits fit at all), the <= 8-bit passes 1024 threads x 4 chunks (25 KB).
template <int BITS> struct RsShape { static constexpr int THREADS = BITS > 8 ? 512 : 1024; };

__device__ __forceinline__ int64_t rs_count(int64_t n_host, const uint32_t* __restrict__ n_dev) {
    return n_dev ? (int64_t)*n_dev : n_host;
}

template <int BITS, bool AOS>
__global__ void __launch_bounds__(1024) rs_upsweep_kernel(int64_t n_host, const uint32_t* __restrict__ n_dev,
                                                               const uint32_t* __restrict__ keys, int shift, uint32_t mask,
                                                               uint32_t* __restrict__ hist) {
    constexpr int RADIX = 1 << BITS;
    const int64_t n = rs_count(n_host, n_dev);
    const int64_t base = (int64_t)blockIdx.x * RS_IPB;
    if (base >= n) return;
    __shared__ uint32_t cnt[RADIX];
    const int t = threadIdx.x;
    constexpr int UT = 1024, UC = RS_IPB / UT;
    for (int i = t; i < RADIX; i += UT) cnt[i] = 0;
    __syncthreads();
    uint32_t k[UC];
#pragma unroll
    for (int c = 0; c < UC; ++c) {                         // (any order: counting only) -- all loads first
        const int64_t i = base + (int64_t)c * UT + t;
        k[c] = i < n ? keys[AOS ? 2 * i : i] : 0u;        // (AoS: the key is the first word of the 8-byte record)
    }
#pragma unroll
    for (int c = 0; c < UC; ++c)
        if (base + (int64_t)c * UT + t < n) atomicAdd(&cnt[(k[c] >> shift) & mask], 1u);
    __syncthreads();
    for (int i = t; i < RADIX; i += UT) hist[(size_t)blockIdx.x * RADIX + i] = cnt[i];
}

// hist[b][d] -> number of items with digit d in the blocks before b (exclusive scan down every digit column, in place);
// totals[d] = the column sum.  One workgroup per 16 digit columns (rows are read in 64-byte pieces), 64 row groups of
// consecutive blocks; a thread keeps its (at most 32) values in registers between the sum and the write-back, and all of
// its loads are in flight together -- the first version looped `sum += hist[..]` (dependent round trips, twice) and took
// 11 us for 256 KB.
template <int BITS>
__global__ void __launch_bounds__(1024) rs_scan_kernel(int64_t n_host, const uint32_t* __restrict__ n_dev,
                                                       uint32_t* __restrict__ hist, uint32_t* __restrict__ totals) {
    constexpr int RADIX = 1 << BITS;
    constexpr int KEEP = 32;                               // rows per thread held in registers (64 x 32 x 4096 = 8.4 M items)
    const int64_t n = rs_count(n_host, n_dev);
    const int nblk = (int)((n + RS_IPB - 1) / RS_IPB);
    __shared__ uint32_t part[64][17];
    const int dd = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int d = blockIdx.x * 16 + dd;
    const int per = (nblk + 63) / 64, b0 = rg * per, b1 = min(nblk, b0 + per);
    uint32_t* col = hist + (size_t)b0 * RADIX + d;
    const int cnt = max(b1 - b0, 0);
    uint32_t v[KEEP];
    uint32_t sum = 0;
    if (per <= KEEP) {
#pragma unroll
        for (int k = 0; k < KEEP; ++k) v[k] = k < cnt ? col[(size_t)k * RADIX] : 0u;
#pragma unroll
        for (int k = 0; k < KEEP; ++k) sum += v[k];
    } else {
        for (int k = 0; k < cnt; ++k) sum += col[(size_t)k * RADIX];
    }
    part[rg][dd] = sum;
    __syncthreads();
    uint32_t run = 0;
    for (int g = 0; g < rg; ++g) run += part[g][dd];
    if (rg == 63) totals[d] = run + sum;
    if (per <= KEEP) {
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            if (k < cnt) col[(size_t)k * RADIX] = run;
            run += v[k];
        }
    } else {
        for (int k = 0; k < cnt; ++k) {
            const uint32_t c = col[(size_t)k * RADIX];
            col[(size_t)k * RADIX] = run;
            run += c;
        }
    }
}

// Exclusive scan over the whole block of the per-thread sums `sum` (each thread holds PER consecutive values).
// `wsum`: one word of LDS per wave.  Returns the exclusive prefix of this thread's first value.
__device__ __forceinline__ uint32_t block_exclusive(uint32_t sum, uint32_t* wsum) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    uint32_t inc = sum;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)inc, o);
        if (lane >= o) inc += v;
    }
    __syncthreads();                                        // (wsum may still be read from a previous call)
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int k = 0; k < w; ++k) base += wsum[k];
    return base + inc - sum;
}

// IN: 0 = separate key / value arrays, 1 = key array + values 0..n-1, 2 = 8-byte (key, value) records in `keys_in`
// OUT: 0 = separate arrays (keys_out may be NULL: values only), 1 = 8-byte records into `keys_out`, 2 = 8-byte records
// into `keys_out` AND the values into `vals_out` (the depth sort's last pass: ids for the emission, records in case a pass
// over the upper key bits has to follow)
template <int BITS, int IN, int OUT>
__global__ void __launch_bounds__(RsShape<BITS>::THREADS) rs_downsweep_kernel(int64_t n_host, const uint32_t* __restrict__ n_dev,
                                                                 const uint32_t* __restrict__ keys_in,
                                                                 const uint32_t* __restrict__ vals_in, int shift, int nbits,
                                                                 const uint32_t* __restrict__ hist,
                                                                 const uint32_t* __restrict__ totals,
                                                                 uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
    constexpr int RADIX = 1 << BITS;
    constexpr int THREADS = RsShape<BITS>::THREADS, WAVES = THREADS / 64, CHUNKS = RS_IPB / THREADS;
    constexpr int PER = RADIX > THREADS ? RADIX / THREADS : 1;             // digits per thread in the block-wide scans
    static_assert(64 * CHUNKS <= 65535, "per-wave digit counts are 16-bit");
    const int64_t n = rs_count(n_host, n_dev);
    const int64_t bbase = (int64_t)blockIdx.x * RS_IPB;
    if (bbase >= n) return;
    __shared__ uint16_t cnt[WAVES][RADIX];                 // running digit counts of each wave's own item stream
    __shared__ uint32_t gbase[RADIX];                       // global position of the block's first item of every digit, minus its local start
    __shared__ uint32_t wsum[WAVES];
    __shared__ uint2 items[RS_IPB / 2];                     // half of the slice in sorted order (two reorder rounds)
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const uint32_t mask = (1u << nbits) - 1u;
    {
        uint32_t* z = reinterpret_cast<uint32_t*>(&cnt[0][0]);
        for (int i = t; i < WAVES * RADIX / 2; i += THREADS) z[i] = 0;
    }
    // loads first: this block's slice, its row of block prefixes and the digit totals
    const int64_t wbase = bbase + (int64_t)w * (64 * CHUNKS);
    uint32_t key[CHUNKS], val[CHUNKS], rank[CHUNKS];
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
        const int64_t i = wbase + c * 64 + lane;
        const bool valid = i < n;
        if (IN == 2) {
            const uint2 kv = valid ? reinterpret_cast<const uint2*>(keys_in)[i] : make_uint2(0xFFFFFFFFu, 0u);
            key[c] = kv.x; val[c] = kv.y;
        } else {
            key[c] = valid ? keys_in[i] : 0xFFFFFFFFu;
            val[c] = valid ? (IN == 1 ? (uint32_t)i : vals_in[i]) : 0u;
        }
    }
    uint32_t tot[PER], pre[PER];
    uint32_t tsum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int d = PER * t + k;
        const bool in = d < RADIX;
        tot[k] = in ? totals[d] : 0u;
        pre[k] = in ? hist[(size_t)blockIdx.x * RADIX + d] : 0u;
        tsum += tot[k];
    }
    // exclusive scan of the digit totals -> start of every digit in the output
    const uint32_t dstart = block_exclusive(tsum, wsum);
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
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
        if (valid && (peers & lt) == 0) cnt[w][d] = (uint16_t)(before + (uint32_t)__popcll(peers));
        __builtin_amdgcn_wave_barrier();
        rank[c] = before + (uint32_t)__popcll(peers & lt);
    }
    __syncthreads();
    // block-local start of every digit; per-wave bases (cnt[w][d] <- items of digit d in the waves before w)
    uint32_t cd[PER];
    uint32_t csum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int d = PER * t + k;
        uint32_t c = 0;
        if (d < RADIX) {
#pragma unroll
            for (int ww = 0; ww < WAVES; ++ww) { const uint32_t x = cnt[ww][d]; cnt[ww][d] = (uint16_t)c; c += x; }
        }
        cd[k] = c;
        csum += c;
    }
    uint32_t ls = block_exclusive(csum, wsum), ds = dstart;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int d = PER * t + k;
        if (d < RADIX) {
#pragma unroll
            for (int ww = 0; ww < WAVES; ++ww) cnt[ww][d] = (uint16_t)(cnt[ww][d] + ls);     // (< RS_IPB = 4096: fits)
            gbase[d] = ds + pre[k] - ls;
        }
        ls += cd[k]; ds += tot[k];
    }
    __syncthreads();
    // reorder through LDS so that the global writes of a wave are runs of consecutive addresses, not 64 scattered words
    const int64_t left = n - bbase;
    const int nvalid = left < RS_IPB ? (int)left : RS_IPB;
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) rank[c] += cnt[w][(key[c] >> shift) & mask];     // position inside the sorted slice
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const uint32_t lo = (uint32_t)half * (RS_IPB / 2);
        if (half) __syncthreads();                          // round 0 has drained `items`
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c) {
            if (wbase + c * 64 + lane < n && rank[c] - lo < (uint32_t)(RS_IPB / 2)) items[rank[c] - lo] = make_uint2(key[c], val[c]);
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CHUNKS / 2; ++c) {
            const int j = (int)lo + c * THREADS + t;
            if (j < nvalid) {
                const uint2 kv = items[j - (int)lo];
                const uint32_t dst = (uint32_t)j + gbase[(kv.x >> shift) & mask];
                if (OUT >= 1) reinterpret_cast<uint2*>(keys_out)[dst] = kv;
                if (OUT == 2) vals_out[dst] = kv.y;
                if (OUT == 0) {
                    if (keys_out) keys_out[dst] = kv.x;
                    vals_out[dst] = kv.y;
                }
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
                                                        uint32_t* __restrict__ meta, unsigned long long instances, int split_slots,
                                                        int lpt, int snake, int gxc) {
    constexpr int BINS = 2048, SH = 2;                 // classes of 4 list entries; lists >= 8188 share the first class
    // quad-list mode (gxc = 8x8 cells per row, 0 = off): `ranges` is per cell; a tile's length is the sum of its four quads'
    // lists (what its workgroup shades), the longest serial chain is the longest CELL list
    auto tile_len = [&](int i, uint32_t& longest) -> uint32_t {
        if (!gxc) { const uint32_t l = ranges[i].y - ranges[i].x; longest = l; return l; }
        const int gxt = gxc >> 1, tx = i % gxt, ty = i / gxt;
        const uint2 a0 = ranges[(2 * ty) * gxc + 2 * tx], a1 = ranges[(2 * ty) * gxc + 2 * tx + 1];
        const uint2 a2 = ranges[(2 * ty + 1) * gxc + 2 * tx], a3 = ranges[(2 * ty + 1) * gxc + 2 * tx + 1];
        const uint32_t l0 = a0.y - a0.x, l1 = a1.y - a1.x, l2 = a2.y - a2.x, l3 = a3.y - a3.x;
        longest = max(max(l0, l1), max(l2, l3));
        return l0 + l1 + l2 + l3;
    };
    __shared__ uint32_t hist[BINS];
    __shared__ uint32_t wsum[16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (!lpt) {
        for (int i = t; i < T; i += 1024) order[i] = (uint32_t)i;
        if (t < VCR_BIN_META_WORDS) meta[t] = t == 3 ? (uint32_t)gxc : 0u;
        return;
    }
    __shared__ uint32_t s_ne, s_max, s_empty;
    hist[t] = 0; hist[t + 1024] = 0;
    if (t == 0) { s_ne = 0; s_max = 0; }
    __syncthreads();
    // Empty tiles (84 % of the 1080p frame of the metric scene) do not go through the histogram: thousands of LDS atomics
    // on ONE word serialise (this kernel measured 14.6 us for 8160 tiles); they are placed behind the non-empty tiles
    // with one atomic per wave instead.
    uint32_t ne = 0, mx = 0;
    for (int i = t; i < T; i += 1024) {
        uint32_t longest;
        const uint32_t len = tile_len(i, longest);
        ne += len > 0; mx = max(mx, longest);
        if (len > 0) atomicAdd(&hist[BINS - 1 - min(len >> SH, (uint32_t)(BINS - 1))], 1u);   // bin 0 = longest
    }
    for (int o = 32; o > 0; o >>= 1) { ne += (uint32_t)__shfl_xor((int)ne, o); mx = max(mx, (uint32_t)__shfl_xor((int)mx, o)); }
    if (lane == 0) { atomicAdd(&s_ne, ne); atomicMax(&s_max, mx); }
    __syncthreads();
    if (t == 0) {
        const int idle = split_slots - (int)s_ne;
        int S = idle <= 0 ? 0 : min(min(idle / 3, (int)s_ne), VCR_SPLIT_MAX);
        if (S < 16) S = 0;                               // (a handful of split items only shifts the launch order of the rest)
        // ... and only when the launch is bound by its longest serial chain rather than by total work: longest list > 4x the
        // list entries per workgroup slot (c2: 6400 against 590 -> split; 1 M / 1080p: 4500 against 1200 -> not: there the
        // SIMDs stay full to the end and the 1.9x work of the split items costs more than their shorter chains return)
        if ((unsigned long long)s_max * (unsigned)split_slots <= (gxc ? 1ull : 4ull) * instances) S = 0;
        meta[0] = (uint32_t)S; meta[1] = s_ne; meta[2] = s_max; meta[3] = (uint32_t)gxc;
        s_empty = s_ne;                                  // the empty tiles follow the non-empty ones in launch order
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
    for (int i0 = 0; i0 < T; i0 += 1024) {
        const int i = i0 + t;
        uint32_t longest_;
        const uint32_t len = i < T ? tile_len(i, longest_) : 1u;
        const unsigned long long em = __builtin_amdgcn_ballot_w64(i < T && len == 0);
        uint32_t ebase = 0;
        if (em) {
            if (lane == 0) ebase = atomicAdd(&s_empty, (uint32_t)__popcll(em));
            ebase = (uint32_t)__builtin_amdgcn_readfirstlane((int)ebase);
        }
        if (i >= T) continue;
        const uint32_t r = len == 0 ? ebase + (uint32_t)__popcll(em & ((1ull << lane) - 1ull))
                                    : atomicAdd(&hist[BINS - 1 - min(len >> SH, (uint32_t)(BINS - 1))], 1u);
        uint32_t pos = r;                                   // rank in launch order
        if (snake) {
            const int band = (int)(r >> 8), j = (int)(r & 255);
            const int band_len = min(256, T - (band << 8));
            if ((band & 1) && j < band_len) pos = (uint32_t)((band << 8) + (band_len - 1 - j));
        }
        order[pos] = (uint32_t)i;
    }
}

template <int BITS, int IN, int OUT>
void rs_pass(int64_t n, const uint32_t* n_dev, const uint32_t* kin, const uint32_t* vin, uint32_t* kout, uint32_t* vout, int shift,
             int nbits, uint32_t* hist, uint32_t* totals, hipStream_t st) {
    constexpr int RADIX = 1 << BITS;
    const int nblk = (int)((n + RS_IPB - 1) / RS_IPB);
    constexpr int DT = RsShape<BITS>::THREADS;
    hipLaunchKernelGGL((rs_upsweep_kernel<BITS, IN == 2>), dim3(nblk), dim3(1024), 0, st, n, n_dev, kin, shift, (1u << nbits) - 1u, hist);
    hipLaunchKernelGGL((rs_scan_kernel<BITS>), dim3(RADIX / 16), dim3(1024), 0, st, n, n_dev, hist, totals);
    hipLaunchKernelGGL((rs_downsweep_kernel<BITS, IN, OUT>), dim3(nblk), dim3(DT), 0, st, n, n_dev, kin, vin, shift, nbits, hist,
                       totals, kout, vout);
}

template <int BITS>
void rs_pass_any(int in, int out, int64_t n, const uint32_t* n_dev, const uint32_t* kin, const uint32_t* vin, uint32_t* kout,
                 uint32_t* vout, int shift, int nbits, uint32_t* hist, uint32_t* totals, hipStream_t st) {
#define VCR_RS(IN, OUT) rs_pass<BITS, IN, OUT>(n, n_dev, kin, vin, kout, vout, shift, nbits, hist, totals, st)
    if (out == 1) { if (in == 0) VCR_RS(0, 1); else if (in == 1) VCR_RS(1, 1); else VCR_RS(2, 1); }
    else if (out == 2) { if (in == 0) VCR_RS(0, 2); else if (in == 1) VCR_RS(1, 2); else VCR_RS(2, 2); }
    else { if (in == 0) VCR_RS(0, 0); else if (in == 1) VCR_RS(1, 0); else VCR_RS(2, 0); }
#undef VCR_RS
}

}  // namespace

// hist[block][digit] of the widest pass
size_t vcr_sort_scratch_bytes(int64_t n) {
    const int64_t nblk = (n + RS_IPB - 1) / RS_IPB;
    return vcr_align(sizeof(uint32_t) * (size_t)(2048 * (nblk > 0 ? nblk : 1)));
}

// Pass plan for `bits` key bits -> number of passes, bits of every pass in `out`.  Digits of at most 9 bits by default (round 4:
// the depth keys are 27 bits wide, vcr_depth_sort -> 3 x 9; 16-17 tile / cell bits -> 2 passes instead of 3; every plan of at
// most 8 bits per pass is unchanged; the A/B against 8 and 11 bits: profiles/r4_sort_digits.txt).  Measured at 1 M keys
// (profiles/r3_sort_ab.txt): 3 x 11 bits 88 us stand-alone / 146 us inside the step against 4 x 8 bits 79 / 137 us -- with
// 2048 digits and 4096 items per workgroup a run of equal digits is 2 items long, so the scatter of the first passes
// degenerates to single stores, and the scan kernel works on 8 KB rows; the wide digits lose more per pass than the
// saved pass returns.
static int rs_plan(int bits, int out[4]) {
    constexpr int max_digit = 9;
    int passes = bits <= 8 ? 1 : (bits <= 16 ? 2 : (bits <= 2 * max_digit ? 2 : (bits <= 3 * max_digit ? 3 : 4)));
    for (int p = 0, left = bits; p < passes; ++p) {
        out[p] = (left + (passes - p) - 1) / (passes - p);
        left -= out[p];
    }
    return passes;
}

int vcr_sort_passes(int bits) { int b[4]; return rs_plan(bits, b); }

// Sorts bits [begin_bit, end_bit) of the keys, stably.  Input: separate arrays (keys_in, vals_in; vals_in == nullptr means
// vals = 0..n-1), or 8-byte (key, value) records (pairs_in, then keys_in / vals_in are ignored).  Output: vals_out and, if
// not NULL, keys_out (separate arrays, distinct from every other buffer).  (pair_a, pair_b): two buffers of n 8-byte records
// for the intermediate passes (pair_b only with >= 3 passes, pair_a with >= 2); the inputs are left untouched.  `hist` holds
// vcr_sort_scratch_bytes(n); `totals`: VCR_SORT_TOTALS_WORDS words (need not be zeroed).  `n` is the host's (upper bound of
// the) element count; `n_dev`, when not NULL, points to the actual count in device memory (<= n).
int vcr_sort_pairs(int64_t n, const uint32_t* keys_in, const uint32_t* vals_in, const uint2* pairs_in, uint2* pair_a, uint2* pair_b,
                   uint32_t* keys_out, uint32_t* vals_out, int begin_bit, int end_bit, uint32_t* hist, uint32_t* totals,
                   hipStream_t st, const uint32_t* n_dev, uint2* pairs_out) {
    if (n <= 0) return 0;
    int bits[4];
    const int passes = rs_plan(end_bit - begin_bit, bits);
    const uint32_t* kin = pairs_in ? reinterpret_cast<const uint32_t*>(pairs_in) : keys_in;
    const uint32_t* vin = pairs_in ? nullptr : vals_in;
    int in = pairs_in ? 2 : (vals_in ? 0 : 1);
    int shift = begin_bit;
    for (int p = 0; p < passes; ++p) {
        const bool last = p == passes - 1;
        // (pairs_out: the last pass leaves its (key, value) records there as well as the values in vals_out; it must not be the
        //  buffer that pass reads)
        uint32_t* kout = last ? (pairs_out ? reinterpret_cast<uint32_t*>(pairs_out) : keys_out)
                              : reinterpret_cast<uint32_t*>((p & 1) ? pair_b : pair_a);
        uint32_t* vout = last ? vals_out : nullptr;
        const int out = last ? (pairs_out ? 2 : 0) : 1;
        if (bits[p] <= 8) rs_pass_any<8>(in, out, n, n_dev, kin, vin, kout, vout, shift, bits[p], hist, totals, st);
        else if (bits[p] <= 9) rs_pass_any<9>(in, out, n, n_dev, kin, vin, kout, vout, shift, bits[p], hist, totals, st);
        else rs_pass_any<11>(in, out, n, n_dev, kin, vin, kout, vout, shift, bits[p], hist, totals, st);
        shift += bits[p];
        kin = kout; vin = nullptr; in = 2;
    }
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}

int vcr_launch_tile_order(int T, const uint2* ranges, uint32_t* order, uint32_t* meta, int64_t instances, bool lpt, bool snake,
                          hipStream_t st, int gxc) {
    // workgroup slots of the compositing kernels on the chip: 5 resident 256-thread workgroups per CU
    static const int slots = [] {
        int dev = 0, cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
        return 5 * cus;
    }();
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, st, T, ranges, order, meta, (unsigned long long)instances, slots,
                       lpt ? 1 : 0, snake ? 1 : 0, gxc);
    VCR_HIP_CHECK(hipGetLastError());
    return 0;
}
