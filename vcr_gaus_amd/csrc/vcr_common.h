// Internal definitions shared by the gfx950 kernels of libvcr_raster.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vcr_raster.h"

#ifndef VCR_T_ANCHOR
#define VCR_T_ANCHOR 0             // 1: experiment build, see composite.hip (transmittance checkpoints)
#endif
#define VCR_TILE 16
#define VCR_TILE_PIX 256
#define VCR_ALPHA_MIN (1.0f / 255.0f)
#define VCR_ALPHA_MAX 0.99f
#define VCR_T_EPS 1e-4f
#define VCR_NEAR 0.2f
#define VCR_LOWPASS 0.3f
#define VCR_PLANE_EPS 1e-4f
#define VCR_MAX_SEM 4
// distortion channel: depth mapped like 2DGS, m = far/(far-near) * (1 - near/d), camera znear/zfar (scene/cameras.py:62-63)
#define VCR_ZNEAR 0.01f
#define VCR_ZFAR 100.0f

// Per-Gaussian screen-space record consumed by the compositing kernels: one 64-byte line, so a
// gather by sorted id touches exactly one half cache line.
struct __attribute__((aligned(16))) GeomRec {
    float px, py, z, opacity;     // q0: pixel-space centre, view-space depth, opacity
    float ca, cb, cc, plane;      // q1: conic (A,B,C), plane offset n . mu_cam
    float r, g, b, pad0;          // q2: colour
    float nx, ny, nz, pad1;       // q3: camera-space normal
};
static_assert(sizeof(GeomRec) == 64, "GeomRec must be one 64-byte line");

// Per-Gaussian screen-space gradient record written by the compositing backward (atomics) and
// consumed by the preprocess backward.
// The first eight slots are RAW sums over pixels of p = dL/dpower (the shading loop is VALU-bound, so the
// per-Gaussian constants are applied once by GradRec::finish() in the consumer): with d = centre - pixel,
//   gx,gy = log2(e) * dL/d(centre); agx,agy = the same with |.| per pixel; ca = sum dx^2 p = -2 dL/dA,
//   cc = sum dy^2 p = -2 dL/dC, cb = sum dx dy p = -dL/dB, opacity = sum p = opacity * dL/dopacity.
struct __attribute__((aligned(16))) GradRec {
    float gx, gy, agx, agy;       // dL/dpx, dL/dpy, sum|dL/dpx|, sum|dL/dpy|
    float ca, cc, cb, opacity;    // dL/dconic (A, C, B), dL/dopacity
    float r, g, b, z;             // dL/drgb, dL/dz (centre depth)
    float plane, nx, ny, nz;      // dL/dplane, dL/dnormal
    __host__ __device__ void finish(float op) {
        const float ln2 = 0.6931471805599453f;
        gx *= ln2; gy *= ln2; agx *= ln2; agy *= ln2;
        ca *= -0.5f; cc *= -0.5f; cb = -cb;
        opacity = op > 0.f ? opacity / op : 0.f;
    }
};
static_assert(sizeof(GradRec) == 64, "GradRec must be 64 bytes");
#define VCR_GRAD_FLOATS 16

static inline size_t vcr_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// ---- state buffer views -------------------------------------------------------------------
struct GeomState {
    GeomRec* rec;        // [N]
    float* sem;          // [N,S]
    uint32_t* tiles;     // [N] tiles touched (0 = culled)
    uint8_t* clamped;    // [N] bit c set when colour channel c was clamped at 0
    uint2* rect;         // [N] tile rectangle, ONE 8-byte gather per Gaussian for the instance emission.  Rectangles of up to
                         //     VCR_RECT_MASK_TILES tiles: {MASKED | xmin | ymin << 10 | (w-1) << 20 | (h-1) << 25, bit k set when
                         //     tile (xmin + k % w, ymin + k / w) can be reached (exact rejection, tile_touch)}; larger ones:
                         //     {xmin | ymin << 10, w | h << 16}, every tile emitted; {0, 0} = culled
    uint32_t* rect_hi;   // [N] quad-list mode: bits 32..63 of the cell mask of rectangles of 33..64 cells (VCR_RECT_MASK64 set)
    float4* cjac;        // [N][3] d(colour channel c) / d(mean), through the normalised view direction (written with the colour by
                         //     whichever kernel evaluates SH -> RGB): the projection backward forms the colour -> mean adjoint
                         //     from these 48 B instead of re-reading the 192 B of SH coefficients
    static size_t bytes(int N, int S) {
        return vcr_align(sizeof(GeomRec) * (size_t)N) + vcr_align(sizeof(float) * (size_t)N * (S > 0 ? S : 1)) +
               vcr_align(sizeof(uint32_t) * (size_t)N) + vcr_align((size_t)N) + vcr_align(sizeof(uint2) * (size_t)N) +
               vcr_align(sizeof(uint32_t) * (size_t)N) + vcr_align(sizeof(float4) * 3 * (size_t)N);
    }
    static GeomState view(void* p, int N, int S) {
        GeomState g;
        char* c = (char*)p;
        g.rec = (GeomRec*)c;      c += vcr_align(sizeof(GeomRec) * (size_t)N);
        g.sem = (float*)c;        c += vcr_align(sizeof(float) * (size_t)N * (S > 0 ? S : 1));
        g.tiles = (uint32_t*)c;   c += vcr_align(sizeof(uint32_t) * (size_t)N);
        g.clamped = (uint8_t*)c;  c += vcr_align((size_t)N);
        g.rect = (uint2*)c;       c += vcr_align(sizeof(uint2) * (size_t)N);
        g.rect_hi = (uint32_t*)c; c += vcr_align(sizeof(uint32_t) * (size_t)N);
        g.cjac = (float4*)c;
        return g;
    }
};

#define VCR_RECT_MASK_TILES 32
#define VCR_RECT_MASKED 0x80000000u
#define VCR_RECT_MASK64 0x40000000u      // (with MASKED) the mask has 64 bits, the upper word in GeomState::rect_hi
#define VCR_MAX_IMAGE_DIM 16384          // tile coordinates have 10 bits in the rectangle record

#define VCR_BIN_META_WORDS 16
// VcrRasterArgs.quad_lists: tile instances are binned per 8x8 QUAD instead of per 16x16 tile -- the projection kernel's exact
// test runs on 8x8 cells, the sort key is the cell index (2 more bits), `ranges` holds one [begin, end) per cell and every
// wave of the compositing kernels walks the list of ITS quad only.  BinState::meta[3] = cells per row when on, 0 when off: the
// compositing kernels (forward and backward) read the mode from there.
#define VCR_SPLIT_MAX 256          // at most this many tiles are split (one band of the launch order)
struct BinState {
    uint2* ranges;         // [4T] per-tile [begin,end) in [0, T); quad-list mode: per 8x8 cell, 4T entries
    uint32_t* tile_order;  // [T] tile ids, longest list first (block scheduling order)
    uint32_t* meta;        // [VCR_BIN_META_WORDS] written by tile_order: [0] number S of heaviest tiles launched as split work
                           //     items (composite.hip), [1] non-empty tiles, [2] longest tile list
    uint32_t* point_list;  // [R'] Gaussian ids, (tile, depth, id)-ordered; LAST, so that the views above do not depend on R'
    // `ckpt`: with the transmittance checkpoints of a training forward behind the lists (composite.hip: ckpt_base) --
    // (R' / 64 + T + 1) records of 256 floats cover both list forms
    static size_t ckpt_floats(int64_t R, int T) { return ((size_t)(R > 0 ? R : 0) / 64 + (size_t)T + 1) * 256; }
    static size_t bytes(int64_t R, int T, bool ckpt = false) {
        return vcr_align(sizeof(uint2) * 4 * (size_t)T) + vcr_align(sizeof(uint32_t) * (size_t)T) +
               vcr_align(sizeof(uint32_t) * VCR_BIN_META_WORDS) + vcr_align(sizeof(uint32_t) * (size_t)(R > 0 ? R : 1)) +
               (ckpt ? vcr_align(sizeof(float) * ckpt_floats(R, T)) : 0);
    }
    static float* ckpt_of(void* p, int64_t R, int T) { return (float*)((char*)p + bytes(R, T, false)); }
    static BinState view(void* p, int T) {
        BinState b;
        char* c = (char*)p;
        b.ranges = (uint2*)c;         c += vcr_align(sizeof(uint2) * 4 * (size_t)T);
        b.tile_order = (uint32_t*)c;  c += vcr_align(sizeof(uint32_t) * (size_t)T);
        b.meta = (uint32_t*)c;        c += vcr_align(sizeof(uint32_t) * VCR_BIN_META_WORDS);
        b.point_list = (uint32_t*)c;
        return b;
    }
};

struct ImageState {
    float* final_T;        // [H*W]
    uint32_t* n_contrib;   // [H*W] index (1-based, within the tile list) of the last contributor
    float* moments;        // [2,H*W] sum w m, sum w m^2 of the mapped depth m (distortion channel, num_dist == 1)
    float* t_ckpt = nullptr;   // per-chunk transmittance checkpoints; lives behind the lists of the BINNING buffer (its size follows R')
    static size_t bytes(int P) {
        return vcr_align(sizeof(float) * (size_t)P) + vcr_align(sizeof(uint32_t) * (size_t)P) +
               vcr_align(2 * sizeof(float) * (size_t)P);
    }
    static ImageState view(void* p, int P) {
        ImageState s;
        char* c = (char*)p;
        s.final_T = (float*)c;      c += vcr_align(sizeof(float) * (size_t)P);
        s.n_contrib = (uint32_t*)c; c += vcr_align(sizeof(uint32_t) * (size_t)P);
        s.moments = (float*)c;
        return s;
    }
};

// ---- SH constants (tools/sh_utils.py:24-52) -------------------------------------------------
#define SH_C0 0.28209479177387814f
#define SH_C1 0.4886025119029199f
#define SH_C2_0 1.0925484305920792f
#define SH_C2_1 -1.0925484305920792f
#define SH_C2_2 0.31539156525252005f
#define SH_C2_3 -1.0925484305920792f
#define SH_C2_4 0.5462742152960396f
#define SH_C3_0 -0.5900435899266435f
#define SH_C3_1 2.890611442640554f
#define SH_C3_2 -0.4570457994644658f
#define SH_C3_3 0.3731763325901154f
#define SH_C3_4 -0.4570457994644658f
#define SH_C3_5 1.445305721320277f
#define SH_C3_6 -0.5900435899266435f

void vcr_set_error(const char* fmt, ...);
#define VCR_HIP_CHECK(expr)                                                                     \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            vcr_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return 1;                                                                           \
        }                                                                                       \
    } while (0)

// ---- stage launchers (defined in the .hip files) ----------------------------------------------
// Counter words of one forward call (LIBRARY-OWNED, zero between calls: the kernels clean up after themselves, no memset launch)
#define VCR_FAR_FLAG_WORD 0        // != 0 when a visible depth key needs more than 27 bits
#define VCR_DONE_WORD 1            // groups of projection workgroups that have all stored their counts (the last one publishes)
#define VCR_DUP_TICKET_WORD 2      // logical block index of the emission kernel (its last block resets it)
// "last workgroup" detection in TWO levels: workgroup b first draws a ticket from group word b % VCR_DONE_GROUPS (<= ~60
// arrivals per address, spread over the kernel's run time), and only the last arrival of a group draws from VCR_DONE_WORD
// (profiles/microbench/ticket_rates.hip: back-to-back atomics on one address cost ~10 ns each)
#define VCR_DONE_GROUPS 64
#define VCR_DONE_GROUP_WORD 64
#define VCR_CTR_WORDS (64 + VCR_DONE_GROUPS)
#define VCR_VIS_SLOTS 1024         // (Readback layout of the debug path)
// What the host polls after the projection: totals + a sequence number published by the device AFTER the totals (system-scope
// fence).  R: tile instances of the 3-sigma rectangles (what the reference counts), E: instances really emitted (exact rejection).
struct VcrPublished { unsigned long long R, E; uint32_t V; uint32_t far; volatile uint32_t seq; };
// `vis_slots`: the counter words above; `blk_counts`: 3 words per workgroup of the projection ((N + 255) / 256 rows, library-owned,
// need not be cleared); `host`: pinned, coherent.  The projection's last workgroup sums the rows and publishes (round 5: neither a
// memset in front of the projection nor a publish kernel behind it).
int vcr_launch_preprocess(const VcrRasterArgs& a, GeomState g, int32_t* radii, uint32_t* depth_key,
                          uint32_t* ids, uint32_t* vis_slots, bool colour, hipStream_t st,
                          uint32_t* blk_counts = nullptr, VcrPublished* host = nullptr, uint32_t seq = 0);
int vcr_launch_depth_keys(const VcrRasterArgs& a, uint32_t* depth_key, hipStream_t st);
int vcr_launch_colour(const VcrRasterArgs& a, GeomState g, hipStream_t st);
int vcr_launch_sh_update_colour(const VcrRasterArgs& a, GeomState g, hipStream_t st);   // a.sh_update + colour in one pass
int vcr_side_grid(int N);     // workgroups of a side-stream kernel (one per CU up to 3 M Gaussians, two above; VCR_SIDE_GRID)
// (`sgrad` / `sgrad_sem` are library-owned accumulators, zero between calls: the kernel clears every record behind its own read)
int vcr_launch_preprocess_backward(const VcrRasterArgs& a, GeomState g, const int32_t* radii,
                                   GradRec* sgrad, float* sgrad_sem, VcrBackwardIO& io,
                                   hipStream_t st);
int vcr_launch_preprocess_backward_tail(const VcrRasterArgs& a, GeomState g, const int32_t* radii, GradRec* sgrad,
                                        float* sgrad_sem, VcrBackwardIO& io, const VcrGeometryStep& t, hipStream_t st);
size_t vcr_binning_temp_bytes(int N, int64_t R, int tile_bits);
size_t vcr_duplicate_status_words(int N);       // look-back words of the emission kernel (library-owned, tagged with the call number)
// Depth keys (round 4): key = bits(z) - bits(VCR_NEAR) of the view-space depth z > VCR_NEAR -- monotone in z, and below
// z = 13 107.2 it fits VCR_DEPTH_KEY_BITS = 27 bits, so the depth order takes THREE 9-bit passes instead of four 8-bit ones
// (culled Gaussians carry 0xFFFFFFFF: last).  A visible Gaussian beyond that depth raises the `far` flag of the projection's
// counters; the host sees it with R / E / V and appends vcr_depth_sort_far -- one more stable pass over the upper five bits of
// the (key, id) records the third pass left in pair_a -- so the order is exact at every depth.
#define VCR_DEPTH_KEY_BITS 27
__host__ __device__ inline uint32_t vcr_depth_key(float z) { return __float_as_uint(z) - 0x3E4CCCCDu; }      // bits(0.2f)
int vcr_depth_sort(int N, const uint32_t* depth_key, uint2* pair_a, uint2* pair_b, uint32_t* ids_sorted, uint32_t* totals,
                   void* temp, hipStream_t st);
int vcr_depth_sort_far(int N, uint2* pair_a, uint32_t* ids_sorted, uint32_t* totals, void* temp, hipStream_t st);
int vcr_duplicate_and_sort(const VcrRasterArgs& a, GeomState g, const int32_t* radii, const uint32_t* ids_sorted,
                           unsigned long long* status, uint32_t* ticket, uint32_t seq, int64_t R /* emitted instances */, int tile_bits, uint2* inst, uint2* pair_a,
                           uint2* pair_b, uint32_t* keys_b, uint32_t* point_list, uint2* ranges, uint32_t* tile_order,
                           uint32_t* meta, int num_tiles, uint32_t* totals, void* temp, size_t temp_bytes, hipStream_t st);
// radix_sort.hip: hand-written stable radix sort of (u32 key, u32 value) pairs and the block-scheduling order
#define VCR_SORT_TOTALS_WORDS 2048            // digit totals of one pass (the widest digit has 11 bits); need not be zeroed
size_t vcr_sort_scratch_bytes(int64_t n);
int vcr_sort_passes(int bits);
int vcr_sort_pairs(int64_t n, const uint32_t* keys_in, const uint32_t* vals_in, const uint2* pairs_in, uint2* pair_a, uint2* pair_b,
                   uint32_t* keys_out, uint32_t* vals_out, int begin_bit, int end_bit, uint32_t* hist, uint32_t* totals,
                   hipStream_t st, const uint32_t* n_dev = nullptr, uint2* pairs_out = nullptr);
// gxc: 8x8 cells per row when `ranges` is per cell (quad-list mode), 0 when it is per tile
int vcr_launch_tile_order(int T, const uint2* ranges, uint32_t* order, uint32_t* meta, int64_t instances, bool lpt, bool snake,
                          hipStream_t st, int gxc = 0);
// two_phase: the form of the f_count = 0 forward (composite.hip); vcr_forward_two_phase decides it from the frame's counts
bool vcr_forward_two_phase(int64_t tile_instances, int64_t visible);
int vcr_launch_composite_forward(const VcrRasterArgs& a, GeomState g, BinState b, ImageState im, VcrForwardOut& o,
                                 hipStream_t st, bool two_phase = false);
int vcr_launch_composite_backward(const VcrRasterArgs& a, GeomState g, BinState b, ImageState im,
                                  const float* dL_dout, GradRec* sgrad, float* sgrad_sem, hipStream_t st);
