// extern "C" entry points of libvcr_raster.so (declared in include/vcr_raster.h).
#include "vcr_common.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <chrono>
#include <vector>

static thread_local char g_err[512] = "";

void vcr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace {

// ---- optional per-stage HIP-event timing (bench.py / profiling only) --------------------------
enum { ST_PREPROCESS = 0, ST_DEPTHSORT, ST_BINNING, ST_COMPOSITE_FWD, ST_COMPOSITE_BWD, ST_PREPROCESS_BWD, ST_COUNT };
struct StageEvt { hipEvent_t a, b; int stage; };
bool g_prof = false;
unsigned g_prof_mask = 0xFFFFFFFFu;     // stages that get events (every event pair costs a few us of stream time)
std::vector<StageEvt> g_used;
std::vector<hipEvent_t> g_free;

hipEvent_t get_event() {
    if (!g_free.empty()) { hipEvent_t e = g_free.back(); g_free.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

struct StageTimer {
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t st;
    int stage;
    StageTimer(int stage_, hipStream_t st_) : st(st_), stage(stage_) {
        if (g_prof && ((g_prof_mask >> stage_) & 1u)) { a = get_event(); b = get_event(); (void)hipEventRecord(a, st); }
    }
    ~StageTimer() {
        if (a) { (void)hipEventRecord(b, st); g_used.push_back({a, b, stage}); }
    }
};

struct Readback { uint32_t V[VCR_VIS_SLOTS]; uint32_t R[VCR_VIS_SLOTS]; };
typedef VcrPublished Published;      // (vcr_common.h: published by the projection's last workgroup)

Published* pinned_published() {
    static thread_local Published* p = nullptr;
    if (!p) {
        // coherent (uncached on the device side) so that the host sees the device's system-scope stores while the stream runs
        if (hipHostMalloc((void**)&p, sizeof(Published), hipHostMallocCoherent | hipHostMallocPortable) != hipSuccess) { p = nullptr; return nullptr; }
        p->R = 0; p->E = 0; p->V = 0; p->far = 0; p->seq = 0;
    }
    return p;
}

// ---- library-owned device scratch (round 5) ------------------------------------------------------------------------------
// Small state that used to be carved out of the caller's per-call scratch and cleared by a memset launch in front of every call:
//   * `ctr`: ticket / flag words (vcr_common.h) -- zero between calls, the kernels clean up after themselves; `blk`: the
//     per-workgroup count rows of the projection (every row of a launch is written before it is read);
//   * `status`: look-back words of the emission kernel, tagged with `seq` instead of being cleared;
//   * `grad`: the backward's per-Gaussian accumulators (GradRec [N] + semantic gradients) -- zero between calls, the projection
//     backward clears every record behind its own read (was a 64 B x N memset per backward).
// One set per (host thread, stream): calls on one stream are ordered, calls on different streams get different sets (a forward
// call returns only after its projection has published, but its emission kernel and a backward run asynchronously).  `dirty`
// flags make the next call clear a block that a failed call may have left half-used.
struct StreamScratch {
    uint32_t* ctr = nullptr; bool ctr_dirty = true;
    uint32_t* blk = nullptr; size_t blk_rows = 0;        // three count words per workgroup of the projection (never cleared)
    unsigned long long* status = nullptr; size_t status_words = 0;
    uint32_t seq = 0;
    char* grad = nullptr; size_t grad_bytes = 0; bool grad_dirty = true;
};
constexpr uint32_t SEQ_MAX = (1u << 30) - 1u;

// Keyed by (device, stream): the NULL stream is handle 0 on every device, so the stream alone would hand a call on cuda:1 the
// blocks that were allocated on cuda:0 (round-5 advisor finding).
struct ScratchKey { int dev; hipStream_t st; };
struct ScratchMap {
    std::vector<std::pair<ScratchKey, StreamScratch*>> sets;
    void release() {                 // hipFree synchronises the device
        for (auto& kv : sets) {
            int cur = 0;
            (void)hipGetDevice(&cur);
            if (cur != kv.first.dev) (void)hipSetDevice(kv.first.dev);
            (void)hipFree(kv.second->ctr); (void)hipFree(kv.second->blk); (void)hipFree(kv.second->status); (void)hipFree(kv.second->grad);
            if (cur != kv.first.dev) (void)hipSetDevice(cur);
            delete kv.second;
        }
        sets.clear();
    }
    ~ScratchMap() {}                 // (thread / process exit: the runtime may already be gone -- callers that want the memory back
                                     //  call vcr_release_scratch() from the thread that made the calls)
};
ScratchMap& scratch_map() { static thread_local ScratchMap m; return m; }

StreamScratch* stream_scratch(hipStream_t st) {
    ScratchMap& m = scratch_map();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    for (auto& kv : m.sets) if (kv.first.st == st && kv.first.dev == dev) return kv.second;
    if (m.sets.size() >= 64) m.release();       // a caller cycling through streams: start over
    StreamScratch* sc = new StreamScratch();
    if (hipMalloc((void**)&sc->ctr, sizeof(uint32_t) * VCR_CTR_WORDS) != hipSuccess) { delete sc; return nullptr; }
    m.sets.emplace_back(ScratchKey{dev, st}, sc);
    return sc;
}

// counter block ready (zero) + status words for `words` look-back entries; advances the call number.  Stream-ordered.
int scratch_begin_forward(StreamScratch* sc, size_t words, int N, hipStream_t st) {
    if (sc->ctr_dirty) { VCR_HIP_CHECK(hipMemsetAsync(sc->ctr, 0, sizeof(uint32_t) * VCR_CTR_WORDS, st)); }
    sc->ctr_dirty = true;            // until this call's projection has published (it then left the block zero)
    const size_t rows = (size_t)(N + 255) / 256 + 1;
    if (rows > sc->blk_rows) {
        if (sc->blk) VCR_HIP_CHECK(hipFree(sc->blk));
        sc->blk = nullptr; sc->blk_rows = 0;
        const size_t cap = rows + rows / 4 + 64;
        VCR_HIP_CHECK(hipMalloc((void**)&sc->blk, sizeof(uint32_t) * 3 * cap));
        sc->blk_rows = cap;
    }
    if (words > sc->status_words) {
        if (sc->status) VCR_HIP_CHECK(hipFree(sc->status));
        sc->status = nullptr; sc->status_words = 0;
        const size_t cap = words + words / 4 + 64;
        VCR_HIP_CHECK(hipMalloc((void**)&sc->status, sizeof(unsigned long long) * cap));
        VCR_HIP_CHECK(hipMemsetAsync(sc->status, 0, sizeof(unsigned long long) * cap, st));
        sc->status_words = cap;
        sc->seq = 0;
    }
    if (sc->seq >= SEQ_MAX) {        // the 30-bit call number wraps: clear the tags once
        VCR_HIP_CHECK(hipMemsetAsync(sc->status, 0, sizeof(unsigned long long) * sc->status_words, st));
        sc->seq = 0;
    }
    ++sc->seq;
    return 0;
}

// accumulators of `bytes` bytes, zero.  Stream-ordered.
int scratch_begin_backward(StreamScratch* sc, size_t bytes, hipStream_t st) {
    if (bytes > sc->grad_bytes) {
        if (sc->grad) VCR_HIP_CHECK(hipFree(sc->grad));
        sc->grad = nullptr; sc->grad_bytes = 0;
        const size_t cap = vcr_align(bytes + bytes / 4);
        VCR_HIP_CHECK(hipMalloc((void**)&sc->grad, cap));
        sc->grad_bytes = cap;
        sc->grad_dirty = true;
    }
    if (sc->grad_dirty) { VCR_HIP_CHECK(hipMemsetAsync(sc->grad, 0, sc->grad_bytes, st)); }
    sc->grad_dirty = true;           // until both kernels of this backward have been accepted
    return 0;
}

Readback* pinned_readback() {
    static thread_local Readback* p = nullptr;
    if (!p && hipHostMalloc((void**)&p, sizeof(Readback), hipHostMallocDefault) != hipSuccess) p = nullptr;
    return p;
}

// events of the optional streams: [0] geometry ready -> [1] colours ready (VcrRasterArgs.colour_stream);
// [2] inputs ready -> [3] depth order ready (VcrRasterArgs.sort_stream)
hipEvent_t colour_event(int k) {
    static thread_local hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr};
    if (!e[k] && hipEventCreateWithFlags(&e[k], hipEventDisableTiming) != hipSuccess) e[k] = nullptr;
    return e[k];
}

__global__ void fill_background_kernel(int P, int C, const float* __restrict__ bg, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    for (int c = 0; c < C; ++c) out[(size_t)c * P + i] = c < 3 ? bg[c] : 0.f;
}

__global__ void max_tile_len_kernel(int T, const uint2* __restrict__ ranges, uint32_t* __restrict__ out) {
    uint32_t m = 0;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < T; t += gridDim.x * 256) m = max(m, ranges[t].y - ranges[t].x);
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

int validate(const VcrRasterArgs* a) {
    if (!a) { vcr_set_error("args is NULL"); return 1; }
    if (a->N < 0 || a->H <= 0 || a->W <= 0) { vcr_set_error("bad sizes N=%d H=%d W=%d", a->N, a->H, a->W); return 1; }
    if (a->H > VCR_MAX_IMAGE_DIM || a->W > VCR_MAX_IMAGE_DIM) { vcr_set_error("image %dx%d exceeds %d pixels per side", a->W, a->H, VCR_MAX_IMAGE_DIM); return 1; }
    if (a->quad_lists && (a->H > VCR_MAX_IMAGE_DIM / 2 || a->W > VCR_MAX_IMAGE_DIM / 2)) {      // (cell coordinates have 10 bits)
        vcr_set_error("image %dx%d exceeds %d pixels per side (quad-list mode)", a->W, a->H, VCR_MAX_IMAGE_DIM / 2); return 1;
    }
    if (a->S < 0 || a->S > VCR_MAX_SEM) { vcr_set_error("semantic channels S=%d unsupported (0..%d)", a->S, VCR_MAX_SEM); return 1; }
    if (a->num_dist < 0 || a->num_dist > 2) {
        vcr_set_error("num_dist=%d unsupported: 0 (none), 1 (depth distortion) or 2 (depth moments sum w d, sum w d^2)",
                      a->num_dist);
        return 1;
    }
    if (a->f_count < 0 || a->f_count > 4) { vcr_set_error("f_count=%d unsupported (0..4)", a->f_count); return 1; }
    if (a->num_dist != 0 && a->f_count != 0) { vcr_set_error("num_dist needs f_count=0"); return 1; }
    if (a->forward_form < 0 || a->forward_form > 2) { vcr_set_error("forward_form=%d unsupported (0 automatic, 1 uniform loop, 2 two-phase)", a->forward_form); return 1; }
    if (a->N == 0) {                            // empty model: data pointers may legitimately be NULL
        if (!a->bg) { vcr_set_error("bg is NULL"); return 1; }
        return 0;
    }
    if ((a->shs == nullptr) == (a->colors_precomp == nullptr)) { vcr_set_error("provide exactly one of shs / colors_precomp"); return 1; }
    const bool sr = a->scales != nullptr && a->rotations != nullptr;
    if (sr == (a->cov3D_precomp != nullptr)) { vcr_set_error("provide exactly one of (scales, rotations) / cov3D_precomp"); return 1; }
    if (a->shs && (a->sh_degree < 0 || a->sh_degree > 3 || (a->sh_degree + 1) * (a->sh_degree + 1) > a->K)) {
        vcr_set_error("sh_degree=%d incompatible with K=%d", a->sh_degree, a->K); return 1;
    }
    if (a->shs_rest && (!a->shs || a->K != 16)) { vcr_set_error("shs_rest needs shs (DC) and K=16"); return 1; }
    if (a->S > 0 && !a->semantics_precomp) { vcr_set_error("S>0 but semantics_precomp is NULL"); return 1; }
    if (!a->bg || !a->viewmatrix || !a->projmatrix || !a->campos || !a->means3D || !a->opacities) {
        vcr_set_error("required pointer is NULL"); return 1;
    }
    return 0;
}


int tile_bits_for(int T) {
    int b = 1;
    while ((1 << b) < T) ++b;
    return b;
}

}  // namespace

extern "C" int vcr_abi_version(void) { return VCR_ABI_VERSION; }
extern "C" const char* vcr_last_error(void) { return g_err; }

extern "C" int vcr_sort_pairs_u32(int64_t n, const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out,
                                  uint32_t* vals_out, int begin_bit, int end_bit, void* scratch, size_t scratch_bytes,
                                  void* stream) {
    if (n <= 0) return 0;
    if (!keys_in || !keys_out || !vals_out || begin_bit < 0 || end_bit > 32 || begin_bit >= end_bit) {
        vcr_set_error("vcr_sort_pairs_u32: bad arguments"); return 1;
    }
    const size_t pb = vcr_align(sizeof(uint2) * (size_t)n), tot = vcr_align(sizeof(uint32_t) * VCR_SORT_TOTALS_WORDS);
    const size_t need = 2 * pb + tot + vcr_sort_scratch_bytes(n);
    if (!scratch || scratch_bytes < need) { vcr_set_error("vcr_sort_pairs_u32: scratch too small (%zu < %zu)", scratch_bytes, need); return 1; }
    char* s = (char*)scratch;
    hipStream_t st = (hipStream_t)stream;
    return vcr_sort_pairs(n, keys_in, vals_in, nullptr, (uint2*)s, (uint2*)(s + pb), keys_out, vals_out, begin_bit, end_bit,
                          (uint32_t*)(s + 2 * pb + tot), (uint32_t*)(s + 2 * pb), st);
}

extern "C" size_t vcr_sort_pairs_u32_scratch_bytes(int64_t n) {
    if (n <= 0) return 0;
    return 2 * vcr_align(sizeof(uint2) * (size_t)n) + vcr_align(sizeof(uint32_t) * VCR_SORT_TOTALS_WORDS) + vcr_sort_scratch_bytes(n);
}

extern "C" int vcr_rasterize_forward(const VcrRasterArgs* args, VcrForwardOut* out, vcr_alloc_fn alloc, void* user,
                                     void* stream) {
    if (validate(args)) return 1;
    if (!out || !alloc) { vcr_set_error("out/alloc is NULL"); return 1; }
    const VcrRasterArgs& a = *args;
    hipStream_t st = (hipStream_t)stream;
    const int N = a.N, P = a.H * a.W;
    const int gx = (a.W + VCR_TILE - 1) / VCR_TILE, gy = (a.H + VCR_TILE - 1) / VCR_TILE;
    const int T = gx * gy;
    const int C = 8 + a.S + a.num_dist;
    if (a.f_count != 3 && a.f_count != 4 && !out->out) { vcr_set_error("out buffer is NULL"); return 1; }
    if (a.f_count != 0 && !out->count) { vcr_set_error("count buffer is NULL for f_count=%d", a.f_count); return 1; }
    if ((a.f_count == 1 || a.f_count == 2) && !out->score) { vcr_set_error("score buffer is NULL"); return 1; }
    out->num_rendered = 0; out->num_visible = 0; out->max_tile_len = -1; out->num_emitted = -1;
    out->geom = out->binning = out->image = nullptr;

    void* geom_p = alloc(user, VCR_BUF_GEOM, GeomState::bytes(N > 0 ? N : 1, a.S));
    void* img_p = alloc(user, VCR_BUF_IMAGE, ImageState::bytes(P));
    if (!geom_p || !img_p) { vcr_set_error("allocator returned NULL"); return 1; }
    GeomState g = GeomState::view(geom_p, N > 0 ? N : 1, a.S);
    ImageState im = ImageState::view(img_p, P);
    out->geom = geom_p; out->image = img_p;

    int64_t R = 0;
    if (N > 0) {
        if (!out->radii) { vcr_set_error("radii buffer is NULL"); return 1; }
        const int tbits = tile_bits_for(T) + (a.quad_lists ? 2 : 0);      // (quad lists: the key is the 8x8 cell)
        const size_t tmp1 = vcr_binning_temp_bytes(N, 0, tbits);
        const size_t nb = vcr_align(sizeof(uint32_t) * (size_t)N);
        const size_t tot_bytes = vcr_align(sizeof(uint32_t) * 2 * VCR_SORT_TOTALS_WORDS);   // digit totals of the two sorts (not zeroed)
        // [depth keys | depth order | two buffers of 8-byte (key, id) records for the sort's passes | digit totals | sort scratch]
        char* s1 = (char*)alloc(user, VCR_BUF_SCRATCH, 6 * nb + tot_bytes + tmp1);
        if (!s1) { vcr_set_error("allocator returned NULL"); return 1; }
        StreamScratch* sc = stream_scratch(st);          // counters + look-back words: library-owned, no memset launch (round 5)
        if (!sc) { vcr_set_error("hipMalloc for the counter block failed"); return 1; }
        if (scratch_begin_forward(sc, vcr_duplicate_status_words(N), N, st)) return 1;
        uint32_t* depth_key = (uint32_t*)s1;
        uint32_t* ids_sorted = (uint32_t*)(s1 + nb);
        uint2* pair_a = (uint2*)(s1 + 2 * nb);
        uint2* pair_b = (uint2*)(s1 + 4 * nb);
        uint32_t* ids = nullptr;                               // (unused by the projection kernel)
        unsigned long long* dup_status = sc->status;
        uint32_t* vis_counter = sc->ctr;
        uint32_t* totals_depth = (uint32_t*)(s1 + 6 * nb);
        uint32_t* totals_tile = totals_depth + VCR_SORT_TOTALS_WORDS;
        void* temp1 = s1 + 6 * nb + tot_bytes;
        // Work launched on the optional streams must be joined on EVERY exit (the scratch buffers go back to the caller's
        // stream-ordered allocator when this call returns): error returns go through join_streams().
        bool sort_launched = false, colour_launched = false;
        auto join_streams = [&]() {
            if (sort_launched) (void)hipStreamWaitEvent(st, colour_event(3), 0);
            if (colour_launched) (void)hipStreamWaitEvent(st, colour_event(1), 0);
            return 1;
        };
        // like VCR_HIP_CHECK, but the error return first makes `st` wait for whatever already runs on the optional streams
#define VCR_HIP_CHECK_JOIN(expr)                                                                        \
        do {                                                                                            \
            hipError_t _e = (expr);                                                                     \
            if (_e != hipSuccess) {                                                                     \
                vcr_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
                return join_streams();                                                                  \
            }                                                                                           \
        } while (0)
        // optional sort stream: depth keys + depth sort of the N Gaussians start now, beside the projection
        const bool split_sort = a.sort_stream && a.sort_stream != stream;
        if (split_sort) {
            hipEvent_t e_in = colour_event(2), e_sorted = colour_event(3);
            if (!e_in || !e_sorted) { vcr_set_error("hipEventCreate for the sort stream failed"); return 1; }
            hipStream_t ss = (hipStream_t)a.sort_stream;
            VCR_HIP_CHECK(hipEventRecord(e_in, st));
            VCR_HIP_CHECK(hipStreamWaitEvent(ss, e_in, 0));
            int rc = vcr_launch_depth_keys(a, depth_key, ss);
            if (!rc) {
                StageTimer tm(ST_DEPTHSORT, ss);
                rc = vcr_depth_sort(N, depth_key, pair_a, pair_b, ids_sorted, totals_depth, temp1, ss);
            }
            sort_launched = true;                           // (from here on every exit joins the sort stream)
            if (hipEventRecord(e_sorted, ss) != hipSuccess) {  // no event to wait on: drain the stream instead
                (void)hipStreamSynchronize(ss);
                sort_launched = false;
                vcr_set_error("hipEventRecord on the sort stream failed");
                return join_streams();
            }
            if (rc) return join_streams();
        }
        // two-stream form: geometry here, SH -> RGB on the colour stream behind whatever the caller queued there
        const bool split_colour = a.colour_stream && a.colour_stream != stream && a.shs && !a.colors_precomp;
        if (a.sh_update && !split_colour) { vcr_set_error("sh_update needs colour_stream and SH colours"); return join_streams(); }
        // R, E and V go back to the host from the projection's last workgroup (no publish kernel, round 5); the depth sort and
        // the offsets scan do not need them and keep the GPU busy while the host wakes up, sizes the instance buffers and
        // enqueues the rest
        Readback* rb = pinned_readback();
        Published* pub = pinned_published();
        if (!rb || !pub) { vcr_set_error("hipHostMalloc for the readback failed"); return join_streams(); }
        static thread_local uint32_t seq_counter = 0;
        const uint32_t seq = ++seq_counter ? seq_counter : ++seq_counter;      // never 0
        {
            StageTimer tm(ST_PREPROCESS, st);
            // (count-only modes 3 / 4 never read a colour: geometry-only projection, 36 against 105 us at 1 M Gaussians)
            const bool colour_here = !split_colour && a.f_count != 3 && a.f_count != 4;
            if (vcr_launch_preprocess(a, g, out->radii, split_sort ? nullptr : depth_key, ids, vis_counter, colour_here, st, sc->blk, pub, seq))
                return join_streams();
        }
        if (split_colour) {
            hipEvent_t e_geo = colour_event(0), e_col = colour_event(1);
            if (!e_geo || !e_col) { vcr_set_error("hipEventCreate for the colour stream failed"); return join_streams(); }
            hipStream_t cs = (hipStream_t)a.colour_stream;
            VCR_HIP_CHECK_JOIN(hipEventRecord(e_geo, st));
            VCR_HIP_CHECK_JOIN(hipStreamWaitEvent(cs, e_geo, 0));
            if (a.colour_stream_hook) a.colour_stream_hook(a.colour_stream_hook_user);
            const int rc = a.sh_update ? vcr_launch_sh_update_colour(a, g, cs) : vcr_launch_colour(a, g, cs);
            colour_launched = true;
            if (hipEventRecord(e_col, cs) != hipSuccess) {
                (void)hipStreamSynchronize(cs);
                colour_launched = false;
                vcr_set_error("hipEventRecord on the colour stream failed");
                return join_streams();
            }
            if (rc) return join_streams();
        }
        // (no event behind the projection: a marker packet between it and the depth sort costs the main stream ~10 us per
        //  step; the slow path of the hand-over below drains the stream instead)
        if (!split_sort) {
            StageTimer tm(ST_DEPTHSORT, st);
            if (vcr_depth_sort(N, depth_key, pair_a, pair_b, ids_sorted, totals_depth, temp1, st)) return join_streams();
        }
        // From here on the colour stream may already be running work that consumed the caller's pending SH update: every
        // error return below first joins it (the caller's retry / error handling must not see an un-joined stream).
        auto fail_joined = join_streams;
        // instance-count dependent buffers, requested NOW for the previous call's count + 1/8 (see below)
        static thread_local int64_t e_guess = 0;
        const bool with_ckpt = VCR_T_ANCHOR && a.f_count == 0;   // (anchored builds: per-chunk transmittance checkpoints for the backward)
        const bool third = vcr_sort_passes(tbits) > 2;      // (more than 16 tile bits: a second intermediate buffer)
        int64_t cap = (e_guess > 0 && !with_ckpt) ? e_guess + e_guess / 8 + 4096 : 0;
        void* bin_p = nullptr;
        char* s2 = nullptr;      // [emitted (tile, id) records | records of the sort's first pass | sorted tile keys | (records of a third pass)]
        if (cap > 0) {
            bin_p = alloc(user, VCR_BUF_BINNING, BinState::bytes(cap, T, with_ckpt));
            s2 = bin_p ? (char*)alloc(user, VCR_BUF_SCRATCH, (third ? 7 : 5) * vcr_align(sizeof(uint32_t) * (size_t)cap) +
                                                             vcr_binning_temp_bytes(N, cap, tbits)) : nullptr;
            if (!bin_p || !s2) { vcr_set_error("allocator returned NULL"); return fail_joined(); }
        }
        {   // spin on the published sequence number for at most ~2 ms of wall time, then sleep in a stream synchronisation (which
            // also surfaces a device fault or a failed launch as an error instead of a hang)
            const auto t_spin = std::chrono::steady_clock::now();
            unsigned spins = 0;
            while (pub->seq != seq) {
#if defined(__x86_64__) || defined(__i386__)
                __builtin_ia32_pause();
#endif
                if ((++spins & 1023u) == 0 &&
                    std::chrono::steady_clock::now() - t_spin > std::chrono::milliseconds(2)) {
                    const hipError_t e = hipStreamSynchronize(st);
                    if (e != hipSuccess) { vcr_set_error("instance-count hand-over: %s", hipGetErrorString(e)); return fail_joined(); }
                    break;
                }
            }
            if (pub->seq != seq) { vcr_set_error("device did not publish the instance count"); return fail_joined(); }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
        sc->ctr_dirty = false;                              // (published: the last workgroup has left the counter block zero)
        R = (int64_t)pub->R;
        const int64_t E = (int64_t)pub->E;                  // what the emission kernel will write: sizes everything below
        if (pub->far) {          // a visible Gaussian beyond the 27-bit key range: one more pass over the upper key bits, behind the sort
            hipStream_t ds = split_sort ? (hipStream_t)a.sort_stream : st;
            StageTimer tm(ST_DEPTHSORT, ds);
            if (vcr_depth_sort_far(N, pair_a, ids_sorted, totals_depth, temp1, ds)) return fail_joined();
            if (split_sort) VCR_HIP_CHECK_JOIN(hipEventRecord(colour_event(3), ds));
        }
        out->num_visible = (int32_t)pub->V;
        out->num_emitted = E;
        if (R >= (1ll << 32)) { vcr_set_error("more than 2^32 tile instances"); return fail_joined(); }

        // (the instance-count dependent buffers were requested BEFORE the hand-over, sized by the previous call's count: the
        //  two allocator callbacks -- Python, ~8 us each -- then run while the GPU sorts, not while it waits for the emission
        //  kernel; only a count beyond that guess asks again)
        if (!bin_p || E > cap) {
            cap = E;
            bin_p = alloc(user, VCR_BUF_BINNING, BinState::bytes(cap, T, with_ckpt));
            s2 = bin_p ? (char*)alloc(user, VCR_BUF_SCRATCH, (third ? 7 : 5) * vcr_align(sizeof(uint32_t) * (size_t)(cap > 0 ? cap : 1)) +
                                                             vcr_binning_temp_bytes(N, cap, tbits)) : nullptr;
            if (!bin_p || !s2) { vcr_set_error("allocator returned NULL"); return fail_joined(); }
        }
        e_guess = E;
        BinState b = BinState::view(bin_p, T);
        out->binning = bin_p;
        im.t_ckpt = with_ckpt ? BinState::ckpt_of(bin_p, cap, T) : nullptr;
        const size_t tmp2 = vcr_binning_temp_bytes(N, cap, tbits);
        const size_t rbts = vcr_align(sizeof(uint32_t) * (size_t)(cap > 0 ? cap : 1));
        if (split_sort) VCR_HIP_CHECK_JOIN(hipStreamWaitEvent(st, colour_event(3), 0));  // the depth order is needed from here on
        {
            StageTimer tm(ST_BINNING, st);
            if (vcr_duplicate_and_sort(a, g, out->radii, ids_sorted, dup_status, sc->ctr + VCR_DUP_TICKET_WORD, sc->seq, E, tbits,
                                       (uint2*)s2, (uint2*)(s2 + 2 * rbts),
                                       third ? (uint2*)(s2 + 5 * rbts) : nullptr, (uint32_t*)(s2 + 4 * rbts),
                                       b.point_list, b.ranges, b.tile_order, b.meta, T, totals_tile, s2 + (third ? 7 : 5) * rbts, tmp2, st))
                return fail_joined();
        }
        out->num_rendered = R;
        if (a.debug) {                       // diagnostics only: longest per-tile list (one extra sync)
            uint32_t* mx = sc->blk;                  // (the projection has published: its count rows are free)
            VCR_HIP_CHECK_JOIN(hipMemsetAsync(mx, 0, sizeof(uint32_t), st));
            hipLaunchKernelGGL(max_tile_len_kernel, dim3(32), dim3(256), 0, st, a.quad_lists ? 4 * T : T, b.ranges, mx);
            VCR_HIP_CHECK_JOIN(hipMemcpyAsync(&rb->R[0], mx, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            VCR_HIP_CHECK_JOIN(hipStreamSynchronize(st));
            out->max_tile_len = (int32_t)rb->R[0];
        }
        if (split_colour) VCR_HIP_CHECK_JOIN(hipStreamWaitEvent(st, colour_event(1), 0));
        {
            StageTimer tm(ST_COMPOSITE_FWD, st);
            if (vcr_launch_composite_forward(a, g, b, im, *out, st, a.forward_form == 2 || (a.forward_form == 0 && vcr_forward_two_phase(R, (int64_t)out->num_visible)))) return fail_joined();
        }
#undef VCR_HIP_CHECK_JOIN
    } else {
        void* bin_p = alloc(user, VCR_BUF_BINNING, BinState::bytes(0, T));
        if (!bin_p) { vcr_set_error("allocator returned NULL"); return 1; }
        BinState b = BinState::view(bin_p, T);
        out->binning = bin_p;
        out->num_emitted = 0;
        VCR_HIP_CHECK(hipMemsetAsync(b.ranges, 0, sizeof(uint2) * 4 * (size_t)T, st));
        if (vcr_launch_tile_order(T, b.ranges, b.tile_order, b.meta, 0, false, false, st,
                                  a.quad_lists ? 2 * ((a.W + VCR_TILE - 1) / VCR_TILE) : 0)) return 1;   // identity order
        VCR_HIP_CHECK(hipMemsetAsync(img_p, 0, ImageState::bytes(P), st));
        if (a.f_count != 3 && a.f_count != 4)
            hipLaunchKernelGGL(fill_background_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, a.f_count ? 3 : C, a.bg,
                               out->out);
        VCR_HIP_CHECK(hipGetLastError());
    }
    return 0;
}

// ---- batched visibility passes ---------------------------------------------------------------------------------------
namespace {

constexpr int VIS_STREAMS = 8, VIS_MAX_SETS = 16;

struct VisPool {
    hipStream_t st[VIS_STREAMS] = {};
    hipEvent_t fork = nullptr, join[VIS_STREAMS] = {};
    Published* pub = nullptr;                       // VIS_MAX_SETS pinned slots
    StreamScratch sc[VIS_MAX_SETS];                 // counter block + look-back words of every buffer set (library-owned)
    bool ok = false;
};

VisPool* vis_pool() {                 // one pool per (host thread, device): streams, events and counter blocks belong to a device
    constexpr int MAX_DEV = 16;
    static thread_local VisPool pools[MAX_DEV];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
    VisPool& p = pools[dev];
    if (!p.ok) {
        for (int k = 0; k < VIS_STREAMS; ++k) {
            if (hipStreamCreateWithFlags(&p.st[k], hipStreamNonBlocking) != hipSuccess) return nullptr;
            if (hipEventCreateWithFlags(&p.join[k], hipEventDisableTiming) != hipSuccess) return nullptr;
        }
        if (hipEventCreateWithFlags(&p.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
        if (hipHostMalloc((void**)&p.pub, sizeof(Published) * VIS_MAX_SETS, hipHostMallocCoherent | hipHostMallocPortable) != hipSuccess) return nullptr;
        memset(p.pub, 0, sizeof(Published) * VIS_MAX_SETS);
        for (int k = 0; k < VIS_MAX_SETS; ++k)
            if (hipMalloc((void**)&p.sc[k].ctr, sizeof(uint32_t) * VCR_CTR_WORDS) != hipSuccess) return nullptr;
        p.ok = true;
    }
    return &p;
}

// host side of the publish / spin hand-over (see vcr_rasterize_forward); `st`: the stream the publishing kernel runs on
int wait_published(volatile Published* pub, uint32_t seq, hipStream_t st) {
    const auto t_spin = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (pub->seq != seq) {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
        if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - t_spin > std::chrono::milliseconds(20)) {
            const hipError_t e = hipStreamSynchronize(st);
            if (e != hipSuccess) { vcr_set_error("visibility batch: %s", hipGetErrorString(e)); return 1; }
            break;
        }
    }
    if (pub->seq != seq) { vcr_set_error("device did not publish the instance count"); return 1; }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return 0;
}

struct VisSet {                   // buffers of one camera in flight; re-used by camera c + sets on the same stream
    void* geom = nullptr;
    char* s1 = nullptr;
    int32_t* radii = nullptr;
    void* bin = nullptr; char* s2 = nullptr; int64_t cap = -1;      // instance-count dependent: grown on demand
    uint32_t seq = 0, dseq = 0;                                     // publish number; tag of the emission kernel's look-back words
};

}  // namespace

extern "C" int vcr_visibility_batch(const VcrVisibilityBatch* vb, vcr_alloc_fn alloc, void* user, void* stream) {
    if (!vb || !alloc) { vcr_set_error("visibility batch: args/alloc is NULL"); return 1; }
    const int N = vb->N, B = vb->B;
    const int max_dim = vb->quad_lists ? VCR_MAX_IMAGE_DIM / 2 : VCR_MAX_IMAGE_DIM;
    if (N < 0 || B < 0 || vb->H <= 0 || vb->W <= 0 || vb->H > max_dim || vb->W > max_dim) {
        vcr_set_error("visibility batch: bad sizes N=%d B=%d H=%d W=%d", N, B, vb->H, vb->W); return 1;
    }
    if (N == 0 || B == 0) return 0;
    const bool sr = vb->scales != nullptr && vb->rotations != nullptr;
    if (sr == (vb->cov3D_precomp != nullptr)) { vcr_set_error("visibility batch: provide exactly one of (scales, rotations) / cov3D_precomp"); return 1; }
    if (!vb->tanfovx || !vb->tanfovy || !vb->viewmatrix || !vb->projmatrix || !vb->campos || !vb->means3D || !vb->opacities || !vb->count) {
        vcr_set_error("visibility batch: required pointer is NULL"); return 1;
    }
    VisPool* pool = vis_pool();
    if (!pool) { vcr_set_error("visibility batch: stream / event / pinned-memory set-up failed"); return 1; }
    hipStream_t st = (hipStream_t)stream;
    const int gx = (vb->W + VCR_TILE - 1) / VCR_TILE, gy = (vb->H + VCR_TILE - 1) / VCR_TILE, T = gx * gy;
    const int tbits = tile_bits_for(T) + (vb->quad_lists ? 2 : 0);
    int sets = vb->inflight > 0 ? vb->inflight : 8;                  // cameras in flight = buffer sets
    sets = sets > VIS_MAX_SETS ? VIS_MAX_SETS : sets;
    if (sets > B) sets = B;
    // camera c uses set (c mod sets) on stream (c mod sets) mod nstreams: camera c and c + sets share a buffer set AND a stream
    const int nstreams = sets < VIS_STREAMS ? sets : VIS_STREAMS;

    // per-set buffer layout: as in vcr_rasterize_forward (no image state: the count modes do not write one)
    const size_t tmp1 = vcr_binning_temp_bytes(N, 0, tbits);
    const size_t nb = vcr_align(sizeof(uint32_t) * (size_t)N);
    const size_t tot_bytes = vcr_align(sizeof(uint32_t) * 2 * VCR_SORT_TOTALS_WORDS);
    const size_t s1_bytes = 7 * nb + tot_bytes + tmp1;           // (+ nb: the radii of this camera)
    VisSet vs[VIS_MAX_SETS];
    for (int k = 0; k < sets; ++k) {
        vs[k].geom = alloc(user, VCR_BUF_SCRATCH, GeomState::bytes(N, 0));
        vs[k].s1 = (char*)alloc(user, VCR_BUF_SCRATCH, s1_bytes);
        if (!vs[k].geom || !vs[k].s1) { vcr_set_error("allocator returned NULL"); return 1; }
        vs[k].radii = (int32_t*)(vs[k].s1 + 6 * nb);
    }
    static thread_local uint32_t seq_counter = 0x40000000u;

    VCR_HIP_CHECK(hipEventRecord(pool->fork, st));
    for (int k = 0; k < nstreams; ++k) VCR_HIP_CHECK(hipStreamWaitEvent(pool->st[k], pool->fork, 0));
    int rc = 0;
    auto join_all = [&]() {
        for (int k = 0; k < nstreams; ++k) {
            if (hipEventRecord(pool->join[k], pool->st[k]) == hipSuccess) (void)hipStreamWaitEvent(st, pool->join[k], 0);
            else (void)hipStreamSynchronize(pool->st[k]);
        }
    };
    auto cam_args = [&](int c) {
        VcrRasterArgs a;
        memset(&a, 0, sizeof(a));
        a.N = N; a.H = vb->H; a.W = vb->W; a.f_count = vb->flags_only ? 4 : 3; a.quad_lists = vb->quad_lists ? 1 : 0;
        a.tanfovx = vb->tanfovx[c]; a.tanfovy = vb->tanfovy[c]; a.scale_modifier = vb->scale_modifier;
        a.viewmatrix = vb->viewmatrix + 16 * (size_t)c; a.projmatrix = vb->projmatrix + 16 * (size_t)c;
        a.campos = vb->campos + 3 * (size_t)c;
        a.means3D = vb->means3D; a.opacities = vb->opacities; a.scales = vb->scales; a.rotations = vb->rotations;
        a.cov3D_precomp = vb->cov3D_precomp;
        return a;
    };
    for (int c0 = 0; c0 < B && !rc; c0 += sets) {
        const int c1 = c0 + sets < B ? c0 + sets : B;
        // front half of every camera of the group: projection (geometry only), counts to the host, depth order
        for (int c = c0; c < c1 && !rc; ++c) {
            VisSet& v = vs[c - c0];
            hipStream_t cs = pool->st[(c - c0) % nstreams];
            const VcrRasterArgs a = cam_args(c);
            GeomState g = GeomState::view(v.geom, N, 0);
            uint32_t* depth_key = (uint32_t*)v.s1;
            uint32_t* ids_sorted = (uint32_t*)(v.s1 + nb);
            StreamScratch& sc = pool->sc[c - c0];
            uint32_t* totals_depth = (uint32_t*)(v.s1 + 7 * nb);
            void* temp1 = v.s1 + 7 * nb + tot_bytes;
            if (scratch_begin_forward(&sc, vcr_duplicate_status_words(N), N, cs)) { rc = 1; break; }
            v.dseq = sc.seq;
            v.seq = ++seq_counter ? seq_counter : ++seq_counter;
            if (vcr_launch_preprocess(a, g, v.radii, depth_key, nullptr, sc.ctr, false, cs, sc.blk, pool->pub + (c - c0), v.seq)) { rc = 1; break; }
            if (vcr_depth_sort(N, depth_key, (uint2*)(v.s1 + 2 * nb), (uint2*)(v.s1 + 4 * nb), ids_sorted, totals_depth, temp1, cs)) { rc = 1; break; }
        }
        // back half: the host sizes the instance buffers of camera c while the later cameras' front halves run
        for (int c = c0; c < c1 && !rc; ++c) {
            VisSet& v = vs[c - c0];
            hipStream_t cs = pool->st[(c - c0) % nstreams];
            Published* pub = pool->pub + (c - c0);
            if (wait_published(pub, v.seq, cs)) { rc = 1; break; }
            StreamScratch& sc = pool->sc[c - c0];
            sc.ctr_dirty = false;                   // (published: the projection left the counter block zero)
            const int64_t R = (int64_t)pub->R, E = (int64_t)pub->E;
            if (pub->far && vcr_depth_sort_far(N, (uint2*)(v.s1 + 2 * nb), (uint32_t*)(v.s1 + nb),
                                               (uint32_t*)(v.s1 + 7 * nb),
                                               v.s1 + 7 * nb + tot_bytes, cs)) { rc = 1; break; }
            if (vb->num_rendered) vb->num_rendered[c] = R;
            if (vb->num_visible) vb->num_visible[c] = (int32_t)pub->V;
            if (R >= (1ll << 32)) { vcr_set_error("more than 2^32 tile instances"); rc = 1; break; }
            if (E <= 0) continue;
            const bool third = vcr_sort_passes(tbits) > 2;
            if (E > v.cap) {                       // grow with head-room: later cameras of the batch re-use the set
                const int64_t cap = E + E / 4 + 1024;
                const size_t rbts = vcr_align(sizeof(uint32_t) * (size_t)cap);
                v.bin = alloc(user, VCR_BUF_SCRATCH, BinState::bytes(cap, T));
                v.s2 = (char*)alloc(user, VCR_BUF_SCRATCH, (third ? 7 : 5) * rbts + vcr_binning_temp_bytes(N, cap, tbits));
                if (!v.bin || !v.s2) { vcr_set_error("allocator returned NULL"); rc = 1; break; }
                v.cap = cap;
            }
            const size_t rbts = vcr_align(sizeof(uint32_t) * (size_t)v.cap);
            const size_t tmp2 = vcr_binning_temp_bytes(N, v.cap, tbits);
            const VcrRasterArgs a = cam_args(c);
            GeomState g = GeomState::view(v.geom, N, 0);
            BinState b = BinState::view(v.bin, T);
            uint32_t* ids_sorted = (uint32_t*)(v.s1 + nb);
            uint32_t* totals_tile = (uint32_t*)(v.s1 + 7 * nb) + VCR_SORT_TOTALS_WORDS;
            if (vcr_duplicate_and_sort(a, g, v.radii, ids_sorted, sc.status, sc.ctr + VCR_DUP_TICKET_WORD, v.dseq, E, tbits,
                                       (uint2*)v.s2, (uint2*)(v.s2 + 2 * rbts),
                                       third ? (uint2*)(v.s2 + 5 * rbts) : nullptr, (uint32_t*)(v.s2 + 4 * rbts), b.point_list,
                                       b.ranges, b.tile_order, b.meta, T, totals_tile, v.s2 + (third ? 7 : 5) * rbts, tmp2, cs)) { rc = 1; break; }
            ImageState im;
            im.final_T = nullptr; im.n_contrib = nullptr; im.moments = nullptr;      // (not written by the count modes)
            VcrForwardOut fo;
            memset(&fo, 0, sizeof(fo));
            fo.count = vb->count;
            if (vcr_launch_composite_forward(a, g, b, im, fo, cs)) { rc = 1; break; }
        }
    }
    join_all();                    // also on errors: the caller's stream-ordered allocator gets the buffers back on return
    return rc;
}

#ifdef VCR_DBG_ACC64
int vcr_dbg_acc64_begin(int N, hipStream_t st);
int vcr_dbg_acc64_end(int N, GradRec* sgrad, hipStream_t st);
#endif
namespace {
// (process-wide, not per thread: autograd runs the backward on its own host thread)
bool g_keep_sgrad = false;
float* g_sgrad_copy = nullptr;
int g_sgrad_copy_n = 0;
// vcr_rasterize_backward (tail == nullptr) and vcr_rasterize_backward_tail
int backward_impl(const VcrRasterArgs* args, VcrBackwardIO* io, const VcrGeometryStep* tail, vcr_alloc_fn alloc, void* user,
                  void* stream) {
    if (validate(args)) return 1;
    if (!io || !alloc) { vcr_set_error("io/alloc is NULL"); return 1; }
    const VcrRasterArgs& a = *args;
    if (a.f_count != 0) { vcr_set_error("backward is defined for f_count=0 only"); return 1; }
    hipStream_t st = (hipStream_t)stream;
    const int N = a.N, P = a.H * a.W;
    if (N == 0) return 0;
    if (!io->dL_dout || !io->geom || !io->binning || !io->image || !io->radii) { vcr_set_error("backward: required pointer is NULL"); return 1; }
    if (!tail && (!io->dL_dmeans3D || !io->dL_dmeans2D || !io->dL_dopacities)) { vcr_set_error("backward: required pointer is NULL"); return 1; }
    if ((io->normals_Rw2c == nullptr) != (io->normals_aux == nullptr)) { vcr_set_error("backward: normals_Rw2c and normals_aux go together"); return 1; }
    if (a.shs && !io->dL_dshs && !io->dL_drgb) { vcr_set_error("backward: dL_dshs and dL_drgb are both NULL"); return 1; }
    if (a.shs_rest && io->dL_dshs && !io->dL_dshs_rest) { vcr_set_error("backward: dL_dshs_rest is NULL"); return 1; }
    if (!tail && a.scales && (!io->dL_dscales || !io->dL_drotations)) { vcr_set_error("backward: dL_dscales/rotations NULL"); return 1; }
    if (a.cov3D_precomp && !io->dL_dcov3D) { vcr_set_error("backward: dL_dcov3D is NULL"); return 1; }
    if (tail) {
        const VcrGeometryStep& t = *tail;
        if (t.N != N) { vcr_set_error("backward tail: N = %d, the render had %d Gaussians", t.N, N); return 1; }
        if (!a.scales || a.cov3D_precomp) { vcr_set_error("backward tail: needs the scale / rotation form of the covariance"); return 1; }
        if (a.shs && io->dL_dshs) { vcr_set_error("backward tail: SH gradients only as dL_drgb"); return 1; }
        if (a.S > 0 && !io->dL_dsemantics) { vcr_set_error("backward tail: dL_dsemantics is NULL"); return 1; }
        if (t.d_means3D || t.d_scales || t.d_rots || t.d_opac || t.d_normals || t.grad2d || t.radii) {
            vcr_set_error("backward tail: the upstream gradient pointers of VcrGeometryStep must be NULL (they stay in registers)"); return 1;
        }
        if (!t.xyz || !t.scaling || !t.rotation || !t.opacity || !t.m_xyz || !t.v_xyz || !t.m_scaling || !t.v_scaling ||
            !t.m_rotation || !t.v_rotation || !t.m_opacity || !t.v_opacity || t.step_xyz < 1 || t.step_scaling < 1 ||
            t.step_rotation < 1 || t.step_opacity < 1 || (a.normals_precomp && (!t.aux || !t.Rw2c)) ||
            (t.scale_reg_sums && (!t.scale_reg_gout || !t.trans || !t.scale)) ||
            (t.accum && (!t.denom || !t.max_radii)) ||
            (t.next_scales && (!t.next_rots || !t.next_opac || (t.next_normals && (!t.next_campos || !t.next_Rw2c || !t.next_aux))))) {
            vcr_set_error("backward tail: inconsistent VcrGeometryStep"); return 1;
        }
        if (!(t.grad_scale > 0.f) || t.normals_world) { vcr_set_error("backward tail: grad_scale must be > 0 and normals_world 0 (single process)"); return 1; }
        if ((((uintptr_t)t.rotation) | ((uintptr_t)t.m_rotation) | ((uintptr_t)t.v_rotation) | ((uintptr_t)t.next_rots)) & 15) {
            vcr_set_error("backward tail: quaternion arrays must be 16-byte aligned"); return 1;
        }
    }
    const int gx = (a.W + VCR_TILE - 1) / VCR_TILE, gy = (a.H + VCR_TILE - 1) / VCR_TILE;
    GeomState g = GeomState::view(const_cast<void*>(io->geom), N, a.S);
    BinState b = BinState::view(const_cast<void*>(io->binning), gx * gy);
    ImageState im = ImageState::view(const_cast<void*>(io->image), P);
    if (io->num_rendered > 0 && io->num_emitted < 0) {       // (quad-list mode counts 8x8 cells: num_emitted may exceed num_rendered)
        vcr_set_error("backward: num_emitted = %lld is not the forward's", (long long)io->num_emitted);
        return 1;
    }
    im.t_ckpt = VCR_T_ANCHOR ? BinState::ckpt_of(const_cast<void*>(io->binning), io->num_emitted, gx * gy) : nullptr;
    const size_t gb = vcr_align(sizeof(GradRec) * (size_t)N);
    const size_t sb = vcr_align(sizeof(float) * (size_t)N * (a.S > 0 ? a.S : 1));
    // the screen-space accumulators: library-owned, zero between calls (the projection backward clears what it reads)
    StreamScratch* sc = stream_scratch(st);
    if (!sc) { vcr_set_error("hipMalloc for the counter block failed"); return 1; }
    if (scratch_begin_backward(sc, gb + sb, st)) return 1;
    char* s = sc->grad;
    GradRec* sgrad = (GradRec*)s;
    float* sgrad_sem = (float*)(s + gb);
    (void)alloc; (void)user;
    if (io->num_rendered > 0) {
        StageTimer tm(ST_COMPOSITE_BWD, st);
#ifdef VCR_DBG_ACC64
        if (vcr_dbg_acc64_begin(N, st)) return 1;
#endif
        if (vcr_launch_composite_backward(a, g, b, im, io->dL_dout, sgrad, sgrad_sem, st)) return 1;
#ifdef VCR_DBG_ACC64
        if (vcr_dbg_acc64_end(N, sgrad, st)) return 1;
#endif
    }
    if (g_keep_sgrad) {                      // diagnostics (vcr_debug_keep_sgrad): the accumulators as the compositing backward left them
        if (g_sgrad_copy_n < N) {
            if (g_sgrad_copy) (void)hipFree(g_sgrad_copy);
            g_sgrad_copy = nullptr; g_sgrad_copy_n = 0;
            VCR_HIP_CHECK(hipMalloc((void**)&g_sgrad_copy, sizeof(GradRec) * (size_t)N));
            g_sgrad_copy_n = N;
        }
        VCR_HIP_CHECK(hipMemcpyAsync(g_sgrad_copy, sgrad, sizeof(GradRec) * (size_t)N, hipMemcpyDeviceToDevice, st));
    }
    VcrBackwardIO io2 = *io;
    if (!a.normals_precomp) io2.dL_dnormals = nullptr;
    if (a.S == 0) io2.dL_dsemantics = nullptr;
    if (a.colors_precomp == nullptr) io2.dL_dcolors = nullptr;
    if (a.shs == nullptr) io2.dL_drgb = nullptr;
    StageTimer tm(ST_PREPROCESS_BWD, st);
    const int rc = tail ? vcr_launch_preprocess_backward_tail(a, g, io->radii, sgrad, sgrad_sem, io2, *tail, st)
                        : vcr_launch_preprocess_backward(a, g, io->radii, sgrad, sgrad_sem, io2, st);
    if (!rc) sc->grad_dirty = false;         // (both kernels accepted: the accumulators are zero again when they have run)
    return rc;
}
}  // namespace

extern "C" int vcr_debug_keep_sgrad(int on) { g_keep_sgrad = on != 0; return 0; }
extern "C" int vcr_debug_read_sgrad(float* host_out, int N) {
    if (!g_sgrad_copy || N > g_sgrad_copy_n || !host_out) { vcr_set_error("vcr_debug_read_sgrad: no copy of %d records held", N); return 1; }
    VCR_HIP_CHECK(hipDeviceSynchronize());
    VCR_HIP_CHECK(hipMemcpy(host_out, g_sgrad_copy, sizeof(GradRec) * (size_t)N, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int vcr_rasterize_backward(const VcrRasterArgs* args, VcrBackwardIO* io, vcr_alloc_fn alloc, void* user,
                                      void* stream) {
    return backward_impl(args, io, nullptr, alloc, user, stream);
}

extern "C" int vcr_rasterize_backward_tail(const VcrRasterArgs* args, VcrBackwardIO* io, const VcrGeometryStep* tail,
                                           vcr_alloc_fn alloc, void* user, void* stream) {
    if (!tail) { vcr_set_error("vcr_rasterize_backward_tail: tail is NULL"); return 1; }
    return backward_impl(args, io, tail, alloc, user, stream);
}

// A HIP stream whose kernels may only run on the compute units whose bits are set in `mask` (`nwords` 32-bit words, bit i of the
// logical CU numbering of the driver, which deals consecutive bits round-robin over the XCDs and their shader engines, so the
// lowest M bits are M CUs spread evenly over the chip).  For the side stream of the two-stream step: a streaming kernel
// confined to part of the chip leaves the rest to the latency-bound sort chain of the main stream.
extern "C" void* vcr_stream_create_cu_masked(const uint32_t* mask, int nwords) {
    hipStream_t s = nullptr;
    if (!mask || nwords <= 0) { vcr_set_error("vcr_stream_create_cu_masked: empty mask"); return nullptr; }
    const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)nwords, mask);
    if (e != hipSuccess) { vcr_set_error("hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e)); return nullptr; }
    return (void*)s;
}
extern "C" int vcr_release_scratch(void) { scratch_map().release(); return 0; }

extern "C" int vcr_stream_destroy(void* stream) {
    if (stream) VCR_HIP_CHECK(hipStreamDestroy((hipStream_t)stream));
    return 0;
}

extern "C" void vcr_profile_enable(int on) { g_prof = on != 0; }
extern "C" void vcr_profile_select(unsigned stage_mask) { g_prof_mask = stage_mask; }

extern "C" int vcr_profile_num_stages(void) { return ST_COUNT; }

// Adds the elapsed milliseconds / launch counts recorded since the last call into ms[0..n) / launches[0..n)
// (stage order: preprocess, depth sort+scan, duplicate+tile sort+ranges, composite fwd, composite bwd,
// preprocess bwd).  The caller must have synchronised the stream(s) first.
extern "C" int vcr_profile_read(float* ms, int32_t* launches, int n) {
    for (auto& e : g_used) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, e.a, e.b) != hipSuccess) { vcr_set_error("hipEventElapsedTime failed (not synchronised?)"); return 1; }
        if (e.stage < n) { ms[e.stage] += t; launches[e.stage] += 1; }
        g_free.push_back(e.a); g_free.push_back(e.b);
    }
    g_used.clear();
    return 0;
}
