// Which pipe should carry the per-survivor record broadcast of the compositing kernels?  (gfx950)
// v_readlane_b32 measured 8.3 SIMD cycles (valu_rates.hip), i.e. the 14 broadcasts per survivor are half of the forward
// shading loop.  This measures the LDS-pipe alternatives, alone and mixed with VALU work, at 4..20 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X X X X X X X X
#define ITERS 2000

// 16 ds_bpermute per body, uniform source lane
__global__ void __launch_bounds__(1024) k_bperm(float* out, float seed, int sel) {
    float a = seed + threadIdx.x, b = a * 2, c = a * 3, d = a + 1;
    const int addr = sel * 4;
    float r0, r1, r2, r3;
    float acc = 0.f;
    for (int i = 0; i < ITERS; ++i) {
        REP8(asm volatile("ds_bpermute_b32 %0, %4, %5\n ds_bpermute_b32 %1, %4, %6\n ds_bpermute_b32 %2, %4, %7\n ds_bpermute_b32 %3, %4, %8\n"
                          : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(addr), "v"(a), "v"(b), "v"(c), "v"(d));)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc += r0 + r1 + r2 + r3;
    }
    if (seed == 12345.f) out[threadIdx.x] = acc;
}

// 8 x ds_read_b128 at a wave-uniform address per body
__global__ void __launch_bounds__(1024) k_lds_b128(float* out, float seed, int sel) {
    __shared__ float4 buf[1024];
    buf[threadIdx.x] = make_float4(seed, seed, seed, seed);
    __syncthreads();
    const int addr = ((threadIdx.x >> 6) * 64 + sel) * 16;      // uniform per wave
    float4 r0, r1, r2, r3;
    float acc = 0.f;
    for (int i = 0; i < ITERS; ++i) {
        REP8(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n"
                          : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(addr));)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc += r0.x + r1.y + r2.z + r3.w;
    }
    if (seed == 12345.f) out[threadIdx.x] = acc;
}

__global__ void __launch_bounds__(1024) k_lds_b32(float* out, float seed, int sel) {
    __shared__ float buf[4096];
    buf[threadIdx.x] = seed;
    __syncthreads();
    const int addr = ((threadIdx.x >> 6) * 64 + sel) * 4;
    float r0, r1, r2, r3;
    float acc = 0.f;
    for (int i = 0; i < ITERS; ++i) {
        REP8(asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:4\n ds_read_b32 %2, %4 offset:8\n ds_read_b32 %3, %4 offset:12\n"
                          : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(addr));)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc += r0 + r1 + r2 + r3;
    }
    if (seed == 12345.f) out[threadIdx.x] = acc;
}

// "survivor iteration" models: 14 broadcasts + 28 VALU (20 v_fma + 8 v_cndmask), broadcasts of iteration i+1 in flight
#define VALU28(A, B, C, D, X, Y, M)                                                                                         \
    asm volatile("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3\n"   \
                 "v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3\n"   \
                 "v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3\n"   \
                 "v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3\n"   \
                 "v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3\n"   \
                 "v_cndmask_b32_e64 %0, %0, %4, %6\n v_cndmask_b32_e64 %1, %1, %4, %6\n v_cndmask_b32_e64 %2, %2, %4, %6\n"          \
                 "v_cndmask_b32_e64 %3, %3, %4, %6\n v_cndmask_b32_e64 %0, %0, %5, %6\n v_cndmask_b32_e64 %1, %1, %5, %6\n"          \
                 "v_cndmask_b32_e64 %2, %2, %5, %6\n v_cndmask_b32_e64 %3, %3, %5, %6\n"                                            \
                 : "+v"(A), "+v"(B), "+v"(C), "+v"(D) : "v"(X), "v"(Y), "s"(M))

__global__ void __launch_bounds__(1024) k_mix_readlane(float* out, float seed, int sel) {
    float a = seed + threadIdx.x, b = a * 2, c = a * 3, d = a + 1;
    float q[14];
    for (int k = 0; k < 14; ++k) q[k] = a + k;
    const unsigned long long m = (unsigned long long)sel * 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < ITERS * 4; ++i) {
        float s[14];
#pragma unroll
        for (int k = 0; k < 14; ++k) s[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q[k]), (sel + i) & 63));
        float x = s[0] + s[2] + s[4] + s[6] + s[8] + s[10] + s[12], y = s[1] + s[3] + s[5] + s[7] + s[9] + s[11] + s[13];
        VALU28(a, b, c, d, x, y, m);
    }
    if (seed == 12345.f) out[threadIdx.x] = a + b + c + d;
}

__global__ void __launch_bounds__(1024) k_mix_bperm(float* out, float seed, int sel) {
    float a = seed + threadIdx.x, b = a * 2, c = a * 3, d = a + 1;
    float q[14];
    for (int k = 0; k < 14; ++k) q[k] = a + k;
    const unsigned long long m = (unsigned long long)sel * 0x9E3779B97F4A7C15ull;
    float s[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) s[k] = __int_as_float(__builtin_amdgcn_ds_bpermute((sel & 63) * 4, __float_as_int(q[k])));
    for (int i = 0; i < ITERS * 4; ++i) {
        float n[14];
#pragma unroll
        for (int k = 0; k < 14; ++k) n[k] = __int_as_float(__builtin_amdgcn_ds_bpermute(((sel + i + 1) & 63) * 4, __float_as_int(q[k])));
        float x = s[0] + s[2] + s[4] + s[6] + s[8] + s[10] + s[12], y = s[1] + s[3] + s[5] + s[7] + s[9] + s[11] + s[13];
        VALU28(a, b, c, d, x, y, m);
#pragma unroll
        for (int k = 0; k < 14; ++k) s[k] = n[k];
    }
    if (seed == 12345.f) out[threadIdx.x] = a + b + c + d;
}

// same, broadcast through 4 x ds_read_b128 (+ the per-chunk LDS write amortised away)
__global__ void __launch_bounds__(1024) k_mix_lds(float* out, float seed, int sel) {
    __shared__ float4 buf[20 * 64 * 4];
    float a = seed + threadIdx.x, b = a * 2, c = a * 3, d = a + 1;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int k = 0; k < 4; ++k) buf[(wv * 64 + lane) * 4 + k] = make_float4(a + k, b, c, d);
    __builtin_amdgcn_wave_barrier();
    const unsigned long long m = (unsigned long long)sel * 0x9E3779B97F4A7C15ull;
    const float4* base = buf + wv * 256;
    float4 s0 = base[(sel & 63) * 4], s1 = base[(sel & 63) * 4 + 1], s2 = base[(sel & 63) * 4 + 2], s3 = base[(sel & 63) * 4 + 3];
    for (int i = 0; i < ITERS * 4; ++i) {
        const int nb = ((sel + i + 1) & 63) * 4;
        const float4 n0 = base[nb], n1 = base[nb + 1], n2 = base[nb + 2], n3 = base[nb + 3];
        float x = s0.x + s0.z + s1.x + s1.z + s2.x + s2.z + s3.x, y = s0.y + s0.w + s1.y + s1.w + s2.y + s2.w + s3.y;
        VALU28(a, b, c, d, x, y, m);
        s0 = n0; s1 = n1; s2 = n2; s3 = n3;
    }
    if (seed == 12345.f) out[threadIdx.x] = a + b + c + d;
}

struct Entry { const char* name; void (*fn)(float*, float, int); double per_iter; const char* unit; };

int main() {
    float* out; hipMalloc(&out, 4096 * sizeof(float));
    int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    std::vector<Entry> es = {
        {"ds_bpermute_b32 (uniform lane)", k_bperm, ITERS * 32.0, "inst"},
        {"ds_read_b128 (wave-uniform address)", k_lds_b128, ITERS * 32.0, "inst"},
        {"ds_read_b32 (wave-uniform address)", k_lds_b32, ITERS * 32.0, "inst"},
        {"iteration: 14 v_readlane + 13 add + 28 VALU", k_mix_readlane, ITERS * 4.0, "iter"},
        {"iteration: 14 ds_bpermute (prefetched) + 13 add + 28 VALU", k_mix_bperm, ITERS * 4.0, "iter"},
        {"iteration: 4 ds_read_b128 (prefetched) + 13 add + 28 VALU", k_mix_lds, ITERS * 4.0, "iter"},
    };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (auto& e : es) {
        for (int waves : {4, 8, 16}) {                       // waves per CU (one block per CU)
            const int threads = 64 * waves;
            hipLaunchKernelGGL(e.fn, dim3(cus), dim3(threads), 0, 0, out, 1.0f, 3);
            hipEventRecord(e0);
            hipLaunchKernelGGL(e.fn, dim3(cus), dim3(threads), 0, 0, out, 1.0f, 3);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            const double cyc = ms * 1e-3 * clk_khz * 1e3;
            printf("%-60s waves/CU %2d  %.3f ms  %.1f CU-cycles per wave-%s  (%.1f per SIMD-wave)\n", e.name, waves, ms,
                   cyc / (e.per_iter * waves), e.unit, cyc / (e.per_iter * waves / 4.0));
        }
    }
    return 0;
}
