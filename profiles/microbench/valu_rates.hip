// Instruction-rate microbenchmark for the VALU forms the compositing kernels are made of (gfx950).
// Build: hipcc --offload-arch=gfx950 -O2 -o valu_rates valu_rates.hip ; run on an MI355X.
// Prints, per instruction form, SIMD cycles per wave64 instruction (assuming the measured sclk) at 1 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 mk2(float s) { f2 r; r.x = s; r.y = s; return r; }
#define REP8(X) X X X X X X X X
#define ITERS 2000

#define KERNEL(NAME, DECL, BODY, SINK)                                                   \
    __global__ void __launch_bounds__(1024) NAME(float* out, float seed, int sel) {      \
        DECL;                                                                            \
        for (int i = 0; i < ITERS; ++i) { REP8(BODY) REP8(BODY) REP8(BODY) REP8(BODY) }  \
        if (seed == 12345.f) out[threadIdx.x] = SINK;                                    \
    }

KERNEL(k_fma, float a = seed; float b = seed * 2; float c = seed * 3; float d = seed + 1; float x = seed + 2; float y = seed + 3,
       asm volatile("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3"
                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));, a + b + c + d)
KERNEL(k_fma_sgpr, float a = seed; float b = seed * 2; float c = seed * 3; float d = seed + 1; float x = seed + 2; float y = seed + 3,
       asm volatile("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3"
                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(x), "v"(y));, a + b + c + d)
KERNEL(k_pk_fma, f2 a = mk2(seed); f2 b = a * 2; f2 c = a * 3; f2 d = a + 1; f2 x = a + 2; f2 y = a + 3,
       asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3"
                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));, a.x + b.y + c.x + d.y)
KERNEL(k_pk_fma_sgpr, f2 a = mk2(seed); f2 b = a * 2; f2 c = a * 3; f2 d = a + 1; f2 x = a + 2; f2 y = a + 3,
       asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3"
                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(x), "v"(y));, a.x + b.y + c.x + d.y)
KERNEL(k_pk_fma_opsel, f2 a = mk2(seed); f2 b = a * 2; f2 c = a * 3; f2 d = a + 1; f2 x = a + 2; f2 y = a + 3,
       asm volatile("v_pk_fma_f32 %0, %4, %5, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %1, %4, %5, %1 op_sel_hi:[0,1,1]\n"
                    "v_pk_fma_f32 %2, %4, %5, %2 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %3, %4, %5, %3 op_sel_hi:[0,1,1]"
                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "s"(y));, a.x + b.y + c.x + d.y)
KERNEL(k_pk_mul, f2 a = mk2(seed); f2 b = a * 2; f2 c = a * 3; f2 d = a + 1; f2 x = a + 2,
       asm volatile("v_pk_mul_f32 %0, %4, %0\n v_pk_mul_f32 %1, %4, %1\n v_pk_mul_f32 %2, %4, %2\n v_pk_mul_f32 %3, %4, %3"
                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x));, a.x + b.y + c.x + d.y)
KERNEL(k_pk_add, f2 a = mk2(seed); f2 b = a * 2; f2 c = a * 3; f2 d = a + 1; f2 x = a + 2,
       asm volatile("v_pk_add_f32 %0, %4, %0\n v_pk_add_f32 %1, %4, %1\n v_pk_add_f32 %2, %4, %2\n v_pk_add_f32 %3, %4, %3"
                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x));, a.x + b.y + c.x + d.y)
KERNEL(k_mul, float a = seed; float b = seed * 2; float c = seed * 3; float d = seed + 1; float x = seed + 2,
       asm volatile("v_mul_f32 %0, %4, %0\n v_mul_f32 %1, %4, %1\n v_mul_f32 %2, %4, %2\n v_mul_f32 %3, %4, %3"
                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x));, a + b + c + d)
KERNEL(k_exp, float a = seed; float b = seed * 2; float c = seed * 3; float d = seed + 1,
       asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3"
                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d));, a + b + c + d)
KERNEL(k_rcp, float a = seed; float b = seed * 2; float c = seed * 3; float d = seed + 1,
       asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3"
                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d));, a + b + c + d)
KERNEL(k_exp_fma_mix, float a = seed; float b = seed * 2; float c = seed * 3; float d = seed + 1; float x = seed + 2; float y = seed + 3,
       asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3"
                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));, a + b + c + d)
KERNEL(k_readlane, float a = seed; float b = seed * 2; int s0 = 0; int s1 = 0; int s2 = 0; int s3 = 0,
       asm volatile("v_readlane_b32 %0, %4, %6\n v_readlane_b32 %1, %5, %6\n v_readlane_b32 %2, %4, %6\n v_readlane_b32 %3, %5, %6"
                    : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(a), "v"(b), "s"(sel));, (float)(s0 + s1 + s2 + s3))
KERNEL(k_readlane_use, float a = seed; float b = seed * 2; float c = seed; float d = seed; int s0 = 0; int s1 = 0,
       asm volatile("v_readlane_b32 %0, %4, %6\n v_readlane_b32 %1, %5, %6\n v_fma_f32 %2, %0, %4, %2\n v_fma_f32 %3, %1, %5, %3"
                    : "=&s"(s0), "=&s"(s1), "+v"(c), "+v"(d) : "v"(a), "v"(b), "s"(sel));, c + d + (float)(s0 + s1))
KERNEL(k_swap32, float a = seed; float b = seed * 2; float c = seed * 3; float d = seed + 1,
       asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3"
                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d));, a + b + c + d)
KERNEL(k_swap16, float a = seed; float b = seed * 2; float c = seed * 3; float d = seed + 1,
       asm volatile("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3"
                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d));, a + b + c + d)
KERNEL(k_add_dpp, float a = seed; float b = seed * 2; float c = seed * 3; float d = seed + 1,
       asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n"
                    "v_add_f32_dpp %2, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_ror:4 row_mask:0xf bank_mask:0xf"
                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d));, a + b + c + d)
KERNEL(k_add_dpp_dep, float a = seed; float b = seed * 2; float c = seed * 3; float d = seed + 1,
       asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                    "v_add_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n s_nop 1"
                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d));, a + b + c + d)
KERNEL(k_cndmask, float a = seed; float b = seed * 2; float c = seed * 3; float d = seed + 1; float x = seed + 2; unsigned long long m = (unsigned long long)sel * 0x9E3779B97F4A7C15ull,
       asm volatile("v_cndmask_b32_e64 %0, %0, %4, %5\n v_cndmask_b32_e64 %1, %1, %4, %5\n v_cndmask_b32_e64 %2, %2, %4, %5\n v_cndmask_b32_e64 %3, %3, %4, %5"
                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "s"(m));, a + b + c + d)
KERNEL(k_cmp, float a = seed; float b = seed * 2; unsigned long long m0 = 0; unsigned long long m1 = 0,
       asm volatile("v_cmp_lt_f32_e64 %0, %2, %3\n v_cmp_gt_f32_e64 %1, %2, %3\n v_cmp_le_f32_e64 %0, %2, %3\n v_cmp_ge_f32_e64 %1, %2, %3"
                    : "=s"(m0), "=s"(m1) : "v"(a), "v"(b));, (float)(m0 + m1))

struct Entry { const char* name; void (*fn)(float*, float, int); int vinst; };

int main() {
    float* out; hipMalloc(&out, 4096 * sizeof(float));
    int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    printf("device clock %d kHz, %d CUs\n", clk_khz, cus);
    std::vector<Entry> es = {
        {"v_fma_f32 (vgpr)", k_fma, 4}, {"v_fma_f32 (sgpr src)", k_fma_sgpr, 4}, {"v_mul_f32", k_mul, 4},
        {"v_pk_fma_f32 (vgpr)", k_pk_fma, 4}, {"v_pk_fma_f32 (sgpr pair)", k_pk_fma_sgpr, 4},
        {"v_pk_fma_f32 (op_sel bcast, sgpr)", k_pk_fma_opsel, 4}, {"v_pk_mul_f32", k_pk_mul, 4}, {"v_pk_add_f32", k_pk_add, 4},
        {"v_exp_f32", k_exp, 4}, {"v_rcp_f32", k_rcp, 4}, {"1 v_exp + 3 v_fma", k_exp_fma_mix, 4},
        {"v_readlane_b32", k_readlane, 4}, {"2 v_readlane + 2 dependent v_fma", k_readlane_use, 4},
        {"v_permlane32_swap", k_swap32, 4}, {"v_permlane16_swap", k_swap16, 4}, {"v_add_f32_dpp (indep)", k_add_dpp, 4},
        {"v_add_f32_dpp (dependent chain + s_nop 1)", k_add_dpp_dep, 4}, {"v_cndmask_b32_e64", k_cndmask, 4}, {"v_cmp_f32_e64", k_cmp, 4},
    };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (auto& e : es) {
        for (int wps : {1, 2, 4}) {                       // waves per SIMD (block = 4 SIMDs x wps waves)
            const int threads = 256 * wps;
            hipLaunchKernelGGL(e.fn, dim3(cus), dim3(threads), 0, 0, out, 1.0f, 3);
            hipEventRecord(e0);
            hipLaunchKernelGGL(e.fn, dim3(cus), dim3(threads), 0, 0, out, 1.0f, 3);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            const double inst = (double)ITERS * 32 * e.vinst * wps;     // wave-instructions per SIMD
            printf("%-44s waves/SIMD %d  %.3f ms  %.2f cycles/inst @%.2f GHz\n", e.name, wps, ms, ms * 1e-3 * clk_khz * 1e3 / inst,
                   clk_khz * 1e-6);
        }
    }
    return 0;
}
