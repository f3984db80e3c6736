// Does a wave64 VALU instruction get cheaper on gfx950 when only part of the EXEC mask is set?  (Round 6: the two-phase compositing
// forward has 30 % of its lanes busy in phase 2; if inactive 16- or 32-lane groups were skipped, masking them would convert the
// idleness into time.)  Build: hipcc --offload-arch=gfx950 -O2 -o exec_rates exec_rates.hip ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(X) X X X X X X X X
#define ITERS 2000

template <int OP>
__global__ void __launch_bounds__(1024) k(float* out, float seed, unsigned long long mask) {
    float a = seed, b = seed * 2, c = seed * 3, d = seed + 1, x = seed + 2, y = seed + 3;
    f2 pa = {seed, seed}, pb = pa * 2, pc = pa * 3, pd = pa + 1, px = pa + 2, py = pa + 3;
    unsigned long long saved;
    asm volatile("s_mov_b64 %0, exec\n s_mov_b64 exec, %1" : "=s"(saved) : "s"(mask));
    for (int i = 0; i < ITERS; ++i) {
        if (OP == 0) {
            REP8(asm volatile("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));)
            REP8(asm volatile("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));)
        } else if (OP == 1) {
            REP8(asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3"
                              : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd) : "v"(px), "v"(py));)
            REP8(asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3"
                              : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd) : "v"(px), "v"(py));)
        } else {
            REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
            REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
        }
    }
    asm volatile("s_mov_b64 exec, %0" :: "s"(saved));
    if (seed == 12345.f) out[threadIdx.x] = a + b + c + d + pa.x + pb.y + pc.x + pd.y;
}

int main() {
    float* out; hipMalloc(&out, 4096 * sizeof(float));
    int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    printf("device clock %d kHz, %d CUs\n", clk_khz, cus);
    const unsigned long long masks[] = {~0ull, 0xFFFFFFFFull, 0xFFFFull, 0x1ull, 0x0001000100010001ull, 0xFFFF0000FFFF0000ull, 0xFFFFFFFF00000000ull};
    const char* mname[] = {"all 64", "low 32", "low 16", "lane 0", "one lane per 16", "rows 1 and 3", "high 32"};
    const char* oname[] = {"v_fma_f32", "v_pk_fma_f32", "v_exp_f32"};
    void (*fns[])(float*, float, unsigned long long) = {k<0>, k<1>, k<2>};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int op = 0; op < 3; ++op)
        for (int mi = 0; mi < 7; ++mi) {
            const int wps = 4, threads = 256 * wps;
            hipLaunchKernelGGL(fns[op], dim3(cus), dim3(threads), 0, 0, out, 1.0f, masks[mi]);
            hipEventRecord(e0);
            hipLaunchKernelGGL(fns[op], dim3(cus), dim3(threads), 0, 0, out, 1.0f, masks[mi]);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            const double inst = (double)ITERS * 64 * wps;
            printf("%-14s exec = %-16s %.3f ms  %.2f cycles/inst @%.2f GHz (4 waves/SIMD)\n", oname[op], mname[mi], ms,
                   ms * 1e-3 * clk_khz * 1e3 / inst, clk_khz * 1e-6);
        }
    return 0;
}
