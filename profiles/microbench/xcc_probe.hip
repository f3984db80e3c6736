#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = x;
}
int main() {
    unsigned* d; hipMalloc(&d, 4096 * 4);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, d);
    unsigned h[4096]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int hist[16] = {0}, rr = 0;
    for (int i = 0; i < 4096; ++i) { hist[h[i] & 15]++; rr += ((h[i] & 15) == (unsigned)(i & 7)); }
    for (int i = 0; i < 16; ++i) printf("%d ", hist[i]);
    printf("\nraw first: "); for (int i = 0; i < 20; ++i) printf("%x ", h[i]);
    printf("\nblockIdx&7 == xcc for %d of 4096\n", rr);
    return 0;
}
