// Calibration of the FETCH_SIZE counter (rocprofv3 --pmc FETCH_SIZE) on the access pattern of the compositing kernels: a GATHER of
// 64-byte records by random index, 16 bytes per lane per load.  Every record is read exactly once (the indices are a permutation), so
// the bytes that must come from memory are known: 64 N (whole records), 32 N (the first half only: what the two-phase forward gathers
// ahead of its culling) -- plus 4 N of coalesced index reads.  A streaming read of the same array is the control.
// Build: hipcc --offload-arch=gfx950 -O2 -o fetch_calib fetch_calib.hip
// Run:   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o f -- ./fetch_calib     (profiles/r6_fetch_calib.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>

__global__ void gather_full(const float4* __restrict__ rec, const uint32_t* __restrict__ idx, float* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4* r = rec + 4 * (size_t)idx[i];
    const float4 a = r[0], b = r[1], c = r[2], d = r[3];
    const float s = a.x + b.y + c.z + d.w;
    if (s == 12345.678f) out[i] = s;
}
__global__ void gather_half(const float4* __restrict__ rec, const uint32_t* __restrict__ idx, float* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4* r = rec + 4 * (size_t)idx[i];
    const float4 a = r[0], b = r[1];
    const float s = a.x + b.y;
    if (s == 12345.678f) out[i] = s;
}
__global__ void stream_full(const float4* __restrict__ rec, float* __restrict__ out, int n4) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 a = rec[i];
    if (a.x == 12345.678f) out[i] = a.y;
}

int main() {
    const int n = 4 << 20;                                 // 4 Mi records = 256 MiB: the size of the Infinity Cache, 64x an L2
    std::vector<uint32_t> h(n);
    std::iota(h.begin(), h.end(), 0u);
    std::mt19937 g(1);
    std::shuffle(h.begin(), h.end(), g);
    float4* rec; uint32_t* idx; float* out;
    hipMalloc(&rec, (size_t)n * 64); hipMalloc(&idx, (size_t)n * 4); hipMalloc(&out, (size_t)n * 16);
    hipMemset(rec, 0, (size_t)n * 64);
    hipMemcpy(idx, h.data(), (size_t)n * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(gather_full, dim3(n / 256), dim3(256), 0, 0, rec, idx, out, n);
        hipLaunchKernelGGL(gather_half, dim3(n / 256), dim3(256), 0, 0, rec, idx, out, n);
        hipLaunchKernelGGL(stream_full, dim3(n * 4 / 256), dim3(256), 0, 0, rec, out, n * 4);
    }
    hipDeviceSynchronize();
    printf("records %d: gather_full must fetch %.1f MiB (+ %.1f MiB of indices), gather_half %.1f MiB (+ indices), stream_full %.1f MiB\n", n,
           n * 64.0 / 1048576, n * 4.0 / 1048576, n * 32.0 / 1048576, n * 64.0 / 1048576);
    return 0;
}
