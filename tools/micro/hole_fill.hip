// Do scattered 8-byte stores into lines that were written (with holes) a moment ago merge in the L2 / Infinity Cache,
// or does each of them cost a read-modify-write at the memory?  K arrays of n doubles; kernel A writes 5 of 6 elements of
// every array (the columns that finish in one pass), kernel B the sixth (the deferred columns), either over the whole
// arrays one after the other or chunk by chunk (A on chunk c, then B on chunk c) with chunks that fit the 256 MB cache.
//   hipcc -O3 --offload-arch=gfx950 -o hole_fill hole_fill.hip && ./hole_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int K = 22;
struct ptrs { double *p[K]; };
__device__ __forceinline__ bool hole(long long i) { return ((unsigned long long)i * 0x9E3779B97F4A7C15ull >> 40) % 6 == 0; }
__global__ void kA(ptrs P, long long i0, long long n)
{
    const long long i = i0 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= i0 + n || hole(i)) return;
#pragma unroll
    for (int k = 0; k < K; ++k) P.p[k][i] = 1.0 + k;
}
__global__ void kB(ptrs P, long long i0, long long n)
{
    const long long i = i0 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= i0 + n || !hole(i)) return;
#pragma unroll
    for (int k = 0; k < K; ++k) P.p[k][i] = 2.0 + k;
}
__global__ void kFull(ptrs P, long long i0, long long n)
{
    const long long i = i0 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= i0 + n) return;
#pragma unroll
    for (int k = 0; k < K; ++k) P.p[k][i] = 3.0 + k;
}
int main()
{
    const long long n = 12000000;
    ptrs P;
    for (int k = 0; k < K; ++k) hipMalloc(&P.p[k], n * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](const char *name, long long chunk, int mode) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            for (long long i0 = 0; i0 < n; i0 += chunk) {
                const long long m = (n - i0 < chunk) ? n - i0 : chunk;
                const unsigned g = (unsigned)((m + 255) / 256);
                if (mode == 0) hipLaunchKernelGGL(kFull, dim3(g), dim3(256), 0, 0, P, i0, m);
                if (mode >= 1) hipLaunchKernelGGL(kA, dim3(g), dim3(256), 0, 0, P, i0, m);
                if (mode == 2) hipLaunchKernelGGL(kB, dim3(g), dim3(256), 0, 0, P, i0, m);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("%-46s chunk %9lld  %.3f ms  (%.0f GB/s of the %d x 8 B per element)\n", name, chunk, best,
               (double)n * K * 8 / best / 1e6, K);
    };
    run("full-line stores only", n, 0);
    run("A alone (5 of 6 elements)", n, 1);
    run("A then B over the whole arrays", n, 2);
    for (long long c : {4000000ll, 2000000ll, 1000000ll, 500000ll, 250000ll, 125000ll}) run("A then B chunk by chunk", c, 2);
    for (long long c : {1000000ll, 250000ll}) run("full-line stores, chunked launches", c, 0);
    return 0;
}
