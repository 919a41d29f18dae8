// How much does the NUMBER of streams cost at the same bytes?  A soil-like kernel: every lane reads NS separate fp64 streams
// plus one record of NR doubles (NS + NR = 55 values read), writes 22 streams.  hipcc -O3 --offload-arch=gfx950 stream_count.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int kRead = 55, kWrite = 22;
struct ptrs { const double *r[kRead]; double *w[kWrite]; const double *rec; };
template <int NR> // NR values of the 55 come from the record (stride NR doubles per lane), the other 55 - NR from streams
__global__ void __launch_bounds__(256) k(ptrs P, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double v[kRead];
#pragma unroll
    for (int s = 0; s < kRead - NR; ++s) v[s] = P.r[s][i];
    if (NR > 0) {
        const double *q = P.rec + i * NR;
#pragma unroll
        for (int s = 0; s < NR; ++s) v[kRead - NR + s] = q[s];
    }
    double acc = 0;
#pragma unroll
    for (int s = 0; s < kRead; ++s) acc += v[s];
#pragma unroll
    for (int s = 0; s < kWrite; ++s) P.w[s][i] = acc + s;
}
template <int NR> float run(ptrs P, long long n, int reps)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k<NR>, dim3((n + 255) / 256), dim3(256), 0, 0, P, n);
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<NR>, dim3((n + 255) / 256), dim3(256), 0, 0, P, n);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}
int main()
{
    const long long n = 12000000;
    ptrs P;
    for (int s = 0; s < kRead; ++s) { hipMalloc((void **)&P.r[s], n * 8); hipMemset((void *)P.r[s], 0, n * 8); }
    for (int s = 0; s < kWrite; ++s) hipMalloc((void **)&P.w[s], n * 8);
    double *rec; hipMalloc((void **)&rec, n * 8 * 24); hipMemset(rec, 0, n * 8 * 24); P.rec = rec;
    const double bytes = (double)n * 8 * (kRead + kWrite);
    for (int rep = 0; rep < 2; ++rep) {
        float t0 = run<0>(P, n, 10), t8 = run<8>(P, n, 10), t21 = run<21>(P, n, 10), t24 = run<24>(P, n, 10);
        printf("55 read streams: %.3f ms %.2f TB/s | 47 + record of 8: %.3f ms %.2f | 34 + record of 21: %.3f ms %.2f | 31 + record of 24: %.3f ms %.2f\n",
               t0, bytes / t0 / 1e9, t8, bytes / t8 / 1e9, t21, bytes / t21 / 1e9, t24, bytes / t24 / 1e9);
    }
    return 0;
}
