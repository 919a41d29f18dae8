// Microbenchmark behind DESIGN.md section 4.1c (round 4): what one level of a cone's chain costs a single wavefront.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I lisflood-code_amd/csrc tools/micro/chain_latency.hip -o chain_latency
// Prints cycles (s_memtime, 100 MHz-independent: uses clock64 = shader clock) per loop trip for a ladder of loop bodies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "lf_math.h"

#define LF_NEWTON_TOL 1e-12
template <int MODE, int KM>
__global__ void __launch_bounds__(64) k(double *out, long long *cyc, int n, double a, double cst)
{
    __shared__ double x[2][65];
    const int tid = threadIdx.x;
    x[0][tid] = 1.0 + tid;
    x[1][tid] = 2.0;
    if (tid == 0) x[0][64] = x[1][64] = 0.0;
    __syncthreads();
    const int base = (tid * 37) & 31;
    int ad[KM];
    for (int k = 0; k < KM; ++k) ad[k] = (k < 2) ? (base + k) * 8 : 64 * 8;
    const float af = (float)a, laf = __builtin_amdgcn_logf(af);
    double q = 1.0;
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        const char *row = (const char *)&x[i & 1][0];
        double t[KM];
#pragma unroll
        for (int k = 0; k < KM; ++k) t[k] = *(const double *)(row + ad[k]);
        double ups = t[0];
#pragma unroll
        for (int k = 1; k < KM; ++k) ups += t[k];
        double c = ups + cst;
        if (MODE == 0) q = c * 0.25;                      // LDS round trip + adds only
        if (MODE == 1) {                                  // + the fp32 part of the solve
            const float cf = (float)c;
            const float lc = __builtin_amdgcn_logf(cf);
            const float ra = __builtin_amdgcn_exp2f(0.2f * lc);
            const float rb = __builtin_amdgcn_exp2f(0.33333334f * (lc - laf));
            float rf = fminf(ra, rb) * 1.000001f;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const float r2 = rf * rf, r3 = r2 * rf;
                const float g = fmaf(r3, r2, fmaf(af, r3, -cf));
                const float gp = r2 * fmaf(5.0f, r2, 3.0f * af);
                rf = fmaf(-g, __builtin_amdgcn_rcpf(gp), rf);
            }
            q = (double)rf;
        }
        if (MODE == 2) q = lf_solve_3_5(c, a);            // the whole solve
        if (MODE == 3) {                                  // the whole solve with masks as in the kernel
            const bool le = c <= LF_NEWTON_TOL;
            const bool quintic = c <= LF_FAST_MAX;
            q = lf_solve_3_5(c, a);
            q = (le || !quintic) ? 0.0 : q;
        }
        x[(i + 1) & 1][tid] = q;
    }
    const long long t1 = clock64();
    out[tid] = q;
    if (tid == 0) cyc[0] = t1 - t0;
}

template <int MODE, int KM>
void run(const char *what, double *d, long long *c)
{
    const int n = 20000;
    hipLaunchKernelGGL((k<MODE, KM>), dim3(1), dim3(64), 0, 0, d, c, n, 3.7, 0.9);
    hipLaunchKernelGGL((k<MODE, KM>), dim3(1), dim3(64), 0, 0, d, c, n, 3.7, 0.9);
    (void)hipDeviceSynchronize();
    long long h;
    (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%-44s KM=%d: %7.1f clock64 ticks per level\n", what, KM, (double)h / n);
}

int main()
{
    double *d;
    long long *c;
    (void)hipMalloc(&d, 512);
    (void)hipMalloc(&c, 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    // calibrate clock64 against wall time
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<2, 4>), dim3(1), dim3(64), 0, 0, d, c, 200000, 3.7, 0.9);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long h;
    (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("clock64: %.1f ticks per us (kernel %.3f ms, %lld ticks)\n", h / (ms * 1e3), ms, h);
    run<0, 1>("LDS write -> read -> 1 add", d, c);
    run<0, 3>("LDS write -> 3 reads -> adds", d, c);
    run<0, 5>("LDS write -> 5 reads -> adds", d, c);
    run<0, 8>("LDS write -> 8 reads -> adds", d, c);
    run<1, 3>("+ fp32 seed and 3 fp32 Newton steps", d, c);
    run<2, 3>("+ 2 fp64 Newton steps, r^5 (lf_solve_3_5)", d, c);
    run<3, 3>("+ masks", d, c);
    run<3, 5>("+ masks", d, c);
    run<3, 8>("+ masks", d, c);
    return 0;
}
