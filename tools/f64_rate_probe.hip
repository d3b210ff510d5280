// Dev probe: issue rate of fp64 vector instructions on gfx950 as a function of waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/f64_rate_probe.hip -o tools/f64_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ void k(double *out, int iters, unsigned long long *cyc) {
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = 1.0 + threadIdx.x * 1e-9 + i;
    const double c1 = 1.0000001, c2 = 1e-9;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) a[i] = __builtin_fma(a[i], c1, c2);
                if (MODE == 1) a[i] = a[i] + c2;
                if (MODE == 2) a[i] = a[i] * c1;
                if (MODE == 3) a[i] = (i & 1) ? a[i] * c1 : a[i] + c2;
                if (MODE == 4) a[i] = fmax(a[i] + c2, 0.5);
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char *name, int threads, int iters) {
    double *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 1024 * 8); hipMalloc(&cyc, 256 * 8);
    k<MODE><<<256, threads>>>(out, iters, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<256, threads>>>(out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= 256;
    const double ninstr = (double)iters * 32 * (MODE == 4 ? 2 : 1);  // per wave
    const int wps = threads / 64 / 4 > 0 ? threads / 64 / 4 : 1;
    printf("%-8s waves/SIMD %d: %.2f cycles (s_memtime @100MHz x24?) per wave-instruction; wall %.1f us -> %.2f ns per instr per SIMD\n", name, wps,
           mean / ninstr, ms * 1e3, ms * 1e6 / (ninstr * wps));
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int threads : {256, 512, 1024}) {
        run<0>("fma", threads, 20000);
        run<1>("add", threads, 20000);
        run<2>("mul", threads, 20000);
        run<3>("add/mul", threads, 20000);
        run<4>("add+max", threads, 20000);
    }
    return 0;
}
