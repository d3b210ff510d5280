// Dev probe: issue rate of v_mfma_f64_16x16x4_f64 on gfx950 (4 independent accumulators per wave) by waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_probe.hip -o tools/mfma_f64_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(double *out, int iters) {
    v4d acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    double *out;
    hipMalloc(&out, 256 * 1024 * 8);
    for (int threads : {256, 512, 1024}) {
        const int iters = 4000;
        k<<<256, threads>>>(out, iters);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k<<<256, threads>>>(out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double n_mfma = (double)iters * 16 * (threads / 64) * 256;
        printf("waves/SIMD %d: %.1f us, %.2f TFLOP/s (2048 flops per MFMA), %.1f ns per MFMA per SIMD\n", threads / 256, ms * 1e3,
               n_mfma * 2048 / (ms * 1e-3) / 1e12, ms * 1e6 / ((double)iters * 16 * (threads / 256)));
    }
    return 0;
}
