// fp64 VALU issue-rate probe: v_fma_f64 with VGPR-only operands vs one SGPR operand, per waves/SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o tools/valu_probe && tools/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(64) void k_probe(double *out, const double *__restrict__ sc, int iters, unsigned long long *clk) {
    double acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = threadIdx.x * 1e-3 + k;
    double v = out[threadIdx.x];
    double s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = sc[k];  // uniform -> SGPRs
    double sv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) sv[k] = s[k] + threadIdx.x * 1e-9;  // VGPR copies
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (MODE == 0) acc[k + 8 * (r & 1)] = __builtin_fma(sv[k], v, acc[k + 8 * (r & 1)]);
                if (MODE == 1) acc[k + 8 * (r & 1)] = __builtin_fma(s[k], v, acc[k + 8 * (r & 1)]);
                if (MODE == 2) acc[k + 8 * (r & 1)] = acc[k + 8 * (r & 1)] + sv[k];
                if (MODE == 3) acc[k + 8 * (r & 1)] = acc[k + 8 * (r & 1)] * sv[k];
            }
            asm volatile("" : "+v"(v));
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    double t = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += acc[k];
    out[blockIdx.x * 64 + threadIdx.x] = t;
    if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    double *out, *sc;
    unsigned long long *clk;
    hipMalloc(&out, 8 * 64 * ncu * 64);
    hipMalloc(&sc, 64);
    hipMalloc(&clk, 8 * ncu * 64);
    hipMemset(out, 0, 8 * 64 * ncu * 64);
    double h[8] = {1.0, 0.5, 0.25, 2.0, 1.5, 0.75, 1.25, 3.0};
    hipMemcpy(sc, h, 64, hipMemcpyHostToDevice);
    const int iters = 20000;
    const char *names[4] = {"fma vgpr operands", "fma one sgpr operand", "add f64 vgpr", "mul f64 vgpr"};
    for (int mode = 0; mode < 4; ++mode)
        for (int wps : {1, 2, 3, 4, 8}) {
            const int blocks = ncu * 4 * wps;
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            auto launch = [&]() {
                if (mode == 0) k_probe<0><<<blocks, 64>>>(out, sc, iters, clk);
                if (mode == 1) k_probe<1><<<blocks, 64>>>(out, sc, iters, clk);
                if (mode == 2) k_probe<2><<<blocks, 64>>>(out, sc, iters, clk);
                if (mode == 3) k_probe<3><<<blocks, 64>>>(out, sc, iters, clk);
            };
            launch();
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> hc(blocks);
            hipMemcpy(hc.data(), clk, 8 * blocks, hipMemcpyDeviceToHost);
            double mean = 0;
            for (auto c : hc) mean += c;
            mean /= blocks;
            const double instr = (double)iters * 64;  // per wave
            const double tflops = 2.0 * instr * 64 * blocks / (ms * 1e-3) / 1e12;
            printf("%-22s %d waves/SIMD: %.2f clocks/instr/wave (memtime), %.3f ms, %.1f TFLOP/s-equivalent, eff. clock if 4 clk/instr: %.0f MHz\n",
                   names[mode], wps, mean / instr, ms, tflops, instr * 4 * wps / (ms * 1e-3) / 1e6);
        }
    return 0;
}
