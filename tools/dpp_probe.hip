#include <hip/hip_runtime.h>
__global__ void k(int *o) {
    int v = threadIdx.x;
    // xor 8 within rows of 16
    int a = __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);   // row_ror:8
    int b = __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);   // row_half_mirror
    int c = __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
    int d = __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
    auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    auto q = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    o[threadIdx.x * 8 + 0] = a; o[threadIdx.x * 8 + 1] = b; o[threadIdx.x * 8 + 2] = c; o[threadIdx.x * 8 + 3] = d;
    o[threadIdx.x * 8 + 4] = r[0]; o[threadIdx.x * 8 + 5] = r[1]; o[threadIdx.x * 8 + 6] = q[0]; o[threadIdx.x * 8 + 7] = q[1];
}
int main() {
    int *d; hipMalloc(&d, 64 * 8 * 4); k<<<1, 64>>>(d); int h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[8] = {"row_ror8", "half_mirror", "qp_xor2", "qp_xor1", "pl16swap[0]", "pl16swap[1]", "pl32swap[0]", "pl32swap[1]"};
    for (int c = 0; c < 8; ++c) { printf("%-12s:", names[c]); for (int l = 0; l < 64; ++l) printf(" %d", h[l * 8 + c]); printf("\n"); }
    return 0;
}
