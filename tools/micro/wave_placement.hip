// Which SIMD does wave w of a workgroup land on?  (developer probe; prints the SIMD id of every wave of a few blocks)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int* out) {
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        const int simd = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);   // HW_REG_HW_ID bits 5:4
        const int cu = __builtin_amdgcn_s_getreg((3 << 11) | (8 << 6) | 4);     // bits 11:8
        out[blockIdx.x * 16 + w] = simd | (cu << 8);
    }
}
int main() {
    int* d; hipMalloc(&d, 64 * 16 * 4); hipMemset(d, 0xff, 64 * 16 * 4);
    for (int nt : {256, 512, 1024}) {
        hipLaunchKernelGGL(k, dim3(8), dim3(nt), 0, 0, d);
        std::vector<int> h(8 * 16); hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        printf("block size %d:\n", nt);
        for (int b = 0; b < 4; ++b) { printf("  block %d simd of waves:", b); for (int w = 0; w < nt / 64; ++w) printf(" %d", h[b * 16 + w] & 3); printf("\n"); }
    }
    return 0;
}
