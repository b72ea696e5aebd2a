// Does the row programs' store pattern cost HBM write bandwidth?  (developer probe)
// A: the MFMA accumulator layout -- lane (j = lane&15, g = lane>>4) stores 16 B at [row j][16m + 4g], m = 0..3: every store
//    instruction writes 16 rows x 64 B.     B: 16 consecutive lanes store one whole 256-B row: every instruction 4 rows x 256 B.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(512) void k(float* __restrict__ out, int ntile, float v) {
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    for (int t = blockIdx.x * 8 + (threadIdx.x >> 6); t < ntile; t += gridDim.x * 8) {
        float* base = out + (size_t)t * 16 * 64;
        const float4 x = make_float4(v + t, v, v, v);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (MODE == 0) *(float4*)(base + j * 64 + 16 * m + 4 * g) = x;
            else *(float4*)(base + (4 * m + g) * 64 + 4 * j) = x;
        }
    }
}
int main() {
    const int ntile = 40000 * 8;   // 5.1M rows x 256 B = 1.3 GB
    float* out; hipMalloc(&out, (size_t)ntile * 16 * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int tensors = 1; tensors <= 1; ++tensors) {
            auto run = [&] { if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(2048), dim3(512), 0, 0, out, ntile, 1.f); else hipLaunchKernelGGL(k<1>, dim3(2048), dim3(512), 0, 0, out, ntile, 1.f); };
            run(); run();
            hipEventRecord(e0, 0);
            for (int i = 0; i < 10; ++i) run();
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("mode %c: %.1f us per launch, %.2f TB/s\n", mode ? 'B' : 'A', ms * 100, (double)ntile * 16 * 256 / (ms * 1e-4) / 1e12);
        }
    return 0;
}
