#pragma once
#include "gcnn_common.hpp"

// ---------------------------------------------------------------------------------------------------------------
// K3/K4/K8'/K11/K12 and their input gradients: the generic 64-wide node GEMM on fp32 MFMA.
//
//   NN mode (forward):   y  = act( (sa*xa) @ wa  [+ xb @ wb] [+ bias] [+ deg (x) bd] )
//   TN mode (backward):  xa' = xa * (ymask > 0) (written back in place when write_back), then
//                        y  (=|+=) so * (xa' @ wa^T)      and optionally      y2 (=|+=) xa' @ wb^T
//
// Block = 4 waves; each wave owns a 32-row tile: stage rows in LDS (full 256-B lines, coalesced), read A fragments as
// b128 along k.  The k index is split between the two lane halves (half h takes k in [32h, 32h+32)): the order of the
// 64 products in each dot product is 0,32,1,33,... (exact fp32 FMA chain, different association than a CPU GEMM).
// Weights sit in LDS as [64][LDW]; B fragments are b32 reads along a row (NN) or b128 reads along k (TN), both
// conflict-free with LDW = 68.
// ---------------------------------------------------------------------------------------------------------------
struct LinArgs {
    const float* xa; const float* ymask; int write_back;
    const float* sa;
    const float* wa; const float* xb; const float* wb;
    const float* bias; const float* bd; const int* seg_ptr;
    const float* so;
    float* y; int beta_y; float* y2; int beta_y2;
    int relu; int n;
};

__device__ __forceinline__ void block_load_w(float* wl, const float* __restrict__ w) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = i * 256 + threadIdx.x;  // float4 index, 1024 of them
        const int r = idx >> 4, c = (idx & 15) * 4;
        *(float4*)(wl + r * LDW + c) = *(const float4*)(w + r * EMB + c);
    }
}

__device__ __forceinline__ void wave_load_tile(float* xs, const float* __restrict__ x, int row0, int n, float scale,
                                               const float* __restrict__ ymask, int write_back, float* xwb, int lane) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = i * 4 + (lane >> 4), c = (lane & 15) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + r < n) {
            const size_t o = (size_t)(row0 + r) * EMB + c;
            v = *(const float4*)(x + o);
            if (ymask) {
                const float4 m = *(const float4*)(ymask + o);
                v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f;
                v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
                if (write_back) *(float4*)(xwb + o) = v;
            }
            v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        }
        *(float4*)(xs + r * LDW + c) = v;
    }
}

template <bool TRANSB>
__device__ __forceinline__ void wave_gemm(const float* xs, const float* wl, f32x16 (&acc)[2], int lane) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float4 a = *(const float4*)(xs + r * LDW + h * 32 + q * 4);
        const float av[4] = {a.x, a.y, a.z, a.w};
        if (TRANSB) {
            const float4 b0 = *(const float4*)(wl + r * LDW + h * 32 + q * 4);
            const float4 b1 = *(const float4*)(wl + (32 + r) * LDW + h * 32 + q * 4);
            const float b0v[4] = {b0.x, b0.y, b0.z, b0.w}, b1v[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[0] = mfma32(av[t], b0v[t], acc[0]);
                acc[1] = mfma32(av[t], b1v[t], acc[1]);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int k = h * 32 + q * 4 + t;
                acc[0] = mfma32(av[t], wl[k * LDW + r], acc[0]);
                acc[1] = mfma32(av[t], wl[k * LDW + 32 + r], acc[1]);
            }
        }
    }
}

// accumulator (C/D layout: col = lane&31, row = (i&3) + 8*(i>>2) + 4*(lane>>5)) -> row-major LDS tile
__device__ __forceinline__ void wave_acc_to_tile(float* xs, const f32x16 (&acc)[2], int lane) {
    const int j = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int i = 0; i < 16; ++i) xs[((i & 3) + 8 * (i >> 2) + 4 * hh) * LDW + ct * 32 + j] = acc[ct][i];
}

__device__ __forceinline__ void wave_store_tile(const float* xs, float* __restrict__ y, int row0, int n, int beta,
                                                int lane) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = i * 4 + (lane >> 4), c = (lane & 15) * 4;
        if (row0 + r < n) {
            float4 v = *(const float4*)(xs + r * LDW + c);
            float* dst = y + (size_t)(row0 + r) * EMB + c;
            if (beta) {
                const float4 o = *(const float4*)dst;
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            *(float4*)dst = v;
        }
    }
}

template <bool TRANSB>
__global__ __launch_bounds__(256) void k_linear(LinArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wla = smem;                                  // [64][LDW]
    float* wlb = smem + 64 * LDW;                       // [64][LDW] (second weight, optional)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* xs = smem + 2 * 64 * LDW + wv * 32 * LDW;    // per-wave [32][LDW] tile

    block_load_w(wla, a.wa);
    if (a.wb) block_load_w(wlb, a.wb);
    __syncthreads();

    const float sa = a.sa ? *a.sa : 1.f;
    const float so = a.so ? *a.so : 1.f;
    const int j = lane & 31, hh = lane >> 5;
    const int ntile = (a.n + 31) >> 5;
    for (int tile = blockIdx.x * 4 + wv; tile < ntile; tile += gridDim.x * 4) {
        const int row0 = tile * 32;
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }

        wave_load_tile(xs, a.xa, row0, a.n, sa, a.ymask, a.write_back, const_cast<float*>(a.xa), lane);
        wave_gemm<TRANSB>(xs, wla, acc, lane);
        if (!TRANSB) {
            if (a.xb) {
                wave_load_tile(xs, a.xb, row0, a.n, 1.f, nullptr, 0, nullptr, lane);
                wave_gemm<false>(xs, wlb, acc, lane);
            }
            // epilogue: bias, degree-weighted bias (the hoisted b_f, model.py:499-500 + 568), activation
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const float bv = a.bias ? a.bias[ct * 32 + j] : 0.f;
                const float bdv = a.bd ? a.bd[ct * 32 + j] : 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float v = acc[ct][i] + bv;
                    if (a.bd) {
                        const int row = row0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                        const float deg = row < a.n ? (float)(a.seg_ptr[row + 1] - a.seg_ptr[row]) : 0.f;
                        v = fmaf(deg, bdv, v);
                    }
                    acc[ct][i] = a.relu ? fmaxf(v, 0.f) : v;
                }
            }
            wave_acc_to_tile(xs, acc, lane);
            wave_store_tile(xs, a.y, row0, a.n, a.beta_y, lane);
        } else {
            f32x16 acc2[2];
            if (a.y2) {
#pragma unroll
                for (int i = 0; i < 16; ++i) { acc2[0][i] = 0.f; acc2[1][i] = 0.f; }
                wave_gemm<true>(xs, wlb, acc2, lane);
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[ct][i] *= so;
            wave_acc_to_tile(xs, acc, lane);
            wave_store_tile(xs, a.y, row0, a.n, a.beta_y, lane);
            if (a.y2) {
                wave_acc_to_tile(xs, acc2, lane);
                wave_store_tile(xs, a.y2, row0, a.n, a.beta_y2, lane);
            }
        }
    }
}

