#pragma once
#include "gcnn_common.hpp"

// ---------------------------------------------------------------------------------------------------------------
// First embedding layer relu(((x+shift)*scale) @ W[f,64] + b) (model.py:174-177 and var/cut twins): its forward opens
// the forward chains (CH_EMBED1 in k_chain.hpp); here its weight gradient on the VALU (K = f <= 14).
// ---------------------------------------------------------------------------------------------------------------
// gradient of the first embedding layer's weights: dW[f][j] = sum_r xn[r][f] * dPre[r][j], db[j] = sum_r dPre[r][j]
// with dPre = dY * (Y > 0).  One block per chunk of rows; per-block partial slab [(F+1)*64] (row F = bias).
template <int F>
__global__ __launch_bounds__(256) void k_embed1_wgrad(const float* __restrict__ x, const float* __restrict__ shift,
                                                      const float* __restrict__ scale, const float* __restrict__ dy,
                                                      const float* __restrict__ yact, float* __restrict__ partial,
                                                      int n, int rows_per_block) {
    __shared__ float red[4][(F + 1) * EMB];
    const int col = threadIdx.x & 63, part = threadIdx.x >> 6;
    float acc[F + 1];
#pragma unroll
    for (int f = 0; f <= F; ++f) acc[f] = 0.f;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(n, r0 + rows_per_block);
#pragma unroll 4
    for (int r = r0 + part; r < r1; r += 4) {
        float d = dy[(size_t)r * EMB + col];
        d = yact[(size_t)r * EMB + col] > 0.f ? d : 0.f;
#pragma unroll
        for (int f = 0; f < F; ++f) acc[f] = fmaf((x[(size_t)r * F + f] + shift[f]) * scale[f], d, acc[f]);
        acc[F] += d;
    }
#pragma unroll
    for (int f = 0; f <= F; ++f) red[part][f * EMB + col] = acc[f];
    __syncthreads();
    for (int i = threadIdx.x; i < (F + 1) * EMB; i += 256)
        partial[(size_t)blockIdx.x * (F + 1) * EMB + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradients: G[64,64] = sum_r (sx*X[r])^T D[r], db = sum_r D[r], dbd = sum_r deg_r D[r]     (B3/B4/B8/B11)
// Grouped launch: one job per (X, D) pair, one block per 256-row chunk of a job.  Wave w owns the 32x32 quadrant
// (w>>1, w&1) of G; rows are the MFMA k dimension.  Per-block partial slab [64*64 + 64 + 64] floats; summed in a
// fixed order by k_reduce (no atomics).
// ---------------------------------------------------------------------------------------------------------------
#define WG_ROWS 256
#define WG_SLAB (EMB * EMB + 2 * EMB)
#define WG_MAX_JOBS 24
struct WgJob { const float* x; const float* sx; const float* d; const int* seg_ptr; const float* d2; int n; int blk0; int slab0; };
struct WgArgs { int njobs; int nblocks; float* partial; WgJob job[WG_MAX_JOBS]; };

__global__ __launch_bounds__(256) void k_wgrad(WgArgs a) {
    __shared__ __attribute__((aligned(16))) float xs[64 * LDW];
    __shared__ __attribute__((aligned(16))) float ds[64 * LDW];
    __shared__ float red[4][2 * EMB];
    int ji = 0;
    while (ji + 1 < a.njobs && (int)blockIdx.x >= a.job[ji + 1].blk0) ++ji;
    const WgJob jb = a.job[ji];
    const int lb = blockIdx.x - jb.blk0;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int qi = wv >> 1, qj = wv & 1, i32 = lane & 31, h = lane >> 5;
    const float sx = jb.sx ? *jb.sx : 1.f;
    const int col = threadIdx.x & 63, part = threadIdx.x >> 6;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float cs = 0.f, cds = 0.f;
    const int rbeg = lb * WG_ROWS, rend = min(jb.n, rbeg + WG_ROWS);
    for (int row0 = rbeg; row0 < rend; row0 += 64) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // 64 rows x 16 float4 = 1024 float4 per matrix
            const int idx = i * 256 + threadIdx.x;
            const int r = idx >> 4, c = (idx & 15) * 4;
            float4 xv = make_float4(0.f, 0.f, 0.f, 0.f), dv = xv;
            if (row0 + r < rend) {
                xv = *(const float4*)(jb.x + (size_t)(row0 + r) * EMB + c);
                dv = *(const float4*)(jb.d + (size_t)(row0 + r) * EMB + c);
                xv.x *= sx; xv.y *= sx; xv.z *= sx; xv.w *= sx;
            }
            *(float4*)(xs + r * LDW + c) = xv;
            *(float4*)(ds + r * LDW + c) = dv;
        }
        __syncthreads();
#pragma unroll 8
        for (int s = 0; s < 32; ++s) {
            const int r = h * 32 + s;
            acc = mfma32(xs[r * LDW + qi * 32 + i32], ds[r * LDW + qj * 32 + i32], acc);
        }
        // column sums of D (bias grads), 16 rows per thread
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
            const int r = part * 16 + s;
            const float dv = ds[r * LDW + col];
            cs += dv;
            const int gr = row0 + r;
            if (jb.seg_ptr) {        // second column sum: degree-weighted (gradient of the hoisted b_f)
                const float deg = gr < rend ? (float)(jb.seg_ptr[gr + 1] - jb.seg_ptr[gr]) : 0.f;
                cds = fmaf(deg, dv, cds);
            } else if (jb.d2) {      // ... or the plain column sum of a second matrix (Q -> d w_edge)
                cds += gr < rend ? jb.d2[(size_t)gr * EMB + col] : 0.f;
            }
        }
    }
    float* slab = a.partial + (size_t)(jb.slab0 + lb) * WG_SLAB;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        slab[(qi * 32 + row) * EMB + qj * 32 + i32] = acc[i];
    }
    red[part][col] = cs; red[part][EMB + col] = cds;
    __syncthreads();
    if (threadIdx.x < 2 * EMB)
        slab[EMB * EMB + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// Fixed-order sum of partial slabs into the flat gradient buffer.  One block per (job, 64-float chunk).
#define RD_MAX_JOBS 96
struct RdJob { const float* src; float* dst; int nparts; int stride; int len; int blk0; };
struct RdArgs { int njobs; RdJob job[RD_MAX_JOBS]; };

__global__ __launch_bounds__(256) void k_reduce(RdArgs a) {
    __shared__ float red[4][EMB];
    int ji = 0;
    while (ji + 1 < a.njobs && (int)blockIdx.x >= a.job[ji + 1].blk0) ++ji;
    const RdJob jb = a.job[ji];
    const int chunk = blockIdx.x - jb.blk0;
    const int col = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int e = chunk * EMB + col;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < jb.len) {
        const float* src = jb.src + e;
        int p = part;
        for (; p + 12 < jb.nparts; p += 16) {  // 4 loads in flight per thread; the order of the adds is fixed
            const float v0 = src[(size_t)p * jb.stride], v1 = src[(size_t)(p + 4) * jb.stride];
            const float v2 = src[(size_t)(p + 8) * jb.stride], v3 = src[(size_t)(p + 12) * jb.stride];
            s0 += v0; s1 += v1; s2 += v2; s3 += v3;
        }
        for (; p < jb.nparts; p += 4) s0 += src[(size_t)p * jb.stride];
    }
    red[part][col] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (part == 0 && e < jb.len) jb.dst[e] = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
}

