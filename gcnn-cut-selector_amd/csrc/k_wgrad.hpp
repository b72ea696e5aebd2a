#pragma once
#include "gcnn_common.hpp"

// ---------------------------------------------------------------------------------------------------------------
// First embedding layer relu(((x+shift)*scale) @ W[f,64] + b) (model.py:174-177 and var/cut twins): its forward opens
// the forward embedding programs (emb_program in k_rows.hpp); here its weight gradient on the VALU (K = f <= 14).
// ---------------------------------------------------------------------------------------------------------------
// gradient of the first embedding layer's weights: dW[f][j] = sum_r xn[r][f] * dPre[r][j], db[j] = sum_r dPre[r][j]
// with dPre = dY * (Y > 0).  One WAVE per chunk of EMB1_ROWS rows, lane = output column; the four waves of a block add up
// in LDS: per-block partial slab [(F+1)*64] (row F = bias).  The three embeddings (F = 4, 14, 6) are extra block ranges of the k_wgrad launch below.
#define EMB1_ROWS 64
#define WG_WAVES 4    // waves (= chunks) per block of the k_wgrad launch
struct Emb1Job { const float* x; const float* shift; const float* scale; const float* dy; const float* yact; float* partial; int n; int f; int blk0; };
struct Emb1Args { int njobs; int nblocks; Emb1Job job[3]; };

template <int F>
__device__ __forceinline__ void embed1_wgrad_body(const Emb1Job& jb, int blk, float* red) {
    const int col = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float acc[F + 1], shift[F], scale[F];
#pragma unroll
    for (int f = 0; f <= F; ++f) acc[f] = 0.f;
#pragma unroll
    for (int f = 0; f < F; ++f) { shift[f] = jb.shift[f]; scale[f] = jb.scale[f]; }
    const int r0 = min(jb.n, (blk * WG_WAVES + wv) * EMB1_ROWS);   // this wave's chunk (may be empty: zeros)
    const int r1 = min(jb.n, r0 + EMB1_ROWS);
#pragma unroll 8
    for (int r = r0; r < r1; ++r) {
        float d = jb.dy[(size_t)r * EMB + col];
        d = jb.yact[(size_t)r * EMB + col] > 0.f ? d : 0.f;
#pragma unroll
        for (int f = 0; f < F; ++f) acc[f] = fmaf((jb.x[(size_t)r * F + f] + shift[f]) * scale[f], d, acc[f]);   // x: wave-uniform
        acc[F] += d;
    }
    // the four waves of the block add up in LDS: one partial slab per block
#pragma unroll
    for (int f = 0; f <= F; ++f) red[wv * 15 * EMB + f * EMB + col] = acc[f];
    __syncthreads();
    for (int i = threadIdx.x; i < (F + 1) * EMB; i += 64 * WG_WAVES)
        jb.partial[(size_t)blk * (F + 1) * EMB + i] = (red[i] + red[15 * EMB + i]) + (red[2 * 15 * EMB + i] + red[3 * 15 * EMB + i]);
}
__device__ __forceinline__ void embed1_wgrad_block(const Emb1Args& a, int b, float* red) {
    int ji = 0;
    while (ji + 1 < a.njobs && b >= a.job[ji + 1].blk0) ++ji;
    const Emb1Job jb = a.job[ji];
    const int blk = b - jb.blk0;
    if (jb.f == 4) embed1_wgrad_body<4>(jb, blk, red);
    else if (jb.f == 6) embed1_wgrad_body<6>(jb, blk, red);
    else embed1_wgrad_body<14>(jb, blk, red);
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradients: G[64,64] = sum_r (sx*X[r])^T D[r], db = sum_r D[r], dbd = sum_r deg_r D[r]     (B3/B4/B8/B11)
// Grouped launch: one job per (X, D) pair, ONE WAVE per WG_ROWS-row chunk of a job; the four waves of a block take four
// consecutive chunks and add their results up in LDS, so a block emits one partial slab per 4 chunks.  Chunk = 128 rows at
// setcov-500 x 32 (balances the 1,024 SIMDs better than 256; 64 drowns in prologue), scaled up with the row count so that
// big row sets (capfac: 650 k rows) do not pay for tens of thousands of slabs.  Rows are the MFMA k dimension (v_mfma_f32_16x16x4_f32, four rows per instruction) and both operands come
// straight from global memory, every row read exactly once as whole 256-B lines: lane (m, g) loads the float4 at columns
// 4m..4m+3 of row 4*step+g of X and of D; component va of the X load and component vb of the D load feed accumulator
// (va, vb), so two loads feed 16 MFMAs.  No LDS or barriers in the main loop; loads run one 16-row batch ahead of the MFMAs.
// Per-block partial slab [64*64 + 64 + 64] floats; summed in a fixed order by k_reduce (no atomics).
// ---------------------------------------------------------------------------------------------------------------
#define WG_ROWS 128   // smallest chunk; big row sets use a multiple (WgArgs.rows_per_wave) so the launch stays at a few thousand waves
#define WG_SLAB (EMB * EMB + 2 * EMB)
#define WG_MAX_JOBS 24
#define WG_STEPS 4   // 4-row MFMA steps per batch of loads
struct WgJob { const float* x; const float* sx; const float* d; const int* seg_ptr; int n; int blk0; int slab0; };
struct WgArgs { int njobs; int nblocks; int rows_per_wave; float* partial; WgJob job[WG_MAX_JOBS]; };

typedef float f32x4w __attribute__((ext_vector_type(4)));
struct WgBatch { float4 x[WG_STEPS], d[WG_STEPS]; int p0[WG_STEPS], p1[WG_STEPS]; };

// EXTRA: 0 = none, 1 = degree-weighted column sum of D (gradient of the hoisted b_f)
template <int EXTRA>
__device__ __forceinline__ void wg_load(WgBatch& t, const WgJob& jb, int row0, int rend, int g, int col) {
#pragma unroll
    for (int s = 0; s < WG_STEPS; ++s) {
        const int r = row0 + 4 * s + g;
        t.x[s] = t.d[s] = make_float4(0.f, 0.f, 0.f, 0.f);
        t.p0[s] = t.p1[s] = 0;
#ifdef WG_ABL_NOLOAD
        if (r < rend && row0 < 0) {
#else
        if (r < rend) {
#endif
            t.x[s] = *(const float4*)(jb.x + (size_t)r * EMB + col);
            t.d[s] = *(const float4*)(jb.d + (size_t)r * EMB + col);
            if (EXTRA == 1) { t.p0[s] = jb.seg_ptr[r]; t.p1[s] = jb.seg_ptr[r + 1]; }   // raw: converting here would wait
        }
    }
}

template <int EXTRA>
__device__ __forceinline__ void wg_body(const WgJob& jb, float* slab, int rbeg, int rend) {   // slab: this wave's LDS region
    const int lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4, col = 4 * m;
    f32x4w acc[4][4];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i >> 2][i & 3] = f32x4w{0.f, 0.f, 0.f, 0.f};
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f), ce = cs;
    WgBatch cur, nxt;
    wg_load<EXTRA>(cur, jb, rbeg, rend, g, col);
    for (int row0 = rbeg; row0 < rend; row0 += 4 * WG_STEPS) {
        wg_load<EXTRA>(nxt, jb, row0 + 4 * WG_STEPS, rend, g, col);   // past the chunk: all lanes load nothing
#pragma unroll
        for (int s = 0; s < WG_STEPS; ++s) {
            const float xa[4] = {cur.x[s].x, cur.x[s].y, cur.x[s].z, cur.x[s].w};
            const float db[4] = {cur.d[s].x, cur.d[s].y, cur.d[s].z, cur.d[s].w};
#pragma unroll
            for (int va = 0; va < 4; ++va)
#pragma unroll
                for (int vb = 0; vb < 4; ++vb)
#ifdef WG_ABL_NOMFMA
                    acc[va][vb][0] += xa[va] * db[vb];
#else
                    acc[va][vb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[va], db[vb], acc[va][vb], 0, 0, 0);
#endif
            cs.x += db[0]; cs.y += db[1]; cs.z += db[2]; cs.w += db[3];
            if (EXTRA == 1) {
                const float deg = (float)(cur.p1[s] - cur.p0[s]);
                ce.x = fmaf(deg, db[0], ce.x); ce.y = fmaf(deg, db[1], ce.y); ce.z = fmaf(deg, db[2], ce.z); ce.w = fmaf(deg, db[3], ce.w);
            }
        }
        cur = nxt;
    }
    const float sx = jb.sx ? *jb.sx : 1.f;
    // acc[va][vb][t] = G[4*(4g+t) + va][4m + vb]
#pragma unroll
    for (int va = 0; va < 4; ++va)
#pragma unroll
        for (int t = 0; t < 4; ++t)
            *(float4*)(slab + (4 * (4 * g + t) + va) * EMB + col) =
                make_float4(acc[va][0][t] * sx, acc[va][1][t] * sx, acc[va][2][t] * sx, acc[va][3][t] * sx);
    // column sums: fold the four row slots (g) of the wave
    float* c8[8] = {&cs.x, &cs.y, &cs.z, &cs.w, &ce.x, &ce.y, &ce.z, &ce.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) { *c8[i] += __shfl_xor(*c8[i], 16); *c8[i] += __shfl_xor(*c8[i], 32); }
    if (g == 0) {
        *(float4*)(slab + EMB * EMB + col) = cs;
        *(float4*)(slab + EMB * EMB + EMB + col) = ce;
    }
}

// d w_edge: the edge-gradient passes leave one 64-float partial per thread block (up to thousands).  Third block type of
// this launch: block j of convolution c adds DW_CHUNK consecutive partial rows in a fixed order, so that k_reduce is left
// with a few dozen rows per convolution like for every other gradient.
#define DW_CHUNK 128
struct DwRedArgs { const float* src[3]; float* dst[3]; int nparts[3]; int blk0[4]; };
__device__ __forceinline__ void dw_reduce_block(const DwRedArgs& d, int b, float* red) {
    int c = 0;
    while (c < 2 && b >= d.blk0[c + 1]) ++c;
    const int chunk = b - d.blk0[c], col = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int p0 = chunk * DW_CHUNK, p1 = min(d.nparts[c], p0 + DW_CHUNK);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const float* src = d.src[c] + col;
    int p = p0 + part;
    for (; p + 12 < p1; p += 16) {   // 4 loads in flight per thread; the order of the adds is fixed
        const float v0 = src[(size_t)p * EMB], v1 = src[(size_t)(p + 4) * EMB], v2 = src[(size_t)(p + 8) * EMB], v3 = src[(size_t)(p + 12) * EMB];
        s0 += v0; s1 += v1; s2 += v2; s3 += v3;
    }
    for (; p < p1; p += 4) s0 += src[(size_t)p * EMB];
    red[part * EMB + col] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (part == 0) d.dst[c][(size_t)chunk * EMB + col] = (red[col] + red[EMB + col]) + (red[2 * EMB + col] + red[3 * EMB + col]);
}

__global__ __launch_bounds__(64 * WG_WAVES) void k_wgrad(WgArgs a, Emb1Args e, DwRedArgs dw) {
    extern __shared__ __attribute__((aligned(16))) float wg_red[];   // [WG_WAVES][WG_SLAB]
    const int ndw = dw.blk0[3];
    if ((int)blockIdx.x < ndw) {   // the d w_edge pre-reduction first: a few short blocks, out of the way before the long chunks fill the chip
        dw_reduce_block(dw, blockIdx.x, wg_red);
        return;
    }
    const int bx = (int)blockIdx.x - ndw;
    if (bx >= a.nblocks) {   // the (short) first-embedding-layer chunks come after the long MFMA chunks (measured: first is worse)
        embed1_wgrad_block(e, bx - a.nblocks, wg_red);
        return;
    }
    int ji = 0;   // last job whose first block is <= this block: binary search (a linear scan is one dependent scalar load per job)
    for (int hi = a.njobs; hi - ji > 1;) {
        const int mid = (ji + hi) >> 1;
        if (bx >= a.job[mid].blk0) ji = mid; else hi = mid;
    }
    const WgJob jb = a.job[ji];
    const int lb = bx - jb.blk0, wv = threadIdx.x >> 6;
    if ((long long)lb * WG_WAVES * a.rows_per_wave >= jb.n) return;   // padding block (jobs start at multiples of 8 blocks): owns no rows, no slab
    const int rbeg = min(jb.n, (lb * WG_WAVES + wv) * a.rows_per_wave), rend = min(jb.n, rbeg + a.rows_per_wave);   // may be empty: zeros
    float* mine = wg_red + wv * WG_SLAB;
    if (jb.seg_ptr) wg_body<1>(jb, mine, rbeg, rend);
    else wg_body<0>(jb, mine, rbeg, rend);
    __syncthreads();
    float4* slab = (float4*)(a.partial + (size_t)(jb.slab0 + lb) * WG_SLAB);
    for (int i = threadIdx.x; i < WG_SLAB / 4; i += 64 * WG_WAVES) {
        const float4 p0 = ((const float4*)wg_red)[i], p1 = ((const float4*)(wg_red + WG_SLAB))[i];
        const float4 p2 = ((const float4*)(wg_red + 2 * WG_SLAB))[i], p3 = ((const float4*)(wg_red + 3 * WG_SLAB))[i];
        slab[i] = make_float4((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y), (p0.z + p1.z) + (p2.z + p3.z),
                              (p0.w + p1.w) + (p2.w + p3.w));
    }
}
static_assert(WG_WAVES == 4, "the block-level sum above is written for four waves");

// Fixed-order sum of partial slabs into the flat gradient buffer.  One block per (job, 64-float chunk).
#define RD_MAX_JOBS 96
struct RdJob { const float* src; float* dst; int nparts; int stride; int len; int blk0; };
// optional fused optimizer step: every reduced element that lies inside the flat gradient buffer [gbase, gbase+gn) is a
// finished gradient, so the Keras-form Adam update (see k_adam) of the same flat slot can follow in the same thread
struct RdAdam { float* p; float* m; float* v; const float* gbase; int gn; float lr_t, b1, b2, eps; };
struct RdArgs { int njobs; float* cdst; float cval; RdAdam adam; RdJob job[RD_MAX_JOBS]; };   // cdst (optional): a constant the launch also stores

__global__ __launch_bounds__(256) void k_reduce(RdArgs a) {
    __shared__ float red[4][EMB];
    if (a.cdst && blockIdx.x == 0 && threadIdx.x == 0) *a.cdst = a.cval;
    int ji = 0;   // binary search over up to 96 jobs
    for (int hi = a.njobs; hi - ji > 1;) {
        const int mid = (ji + hi) >> 1;
        if ((int)blockIdx.x >= a.job[mid].blk0) ji = mid; else hi = mid;
    }
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
    if (part == 0 && e < jb.len) {
        const float gi = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
        jb.dst[e] = gi;
        const long long idx = (jb.dst + e) - a.adam.gbase;
        if (a.adam.p && idx >= 0 && idx < a.adam.gn) {
            const float mi = a.adam.b1 * a.adam.m[idx] + (1.f - a.adam.b1) * gi;
            const float vi = a.adam.b2 * a.adam.v[idx] + (1.f - a.adam.b2) * gi * gi;
            a.adam.m[idx] = mi; a.adam.v[idx] = vi;
            a.adam.p[idx] -= a.adam.lr_t * mi / (sqrtf(vi) + a.adam.eps);
        }
    }
}

