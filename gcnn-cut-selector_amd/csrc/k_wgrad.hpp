#pragma once
#include "gcnn_common.hpp"

#define WG_WAVES 4    // waves (= chunks) per block of the k_wgrad launch

// ---------------------------------------------------------------------------------------------------------------
// Weight gradients: G[64,64] = sum_r (sx*X[r])^T D[r], db = sum_r D[r], dbd = sum_r deg_r D[r]     (B3/B4/B8/B11)
// Grouped launch: one job per (X, D) pair, ONE WAVE per chunk of a job's rows; the four waves of a block take four
// consecutive chunks and add their results up in LDS, so a block emits one partial slab per 4 chunks.  The host sizes the
// chunks so that the whole launch is ONE resident round (two blocks per CU; gcnn_capi.hip, place_wg): every SIMD holds two
// waves of nearly equal length from start to end.  Rows are the MFMA k dimension (v_mfma_f32_16x16x4_f32, four rows per
// instruction) and both operands come straight from global memory, every row read exactly once as whole 256-B lines: lane
// (m, g) loads the float4 at columns 4m..4m+3 of row 4*step+g of X and of D; component va of the X load and component vb of
// the D load feed accumulator (va, vb), so two loads feed 16 MFMAs.  No LDS or barriers in the main loop; the loads of the
// next 16-row batch are in flight while the 64 MFMAs of the current one issue (see wg_load / the scheduling barriers in
// wg_body for what that takes).
// Per-block partial slab [64*64 + 64 + 64] floats; summed in a fixed order by k_reduce (no atomics).
// ---------------------------------------------------------------------------------------------------------------
#define WG_ROWS 64    // smallest chunk (rows per wave)
#define WG_SLAB (EMB * EMB + 2 * EMB)
#define WG_MAX_JOBS 28
#define WG_STEPS 4   // 4-row MFMA steps per batch of loads
#ifndef WG_RING
#define WG_RING 2    // batches in the load ring (WG_RING - 1 in flight behind the one being multiplied; 3 and 4 measured: no gain)
#endif
// f > 0: the job is the FIRST layer of an embedding, relu(((x+shift)*scale) @ W[f,64] + b) (model.py:174-177 and twins):
//        x = the raw features [n][f], d = dE1 still unmasked, mask = the ReLU pattern of E1 (k_rows.hpp, mask16: 8 B per row);
//        G is [f,64] (rows >= f: zeros)
// f2 > 0: the job is the SECOND layer of an embedding, G = E1^T D with E1 = relu(((x+shift)*scale) @ W1[f2,64] + b1) -- the first
//        layer's output is not stored by the forward pass (an [N,64] write, and this job's read of it, for something that f2 <= 14
//        raw features determine): x = the raw features [n][f2], and E1 is recomputed 16 rows at a time (EXTRA == 3 below)
struct WgJob { const float* x; const float* sx; const float* d; const int* seg_ptr; const unsigned short* mask; const float *shift, *scale;
               const float *w1, *b1;
               int n; int blk0; int slab0; int f, f2;
               int nb, rows; };   // blocks of the job; rows per wave (a multiple of 16: the job's rows spread evenly over nb * 4 waves)
struct WgArgs { int njobs; int nblocks; float* partial; WgJob job[WG_MAX_JOBS]; };

typedef float f32x4w __attribute__((ext_vector_type(4)));
#define WG_KS_MAX 4   // raw features come in groups of four (one MFMA k-step): at most 14 of them
struct WgBatch { float4 x[WG_STEPS], d[WG_STEPS]; int p0[WG_STEPS], p1[WG_STEPS];   // p0 / p1: segment offsets (EXTRA 1) or the raw mask word (EXTRA 2)
                 float xr[WG_KS_MAX]; };                                           // EXTRA 3: raw features 4*ks + (lane >> 4) of row (lane & 15) of the batch

// EXTRA: 0 = none, 1 = degree-weighted column sum of D (gradient of the hoisted b_f),
//        2 = first embedding layer: X^T has only f <= 14 (padded to 16) rows, so lane (m, g) loads ONE raw feature x[row][m]
//            (normalised at use) as the A operand of the single accumulator row, and D is masked by the layer's output
//        3 = second embedding layer with X = E1 recomputed: per batch of 16 rows the wave evaluates the tile
//            E1[16 rows][64] = relu(Xn[16][F] W1[F][64] + b1) with ceil(F/4) * 4 MFMAs (A: lane (j, k) holds the normalised feature
//            4*ks + k of row j; B: the lane's W1 entries, loop invariant, in registers), and the tile passes through the wave's own
//            LDS region into the operand layout (row 4*step + g, columns 4m..4m+3).  The forward program sums the same F products
//            with a VALU FMA chain: the recomputed E1 equals the forward's up to fp32 rounding (~1e-7 relative)
// Loads are UNCONDITIONAL (rows past the chunk are clamped to its last row and zeroed at use): a load under a branch would
// make the number of loads in flight unknown to the compiler, which then waits for all of them (s_waitcnt vmcnt(0)) right
// after issuing the next batch -- no overlap with the MFMAs at all.
template <int EXTRA, int F>
__device__ __forceinline__ void wg_load(WgBatch& t, const WgJob& jb, int row0, int rend, int g, int col) {
    if (EXTRA == 3) {
        const size_t r = (size_t)min(row0 + (int)(threadIdx.x & 15), rend - 1) * F;
#pragma unroll
        for (int ks = 0; ks < (F + 3) / 4; ++ks) t.xr[ks] = jb.x[r + min(4 * ks + g, F - 1)];   // features >= F: scale 0 at use
    }
#pragma unroll
    for (int s = 0; s < WG_STEPS; ++s) {
        const int r = min(row0 + 4 * s + g, rend - 1);   // callers guarantee rend > 0
        if (EXTRA == 3) {
        } else if (EXTRA == 2) {
            t.x[s] = make_float4(jb.x[(size_t)r * jb.f + min(col >> 2, jb.f - 1)], 0.f, 0.f, 0.f);   // lanes m >= f: scale 0 below
            // columns 4m..4m+3 of row r (m = col/4 = 4a + b) are nibble a of the word at mask[r][b] (bit 4a+i <-> feature 16a+4b+i);
            // kept raw here (decoding would wait for the load), decoded at use
            t.p0[s] = jb.mask[(size_t)r * 4 + ((col >> 2) & 3)];
        } else t.x[s] = *(const float4*)(jb.x + (size_t)r * EMB + col);
        t.d[s] = *(const float4*)(jb.d + (size_t)r * EMB + col);
        if (EXTRA == 1) { t.p0[s] = jb.seg_ptr[r]; t.p1[s] = jb.seg_ptr[r + 1]; }   // raw: converting here would wait
    }
}

#define WG_SCR 68   // padded row of the EXTRA == 3 scratch tile (floats)
template <int EXTRA, int F = 1>
__device__ __forceinline__ void wg_body(const WgJob& jb, float* slab, int rbeg, int rend) {   // slab: this wave's LDS region
    const int lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4, col = 4 * m;
    f32x4w acc[4][4];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i >> 2][i & 3] = f32x4w{0.f, 0.f, 0.f, 0.f};
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f), ce = cs;
    float shift = 0.f, scale = 0.f;   // EXTRA == 2: this lane's feature (m); features >= f contribute zeros
    if (EXTRA == 2 && m < jb.f) { shift = jb.shift[m]; scale = jb.scale[m]; }
    // EXTRA == 3, loop invariant: lane (j = m, k = g) holds W1[4*ks + k][16*blk + j], b1[16*blk + j], and shift / scale of its features
    constexpr int KS = EXTRA == 3 ? (F + 3) / 4 : 1;
    float w1r[KS][4], b1r[4], sh3[KS], sc3[KS];
    if (EXTRA == 3) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int f = 4 * ks + g;
            const bool in = f < F;
            sh3[ks] = in ? jb.shift[f] : 0.f; sc3[ks] = in ? jb.scale[f] : 0.f;
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) w1r[ks][blk] = in ? jb.w1[f * EMB + 16 * blk + m] : 0.f;
        }
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) b1r[blk] = jb.b1[16 * blk + m];
    }
    auto compute = [&](const WgBatch& cur, int row0) {
        if (EXTRA == 3) {   // E1 of the batch's 16 rows, into the wave's scratch tile (the wave's LDS operations complete in order)
            f32x4w ev[4];
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) ev[blk] = f32x4w{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const float xn = (cur.xr[ks] + sh3[ks]) * sc3[ks];
#pragma unroll
                for (int blk = 0; blk < 4; ++blk) ev[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(xn, w1r[ks][blk], ev[blk], 0, 0, 0);
            }
            // ev[blk][t] = E1[row 4g + t][16*blk + m]: banks (4g+t) * 68 + 16*blk + m are distinct over the 64 lanes
#pragma unroll
            for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                for (int t = 0; t < 4; ++t) slab[(4 * g + t) * WG_SCR + 16 * blk + m] = fmaxf(ev[blk][t] + b1r[blk], 0.f);
        }
#pragma unroll
        for (int s = 0; s < WG_STEPS; ++s) {
            const bool live = row0 + 4 * s + g < rend;
            const float4 xs = EXTRA == 3 ? *(const float4*)(slab + (4 * s + g) * WG_SCR + col) : cur.x[s];
            float xa[4] = {xs.x, xs.y, xs.z, xs.w};
            float db[4] = {cur.d[s].x, cur.d[s].y, cur.d[s].z, cur.d[s].w};
#pragma unroll
            for (int v = 0; v < 4; ++v) { xa[v] = live ? xa[v] : 0.f; db[v] = live ? db[v] : 0.f; }
            if (EXTRA == 2) {
                xa[0] = (xa[0] + shift) * scale;
                const unsigned nib = (unsigned)cur.p0[s] >> ((col >> 2) & ~3);
#pragma unroll
                for (int vb = 0; vb < 4; ++vb) db[vb] = (nib >> vb) & 1u ? db[vb] : 0.f;
            }
#pragma unroll
            for (int va = 0; va < (EXTRA == 2 ? 1 : 4); ++va)
#pragma unroll
                for (int vb = 0; vb < 4; ++vb)
                    acc[va][vb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[va], db[vb], acc[va][vb], 0, 0, 0);
            cs.x += db[0]; cs.y += db[1]; cs.z += db[2]; cs.w += db[3];
            if (EXTRA == 1) {
                const float deg = (float)(cur.p1[s] - cur.p0[s]);
                ce.x = fmaf(deg, db[0], ce.x); ce.y = fmaf(deg, db[1], ce.y); ce.z = fmaf(deg, db[2], ce.z); ce.w = fmaf(deg, db[3], ce.w);
            }
        }
    };
    // Ring of WG_RING batches, the loop unrolled WG_RING times so that no batch is ever copied (a copy of a batch still in
    // flight would wait for it): while batch u is multiplied, the loads of the next WG_RING-1 batches are outstanding.  The
    // loads are unconditional and outside every branch (the wait counts are exact); only the MFMAs of batches past the chunk
    // are skipped (a wave-uniform branch).  The scheduling barriers keep each group of loads where it is written: the
    // scheduler otherwise sinks them to their first use, i.e. behind the MFMAs they are meant to overlap.
    constexpr int BR = 4 * WG_STEPS;   // rows per batch
    WgBatch ring[WG_RING];
    if (rbeg < rend) {   // (an empty chunk stores a zero slab)
#pragma unroll
        for (int u = 0; u < WG_RING - 1; ++u) wg_load<EXTRA, F>(ring[u], jb, rbeg + u * BR, rend, g, col);
    }
    for (int row0 = rbeg; row0 < rend; row0 += WG_RING * BR) {
#pragma unroll
        for (int u = 0; u < WG_RING; ++u) {
            wg_load<EXTRA, F>(ring[(u + WG_RING - 1) % WG_RING], jb, row0 + (u + WG_RING - 1) * BR, rend, g, col);   // past the chunk: its last row again, unused
            __builtin_amdgcn_sched_barrier(0);
            if (row0 + u * BR < rend) compute(ring[u], row0 + u * BR);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const float sx = jb.sx ? *jb.sx : 1.f;
    if (EXTRA == 2) {   // acc[0][vb][t] = G[4g + t][4m + vb]: 16 rows, the first f of them real
#pragma unroll
        for (int t = 0; t < 4; ++t)
            *(float4*)(slab + (4 * g + t) * EMB + col) = make_float4(acc[0][0][t], acc[0][1][t], acc[0][2][t], acc[0][3][t]);
    } else {            // acc[va][vb][t] = G[4*(4g+t) + va][4m + vb]
#pragma unroll
        for (int va = 0; va < 4; ++va)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                *(float4*)(slab + (4 * (4 * g + t) + va) * EMB + col) =
                    make_float4(acc[va][0][t] * sx, acc[va][1][t] * sx, acc[va][2][t] * sx, acc[va][3][t] * sx);
    }
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

__global__ __launch_bounds__(64 * WG_WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_wgrad(WgArgs a, DwRedArgs dw) {
    extern __shared__ __attribute__((aligned(16))) float wg_red[];   // [WG_WAVES][WG_SLAB]
    const int ndw = dw.blk0[3];
    if ((int)blockIdx.x < ndw) {   // the d w_edge pre-reduction first: a few short blocks, out of the way before the long chunks fill the chip
        dw_reduce_block(dw, blockIdx.x, wg_red);
        return;
    }
    const int bx = (int)blockIdx.x - ndw;
    int ji = 0;   // last job whose first block is <= this block: binary search (a linear scan is one dependent scalar load per job)
    for (int hi = a.njobs; hi - ji > 1;) {
        const int mid = (ji + hi) >> 1;
        if (bx >= a.job[mid].blk0) ji = mid; else hi = mid;
    }
    const WgJob jb = a.job[ji];
    // Which of the job's row blocks this block takes: rotated so that (row block) = (block index) mod 8 -- blocks are dealt to
    // the eight XCDs round-robin, so the SAME rows of two jobs that share an operand (placed next to each other by the host:
    // dZ1 feeds both halves of W1, a raw embedding X up to three products) are read on the same XCD at about the same time,
    // and the second read is an L2 hit.  (All but at most 7 blocks of a job; padding jobs to 8-block boundaries instead
    // unbalances the XCDs.)
    const int wv = threadIdx.x >> 6;
    int lb = bx - jb.blk0 + (jb.blk0 & 7) % jb.nb;
    if (lb >= jb.nb) lb -= jb.nb;
    const int rbeg = min(jb.n, (lb * WG_WAVES + wv) * jb.rows), rend = min(jb.n, rbeg + jb.rows);   // may be empty: zeros
    float* mine = wg_red + wv * WG_SLAB;
    if (jb.f2) {   // (one job per block: uniform)
        if (jb.f2 == 14) wg_body<3, 14>(jb, mine, rbeg, rend);
        else if (jb.f2 == 6) wg_body<3, 6>(jb, mine, rbeg, rend);
        else wg_body<3, 4>(jb, mine, rbeg, rend);
    } else if (jb.f) wg_body<2>(jb, mine, rbeg, rend);
    else if (jb.seg_ptr) wg_body<1>(jb, mine, rbeg, rend);
    else wg_body<0>(jb, mine, rbeg, rend);
    __syncthreads();
    float4* slab = (float4*)(a.partial + (size_t)(jb.slab0 + lb) * WG_SLAB);
    const int head = jb.f ? 16 * EMB / 4 : EMB * EMB / 4;   // float4s of G that hold anything (first-layer jobs: 16 rows)
    for (int i = threadIdx.x; i < head + 2 * EMB / 4; i += 64 * WG_WAVES) {
        const int q = i < head ? i : EMB * EMB / 4 + (i - head);
        const float4 p0 = ((const float4*)wg_red)[q], p1 = ((const float4*)(wg_red + WG_SLAB))[q];
        const float4 p2 = ((const float4*)(wg_red + 2 * WG_SLAB))[q], p3 = ((const float4*)(wg_red + 3 * WG_SLAB))[q];
        slab[q] = make_float4((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y), (p0.z + p1.z) + (p2.z + p3.z),
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

// ---------------------------------------------------------------------------------------------------------------
// Gradients of the folded layers (k_rows.hpp, Program 2): the forward uses M = s2*Wf*W1a and u = s2*bf*W1a, the weight-gradient
// launch leaves, per convolution, partial slabs of G1 = S^T dZ1 [64,64] and g2 = sum_r deg_r dZ1[r] [64] (one job instead of the
// two that needed A and dA).  By the chain rule through M and u:
//   dWf  = s2 * G1 W1a^T          dW1a = s2 * (Wf^T G1 + bf (x) g2)          dbf = s2 * W1a g2
// A few 64^3 products on reduced data, done by FOLD_BLOCKS blocks per convolution at the front of the k_reduce launch (a launch
// of its own behind k_reduce cost 6.8 us per step): 16 blocks for four rows of dWf each, 16 for four columns of dW1a each, one
// for dbf.  A block adds up exactly the 4 x 64 entries of G1 its outputs need straight from the job's slabs -- 64 float4 positions
// x 4 groups of slabs, a handful of loads per thread, all in flight together with the weight matrix it stages through LDS: one
// memory round trip -- in a fixed order (slabs p, p+4, ... per group, then (g0+g1)+(g2+g3), like k_reduce).  Carries the Adam
// update of these entries when the backward pass was asked to apply it (k_reduce does the same for every other gradient).
// ---------------------------------------------------------------------------------------------------------------
struct FoldArgs { const float* slab[3]; int nparts[3]; const float *wf[3], *bf[3], *w1a[3], *s2[3]; float *gwf[3], *gbf[3], *gw1a[3]; int n; };
#define FOLD_BLOCKS 33
#define FOLD_LDS_FLOATS (EMB * LDW + 1024 + 256 + 2 * EMB)   // a [64,64] matrix | partial sums | the slice of G1 | g2, bf
// sum over the slabs p = first, first + step, ... of the float4 at element offset `off`; eight loads in flight, unconditional
// (a slab index past the end re-reads the last slab and is dropped)
__device__ __forceinline__ float4 fold_chain(const float* __restrict__ slab, int np, int first, int step, int off) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = first; p < np; p += 8 * step) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *(const float4*)(slab + (size_t)min(p + u * step, np - 1) * WG_SLAB + off);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (p + u * step < np) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    return s;
}
__device__ __forceinline__ float4 f4_add(const float4 a, const float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ void fold_block(const FoldArgs& a, const RdAdam& adam, const int b, float* lds) {
    float* big = lds;                        // a whole [64,64] weight matrix, rows padded to LDW
    float4* psum = (float4*)(lds + EMB * LDW);   // [4][64] (dbf block: [16][16]) partial sums
    float* sl = lds + EMB * LDW + 1024;      // the block's 4 x 64 entries of G1
    float* sv = sl + 256;                    // g2 | bf
    const int c = b / FOLD_BLOCKS, part = b % FOLD_BLOCKS, t = threadIdx.x;
    const float s2 = *a.s2[c];
    const float* slab = a.slab[c];
    const int np = a.nparts[c];
    auto emit = [&](float* dst, float gi) {
        *dst = gi;
        const long long idx = dst - adam.gbase;
        if (adam.p && idx >= 0 && idx < adam.gn) {
            const float mi = adam.b1 * adam.m[idx] + (1.f - adam.b1) * gi;
            const float vi = adam.b2 * adam.v[idx] + (1.f - adam.b2) * gi * gi;
            adam.m[idx] = mi; adam.v[idx] = vi;
            adam.p[idx] -= adam.lr_t * mi / (sqrtf(vi) + adam.eps);
        }
    };
    const float* wsrc = (part >= 16 && part < 32) ? a.wf[c] : a.w1a[c];
    float4 tw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) tw[q] = *(const float4*)(wsrc + (q * 256 + t) * 4);
    const int pos = t & 63, pg = t >> 6;
    if (part < 16) {           // dWf[i][j] = s2 * sum_k G1[i][k] * W1a[j][k], rows i = 4*part .. 4*part+3
        const float4 g = fold_chain(slab, np, pg, 4, (4 * part + (pos >> 4)) * EMB + (pos & 15) * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {   // W1a transposed ([k][j]): the loop below reads consecutive j across the lanes
            const int idx = q * 256 + t, r = idx >> 4, col = (idx & 15) * 4;
            big[(col + 0) * LDW + r] = tw[q].x; big[(col + 1) * LDW + r] = tw[q].y; big[(col + 2) * LDW + r] = tw[q].z; big[(col + 3) * LDW + r] = tw[q].w;
        }
        psum[pg * 64 + pos] = g;
        __syncthreads();
        if (t < 64) *(float4*)(sl + 4 * t) = f4_add(f4_add(psum[t], psum[64 + t]), f4_add(psum[128 + t], psum[192 + t]));   // sl[ii][k]
        __syncthreads();
        const int ii = t >> 6, j = t & 63;
        float acc = 0.f;
#pragma unroll 8
        for (int k = 0; k < EMB; ++k) acc = fmaf(sl[ii * EMB + k], big[k * LDW + j], acc);
        emit(a.gwf[c] + (4 * part + ii) * EMB + j, s2 * acc);
    } else if (part < 32) {    // dW1a[j][k] = s2 * (sum_i Wf[i][j] * G1[i][k] + bf[j] * g2[k]), columns k = k0 .. k0+3
        const int k0 = 4 * (part - 16);
        const float4 g = fold_chain(slab, np, pg, 4, pos * EMB + k0);
        float4 g2 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pos == 0) g2 = fold_chain(slab, np, pg, 4, EMB * EMB + EMB + k0);
        const float vb = t < EMB ? a.bf[c][t] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int idx = q * 256 + t; *(float4*)(big + (idx >> 4) * LDW + (idx & 15) * 4) = tw[q]; }   // Wf[i][j]
        psum[pg * 64 + pos] = g;
        if (pos == 0) ((float4*)sv)[pg] = g2;   // sv[0..15]: the four groups' sums of g2[k0..k0+3]; bf behind them
        if (t < EMB) sv[EMB + t] = vb;
        __syncthreads();
        if (t < 64) *(float4*)(sl + 4 * t) = f4_add(f4_add(psum[t], psum[64 + t]), f4_add(psum[128 + t], psum[192 + t]));   // sl[i][kk]
        __syncthreads();
        const int j = t >> 2, kk = t & 3;
        const float4 g2s = f4_add(f4_add(((const float4*)sv)[0], ((const float4*)sv)[1]), f4_add(((const float4*)sv)[2], ((const float4*)sv)[3]));
        const float g2k = kk == 0 ? g2s.x : kk == 1 ? g2s.y : kk == 2 ? g2s.z : g2s.w;
        float acc = sv[EMB + j] * g2k;
#pragma unroll 8
        for (int i = 0; i < EMB; ++i) acc = fmaf(big[i * LDW + j], sl[4 * i + kk], acc);
        emit(a.gw1a[c] + j * EMB + k0 + kk, s2 * acc);
    } else {                   // dbf[j] = s2 * sum_k W1a[j][k] * g2[k]
        const int p16 = t & 15, g16 = t >> 4;
        const float4 g = fold_chain(slab, np, g16, 16, EMB * EMB + EMB + 4 * p16);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int idx = q * 256 + t; *(float4*)(big + (idx >> 4) * LDW + (idx & 15) * 4) = tw[q]; }   // W1a[j][k]
        psum[g16 * 16 + p16] = g;
        __syncthreads();
        if (t < 16) {
            float4 s = psum[t];
            for (int q = 1; q < 16; ++q) s = f4_add(s, psum[q * 16 + t]);
            *(float4*)(sv + 4 * t) = s;
        }
        __syncthreads();
        if (t < EMB) {
            float acc = 0.f;
#pragma unroll 8
            for (int k = 0; k < EMB; ++k) acc = fmaf(big[t * LDW + k], sv[k], acc);
            emit(a.gbf[c] + t, s2 * acc);
        }
    }
}

static_assert(sizeof(RdArgs) + sizeof(FoldArgs) <= 4096, "k_reduce: kernel arguments beyond the 4 KB a launch carries");
__global__ __launch_bounds__(256) void k_reduce(RdArgs a, FoldArgs f) {
    __shared__ __attribute__((aligned(16))) float lds[FOLD_LDS_FLOATS];
    if ((int)blockIdx.x < FOLD_BLOCKS * f.n) { fold_block(f, a.adam, blockIdx.x, lds); return; }   // the longest blocks first
    float (*red)[EMB] = (float (*)[EMB])lds;
    const int bid = blockIdx.x - FOLD_BLOCKS * f.n;
    if (a.cdst && bid == 0 && threadIdx.x == 0) *a.cdst = a.cval;
    int ji = 0;   // binary search over up to 96 jobs
    for (int hi = a.njobs; hi - ji > 1;) {
        const int mid = (ji + hi) >> 1;
        if (bid >= a.job[mid].blk0) ji = mid; else hi = mid;
    }
    const RdJob jb = a.job[ji];
    const int chunk = bid - jb.blk0;
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


