#pragma once
#include "gcnn_common.hpp"

// ---------------------------------------------------------------------------------------------------------------
// Edge pass (K5-K7 + K9 fused, K8 hoisted): S[r] = sum_{e in seg(r)} relu(s1 * (PL[l_e] + c_e*w + PR[v_e]))
// with c_e = (coef_e + e_shift) * e_scale (the edge PreNorm, model.py:288/291).
// G = 16*SLOTS lanes cooperate on one receiver: 16 lanes x float4 cover the 64 channels, SLOTS edges in flight per
// step and 4 steps unrolled => up to 4*SLOTS independent 256-B row gathers per receiver.  The segment's (index, coef)
// pairs are loaded coalesced, one per lane, and broadcast with wave shuffles (loops have group-uniform trip counts: a
// shuffle must never read a lane that has left the loop).  Slot partial sums are combined in a fixed order.
//
// Nothing per edge is stored for the backward pass.  With J_e = (c_e*w + P_send[u_e]) + P_recv[r_e] evaluated by the same two
// instructions in both passes, the sender-ordered backward pass RECOMPUTES the ReLU pattern bit for bit from the two
// projected tables: it gathers P_recv[r_e] next to dS[r_e] (same row index, both L2-resident).  That costs one more 256-B
// gather per edge in the backward pass and saves the forward pass the per-edge mask assembly (which had doubled its time),
// 8 B/edge of stores, the cross maps between the two edge orders and their build.  The forward only counts, per receiver and
// channel, the active edges N: dS[r] is constant over a receiver's segment, so dP_recv[r] = s1*dS[r]*N[r] -- the
// receiver-ordered half of the backward pass is an element-wise epilogue of the row program that produces dS.
// ---------------------------------------------------------------------------------------------------------------
#ifndef EDGE_U
#define EDGE_U 4    // independent row gathers per lane and step, forward pass
#endif
#ifndef SEG_U
#define SEG_U 16    // rows per lane group in flight in the standalone scatter-sum pass (k_seg_sum); 4 / 8 / 16 measured: 5.52 / 5.60 / 5.80 TB/s
#endif
#ifndef EDGE_UB
#define EDGE_UB 2   // ... sender-ordered backward pass (two gathers per edge).  4 needs 100 VGPRs (four waves per SIMD): 23.7 vs 22.3 us per 800 k-edge pass
#endif
struct EdgeArgs {
    const int* seg_ptr; const int* oth; const float* coef;
    const float* p_own; const float* p_oth;       // projected table of the segment owner [n_own,64] / of the other side (gathered)
    const float* w_edge; const float* e_shift; const float* e_scale; const float* s1;
    const float* d_s;                              // send pass: dS of the forward's receivers, gathered by oth like p_oth
    float* out;                                    // S (fwd) / dP_send
    float* dw_partial;                             // send pass: [gridDim.x][64], each block's share of d w_edge (summed by k_reduce)
    float* cnt_rows;                               // fwd (COUNT): N[r] = number of active edges per channel
    int n_own;
};

template <int SLOTS>
__device__ __forceinline__ float4 slot_reduce(float4 v) {
    if (SLOTS >= 2) {
        v.x += __shfl_xor(v.x, 16); v.y += __shfl_xor(v.y, 16); v.z += __shfl_xor(v.z, 16); v.w += __shfl_xor(v.w, 16);
    }
    if (SLOTS >= 4) {
        v.x += __shfl_xor(v.x, 32); v.y += __shfl_xor(v.y, 32); v.z += __shfl_xor(v.z, 32); v.w += __shfl_xor(v.w, 32);
    }
    return v;
}

// Long segments.  With fewer than 64 lanes per segment (SLOTS < 4: graphs of low mean degree) a segment of hundreds of edges
// would keep one 16- or 32-lane group busy for dozens of dependent gather rounds while the rest of the chip has finished:
// capfac has 201 rows of 100 edges among 10,000 rows of 2 (instance_generator.py:569-647), indset Barabasi-Albert hubs
// (:44-143).  The main kernels therefore SKIP segments longer than edge_long_threshold(SLOTS); a second launch (k_edge_*_long)
// finds them -- every wave looks at rows w, w+W, w+2W, ... so that neighbouring hub rows land in different waves -- and gives
// each a whole wave.  The host launches it only when the graph has such a segment (gcnn_graph.*_max_deg; 0 = unknown).
__host__ __device__ constexpr int edge_long_threshold(int slots) { return slots >= 4 ? 0x7fffffff : 32 * slots; }

template <int SLOTS>
struct EdgeLane {   // lane geometry of a G = 16*SLOTS lane group
    int gl, gbase, slot, ch;
    __device__ __forceinline__ EdgeLane() {
        const int lane = threadIdx.x & 63;
        gl = lane % (16 * SLOTS); gbase = lane - gl; slot = gl >> 4; ch = (gl & 15) * 4;
    }
};

// Forward edge pass.  relu(s1*J) = s1*max(J,0) for s1 >= 0 and s1*min(J,0) for s1 < 0, so the scale is applied once per
// receiver.  J_e = (c_e*w + P_oth[oth_e]) + P_own[r].  COUNT also emits N (training; inference skips it).
// One segment [beg, end) of receiver r by one lane group (every lane of the group must call).
struct EdgeSum { float4 acc, cnt; };   // per lane group, after the slot reduction: sum_e max/min(J_e, 0) and the active-edge counts
// Row gathers address the table as base (uniform) + 32-bit byte offset, the offset (index << 8) made ONCE per loaded edge before it
// is broadcast, and a segment runs in FULL steps (EDGE_U * SLOTS edges, no lane masks, all gathers issued back to back) followed by
// at most one masked step: per gathered row that leaves one add of address arithmetic where 64-bit shifts, per-step bounds
// compares and exec-mask branches used to be (-25 % vector instructions per edge; tables are limited to 2^24 rows).
__device__ __forceinline__ float4 edge_row(const float* __restrict__ tab, unsigned byte_off) {
    return *(const float4*)((const char*)tab + byte_off);
}
template <bool COUNT, bool NEG>
__device__ __forceinline__ void edge_fwd_term(const float c, const float4 p, const float4 w, const float4 pown, float4& acc, unsigned (&n)[4]) {
    float h0 = fmaf(c, w.x, p.x) + pown.x, h1 = fmaf(c, w.y, p.y) + pown.y;
    float h2 = fmaf(c, w.z, p.z) + pown.z, h3 = fmaf(c, w.w, p.w) + pown.w;
    h0 = NEG ? fminf(h0, 0.f) : fmaxf(h0, 0.f); h1 = NEG ? fminf(h1, 0.f) : fmaxf(h1, 0.f);
    h2 = NEG ? fminf(h2, 0.f) : fmaxf(h2, 0.f); h3 = NEG ? fminf(h3, 0.f) : fmaxf(h3, 0.f);
    acc.x += h0; acc.y += h1; acc.z += h2; acc.w += h3;
    if (COUNT) {   // after the clamp "active" is "non-zero" (h = +-0 when clamped)
        n[0] += h0 != 0.f; n[1] += h1 != 0.f; n[2] += h2 != 0.f; n[3] += h3 != 0.f;
    }
}
template <int SLOTS, bool COUNT, bool NEG>
__device__ __forceinline__ EdgeSum edge_fwd_partial(const EdgeArgs& a, const EdgeLane<SLOTS>& L, const float4 w, const float esh,
                                                    const float esc, const int r, const int beg, const int end) {
    constexpr int G = 16 * SLOTS, STEP = EDGE_U * SLOTS;
    const int gl = L.gl, slot = L.slot, ch = L.ch;
    const float4 pown = *(const float4*)(a.p_own + (size_t)r * EMB + ch);
    const unsigned chb = 4u * ch;                       // this lane's column, in bytes
    const int src0 = (L.gbase + slot) << 2;             // ds_bpermute address of the lane holding edge `slot` of the chunk
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned n[4] = {0, 0, 0, 0};
    for (int base = beg; base < end; base += G) {
        const int e = base + gl;
        unsigned ob = 0; float c = 0.f;
        if (e < end) { ob = (unsigned)a.oth[e] << 8; c = (a.coef[e] + esh) * esc; }   // row offset in bytes, PreNorm'ed coefficient
        const int cnt = min(G, end - base);
        int i0 = 0;
        for (; i0 + STEP <= cnt; i0 += STEP) {   // full steps
            unsigned oi[EDGE_U]; float ci[EDGE_U]; float4 p[EDGE_U];
#pragma unroll
            for (int u = 0; u < EDGE_U; ++u) {
                const int src = src0 + ((i0 + u * SLOTS) << 2);
                oi[u] = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)ob);
                ci[u] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, c)));
            }
#pragma unroll
            for (int u = 0; u < EDGE_U; ++u) p[u] = edge_row(a.p_oth, oi[u] + chb);
#pragma unroll
            for (int u = 0; u < EDGE_U; ++u) edge_fwd_term<COUNT, NEG>(ci[u], p[u], w, pown, acc, n);
        }
        if (i0 < cnt) {   // the rest of the chunk: fewer than STEP edges, slots past the end are masked
            unsigned oi[EDGE_U]; float ci[EDGE_U]; bool ok[EDGE_U]; float4 p[EDGE_U];
#pragma unroll
            for (int u = 0; u < EDGE_U; ++u) {
                const int i = i0 + u * SLOTS + slot;
                ok[u] = i < cnt;
                const int src = ok[u] ? src0 + ((i0 + u * SLOTS) << 2) : (L.gbase << 2);   // by every lane: a shuffle must not sit under a lane mask
                oi[u] = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)ob);
                ci[u] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, c)));
            }
            // Wide lane groups (mean degree >= 12): unconditional gathers -- a slot past the end re-reads the row of the
            // group's first edge, unused -- because loads under a lane mask hide from the compiler how many are in flight.
            // SLOTS == 1 serves graphs of degree 2-3 (capfac, indset), where most slots of a step are past the end and the
            // wasted gathers cost more than the waits (capfac x 32: +2 % per step when unconditional).
#pragma unroll
            for (int u = 0; u < EDGE_U; ++u)
                if (SLOTS > 1 || ok[u]) p[u] = edge_row(a.p_oth, oi[u] + chb);
#pragma unroll
            for (int u = 0; u < EDGE_U; ++u)
                if (ok[u]) edge_fwd_term<COUNT, NEG>(ci[u], p[u], w, pown, acc, n);
        }
    }
    EdgeSum out;
    out.acc = slot_reduce<SLOTS>(acc);
    out.cnt = COUNT ? slot_reduce<SLOTS>(make_float4((float)n[0], (float)n[1], (float)n[2], (float)n[3])) : make_float4(0.f, 0.f, 0.f, 0.f);
    return out;
}
template <int SLOTS, bool COUNT, bool NEG>
__device__ __forceinline__ void edge_fwd_segment(const EdgeArgs& a, const float s1, const EdgeLane<SLOTS>& L, const float4 w,
                                                 const float esh, const float esc, const int r, const int beg, const int end) {
    const EdgeSum t = edge_fwd_partial<SLOTS, COUNT, NEG>(a, L, w, esh, esc, r, beg, end);
    if (L.slot == 0) {
        *(float4*)(a.out + (size_t)r * EMB + L.ch) = make_float4(s1 * t.acc.x, s1 * t.acc.y, s1 * t.acc.z, s1 * t.acc.w);
        if (COUNT) *(float4*)(a.cnt_rows + (size_t)r * EMB + L.ch) = t.cnt;
    }
}

template <int SLOTS, bool COUNT, bool NEG>
__device__ __forceinline__ void edge_fwd_impl(const EdgeArgs& a, const float s1, const int bid, const int nblk) {
    constexpr int G = 16 * SLOTS, RPW = 64 / G;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const EdgeLane<SLOTS> L;
    const float4 w = *(const float4*)(a.w_edge + L.ch);
    const float esh = *a.e_shift, esc = *a.e_scale;
    const int nwork = (a.n_own + RPW - 1) / RPW;  // one work item = one wave's RPW receivers
    for (int item = xcd_remap(bid, nblk) * 4 + wv; item < nwork; item += nblk * 4) {
        const int r = item * RPW + lane / G;
        if (r < a.n_own) {
            const int beg = a.seg_ptr[r], end = a.seg_ptr[r + 1];
            if (end - beg <= edge_long_threshold(SLOTS)) edge_fwd_segment<SLOTS, COUNT, NEG>(a, s1, L, w, esh, esc, r, beg, end);
        }
    }
}

template <int SLOTS, bool COUNT> __device__ __forceinline__ void edge_fwd_long_body(const EdgeArgs& a, int nb, int b, int thresh);
// LONG: the first `long_blocks` blocks (a multiple of 8, so that the XCD-aware remap of the others still holds) serve the segments
// the others skip, see edge_long_rows: the same launch instead of a second one (4 launches fewer per capfac / combauc step).
// A pass whose list has no long segment runs the LONG = false instantiation: the second body costs registers (a wave of occupancy
// in the backward pass) and 2 % of a setcov step when it is merely present.
template <int SLOTS, bool COUNT, bool LONG = false>
__global__ __launch_bounds__(256) void k_edge_fwd(EdgeArgs a, int long_blocks) {
    if (LONG && (int)blockIdx.x < long_blocks) { edge_fwd_long_body<SLOTS, COUNT>(a, long_blocks, blockIdx.x, edge_long_threshold(SLOTS)); return; }
    const float s1 = *a.s1;
    const int bid = blockIdx.x - (LONG ? long_blocks : 0), nblk = gridDim.x - (LONG ? long_blocks : 0);
    if (s1 < 0.f) edge_fwd_impl<SLOTS, COUNT, true>(a, s1, bid, nblk);
    else edge_fwd_impl<SLOTS, COUNT, false>(a, s1, bid, nblk);
}

// One segment [beg, end) of receiver r by ALL lane groups of a 256-thread block (4 waves x 64 / (16*SLOTS) groups): each group a
// contiguous share, the partial sums (and active-edge counts: COUNT, the training forward) added in a fixed order.  Block-uniform
// call.  (The groups run the very instantiation of edge_fwd_partial the one-group-per-segment blocks of the same kernel run.)
template <int SLOTS, bool COUNT>
__device__ __forceinline__ void edge_fwd_block_segment(const EdgeArgs& a, const float s1, const EdgeLane<SLOTS>& L, const float4 w, const float esh,
                                                       const float esc, const int r, const int beg, const int end,
                                                       float4 (*red)[16], float4 (*redn)[16]) {
    constexpr int G = 16 * SLOTS, NG = 256 / G;
    const int grp = threadIdx.x / G, lane = threadIdx.x & 63;
    const int share = (end - beg + NG - 1) / NG;
    const int b = min(end, beg + grp * share), e = min(end, b + share);
    const EdgeSum t = s1 < 0.f ? edge_fwd_partial<SLOTS, COUNT, true>(a, L, w, esh, esc, r, b, e)
                               : edge_fwd_partial<SLOTS, COUNT, false>(a, L, w, esh, esc, r, b, e);
    if (L.gl < 16) { red[grp][L.gl] = t.acc; if (COUNT) redn[grp][L.gl] = t.cnt; }
    __syncthreads();
    if (threadIdx.x < 16) {
        float4 p = red[0][lane];
#pragma unroll
        for (int k = 1; k < NG; ++k) { const float4 q = red[k][lane]; p.x += q.x; p.y += q.y; p.z += q.z; p.w += q.w; }
        *(float4*)(a.out + (size_t)r * EMB + 4 * lane) = make_float4(s1 * p.x, s1 * p.y, s1 * p.z, s1 * p.w);
    } else if (COUNT && threadIdx.x >= 64 && threadIdx.x < 80) {   // counts: integer-valued, any order gives the same bits
        float4 p = redn[0][lane];
#pragma unroll
        for (int k = 1; k < NG; ++k) { const float4 q = redn[k][lane]; p.x += q.x; p.y += q.y; p.z += q.z; p.w += q.w; }
        *(float4*)(a.cnt_rows + (size_t)r * EMB + 4 * lane) = p;
    }
    __syncthreads();
}
// A few long segments (the cut rows of conv v->k: a few dozen to a couple of thousand cuts of 10-200 nonzeros each): a wave per
// segment would walk up to a dozen dependent gather rounds with most of the chip idle.  Here a 4-wave block serves one segment.
template <bool COUNT>
__device__ __forceinline__ void edge_fwd_block_body(const EdgeArgs& a, const int bid, const int nblk) {
    __shared__ float4 red[4][16], redn[COUNT ? 4 : 1][16];
    const float s1 = *a.s1;
    const EdgeLane<4> L;
    const float4 w = *(const float4*)(a.w_edge + L.ch);
    const float esh = *a.e_shift, esc = *a.e_scale;
    // a block per segment (nblk = n_own): every XCD takes a contiguous range of segments, so the rows a range gathers -- the
    // variable rows of its own samples -- stay in that XCD's L2 (round-robin placement had every XCD fetch every sample's table:
    // 37 MB of HBM reads per pass at setcov x 32 instead of 11)
    for (int i = bid; i < a.n_own; i += nblk) {
        const int r = nblk == a.n_own ? xcd_remap(i, nblk) : i;
        edge_fwd_block_segment<4, COUNT>(a, s1, L, w, esh, esc, r, a.seg_ptr[r], a.seg_ptr[r + 1], red, redn);
    }
}
template <bool COUNT>
__global__ __launch_bounds__(256) void k_edge_fwd_block(EdgeArgs a) { edge_fwd_block_body<COUNT>(a, blockIdx.x, gridDim.x); }

// The rows the main blocks leave out (longer than `thresh`): found and served by whole BLOCKS of LONG_NW waves.  Thread t of block b
// (of nb) looks at rows (q + t) * nb + b, q = 0, LONG_NW*64, ... -- neighbouring hub rows (capfac's 201 long rows open every
// sample) land in different blocks -- the block lists its long rows in row order (ballots: a fixed order) and serves them one
// after the other, all its lane groups on one row (`body(r, beg, end)`, block-uniform): a 493-edge hub row of a combauc batch is a
// handful of gather rounds per group instead of 31 for one wave alone.
#define LONG_NW 4
template <class Body>
__device__ __forceinline__ void edge_long_rows(const int* __restrict__ seg_ptr, int n_own, int thresh, int nb, int blk, Body body) {
    constexpr int NT = 64 * LONG_NW;
    __shared__ int lr[NT], lb[NT], le[NT], wcnt[LONG_NW];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const long long NB = nb, b = blk;
    for (long long q0 = 0; q0 * NB + b < n_own; q0 += NT) {
        const long long rr = (q0 + t) * NB + b;
        const int r = rr < n_own ? (int)rr : -1;
        int beg = 0, end = 0;
        if (r >= 0) { beg = seg_ptr[r]; end = seg_ptr[r + 1]; }
        const bool is_long = end - beg > thresh;
        const unsigned long long bal = __ballot(is_long);
        if (lane == 0) wcnt[wv] = __popcll(bal);
        __syncthreads();
        int pos = __popcll(bal & ((1ull << lane) - 1)), total = 0;
#pragma unroll
        for (int k = 0; k < LONG_NW; ++k) { if (k < wv) pos += wcnt[k]; total += wcnt[k]; }
        if (is_long) { lr[pos] = r; lb[pos] = beg; le[pos] = end; }
        __syncthreads();
        for (int i = 0; i < total; ++i) body(lr[i], lb[i], le[i]);
        __syncthreads();   // the lists are rewritten by the next round
    }
}
template <int SLOTS, bool COUNT>
__device__ __forceinline__ void edge_fwd_long_body(const EdgeArgs& a, int nb, int b, int thresh) {
    constexpr int NG = 256 / (16 * SLOTS);
    __shared__ float4 red[NG][16], redn[COUNT ? NG : 1][16];
    const float s1 = *a.s1;
    const EdgeLane<SLOTS> L;
    const float4 w = *(const float4*)(a.w_edge + L.ch);
    const float esh = *a.e_shift, esc = *a.e_scale;
    edge_long_rows(a.seg_ptr, a.n_own, thresh, nb, b, [&](int r, int beg, int end) {
        edge_fwd_block_segment<SLOTS, COUNT>(a, s1, L, w, esh, esc, r, beg, end, red, redn);
    });
}

// Backward, receiver-ordered half, element-wise: dP_recv[r] = s1*dS[r]*N[r].  (The model fuses this into the epilogue of
// the row program that produces dS; this kernel serves the per-op entry point.)
__global__ __launch_bounds__(256) void k_edge_bwd_recv(const float* __restrict__ d_s, const float* __restrict__ nrows,
                                                       const float* __restrict__ s1p, float* __restrict__ d_p, int n4) {
    const float s1 = *s1p;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
        const float4 d = ((const float4*)d_s)[i], nn = ((const float4*)nrows)[i];
        ((float4*)d_p)[i] = make_float4(s1 * d.x * nn.x, s1 * d.y * nn.y, s1 * d.z * nn.z, s1 * d.w * nn.w);
    }
}

// Backward, sender-ordered half.  Segment owner = the SENDING node u; per edge the receiver r = oth[e]:
//   J_e = (c_e*w + P_send[u]) + P_recv[r]   -- the forward's expression, instruction for instruction: same bits
//   t_e = [s1*J_e > 0] * dS[r]
//   dP_send[u] = s1 * sum_{e in seg(u)} t_e          d w_edge = s1 * sum_e c_e * t_e   (per-block partials, see edge_dw_block_store)
// two 256-B row gathers per edge (dS and P_recv, same row index), nothing else.
// Returns the segment's share of d w_edge (before the s1 factor), the same value in every lane of the group.
template <bool NEG>
__device__ __forceinline__ void edge_bwd_term(const float c, const float4 d, const float4 q, const float4 w, const float4 psend, float4& acc, float4& dw) {
    const float j0 = fmaf(c, w.x, psend.x) + q.x, j1 = fmaf(c, w.y, psend.y) + q.y;
    const float j2 = fmaf(c, w.z, psend.z) + q.z, j3 = fmaf(c, w.w, psend.w) + q.w;
    const float t0 = (NEG ? j0 < 0.f : j0 > 0.f) ? d.x : 0.f, t1 = (NEG ? j1 < 0.f : j1 > 0.f) ? d.y : 0.f;
    const float t2 = (NEG ? j2 < 0.f : j2 > 0.f) ? d.z : 0.f, t3 = (NEG ? j3 < 0.f : j3 > 0.f) ? d.w : 0.f;
    acc.x += t0; acc.y += t1; acc.z += t2; acc.w += t3;
    dw.x = fmaf(c, t0, dw.x); dw.y = fmaf(c, t1, dw.y); dw.z = fmaf(c, t2, dw.z); dw.w = fmaf(c, t3, dw.w);
}
struct BwdSum { float4 acc, dw; };   // per lane group, after the slot reduction: sum_e t_e and sum_e c_e*t_e (both before the s1 factor)
template <int SLOTS, bool NEG>
__device__ __forceinline__ BwdSum edge_bwd_send_partial(const EdgeArgs& a, const EdgeLane<SLOTS>& L, const float4 w,
                                                        const float esh, const float esc, const int u, const int beg, const int end) {
    constexpr int G = 16 * SLOTS, STEP = EDGE_UB * SLOTS;
    const int gl = L.gl, slot = L.slot, ch = L.ch;
    const float4 psend = *(const float4*)(a.p_own + (size_t)u * EMB + ch);
    const unsigned chb = 4u * ch;
    const int src0 = (L.gbase + slot) << 2;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), dw = acc;
    for (int base = beg; base < end; base += G) {
        const int e = base + gl;
        unsigned ob = 0; float c = 0.f;
        if (e < end) { ob = (unsigned)a.oth[e] << 8; c = (a.coef[e] + esh) * esc; }
        const int cnt = min(G, end - base);
        int i0 = 0;
        for (; i0 + STEP <= cnt; i0 += STEP) {   // full steps (see edge_fwd_partial)
            unsigned oi[EDGE_UB]; float ci[EDGE_UB]; float4 d[EDGE_UB], q[EDGE_UB];
#pragma unroll
            for (int v = 0; v < EDGE_UB; ++v) {
                const int src = src0 + ((i0 + v * SLOTS) << 2);
                oi[v] = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)ob);
                ci[v] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, c)));
            }
#pragma unroll
            for (int v = 0; v < EDGE_UB; ++v) { d[v] = edge_row(a.d_s, oi[v] + chb); q[v] = edge_row(a.p_oth, oi[v] + chb); }
#pragma unroll
            for (int v = 0; v < EDGE_UB; ++v) edge_bwd_term<NEG>(ci[v], d[v], q[v], w, psend, acc, dw);
        }
        if (i0 < cnt) {
            unsigned oi[EDGE_UB]; float ci[EDGE_UB]; bool ok[EDGE_UB]; float4 d[EDGE_UB], q[EDGE_UB];
#pragma unroll
            for (int v = 0; v < EDGE_UB; ++v) {
                const int i = i0 + v * SLOTS + slot;
                ok[v] = i < cnt;
                const int src = ok[v] ? src0 + ((i0 + v * SLOTS) << 2) : (L.gbase << 2);   // by every lane: a shuffle must not sit under a lane mask
                oi[v] = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)ob);
                ci[v] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, c)));
            }
#pragma unroll
            for (int v = 0; v < EDGE_UB; ++v)   // unconditional for wide lane groups, see edge_fwd_partial
                if (SLOTS > 1 || ok[v]) { d[v] = edge_row(a.d_s, oi[v] + chb); q[v] = edge_row(a.p_oth, oi[v] + chb); }
#pragma unroll
            for (int v = 0; v < EDGE_UB; ++v)
                if (ok[v]) edge_bwd_term<NEG>(ci[v], d[v], q[v], w, psend, acc, dw);
        }
    }
    BwdSum out;
    out.acc = slot_reduce<SLOTS>(acc); out.dw = slot_reduce<SLOTS>(dw);
    return out;
}
// ... of a whole segment by one lane group: stores dP_send[u], returns the segment's share of d w_edge
template <int SLOTS, bool NEG>
__device__ __forceinline__ float4 edge_bwd_send_segment(const EdgeArgs& a, const float s1, const EdgeLane<SLOTS>& L, const float4 w,
                                                        const float esh, const float esc, const int u, const int beg, const int end) {
    const BwdSum t = edge_bwd_send_partial<SLOTS, NEG>(a, L, w, esh, esc, u, beg, end);
    if (L.slot == 0) *(float4*)(a.out + (size_t)u * EMB + L.ch) = make_float4(s1 * t.acc.x, s1 * t.acc.y, s1 * t.acc.z, s1 * t.acc.w);
    return t.dw;
}
// d w_edge = s1 * sum over ALL edges of c_e * t_e.  Every lane group adds up the shares of the segments it serves (a fixed
// assignment, hence a fixed order), the groups of a wave and the four waves of a block are combined in a fixed order and the
// block stores ONE 64-float partial -- instead of a [n_send,64] matrix written here and column-summed by another kernel.
template <int SLOTS, int NW = 4>
__device__ __forceinline__ void edge_dw_block_store(float4 dw, const float s1, float* __restrict__ dst, const int ch) {
    __shared__ float4 red[NW][16];
    if (SLOTS <= 2) { dw.x += __shfl_xor(dw.x, 32); dw.y += __shfl_xor(dw.y, 32); dw.z += __shfl_xor(dw.z, 32); dw.w += __shfl_xor(dw.w, 32); }
    if (SLOTS == 1) { dw.x += __shfl_xor(dw.x, 16); dw.y += __shfl_xor(dw.y, 16); dw.z += __shfl_xor(dw.z, 16); dw.w += __shfl_xor(dw.w, 16); }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane < 16) red[wv][lane] = dw;
    __syncthreads();
    if (threadIdx.x < 16) {
        float4 p = make_float4((red[0][lane].x + red[1][lane].x) + (red[2][lane].x + red[3][lane].x), (red[0][lane].y + red[1][lane].y) + (red[2][lane].y + red[3][lane].y),
                               (red[0][lane].z + red[1][lane].z) + (red[2][lane].z + red[3][lane].z), (red[0][lane].w + red[1][lane].w) + (red[2][lane].w + red[3][lane].w));
#pragma unroll
        for (int k = 4; k < NW; ++k) { const float4 q = red[k][lane]; p.x += q.x; p.y += q.y; p.z += q.z; p.w += q.w; }
        *(float4*)(dst + ch) = make_float4(s1 * p.x, s1 * p.y, s1 * p.z, s1 * p.w);
    }
}
template <int SLOTS, bool NEG>
__device__ __forceinline__ void edge_bwd_send_impl(const EdgeArgs& a, const float s1, const int bid, const int nblk) {
    constexpr int G = 16 * SLOTS, RPW = 64 / G;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const EdgeLane<SLOTS> L;
    const float esh = *a.e_shift, esc = *a.e_scale;
    const float4 w = *(const float4*)(a.w_edge + L.ch);
    const int nwork = (a.n_own + RPW - 1) / RPW;
    float4 dwsum = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int item = xcd_remap(bid, nblk) * 4 + wv; item < nwork; item += nblk * 4) {
        const int u = item * RPW + lane / G;
        if (u < a.n_own) {
            const int beg = a.seg_ptr[u], end = a.seg_ptr[u + 1];
            if (end - beg <= edge_long_threshold(SLOTS)) {
                const float4 dw = edge_bwd_send_segment<SLOTS, NEG>(a, s1, L, w, esh, esc, u, beg, end);
                dwsum.x += dw.x; dwsum.y += dw.y; dwsum.z += dw.z; dwsum.w += dw.w;
            }
        }
    }
    edge_dw_block_store<SLOTS>(dwsum, s1, a.dw_partial + (size_t)blockIdx.x * EMB, L.ch);
}
template <int SLOTS>
__device__ __forceinline__ void edge_bwd_send_long_body(const EdgeArgs& a, int nb, int blk, int thresh) {
    constexpr int G = 16 * SLOTS, NG = 256 / G;
    __shared__ float4 red[NG][16];
    const float s1 = *a.s1;
    const EdgeLane<SLOTS> L;
    const float4 w = *(const float4*)(a.w_edge + L.ch);
    const float esh = *a.e_shift, esc = *a.e_scale;
    const int grp = threadIdx.x / G, lane = threadIdx.x & 63;
    float4 dwsum = make_float4(0.f, 0.f, 0.f, 0.f);   // this lane group's share of d w_edge over every long row of the block
    edge_long_rows(a.seg_ptr, a.n_own, thresh, nb, blk, [&](int u, int beg, int end) {
        const int share = (end - beg + NG - 1) / NG;
        const int b = min(end, beg + grp * share), e = min(end, b + share);
        const BwdSum t = s1 < 0.f ? edge_bwd_send_partial<SLOTS, true>(a, L, w, esh, esc, u, b, e) : edge_bwd_send_partial<SLOTS, false>(a, L, w, esh, esc, u, b, e);
        dwsum.x += t.dw.x; dwsum.y += t.dw.y; dwsum.z += t.dw.z; dwsum.w += t.dw.w;
        if (L.gl < 16) red[grp][L.gl] = t.acc;
        __syncthreads();
        if (threadIdx.x < 16) {
            float4 p = red[0][lane];
#pragma unroll
            for (int k = 1; k < NG; ++k) { const float4 q = red[k][lane]; p.x += q.x; p.y += q.y; p.z += q.z; p.w += q.w; }
            *(float4*)(a.out + (size_t)u * EMB + 4 * lane) = make_float4(s1 * p.x, s1 * p.y, s1 * p.z, s1 * p.w);
        }
        __syncthreads();
    });
    edge_dw_block_store<SLOTS>(dwsum, s1, a.dw_partial + (size_t)blockIdx.x * EMB, L.ch);
}
// every block (long-row blocks first, see k_edge_fwd) leaves one partial row of d w_edge at dw_partial[blockIdx.x]
template <int SLOTS, bool LONG = false>
__global__ __launch_bounds__(256) void k_edge_bwd_send(EdgeArgs a, int long_blocks) {
    if (LONG && (int)blockIdx.x < long_blocks) { edge_bwd_send_long_body<SLOTS>(a, long_blocks, blockIdx.x, edge_long_threshold(SLOTS)); return; }
    const float s1 = *a.s1;
    const int bid = blockIdx.x - (LONG ? long_blocks : 0), nblk = gridDim.x - (LONG ? long_blocks : 0);
    if (s1 < 0.f) edge_bwd_send_impl<SLOTS, true>(a, s1, bid, nblk); else edge_bwd_send_impl<SLOTS, false>(a, s1, bid, nblk);
}

// ---------------------------------------------------------------------------------------------------------------
// K9 standalone: the scatter-sum pass as the reference defines it (tf.scatter_nd over [E,64] messages,
// model.py:568-569) on receiver-sorted segments.  Pure streaming: 260 B/edge in, 256 B/receiver out.
// ---------------------------------------------------------------------------------------------------------------
template <int SLOTS, bool PERM>
__global__ __launch_bounds__(256) void k_seg_sum(const float* __restrict__ msg, const int* __restrict__ seg_ptr,
                                                 const int* __restrict__ perm, int n_recv, float* __restrict__ out) {
    constexpr int G = 16 * SLOTS, RPW = 64 / G;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gl = lane % G, slot = gl >> 4, ch = (gl & 15) * 4;
    const int nwork = (n_recv + RPW - 1) / RPW;
    for (int item = blockIdx.x * 4 + wv; item < nwork; item += gridDim.x * 4) {
        const int r = item * RPW + lane / G;
        if (r >= n_recv) continue;
        const int beg = seg_ptr[r], end = seg_ptr[r + 1];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // SEG_U rows per lane group in flight; the loads are unconditional (rows past the segment re-read its last row and are
        // not added): under a lane mask the compiler cannot count what is in flight and drains every round before the next
        for (int e0 = beg + slot; e0 < end; e0 += SEG_U * SLOTS) {
            float4 p[SEG_U];
#pragma unroll
            for (int u = 0; u < SEG_U; ++u) {
                const int e = min(e0 + u * SLOTS, end - 1);
                const size_t row = PERM ? (size_t)perm[e] : (size_t)e;
                p[u] = *(const float4*)(msg + row * EMB + ch);
            }
#pragma unroll
            for (int u = 0; u < SEG_U; ++u)
                if (e0 + u * SLOTS < end) { acc.x += p[u].x; acc.y += p[u].y; acc.z += p[u].z; acc.w += p[u].w; }
        }
        acc = slot_reduce<SLOTS>(acc);
        if (slot == 0) *(float4*)(out + (size_t)r * EMB + ch) = acc;
    }
}

// transpose of the pass (gradient of tf.scatter_nd = row gather): d_msg[row(e)] = d_out[recv(e)]
template <bool PERM>
__global__ __launch_bounds__(256) void k_seg_bcast(const float* __restrict__ d_out, const int* __restrict__ seg_ptr,
                                                   const int* __restrict__ perm, int n_recv, float* __restrict__ d_msg) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int slot = lane >> 4, ch = (lane & 15) * 4;
    for (int r = blockIdx.x * 4 + wv; r < n_recv; r += gridDim.x * 4) {
        const int beg = seg_ptr[r], end = seg_ptr[r + 1];
        const float4 v = *(const float4*)(d_out + (size_t)r * EMB + ch);
        for (int e = beg + slot; e < end; e += 4) {
            const size_t row = PERM ? (size_t)perm[e] : (size_t)e;
            *(float4*)(d_msg + row * EMB + ch) = v;
        }
    }
}

