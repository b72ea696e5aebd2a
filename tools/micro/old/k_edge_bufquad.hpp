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
struct EdgeArgs {
    const int* seg_ptr; const int* oth; const float* coef;
    const float* p_own; const float* p_oth;       // projected table of the segment owner [n_own,64] / of the other side [n_oth,64] (gathered)
    const float* w_edge; const float* e_shift; const float* e_scale; const float* s1;
    const float* d_s;                              // send pass: dS [n_oth,64], gathered by oth like p_oth
    float* out;                                    // S (fwd) / dP_send
    float* dw_rows;                                // send pass: Q [n_own,64], per-sender share of d w_edge
    float* cnt_rows;                               // fwd (COUNT): N[r] = number of active edges per channel
    int n_own, n_oth, n_edges;
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

// Buffer (SRD) addressing for everything an edge pass gathers: a 32-bit byte offset per lane instead of a 64-bit address
// (one VALU op per row gather), and the hardware range check makes reads past the end of a list return 0 -- the index
// quads below are fetched without looking at the segment end first.  Tables must stay below 4 GB (checked on the host).
// (The intrinsics are declared by their LLVM names: this compiler's __builtin_amdgcn_raw_buffer_load_b128 lowers to a
// one-dword load.)
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4e __attribute__((ext_vector_type(4)));
__device__ f32x4e gcnn_buf_load_f4(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ i32x4 gcnn_buf_load_i4(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4i32");
typedef i32x4 edge_rsrc_t;
__device__ __forceinline__ edge_rsrc_t edge_rsrc(const void* p, unsigned bytes) {   // raw buffer: base, no stride, byte range
    const unsigned long long a = (unsigned long long)p;
    edge_rsrc_t r;
    r.x = (int)(unsigned)a; r.y = (int)((unsigned)(a >> 32) & 0xffffu); r.z = (int)bytes; r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ float4 ld_row4(edge_rsrc_t r, unsigned byte_off) {
    const f32x4e v = gcnn_buf_load_f4(r, (int)byte_off, 0, 0);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ i32x4 ld_idx4(edge_rsrc_t r, int byte_off) { return gcnn_buf_load_i4(r, byte_off, 0, 0); }

// Work decomposition shared by both passes.  G = 16*SLOTS lanes serve one segment (16 lanes x float4 = the 64 channels);
// per iteration each of the SLOTS 16-lane groups takes FOUR CONSECUTIVE edges: one (dword-aligned) 16-byte load fetches
// their four indices, another their four coefficients -- the same address in all 16 lanes, one request -- and the quad of
// the next iteration is requested before the current rows are gathered.  No cross-lane traffic and no LDS inside the loop,
// so segments of a wave may run different trip counts; a group's last, partial quad goes through a predicated copy of the
// body.  SLOTS = 4: one wave per segment, 16 row gathers in flight per iteration.

// Forward edge pass.  relu(s1*J) = s1*max(J,0) for s1 >= 0 and s1*min(J,0) for s1 < 0, so the scale is applied once per
// receiver.  J_e = (c_e*w + P_oth[oth_e]) + P_own[r].  COUNT also emits N (training; inference skips it).
template <int SLOTS, bool COUNT, bool NEG>
__device__ __forceinline__ void edge_fwd_impl(const EdgeArgs& a, const float s1) {
    constexpr int G = 16 * SLOTS, RPW = 64 / G, EPI = 4 * SLOTS;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gl = lane % G, slot = gl >> 4, cl = gl & 15, ch = cl * 4;
    const float4 w = *(const float4*)(a.w_edge + ch);
    const float esh = *a.e_shift, esc = *a.e_scale;
    const edge_rsrc_t r_oth = edge_rsrc(a.oth, (unsigned)a.n_edges * 4u), r_coef = edge_rsrc(a.coef, (unsigned)a.n_edges * 4u);
    const edge_rsrc_t r_tab = edge_rsrc(a.p_oth, (unsigned)a.n_oth * 256u);
    const unsigned lane_off = (unsigned)ch * 4u;
    const int nwork = (a.n_own + RPW - 1) / RPW;  // one work item = one wave's RPW receivers
    for (int item = xcd_remap(blockIdx.x, gridDim.x) * 4 + wv; item < nwork; item += gridDim.x * 4) {
        const int r = item * RPW + lane / G;
        if (r < a.n_own) {
            const int end = a.seg_ptr[r + 1];
            int e = a.seg_ptr[r] + 4 * slot;      // this group's first edge of the current iteration
            i32x4 idx = ld_idx4(r_oth, e * 4);
            float4 cf = ld_row4(r_coef, (unsigned)e * 4u);
            const float4 pown = *(const float4*)(a.p_own + (size_t)r * EMB + ch);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            unsigned n0 = 0, n1 = 0, n2 = 0, n3 = 0;
            auto edge = [&](const float4& p, float c) {
                float h0 = fmaf(c, w.x, p.x) + pown.x, h1 = fmaf(c, w.y, p.y) + pown.y;
                float h2 = fmaf(c, w.z, p.z) + pown.z, h3 = fmaf(c, w.w, p.w) + pown.w;
                h0 = NEG ? fminf(h0, 0.f) : fmaxf(h0, 0.f); h1 = NEG ? fminf(h1, 0.f) : fmaxf(h1, 0.f);
                h2 = NEG ? fminf(h2, 0.f) : fmaxf(h2, 0.f); h3 = NEG ? fminf(h3, 0.f) : fmaxf(h3, 0.f);
                acc.x += h0; acc.y += h1; acc.z += h2; acc.w += h3;
                if (COUNT) {   // after the clamp "active" is "non-zero" (h = +-0 when clamped)
                    n0 += h0 != 0.f; n1 += h1 != 0.f; n2 += h2 != 0.f; n3 += h3 != 0.f;
                }
            };
            while (e + 3 < end) {      // full quads
                const int en = e + EPI;
                const i32x4 idx_n = ld_idx4(r_oth, en * 4);
                const float4 cf_n = ld_row4(r_coef, (unsigned)en * 4u);
                const float4 p0 = ld_row4(r_tab, (unsigned)idx.x * 256u + lane_off), p1 = ld_row4(r_tab, (unsigned)idx.y * 256u + lane_off);
                const float4 p2 = ld_row4(r_tab, (unsigned)idx.z * 256u + lane_off), p3 = ld_row4(r_tab, (unsigned)idx.w * 256u + lane_off);
                edge(p0, (cf.x + esh) * esc); edge(p1, (cf.y + esh) * esc);
                edge(p2, (cf.z + esh) * esc); edge(p3, (cf.w + esh) * esc);
                e = en; idx = idx_n; cf = cf_n;
            }
            if (e < end) {             // this group's last quad: 1-3 edges
                const int m = end - e;
                float4 p0, p1 = make_float4(0.f, 0.f, 0.f, 0.f), p2 = p1;
                p0 = ld_row4(r_tab, (unsigned)idx.x * 256u + lane_off);
                if (m > 1) p1 = ld_row4(r_tab, (unsigned)idx.y * 256u + lane_off);
                if (m > 2) p2 = ld_row4(r_tab, (unsigned)idx.z * 256u + lane_off);
                edge(p0, (cf.x + esh) * esc);
                if (m > 1) edge(p1, (cf.y + esh) * esc);
                if (m > 2) edge(p2, (cf.z + esh) * esc);
            }
            acc = slot_reduce<SLOTS>(acc);
            if (slot == 0) *(float4*)(a.out + (size_t)r * EMB + ch) = make_float4(s1 * acc.x, s1 * acc.y, s1 * acc.z, s1 * acc.w);
            if (COUNT) {
                const float4 nacc = slot_reduce<SLOTS>(make_float4((float)n0, (float)n1, (float)n2, (float)n3));
                if (slot == 0) *(float4*)(a.cnt_rows + (size_t)r * EMB + ch) = nacc;
            }
        }
    }
}

#ifndef EDGE_FWD_WAVES
#define EDGE_FWD_WAVES 1
#endif
#ifndef EDGE_BWD_WAVES
#define EDGE_BWD_WAVES 1
#endif
template <int SLOTS, bool COUNT>
__global__ __launch_bounds__(256, EDGE_FWD_WAVES) void k_edge_fwd(EdgeArgs a) {
    const float s1 = *a.s1;
    if (s1 < 0.f) edge_fwd_impl<SLOTS, COUNT, true>(a, s1); else edge_fwd_impl<SLOTS, COUNT, false>(a, s1);
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
//   dP_send[u] = s1 * sum_{e in seg(u)} t_e          Q[u] = s1 * sum_{e in seg(u)} c_e * t_e   (share of d w_edge)
// two 256-B row gathers per edge (dS and P_recv, same row index), nothing else.
template <int SLOTS, bool NEG>
__device__ __forceinline__ void edge_bwd_send_impl(const EdgeArgs& a, const float s1) {
    constexpr int G = 16 * SLOTS, RPW = 64 / G, EPI = 4 * SLOTS;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gl = lane % G, slot = gl >> 4, cl = gl & 15, ch = cl * 4;
    const float esh = *a.e_shift, esc = *a.e_scale;
    const float4 w = *(const float4*)(a.w_edge + ch);
    const edge_rsrc_t r_oth = edge_rsrc(a.oth, (unsigned)a.n_edges * 4u), r_coef = edge_rsrc(a.coef, (unsigned)a.n_edges * 4u);
    const edge_rsrc_t r_tab = edge_rsrc(a.p_oth, (unsigned)a.n_oth * 256u), r_ds = edge_rsrc(a.d_s, (unsigned)a.n_oth * 256u);
    const unsigned lane_off = (unsigned)ch * 4u;
    const int nwork = (a.n_own + RPW - 1) / RPW;
    for (int item = xcd_remap(blockIdx.x, gridDim.x) * 4 + wv; item < nwork; item += gridDim.x * 4) {
        const int u = item * RPW + lane / G;
        if (u < a.n_own) {
            const int end = a.seg_ptr[u + 1];
            int e = a.seg_ptr[u] + 4 * slot;
            i32x4 idx = ld_idx4(r_oth, e * 4);
            float4 cf = ld_row4(r_coef, (unsigned)e * 4u);
            const float4 psend = *(const float4*)(a.p_own + (size_t)u * EMB + ch);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), dw = acc;
            auto edge = [&](const float4& d, const float4& q, float c) {
                const float j0 = fmaf(c, w.x, psend.x) + q.x, j1 = fmaf(c, w.y, psend.y) + q.y;
                const float j2 = fmaf(c, w.z, psend.z) + q.z, j3 = fmaf(c, w.w, psend.w) + q.w;
                const float t0 = (NEG ? j0 < 0.f : j0 > 0.f) ? d.x : 0.f, t1 = (NEG ? j1 < 0.f : j1 > 0.f) ? d.y : 0.f;
                const float t2 = (NEG ? j2 < 0.f : j2 > 0.f) ? d.z : 0.f, t3 = (NEG ? j3 < 0.f : j3 > 0.f) ? d.w : 0.f;
                acc.x += t0; acc.y += t1; acc.z += t2; acc.w += t3;
                dw.x = fmaf(c, t0, dw.x); dw.y = fmaf(c, t1, dw.y); dw.z = fmaf(c, t2, dw.z); dw.w = fmaf(c, t3, dw.w);
            };
            while (e + 3 < end) {
                const int en = e + EPI;
                const i32x4 idx_n = ld_idx4(r_oth, en * 4);
                const float4 cf_n = ld_row4(r_coef, (unsigned)en * 4u);
                const unsigned o0 = (unsigned)idx.x * 256u + lane_off, o1 = (unsigned)idx.y * 256u + lane_off, o2 = (unsigned)idx.z * 256u + lane_off, o3 = (unsigned)idx.w * 256u + lane_off;
                const float4 d0 = ld_row4(r_ds, o0), q0 = ld_row4(r_tab, o0), d1 = ld_row4(r_ds, o1), q1 = ld_row4(r_tab, o1);
                const float4 d2 = ld_row4(r_ds, o2), q2 = ld_row4(r_tab, o2), d3 = ld_row4(r_ds, o3), q3 = ld_row4(r_tab, o3);
                edge(d0, q0, (cf.x + esh) * esc); edge(d1, q1, (cf.y + esh) * esc);
                edge(d2, q2, (cf.z + esh) * esc); edge(d3, q3, (cf.w + esh) * esc);
                e = en; idx = idx_n; cf = cf_n;
            }
            if (e < end) {
                const int m = end - e;
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                float4 d0, q0, d1 = z, q1 = z, d2 = z, q2 = z;
                const unsigned o0 = (unsigned)idx.x * 256u + lane_off, o1 = (unsigned)idx.y * 256u + lane_off, o2 = (unsigned)idx.z * 256u + lane_off;
                d0 = ld_row4(r_ds, o0); q0 = ld_row4(r_tab, o0);
                if (m > 1) { d1 = ld_row4(r_ds, o1); q1 = ld_row4(r_tab, o1); }
                if (m > 2) { d2 = ld_row4(r_ds, o2); q2 = ld_row4(r_tab, o2); }
                edge(d0, q0, (cf.x + esh) * esc);
                if (m > 1) edge(d1, q1, (cf.y + esh) * esc);
                if (m > 2) edge(d2, q2, (cf.z + esh) * esc);
            }
            acc = slot_reduce<SLOTS>(acc); dw = slot_reduce<SLOTS>(dw);
            if (slot == 0) {
                *(float4*)(a.out + (size_t)u * EMB + ch) = make_float4(s1 * acc.x, s1 * acc.y, s1 * acc.z, s1 * acc.w);
                *(float4*)(a.dw_rows + (size_t)u * EMB + ch) = make_float4(s1 * dw.x, s1 * dw.y, s1 * dw.z, s1 * dw.w);
            }
        }
    }
}
template <int SLOTS>
__global__ __launch_bounds__(256, EDGE_BWD_WAVES) void k_edge_bwd_send(EdgeArgs a) {
    const float s1 = *a.s1;
    if (s1 < 0.f) edge_bwd_send_impl<SLOTS, true>(a, s1); else edge_bwd_send_impl<SLOTS, false>(a, s1);
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
        for (int e0 = beg + slot; e0 < end; e0 += 4 * SLOTS) {
            float4 p[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * SLOTS;
                p[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < end) {
                    const size_t row = PERM ? (size_t)perm[e] : (size_t)e;
                    p[u] = *(const float4*)(msg + row * EMB + ch);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { acc.x += p[u].x; acc.y += p[u].y; acc.z += p[u].z; acc.w += p[u].w; }
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

