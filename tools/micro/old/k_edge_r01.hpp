#pragma once
// (gcnn_common.hpp comes from the including file)

// ---------------------------------------------------------------------------------------------------------------
// Edge pass (K5-K7 + K9 fused, K8 hoisted): S[r] = sum_{e in seg(r)} relu(s1 * (PL[l_e] + c_e*w + PR[v_e]))
// with c_e = (coef_e + e_shift) * e_scale (the edge PreNorm, model.py:288/291).
// G = 16*SLOTS lanes cooperate on one receiver: 16 lanes x float4 cover the 64 channels, SLOTS edges in flight per
// step and 4 steps unrolled => up to 4*SLOTS independent 256-B row gathers per receiver.  The segment's (index, coef)
// pairs are loaded coalesced, one per lane, and broadcast with wave shuffles (loops have group-uniform trip counts: a
// shuffle must never read a lane that has left the loop).  Slot partial sums are combined in a fixed order.
// ---------------------------------------------------------------------------------------------------------------
struct EdgeArgs {
    const int* seg_ptr; const int* oth; const float* coef;
    const float* p_recv; const float* p_oth;      // forward: projected tables of the segment owner [R,64] / the gathered side
    const float* w_edge; const float* e_shift; const float* e_scale; const float* s1;
    const float* d_s;                              // send pass: dS [R,64], gathered by oth
    const int* xpos;                               // send pass: position of each edge in the receiver-ordered list
    unsigned long long* mask;                      // [E] ReLU bits in receiver order (nibble c = channels 4c..4c+3): fwd writes, send pass reads
    float* out;                                    // S (fwd) / dP_send
    float* dw_rows;                                // send pass: Q [n_send,64], per-sender share of d w_edge
    float* cnt_rows;                               // fwd (SAVE): N[r] = number of active edges per channel
    int n_recv;
};

template <int SLOTS>
__device__ __forceinline__ float4 slot_reduce_r01(float4 v) {
    if (SLOTS >= 2) {
        v.x += __shfl_xor(v.x, 16); v.y += __shfl_xor(v.y, 16); v.z += __shfl_xor(v.z, 16); v.w += __shfl_xor(v.w, 16);
    }
    if (SLOTS >= 4) {
        v.x += __shfl_xor(v.x, 32); v.y += __shfl_xor(v.y, 32); v.z += __shfl_xor(v.z, 32); v.w += __shfl_xor(v.w, 32);
    }
    return v;
}

// Forward edge pass.  relu(s1*J) = s1*max(J,0) for s1 >= 0 and s1*min(J,0) for s1 < 0, so the scale is applied once per
// receiver.  J_e = (c_e*w + P_oth[oth_e]) + P_own[r].
// SAVE also emits what the backward pass needs: per edge one 64-bit word (nibble c = the ReLU bits of channels 4c..4c+3,
// gathered from the edge's 16 lanes through a per-wave LDS scratch) and per receiver/channel the number N of active edges.  Because dS[r] is
// constant over a receiver's segment, dP_recv[r] = s1*dS[r]*N[r]: the receiver-ordered half of the backward pass is an
// element-wise epilogue (of the row program that produces dS), not an edge pass.
template <int SLOTS, bool SAVE, bool NEG>
__device__ __forceinline__ void edge_fwd_impl(const EdgeArgs& a, const float s1) {
    constexpr int G = 16 * SLOTS, RPW = 64 / G;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gl = lane % G, gbase = lane - gl, slot = gl >> 4, cl = gl & 15, ch = cl * 4;
    const float4 w = *(const float4*)(a.w_edge + ch);
    const float esh = *a.e_shift, esc = *a.e_scale;
    // per-wave LDS scratch for the ReLU nibbles of one step: [u][lane] bytes; storing lane t = gl (< 4*SLOTS) of a receiver
    // handles the edge of step-iteration u = t / SLOTS, slot = t % SLOTS, i.e. the 16 bytes at [u][gbase + 16*slot ..]
    __shared__ __attribute__((aligned(16))) unsigned char nib_all[4][4 * 64];
    unsigned char* nib_lds = nib_all[wv];
    const int st_off = ((gl / SLOTS) & 3) * 64 + gbase + 16 * (gl % SLOTS);

    const int nwork = (a.n_recv + RPW - 1) / RPW;  // one work item = one wave's RPW receivers
    for (int item = xcd_remap(blockIdx.x, gridDim.x) * 4 + wv; item < nwork; item += gridDim.x * 4) {
        const int r = item * RPW + lane / G;
        if (r < a.n_recv) {
            const int beg = a.seg_ptr[r], end = a.seg_ptr[r + 1];
            const float4 pown = *(const float4*)(a.p_recv + (size_t)r * EMB + ch);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            unsigned n0 = 0, n1 = 0, n2 = 0, n3 = 0;
            for (int base = beg; base < end; base += G) {
                const int e = base + gl;
                int o = 0; float c = 0.f;
                if (e < end) { o = a.oth[e]; c = (a.coef[e] + esh) * esc; }
                const int cnt = min(G, end - base);
                for (int i0 = 0; i0 < cnt; i0 += 4 * SLOTS) {
                    int oi[4]; float ci[4]; bool ok[4]; float4 p[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = i0 + u * SLOTS + slot;
                        ok[u] = i < cnt;
                        const int src = gbase + (ok[u] ? i : 0);
                        oi[u] = __shfl(o, src); ci[u] = __shfl(c, src);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (ok[u]) p[u] = *(const float4*)(a.p_oth + (size_t)oi[u] * EMB + ch);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        float h0 = 0.f, h1 = 0.f, h2 = 0.f, h3 = 0.f;
                        if (ok[u]) {
                            h0 = fmaf(ci[u], w.x, p[u].x) + pown.x; h1 = fmaf(ci[u], w.y, p[u].y) + pown.y;
                            h2 = fmaf(ci[u], w.z, p[u].z) + pown.z; h3 = fmaf(ci[u], w.w, p[u].w) + pown.w;
                            h0 = NEG ? fminf(h0, 0.f) : fmaxf(h0, 0.f); h1 = NEG ? fminf(h1, 0.f) : fmaxf(h1, 0.f);
                            h2 = NEG ? fminf(h2, 0.f) : fmaxf(h2, 0.f); h3 = NEG ? fminf(h3, 0.f) : fmaxf(h3, 0.f);
                            acc.x += h0; acc.y += h1; acc.z += h2; acc.w += h3;
                        }
                        if (SAVE) {
                            // this lane's four ReLU bits as a nibble; h >= +0 after the ReLU, so "h > 0" can be read off the
                            // bit pattern: (bits + 0x7fffffff) >> 31  (+0 -> 0, anything positive -> 1)
                            unsigned b0, b1, b2, b3;
                            if (NEG) { b0 = h0 < 0.f; b1 = h1 < 0.f; b2 = h2 < 0.f; b3 = h3 < 0.f; }
                            else {
                                b0 = (__float_as_uint(h0) + 0x7fffffffu) >> 31; b1 = (__float_as_uint(h1) + 0x7fffffffu) >> 31;
                                b2 = (__float_as_uint(h2) + 0x7fffffffu) >> 31; b3 = (__float_as_uint(h3) + 0x7fffffffu) >> 31;
                            }
                            n0 += b0; n1 += b1; n2 += b2; n3 += b3;
                            nib_lds[u * 64 + lane] = (unsigned char)(b0 | (b1 << 1) | (b2 << 2) | (b3 << 3));
                        }
                    }
                    if (SAVE) {
                        // The nibbles of one edge (16 lanes) are 16 consecutive bytes in LDS: storing lane t = gl (< 4*SLOTS) of a
                        // receiver reads those of edge t = u*SLOTS + slot, squeezes them into the edge's 64-bit word (nibble c =
                        // channels 4c..4c+3) and ONE store instruction writes the step's consecutive words.  No cross-lane
                        // VALU traffic: LDS operations of a wave execute in order, so the reads see the writes above.
                        const uint4 q = *(const uint4*)(nib_lds + st_off);
                        unsigned x0 = q.x, x1 = q.y, x2 = q.z, x3 = q.w;
                        x0 = (x0 | (x0 >> 4)) & 0x00ff00ffu; x0 = (x0 | (x0 >> 8)) & 0xffffu;
                        x1 = (x1 | (x1 >> 4)) & 0x00ff00ffu; x1 = (x1 | (x1 >> 8)) & 0xffffu;
                        x2 = (x2 | (x2 >> 4)) & 0x00ff00ffu; x2 = (x2 | (x2 >> 8)) & 0xffffu;
                        x3 = (x3 | (x3 >> 4)) & 0x00ff00ffu; x3 = (x3 | (x3 >> 8)) & 0xffffu;
                        if (gl < 4 * SLOTS && i0 + gl < cnt)
                            a.mask[base + i0 + gl] = (unsigned long long)(x0 | (x1 << 16)) | ((unsigned long long)(x2 | (x3 << 16)) << 32);
                    }
                }
            }
            acc = slot_reduce_r01<SLOTS>(acc);
            if (slot == 0) *(float4*)(a.out + (size_t)r * EMB + ch) = make_float4(s1 * acc.x, s1 * acc.y, s1 * acc.z, s1 * acc.w);
            if (SAVE) {
                const float4 nacc = slot_reduce_r01<SLOTS>(make_float4((float)n0, (float)n1, (float)n2, (float)n3));
                if (slot == 0) *(float4*)(a.cnt_rows + (size_t)r * EMB + ch) = nacc;
            }
        }
    }
}

template <int SLOTS, bool SAVE>
__global__ __launch_bounds__(256) void k_edge_fwd(EdgeArgs a) {
    const float s1 = *a.s1;
    if (s1 < 0.f) edge_fwd_impl<SLOTS, SAVE, true>(a, s1); else edge_fwd_impl<SLOTS, SAVE, false>(a, s1);
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

// Backward, sender-ordered half: with t_e = mask_e * dS[recv(e)],
//   dP_send[u] = s1 * sum_{e in seg(u)} t_e          Q[u] = s1 * sum_{e in seg(u)} c_e * t_e   (share of d w_edge)
// one 256-B row gather and one 8-B mask gather (through xpos, the edge's position in the receiver-ordered list) per edge.
template <int SLOTS>
__global__ __launch_bounds__(256) void k_edge_bwd_send(EdgeArgs a) {
    constexpr int G = 16 * SLOTS, RPW = 64 / G;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gl = lane % G, gbase = lane - gl, slot = gl >> 4, cl = gl & 15, ch = cl * 4;
    const float s1 = *a.s1, esh = *a.e_shift, esc = *a.e_scale;
    const unsigned nib_sh = 4u * (cl & 7);   // mask word: nibble c <=> channels 4c..4c+3 (see k_edge_fwd); c < 8 in the low half
    const int nwork = (a.n_recv + RPW - 1) / RPW;
    for (int item = xcd_remap(blockIdx.x, gridDim.x) * 4 + wv; item < nwork; item += gridDim.x * 4) {
        const int u = item * RPW + lane / G;
        if (u < a.n_recv) {
            const int beg = a.seg_ptr[u], end = a.seg_ptr[u + 1];
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), dw = acc;
            for (int base = beg; base < end; base += G) {
                const int e = base + gl;
                int o = 0; unsigned mlo = 0u, mhi = 0u; float c = 0.f;
                if (e < end) {
                    o = a.oth[e]; c = (a.coef[e] + esh) * esc;
                    const unsigned long long m = a.mask[a.xpos[e]];
                    mlo = (unsigned)m; mhi = (unsigned)(m >> 32);
                }
                const int cnt = min(G, end - base);
                for (int i0 = 0; i0 < cnt; i0 += 4 * SLOTS) {
                    int oi[4]; float ci[4]; bool ok[4]; float4 d[4]; unsigned wlo[4], whi[4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int i = i0 + v * SLOTS + slot;
                        ok[v] = i < cnt;
                        const int src = gbase + (ok[v] ? i : 0);
                        oi[v] = __shfl(o, src); ci[v] = __shfl(c, src);
                        wlo[v] = __shfl(mlo, src); whi[v] = __shfl(mhi, src);   // by every lane: a shuffle must not sit
                    }                                                           // under a lane mask
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        d[v] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (ok[v]) d[v] = *(const float4*)(a.d_s + (size_t)oi[v] * EMB + ch);
                    }
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        // rows of inactive slots are zero, so their mask bits do not matter
                        const unsigned mb = (cl < 8 ? wlo[v] : whi[v]) >> nib_sh;
                        const float t0 = (mb & 1u) ? d[v].x : 0.f, t1 = (mb & 2u) ? d[v].y : 0.f;
                        const float t2 = (mb & 4u) ? d[v].z : 0.f, t3 = (mb & 8u) ? d[v].w : 0.f;
                        acc.x += t0; acc.y += t1; acc.z += t2; acc.w += t3;
                        dw.x = fmaf(ci[v], t0, dw.x); dw.y = fmaf(ci[v], t1, dw.y); dw.z = fmaf(ci[v], t2, dw.z); dw.w = fmaf(ci[v], t3, dw.w);
                    }
                }
            }
            acc = slot_reduce_r01<SLOTS>(acc); dw = slot_reduce_r01<SLOTS>(dw);
            if (slot == 0) {
                *(float4*)(a.out + (size_t)u * EMB + ch) = make_float4(s1 * acc.x, s1 * acc.y, s1 * acc.z, s1 * acc.w);
                *(float4*)(a.dw_rows + (size_t)u * EMB + ch) = make_float4(s1 * dw.x, s1 * dw.y, s1 * dw.z, s1 * dw.w);
            }
        }
    }
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
        acc = slot_reduce_r01<SLOTS>(acc);
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

