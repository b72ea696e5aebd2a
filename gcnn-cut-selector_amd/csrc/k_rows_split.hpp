#pragma once
#include "k_rows.hpp"

// ---------------------------------------------------------------------------------------------------------------
// Row programs for FEW tiles (the cut rows of a training batch; every row set of a single-state inference call).
// With fewer tiles than SIMDs the programs of k_rows.hpp are a latency chain: one wave walks a tile through 4-10 dependent
// 64x64 products of 64 MFMAs each while three quarters of the chip idle.  Here the FOUR waves of a block share one tile: wave
// w computes output features 16w..16w+15 of every stage (16 MFMAs instead of 64, its own quarter of every epilogue, load and
// store) and the full tile the next stage needs as its B operand is put together through a 4 KB LDS exchange tile
// (double-buffered: one barrier per stage).  Same MFMA order per output element as the one-wave programs -- same bits.
// Blocks = tiles (one block per CU at these sizes), 256 threads.
// ---------------------------------------------------------------------------------------------------------------
#define SX_ROW 68                       // padded row of the exchange tile (floats): lanes j = 0..15 of a float4 read hit distinct banks
#define SX_FLOATS (2 * 16 * SX_ROW)     // two exchange tiles
struct RQuart { float v[4]; };          // one wave's quarter of a tile: lane (j, g) holds X[row0 + j][16*w + 4*g + i]

struct SplitLane {
    int lane, wv, j, g, phase;
    float* xbuf;
    __device__ __forceinline__ SplitLane(float* exchange) : lane(threadIdx.x & 63), wv(threadIdx.x >> 6), j(lane & 15), g(lane >> 4), phase(0), xbuf(exchange) {}
};

// acc += Wop[16*w + (lane&15)][kf] * (scale * T[kf]) over the whole k range: the wave's quarter of rt_gemm
template <int MODE>
__device__ __forceinline__ void rq_gemm(const RTile& t, float scale, const float* wl, f32x4& acc, const SplitLane& L) {
    const int m = L.j, g = L.g, mo = L.wv;
    wl = lds_here(wl);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        float av[4];
        if (MODE == GEMM_BWD) {
            const float4 w4 = *(const float4*)(wl + (16 * mo + m) * LDW + 16 * mt + 4 * g);
            av[0] = w4.x; av[1] = w4.y; av[2] = w4.z; av[3] = w4.w;
        } else if (MODE == GEMM_FWD) {
            const float4 w4 = *(const float4*)(wl + ((4 * mt + g) * 64 + 16 * mo + m) * 4);
            av[0] = w4.x; av[1] = w4.y; av[2] = w4.z; av[3] = w4.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = wl[(16 * mt + 4 * g + i) * LDW + 16 * mo + m];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = mfma16(av[i], t.v[mt][i] * scale, acc);
    }
}
template <int MODE>
__device__ __forceinline__ void rq_mm(RQuart& o, const RTile& in, float scale, const float* wl, const SplitLane& L) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    rq_gemm<MODE>(in, scale, wl, acc, L);
#pragma unroll
    for (int i = 0; i < 4; ++i) o.v[i] = acc[i];
}
template <int MODE>
__device__ __forceinline__ void rq_mm2(RQuart& o, const RTile& a, float sa, const float* wa, const RTile& b, const float* wb, const SplitLane& L) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    rq_gemm<MODE>(a, sa, wa, acc, L);
    rq_gemm<MODE>(b, 1.f, wb, acc, L);
#pragma unroll
    for (int i = 0; i < 4; ++i) o.v[i] = acc[i];
}
template <bool RELU>
__device__ __forceinline__ void rq_bias(RQuart& o, const float* vec, const SplitLane& L) {
    const float4 b = *(const float4*)(lds_here(vec) + 16 * L.wv + 4 * L.g);
    const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float v = o.v[i] + bv[i]; o.v[i] = RELU ? fmaxf(v, 0.f) : v; }
}
__device__ __forceinline__ void rq_clear_unless(RQuart& o, bool ok) {
#pragma unroll
    for (int i = 0; i < 4; ++i) o.v[i] = ok ? o.v[i] : 0.f;
}
__device__ __forceinline__ void rq_mask(RQuart& o, const RQuart& act) {
#pragma unroll
    for (int i = 0; i < 4; ++i) o.v[i] = act.v[i] > 0.f ? o.v[i] : 0.f;
}
__device__ __forceinline__ void rq_load(RQuart& q, const float* x, int row, bool ok, const SplitLane& L) {
    float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) f = *(const float4*)(x + (size_t)row * EMB + 16 * L.wv + 4 * L.g);
    q.v[0] = f.x; q.v[1] = f.y; q.v[2] = f.z; q.v[3] = f.w;
}
__device__ __forceinline__ void rq_store(const RQuart& q, float* x, int row, bool ok, const SplitLane& L) {
    if (!ok || !x) return;
    *(float4*)(x + (size_t)row * EMB + 16 * L.wv + 4 * L.g) = make_float4(q.v[0], q.v[1], q.v[2], q.v[3]);
}
// every wave contributes its quarter; after the block barrier every wave holds the full tile (the next stage's B operand)
__device__ __forceinline__ void rq_exchange(RTile& full, const RQuart& q, SplitLane& L) {
    float* b = L.xbuf + L.phase * 16 * SX_ROW;
    *(float4*)(b + L.j * SX_ROW + 16 * L.wv + 4 * L.g) = make_float4(q.v[0], q.v[1], q.v[2], q.v[3]);
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const float4 f = *(const float4*)(b + L.j * SX_ROW + 16 * mt + 4 * L.g);
        full.v[mt][0] = f.x; full.v[mt][1] = f.y; full.v[mt][2] = f.z; full.v[mt][3] = f.w;
    }
    L.phase ^= 1;   // the next stage writes the other tile: a fast wave never overwrites what a slow one still reads
}

// ---- Program 1, split: embedding + projections (emb_program) -------------------------------------------------------------
template <int F, int NPROJ>
__device__ __forceinline__ void emb_split(const EmbArgs& a, float* smem, int bid, int nblk) {
    constexpr int NT = 256, NM = 1 + NPROJ, NV = 2 + NPROJ;
    const int ntile = (a.n + 15) >> 4;
    float* vecs = smem + NM * 64 * LDW;
    float* w1s = smem + ROWS_LDS_FLOATS(NM, NV);
    SplitLane L(smem + EMB_LDS_FLOATS);
    if (NPROJ == 2) {
        const float* const w[3] = {a.w2, a.wp[0], a.wp[1]};
        const float* const v[4] = {a.b1, a.b2, a.bp[0], a.bp[1]};
        stage_lds<3, 4, NT, true>((float*)smem, w, v);
    } else {
        const float* const w[2] = {a.w2, a.wp[0]};
        const float* const v[3] = {a.b1, a.b2, a.bp[0]};
        stage_lds<2, 3, NT, true>((float*)smem, w, v);
    }
    for (int i = threadIdx.x; i < F * 64; i += NT) w1s[i] = a.w1[i];
    float shift[F], scale[F];
#pragma unroll
    for (int f = 0; f < F; ++f) { shift[f] = a.shift[f]; scale[f] = a.scale[f]; }
    __syncthreads();
    for (int tile = bid; tile < ntile; tile += nblk) {
        const int row = tile * 16 + L.j;
        const bool ok = row < a.n;
        RQuart q;
        RTile full, t;
#pragma unroll
        for (int i = 0; i < 4; ++i) q.v[i] = 0.f;
        const float* w1h = lds_here(w1s);
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const float xn = ok ? (a.x[(size_t)row * F + f] + shift[f]) * scale[f] : 0.f;
            const float4 w = *(const float4*)(w1h + f * EMB + 16 * L.wv + 4 * L.g);
            q.v[0] = fmaf(xn, w.x, q.v[0]); q.v[1] = fmaf(xn, w.y, q.v[1]); q.v[2] = fmaf(xn, w.z, q.v[2]); q.v[3] = fmaf(xn, w.w, q.v[3]);
        }
        rq_bias<true>(q, vecs, L);
        rq_clear_unless(q, ok);
        rq_exchange(full, q, L);
        if (L.wv == 0) rt_mask_store(full, a.m_e1, row, ok, L.g);   // every wave holds the whole tile now: one of them writes its pattern
        rq_mm<GEMM_FWD>(q, full, 1.f, smem, L);
        rq_bias<true>(q, vecs + 64, L);
        rq_clear_unless(q, ok);
        rq_store(q, a.xo, row, ok, L);
        rq_exchange(t, q, L);
        if (L.wv == 0) rt_mask_store(t, a.m_x, row, ok, L.g);
#pragma unroll
        for (int k = 0; k < NPROJ; ++k) {
            rq_mm<GEMM_FWD>(q, t, 1.f, smem + (1 + k) * 64 * LDW, L);
            rq_bias<false>(q, vecs + (2 + k) * 64, L);
            rq_store(q, a.po[k], row, ok, L);
        }
    }
}

// ---- Program 2, split: receiver-side update (convf_program; CF_PROJ and CF_READOUT -- the training tail is convturn_split) ----
template <int TAIL, bool KEEP_A = false>
__device__ __forceinline__ void convf_split(const ConvFArgs& a, float* smem, int bid, int nblk) {
    static_assert(TAIL == CF_PROJ || TAIL == CF_READOUT, "the loss tail runs in convturn_split");
    constexpr int NT = 256, NM = KEEP_A ? 5 : 4;
    constexpr int iW1B = NM - 3, iW2 = NM - 2, iWT = NM - 1;
    const int ntile = (a.n + 15) >> 4;
    float* vecs = smem + NM * 64 * LDW;
    SplitLane L(smem + ROWS_LDS_FLOATS(5, 5));   // the exchange tiles sit behind the larger (KEEP_A) image in either form
    if (KEEP_A) {
        const float* const w[5] = {a.wf, a.w1a, a.w1b, a.w2, a.wt};
        const float* const v[5] = {a.bf, a.b1, a.b2, a.bt, TAIL != CF_PROJ ? a.ws : nullptr};
        stage_lds<5, 5, NT, true>(smem, w, v);
    } else {
        const float* const w[4] = {a.mfuse, a.w1b, a.w2, a.wt};
        const float* const v[5] = {a.ufuse, a.b1, a.b2, a.bt, TAIL != CF_PROJ ? a.ws : nullptr};
        stage_lds<4, 5, NT, true>(smem, w, v);
    }
    const float s2 = KEEP_A ? *a.s2 : 1.f;
    const float bs = TAIL != CF_PROJ ? *a.bs : 0.f;
    __syncthreads();
    for (int tile = bid; tile < ntile; tile += nblk) {
        const int row = tile * 16 + L.j;
        const bool ok = row < a.n;
        RTile s_in, xr, full, z;
        rt_load(s_in, a.s, row, ok, L.g);
        rt_load(xr, a.xrecv, row, ok, L.g);
        const float deg = ok ? (float)(a.seg_ptr[row + 1] - a.seg_ptr[row]) : 0.f;
        RQuart q;
        auto add_deg = [&](RQuart& t) {   // + deg * (bf | u)
            const float4 b = *(const float4*)(lds_here(vecs) + 16 * L.wv + 4 * L.g);
            t.v[0] = fmaf(deg, b.x, t.v[0]); t.v[1] = fmaf(deg, b.y, t.v[1]); t.v[2] = fmaf(deg, b.z, t.v[2]); t.v[3] = fmaf(deg, b.w, t.v[3]);
        };
        if (KEEP_A) {
            rq_mm<GEMM_FWD>(q, s_in, 1.f, smem, L);
            add_deg(q);
            rq_clear_unless(q, ok);
            rq_store(q, a.a_out, row, ok, L);
            rq_exchange(full, q, L);
            rq_mm2<GEMM_FWD>(q, full, s2, smem + 64 * LDW, xr, smem + 2 * 64 * LDW, L);
        } else {
            rq_mm2<GEMM_FWD>(q, s_in, 1.f, smem, xr, smem + iW1B * 64 * LDW, L);
            add_deg(q);
        }
        rq_bias<true>(q, vecs + 64, L);
        rq_clear_unless(q, ok);
        rq_store(q, a.z1, row, ok, L);
        rq_exchange(z, q, L);
        if (L.wv == 0) rt_mask_store(z, a.m_z1, row, ok, L.g);
        rq_mm<GEMM_FWD>(q, z, 1.f, smem + iW2 * 64 * LDW, L);
        rq_bias<true>(q, vecs + 2 * 64, L);
        rq_clear_unless(q, ok);
        rq_store(q, a.out, row, ok, L);
        rq_exchange(full, q, L);
        if (L.wv == 0) rt_mask_store(full, a.m_out, row, ok, L.g);
        rq_mm<GEMM_FWD>(q, full, 1.f, smem + iWT * 64 * LDW, L);
        if (TAIL == CF_PROJ) {
            rq_bias<false>(q, vecs + 3 * 64, L);
            rq_store(q, a.t_out, row, ok, L);
        } else {
            rq_bias<true>(q, vecs + 3 * 64, L);
            rq_clear_unless(q, ok);
            rq_store(q, a.t_out, row, ok, L);
            rq_exchange(full, q, L);   // O1: every wave gets the whole tile, wave 0 finishes the dot product
            if (L.wv == 0) {
                const float score = readout_score(full, vecs + 4 * 64, bs, L.g);
                if (L.g == 0 && ok) a.scores[row] = score;
            }
        }
    }
}

// ---- Program 2+3, split: the training turnaround of the cut rows (convturn_program) -----------------------------------------
__device__ __forceinline__ void convturn_split(const ConvFArgs& a, const ConvBArgs& b, float* smem, int bid, int nblk) {
    constexpr int NT = 256, NM = 4;
    const int ntile = (a.n + 15) >> 4;
    float* vecs = smem + NM * 64 * LDW;
    SplitLane L(smem + ROWS_LDS_FLOATS(5, 5));
    {
        const float* const w[4] = {a.mfuse, a.w1b, a.w2, a.wt};
        const float* const v[5] = {a.ufuse, a.b1, a.b2, a.bt, a.ws};
        stage_lds<4, 5, NT>(smem, w, v);
    }
    const float s1 = *b.s1, bs = *a.bs;
    float* const MF = smem; float* const W1B = smem + 64 * LDW;
    float* const W2 = smem + 2 * 64 * LDW; float* const WT = smem + 3 * 64 * LDW;
    __syncthreads();
    for (int tile = bid; tile < ntile; tile += nblk) {
        const int row = tile * 16 + L.j;
        const bool ok = row < a.n;
        RTile s_in, xr, full, o1;
        RQuart nr, q, z1q, xoq;
        rt_load(s_in, a.s, row, ok, L.g);
        rt_load(xr, a.xrecv, row, ok, L.g);
        rq_load(nr, b.nrows, row, ok, L);
        const float deg = ok ? (float)(a.seg_ptr[row + 1] - a.seg_ptr[row]) : 0.f;
        // ---- forward half
        rq_mm2<GEMM_FWD_RM>(z1q, s_in, 1.f, MF, xr, W1B, L);
        {
            const float4 bb = *(const float4*)(lds_here(vecs) + 16 * L.wv + 4 * L.g);
            z1q.v[0] = fmaf(deg, bb.x, z1q.v[0]); z1q.v[1] = fmaf(deg, bb.y, z1q.v[1]); z1q.v[2] = fmaf(deg, bb.z, z1q.v[2]); z1q.v[3] = fmaf(deg, bb.w, z1q.v[3]);
        }
        rq_bias<true>(z1q, vecs + 64, L);
        rq_clear_unless(z1q, ok);
        rq_store(z1q, a.z1, row, ok, L);
        rq_exchange(full, z1q, L);
        rq_mm<GEMM_FWD_RM>(xoq, full, 1.f, W2, L);
        rq_bias<true>(xoq, vecs + 2 * 64, L);
        rq_clear_unless(xoq, ok);
        rq_store(xoq, a.out, row, ok, L);
        rq_exchange(full, xoq, L);
        rq_mm<GEMM_FWD_RM>(q, full, 1.f, WT, L);
        rq_bias<true>(q, vecs + 3 * 64, L);
        rq_clear_unless(q, ok);
        rq_store(q, a.t_out, row, ok, L);
        rq_exchange(o1, q, L);
        // the MSE head: every wave needs dO1pre (the first backward product's B operand) -- it is element-wise in O1 and the
        // row's score, so each computes it; wave 0 also writes the scores, the tile's partial slab and dO1pre itself
        const float score = readout_score(o1, vecs + 4 * 64, bs, L.g);
        RTile go;
        if (L.wv == 0) {
            if (L.g == 0 && ok) a.scores[row] = score;
            loss_head_tile(go, o1, score, a, vecs + 4 * 64, tile, row, ok, L.lane);
            rt_store(go, a.g_o1, row, ok, L.g);
        } else {
            const float ds = ok ? 2.f * (score - a.targets[row]) * a.loss_scale : 0.f;
            const float* wsv = lds_here(vecs + 4 * 64);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float4 w = *(const float4*)(wsv + 16 * m + 4 * L.g);
                const float wv4[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) go.v[m][i] = o1.v[m][i] > 0.f ? ds * wv4[i] : 0.f;
            }
        }
        // ---- backward half
        rq_mm<GEMM_BWD>(q, go, 1.f, WT, L);
        rq_mask(q, xoq);
        rq_store(q, b.g_out, row, ok, L);
        rq_exchange(full, q, L);
        rq_mm<GEMM_BWD>(q, full, 1.f, W2, L);
        rq_mask(q, z1q);
        rq_store(q, b.g_z1, row, ok, L);
        rq_exchange(full, q, L);
        rq_mm<GEMM_BWD>(q, full, 1.f, W1B, L);
        rq_store(q, b.g_xrecv, row, ok, L);
        rq_mm<GEMM_BWD>(q, full, 1.f, MF, L);
        rq_store(q, b.g_s, row, ok, L);
#pragma unroll
        for (int i = 0; i < 4; ++i) q.v[i] = s1 * q.v[i] * nr.v[i];
        rq_store(q, b.g_precv, row, ok, L);
    }
}
__global__ __launch_bounds__(256) void k_conv_turn_split(ConvFArgs f, ConvBArgs b) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    convturn_split(f, b, smem, blockIdx.x, gridDim.x);
}
template <int TAIL, bool KEEP_A = false>
__global__ __launch_bounds__(256) void k_conv_fwd_split(ConvFArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    convf_split<TAIL, KEEP_A>(a, smem, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(256) void k_embed_fwd_split(EmbGroupArgs m) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;
    if (b < m.blk0[1]) emb_split<14, 2>(m.v, smem, b, m.blk0[1]);
    else if (b < m.blk0[2]) emb_split<4, 1>(m.c, smem, b - m.blk0[1], m.blk0[2] - m.blk0[1]);
    else if (b < m.blk0[3]) emb_split<6, 1>(m.k, smem, b - m.blk0[2], m.blk0[3] - m.blk0[2]);
    else fuse_weights(m.fz, b - m.blk0[3], smem);
}
#define EMB_SPLIT_LDS_FLOATS (EMB_LDS_FLOATS + SX_FLOATS)
#define CONV_SPLIT_LDS_FLOATS (ROWS_LDS_FLOATS(5, 5) + SX_FLOATS)
