#pragma once
#include "../../gcnn-cut-selector_amd/csrc/k_edge.hpp"

// ---------------------------------------------------------------------------------------------------------------
// Edge passes with the gathered table staged through LDS ("window" passes).
//
// A stacked mini-batch is a disjoint union of per-sample graphs (utils.py:401-407): the receivers of one sample gather rows of
// that sample's other side only, i.e. from a WINDOW of a few hundred to a thousand consecutive rows of the projected table --
// and every row of the window is gathered dozens of times (once per incident edge).  The plain passes (k_edge.hpp) fetch each of
// those rows from the XCD's L2 per edge, 256 B at a time, and run at the L2's row-gather rate (13-16 TB/s chip-wide).  Here a
// workgroup owns a CHUNK of consecutive receivers and CH of the 64 channels: it copies the window's CH-channel slice into LDS once
// (coalesced), then serves every edge of the chunk from LDS (150 TB/s chip-wide), so the L2 sees each window row once per
// chunk instead of once per edge.
//
// Nothing about samples is passed in: the plan carries, per receiver, the smallest and largest gathered index of its segment
// (lo / hi; an empty segment has hi < lo).  Inside a chunk the workgroup finds the diagonal blocks by itself -- receiver t starts
// a new block when lo[t] lies past every hi before it -- merges consecutive blocks while their joint window fits the LDS table,
// and runs stage -> barrier -> gather for each merged piece.  A piece whose window does not fit (one huge sample) is served
// from global memory by the same code (`direct`), so any graph is handled; the host only picks this path when the plan says
// the windows are small (gcnn_graph.*_win).
//
// Lane geometry: a row slice is CH floats = CH/4 lanes x float4; SLOTS edges of one receiver are in flight per step;
// G = SLOTS*CH/4 lanes serve one receiver, 64/G receivers share a wave.  Sums run over the segment in the same slot-strided
// order for every launch => bitwise reproducible, no float atomics.
// ---------------------------------------------------------------------------------------------------------------
#define WIN_T_MAX 256          // receivers per chunk, at most (one per thread of the block while the pieces are found)
#ifndef WIN_U
#define WIN_U 4                // steps (SLOTS edges each) whose index loads / LDS reads are issued before the first is used
#endif

#ifdef WIN_STAMPS   // diagnostic build of tools/micro/bench_edge_win.hip: where a block spends its cycles
#define WSTAMP(i) do { if (threadIdx.x == 0 && a.stamps) a.stamps[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WSTAMP(i) do { } while (0)
#endif
struct WinArgs {
    EdgeArgs e;
    unsigned long long* stamps;
    const int* lo; const int* hi;     // per segment owner: smallest / largest gathered index (hi < lo: empty segment)
    int chunk;                        // owners per chunk (<= WIN_T_MAX)
    int win_rows;                     // rows the LDS table holds (per table)
};

struct WinPiece { int t0, t1, lo, rows; };   // owners [t0, t1) of the chunk gather from rows [lo, lo + rows); rows < 0: serve from global memory

// Finds the pieces of one chunk (see above).  Every thread of the block calls (the first WIN_T_MAX of them hold one owner each);
// `slo` / `shi`: LDS scratch, WIN_T_MAX ints each; pieces in `pc`, their number returned.
__device__ __forceinline__ int win_find_pieces(const WinArgs& a, const int r0, const int n, int* slo, int* shi, WinPiece* pc, int* npc) {
    const int t = threadIdx.x, lane = t & 63, wv = min(t >> 6, 4);   // waves past the fourth hold no owner
    __shared__ int wred[2][5];
    int lo = 0x7fffffff, hi = -1;
    if (t < n) { lo = a.lo[r0 + t]; hi = a.hi[r0 + t]; }
    const bool empty = hi < lo;
    if (empty) { lo = 0x7fffffff; hi = -1; }
    // inclusive prefix max of hi, inclusive suffix min of lo over the chunk (wave scans, then the other waves' totals)
    int pm = hi, sm = lo;
    for (int off = 1; off < 64; off <<= 1) {
        const int u = __shfl_up(pm, off), d = __shfl_down(sm, off);
        if (lane >= off) pm = max(pm, u);
        if (lane + off < 64) sm = min(sm, d);
    }
    if (lane == 63) wred[0][wv] = pm;
    if (lane == 0) wred[1][wv] = sm;
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
        if (w < wv) pm = max(pm, wred[0][w]);
        if (w > wv) sm = min(sm, wred[1][w]);
    }
    const int before = __shfl_up(pm, 1);
    int pm_prev = lane == 0 ? -1 : before;          // prefix max over owners < t
    if (lane == 0) for (int w = 0; w < wv; ++w) pm_prev = max(pm_prev, wred[0][w]);
    const bool cut = t < n && (t == 0 || (!empty && lo > pm_prev));
    if (t < WIN_T_MAX) {
        slo[t] = cut ? sm : 0;      // a block's window: from the suffix min at its first owner ...
        shi[t] = pm;                // ... to the prefix max at its last owner
    }
    // block starts, compacted in order: ballot per wave + the waves before
    const unsigned long long bal = __ballot(cut);
    __shared__ int wcnt[5];
    if (lane == 0) wcnt[wv] = __popcll(bal);
    __syncthreads();
    int pos = __popcll(bal & ((1ull << lane) - 1));
    for (int w = 0; w < wv; ++w) pos += wcnt[w];
    const int nblk = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    __shared__ short start[WIN_T_MAX + 1];
    if (cut) start[pos] = (short)t;
    if (t == 0) start[nblk] = (short)n;
    __syncthreads();
    if (t == 0) {   // greedy merge of consecutive blocks while the joint window fits (usually one to three blocks per chunk)
        int np = 0, i = 0;
        while (i < nblk) {
            const int t0 = start[i], wlo = slo[t0];
            int j = i + 1, whi = shi[start[j] - 1];
            while (j < nblk && shi[start[j + 1] - 1] - wlo + 1 <= a.win_rows) { ++j; whi = shi[start[j] - 1]; }
            const int rows = max(whi - wlo + 1, 0);
            pc[np++] = WinPiece{t0, (int)start[j], wlo, rows <= a.win_rows ? rows : -1};
            i = j;
        }
        *npc = np;
    }
    __syncthreads();
    return *npc;
}

// Copies rows [lo, lo + rows) x channels [c0, c0 + CH) of `src` ([n,64] fp32) into the LDS table `tab` ([rows][CH]).
template <int CH, int NT>
__device__ __forceinline__ void win_stage(const float* __restrict__ src, const int lo, const int rows, const int c0, float* tab) {
    constexpr int PPR = CH / 4;   // float4 pieces per row
    const int n = rows * PPR;
    int i = threadIdx.x;
    for (; i + 3 * NT < n; i += 4 * NT) {   // four loads in flight per thread
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int k = i + u * NT; v[u] = *(const float4*)(src + (size_t)(lo + k / PPR) * EMB + c0 + 4 * (k % PPR)); }
#pragma unroll
        for (int u = 0; u < 4; ++u) *(float4*)(tab + (size_t)(i + u * NT) * 4) = v[u];
    }
    for (; i < n; i += NT) *(float4*)(tab + (size_t)i * 4) = *(const float4*)(src + (size_t)(lo + i / PPR) * EMB + c0 + 4 * (i % PPR));
}

// Forward: S[r] = s1 * sum_e max/min(J_e, 0), N[r] = number of active edges per channel, J_e = (c_e*w + P_oth[oth_e]) + P_own[r]
// grid = (chunks, 64 / CH); block = NT threads; dynamic LDS = win_rows * CH floats
template <int CH, int SLOTS, bool COUNT, bool NEG, int NT>
__device__ __forceinline__ void edge_fwd_win_impl(const WinArgs& a, const float s1, float* tab) {
    constexpr int PPR = CH / 4, G = SLOTS * PPR, RPW = 64 / G, NW = NT / 64;
    static_assert(G <= 64 && 64 % G == 0, "lane group");
    __shared__ int slo[WIN_T_MAX], shi[WIN_T_MAX];
    __shared__ WinPiece pc[WIN_T_MAX];
    __shared__ int npc;
    const EdgeArgs& e = a.e;
    const int r0 = blockIdx.x * a.chunk, n = min(a.chunk, e.n_own - r0), c0 = blockIdx.y * CH;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gl = lane % G, sub = lane / G, slot = gl / PPR, ch = c0 + 4 * (gl % PPR), lch = 4 * (gl % PPR);
    const float4 w = *(const float4*)(e.w_edge + ch);
    const float esh = *e.e_shift, esc = *e.e_scale;
    WSTAMP(0);
    const int np = win_find_pieces(a, r0, n, slo, shi, pc, &npc);
    WSTAMP(1);
    for (int p = 0; p < np; ++p) {
        const WinPiece q = pc[p];
        const bool direct = q.rows < 0;
        if (!direct) win_stage<CH, NT>(e.p_oth, q.lo, q.rows, c0, tab);
        __syncthreads();
        if (p == 0) WSTAMP(2);
        for (int t = q.t0 + wv * RPW + sub; t - sub < q.t1; t += NW * RPW) {   // every lane of a wave runs the same trips
            const bool live = t < q.t1;
            const int r = r0 + (live ? t : q.t0);
            int beg = e.seg_ptr[r], end = e.seg_ptr[r + 1];
            if (!live) end = beg;
            const float4 pown = *(const float4*)(e.p_own + (size_t)r * EMB + ch);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            unsigned n0 = 0, n1 = 0, n2 = 0, n3 = 0;
            int len = end - beg;
            if (RPW > 1) {   // the receivers sharing a wave loop together: as many trips as the longest of them needs
#pragma unroll
                for (int off = G; off < 64; off <<= 1) len = max(len, __shfl_xor(len, off));
            }
            for (int b0 = 0; b0 < len; b0 += WIN_U * SLOTS) {
                int oi[WIN_U]; float ci[WIN_U]; bool ok[WIN_U]; float4 v[WIN_U];
#pragma unroll
                for (int u = 0; u < WIN_U; ++u) {
                    const int ed = beg + b0 + u * SLOTS + slot;
                    ok[u] = ed < end;
                    oi[u] = ok[u] ? e.oth[ed] : q.lo;
                    ci[u] = ok[u] ? e.coef[ed] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < WIN_U; ++u) {
                    if (direct) { if (ok[u]) v[u] = *(const float4*)(e.p_oth + (size_t)oi[u] * EMB + ch); }
                    else v[u] = *(const float4*)(tab + (size_t)(oi[u] - q.lo) * CH + lch);
                }
#pragma unroll
                for (int u = 0; u < WIN_U; ++u)
                    if (ok[u]) {
                        const float c = (ci[u] + esh) * esc;
                        float h0 = fmaf(c, w.x, v[u].x) + pown.x, h1 = fmaf(c, w.y, v[u].y) + pown.y;
                        float h2 = fmaf(c, w.z, v[u].z) + pown.z, h3 = fmaf(c, w.w, v[u].w) + pown.w;
                        h0 = NEG ? fminf(h0, 0.f) : fmaxf(h0, 0.f); h1 = NEG ? fminf(h1, 0.f) : fmaxf(h1, 0.f);
                        h2 = NEG ? fminf(h2, 0.f) : fmaxf(h2, 0.f); h3 = NEG ? fminf(h3, 0.f) : fmaxf(h3, 0.f);
                        acc.x += h0; acc.y += h1; acc.z += h2; acc.w += h3;
                        if (COUNT) { n0 += h0 != 0.f; n1 += h1 != 0.f; n2 += h2 != 0.f; n3 += h3 != 0.f; }
                    }
            }
            float4 cn = make_float4((float)n0, (float)n1, (float)n2, (float)n3);
#pragma unroll
            for (int off = PPR; off < G; off <<= 1) {   // fold the slots
                acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off); acc.z += __shfl_xor(acc.z, off); acc.w += __shfl_xor(acc.w, off);
                if (COUNT) { cn.x += __shfl_xor(cn.x, off); cn.y += __shfl_xor(cn.y, off); cn.z += __shfl_xor(cn.z, off); cn.w += __shfl_xor(cn.w, off); }
            }
            if (live && slot == 0) {
                *(float4*)(e.out + (size_t)r * EMB + ch) = make_float4(s1 * acc.x, s1 * acc.y, s1 * acc.z, s1 * acc.w);
                if (COUNT) *(float4*)(e.cnt_rows + (size_t)r * EMB + ch) = cn;
            }
        }
        __syncthreads();   // the table is re-staged by the next piece
        if (p == 0) WSTAMP(3);
    }
    WSTAMP(4);
}
template <int CH, int SLOTS, bool COUNT, int NT>
__global__ __launch_bounds__(NT) void k_edge_fwd_win(WinArgs a) {
    extern __shared__ __attribute__((aligned(16))) float win_tab[];
    const float s1 = *a.e.s1;
    if (s1 < 0.f) edge_fwd_win_impl<CH, SLOTS, COUNT, true, NT>(a, s1, win_tab);
    else edge_fwd_win_impl<CH, SLOTS, COUNT, false, NT>(a, s1, win_tab);
}
