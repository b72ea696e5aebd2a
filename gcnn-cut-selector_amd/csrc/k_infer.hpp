#pragma once
#include "gcnn_common.hpp"
#include "k_rows.hpp"
#include "k_rows_split.hpp"
#include "k_edge.hpp"

// ---------------------------------------------------------------------------------------------------------------
// Graph plan for ONE sampled state (the SCIP cut selector calls the model once per separation round on a single graph,
// model_evaluator.py:82-103): three launches instead of a device-wide radix sort.  Inference needs three segment
// structures: constraint edges by constraint (conv v->c), constraint edges by variable (conv c->v) and cut edges by cut
// (conv v->k).  The reference's get_state emits (row, col)-sorted COO (utils.py:102-104), so the two by-left structures are
// the input lists themselves plus segment offsets; only the by-variable order of the constraint edges has to be built:
//   1. count : one pass over both lists -- range check, sortedness check, by-left offsets, per-variable counts
//   2. place : every block scans the counts in LDS (no separate scan launch), then each edge takes a slot of its
//              variable's segment with an integer atomic (arrival order)
//   3. order : per variable, ranks the segment's input positions (unique, so rank = stable order) and gathers
//              (left id, coefficient) into place -- the result equals a stable sort by variable id
// The steps run as extra blocks of the forward pass's first three launches (bottom of this file); for states with many
// variables, place and order are launches of their own (k_iplan_place, k_iplan_order).
// flags[0] index out of range, flags[1] constraint list not sorted by row, flags[2] cut list not sorted by row, flags[3] a
// variable with more edges than IPLAN_MAX_DEG: in all four cases the caller takes the general path (gcnn_graph_build).
// ---------------------------------------------------------------------------------------------------------------
#define IPLAN_MAX_VARS 32768   // the per-block scan of the counts lives in LDS (128 KB)
#define IPLAN_MAX_DEG 2048
#ifndef IPLAN_FUSE_MAX_VARS
#define IPLAN_FUSE_MAX_VARS 4096   // above: place / order as launches of their own (small LDS footprint, all lane groups resident)
#endif

struct IplanSet { int* inds; int n_edges, n_left; int* l_ptr; };   // inds = [2,E]: left ids then variable ids
struct IplanArgs {
    IplanSet s[2];        // 0: constraint edges, 1: cut edges
    int n_vars;
    int* vcount;          // [n_vars]   zero on entry
    int* cursor;          // [n_vars]   zero on entry
    int* flags;           // [4]        zero on entry
    int* v_ptr; int* v_pos; int* v_oth; float* v_coef;   // by-variable structure of the constraint edges
    const float* cons_coef;
    int blocks0, blocks1; // blocks per edge set in the count step
};

// Robustness: the forward pass is launched right behind the plan without looking at the flags, so whatever the input, every
// structure the forward reads must stay inside its arrays.  Out-of-range ids are replaced by 0 in the uploaded copy (and
// counted as such), the by-left offset arrays arrive zero-filled with the upload (an unsorted list leaves gaps), and an
// over-long variable segment is filled with zeros.  The flags tell the host that the scores of such a call mean nothing.
// block `bid` of blocks0 + blocks1 blocks of `nt` threads: the first blocks0 sweep the constraint list, the others the cut list
__device__ __forceinline__ void iplan_count_body(const IplanArgs& a, const int bid, const int nt) {
    const int set = bid >= a.blocks0;
    const IplanSet s = a.s[set];
    const int lb = bid - (set ? a.blocks0 : 0), nb = set ? a.blocks1 : a.blocks0;
    int* left = s.inds;
    int* var = s.inds + s.n_edges;
    int bad = 0, unsorted = 0;
    for (int i = lb * nt + (int)threadIdx.x; i <= s.n_edges; i += nb * nt) {
        // by-left offsets: ptr[k] = first position whose left id >= k (ids clamped so that a bad index cannot write out of range)
        const int lo = i == 0 ? -1 : min(max(left[i - 1], -1), s.n_left);
        const int hi = i == s.n_edges ? s.n_left : min(max(left[i], -1), s.n_left);
        if (i < s.n_edges) {
            const int l = left[i];
            int v = var[i];
            const bool bad_l = l < 0 || l >= s.n_left, bad_v = v < 0 || v >= a.n_vars;
            bad |= bad_l | bad_v;
            if (i + 1 < s.n_edges) unsorted |= left[i + 1] < l;
            if (bad_v) { v = 0; var[i] = 0; }
            if (set == 0 && a.n_vars > 0) atomicAdd(&a.vcount[v], 1);
        }
        for (int k = lo + 1; k <= hi; ++k) s.l_ptr[k] = i;
    }
    if (bad) atomicOr(&a.flags[0], 1);
    if (unsorted) atomicOr(&a.flags[1 + set], 1);
}
// (left ids out of range are clamped on use: they only ever index the offset array above; the lists keep them, the by-variable
// structure below stores them clamped.)

// exclusive scan of cnt[0..n) into LDS `pre` (n <= IPLAN_MAX_VARS), total in pre[n]; NT threads (a multiple of 64, <= 1024)
template <int NT>
__device__ __forceinline__ void iplan_scan(const int* __restrict__ cnt, int n, int* pre) {
    __shared__ int wsum[NT / 64];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    for (int i = t; i < n; i += NT) pre[i] = cnt[i];        // coalesced, independent loads; the serial part below runs in LDS
    __syncthreads();
    const int per = (n + NT - 1) / NT, b = min(n, t * per), e = min(n, b + per);
    int sum = 0;
    for (int i = b; i < e; ++i) { const int c = pre[i]; pre[i] = sum; sum += c; }
    int inc = sum;                                   // inclusive scan over the wave, then over the waves
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(inc, off); if (lane >= off) inc += o; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int base = inc - sum;
    for (int w = 0; w < wv; ++w) base += wsum[w];
    for (int i = b; i < e; ++i) pre[i] += base;
    if (t == NT - 1) pre[n] = base + sum;
    __syncthreads();
}
// block `bid` of `nblk` blocks of NT threads; `pre`: dynamic LDS, n_vars + 1 ints
template <int NT>
__device__ __forceinline__ void iplan_place_body(const IplanArgs& a, int* pre, const int bid, const int nblk) {
    iplan_scan<NT>(a.vcount, a.n_vars, pre);
    if (bid == 0)
        for (int i = threadIdx.x; i <= a.n_vars; i += NT) a.v_ptr[i] = pre[i];
    const int n = a.s[0].n_edges;
    for (int e = bid * NT + threadIdx.x; e < n; e += nblk * NT) {
        const int v = a.s[0].inds[n + e];           // sanitised by the count pass
        a.v_pos[pre[v] + atomicAdd(&a.cursor[v], 1)] = e;
    }
}
__global__ __launch_bounds__(1024) void k_iplan_place(IplanArgs a) {
    extern __shared__ int pre[];   // [n_vars + 1]
    iplan_place_body<1024>(a, pre, blockIdx.x, gridDim.x);
}

// 16 lanes per variable.  Input positions are unique, so an element's rank among its segment is its place in the stable
// order.  The segment is staged in LDS (padded with INT_MAX) and every element compares itself with all of it, four at a
// time -- quadratic in the degree, which is what IPLAN_MAX_DEG bounds (segments beyond IPLAN_LDS_DEG re-read global memory).
#define IPLAN_LDS_DEG 256
template <int NT>   // block `bid` of `nblk`: its NT/16 lane groups take variables bid*NT/16 + grp, then + nblk*NT/16, ...
__device__ __forceinline__ void iplan_order_body(const IplanArgs& a, const int bid, const int nblk) {
    __shared__ __attribute__((aligned(16))) int seg[NT / 16][IPLAN_LDS_DEG];
    const int grp = threadIdx.x >> 4, sub = threadIdx.x & 15;
    const int* left = a.s[0].inds;
    const int n_left = a.s[0].n_left;
    for (int v = bid * (NT / 16) + grp; v < a.n_vars; v += nblk * (NT / 16)) {
        const int beg = a.v_ptr[v], n = a.v_ptr[v + 1] - beg;
        if (n > IPLAN_MAX_DEG) {
            if (sub == 0) atomicOr(&a.flags[3], 1);
            for (int k = sub; k < n; k += 16) { a.v_oth[beg + k] = 0; a.v_coef[beg + k] = 0.f; }
        } else if (n <= IPLAN_LDS_DEG) {
            const int n4 = (n + 3) & ~3;
            // the 16 lanes of a group sit in one wave: LDS operations of a wave complete in order, no barrier needed
            for (int k = sub; k < n4; k += 16) seg[grp][k] = k < n ? a.v_pos[beg + k] : 0x7fffffff;
            for (int k = sub; k < n; k += 16) {
                const int x = seg[grp][k];
                int rank = 0;
                for (int j = 0; j < n4; j += 4) {
                    const int4 y = *(const int4*)&seg[grp][j];
                    rank += (y.x < x) + (y.y < x) + (y.z < x) + (y.w < x);
                }
                a.v_oth[beg + rank] = min(max(left[x], 0), max(n_left - 1, 0));
                a.v_coef[beg + rank] = a.cons_coef[x];
            }
        } else {
            for (int k = sub; k < n; k += 16) {
                const int x = a.v_pos[beg + k];
                int rank = 0;
                for (int j = 0; j < n; ++j) rank += a.v_pos[beg + j] < x;
                a.v_oth[beg + rank] = min(max(left[x], 0), max(n_left - 1, 0));
                a.v_coef[beg + rank] = a.cons_coef[x];
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_iplan_order(IplanArgs a) { iplan_order_body<256>(a, blockIdx.x, gridDim.x); }

// Descending stable ranking of the scores (model_evaluator.py:110-111: sorted(range(n), key=quality, reverse=True) keeps equal
// scores in index order): the same bitonic network and total order as the ranking metric.  One block, n <= RK_MAX.
__global__ __launch_bounds__(256) void k_rank_scores(const float* __restrict__ scores, int n, int* __restrict__ order) {
    __shared__ float v[4096];
    __shared__ int ix[4096];
    int m = 1;
    while (m < n) m <<= 1;
    for (int i = threadIdx.x; i < m; i += 256) {
        const float x = i < n ? scores[i] : -INFINITY;
        v[i] = x != x ? -INFINITY : x; ix[i] = i < n ? i : 0x7fffffff;
    }
    __syncthreads();
    for (int k = 2; k <= m; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < m; i += 256) {
                const int p = i ^ j;
                if (p > i) {
                    const bool up = (i & k) == 0;
                    const float vi = v[i], vp = v[p];
                    const int ii = ix[i], ip = ix[p];
                    const bool before_pi = vp > vi || (vp == vi && ip < ii);   // p's element ranks before i's
                    const bool before_ip = vi > vp || (vi == vp && ii < ip);
                    if (up ? before_pi : before_ip) { v[i] = vp; v[p] = vi; ix[i] = ip; ix[p] = ii; }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < n; i += 256) order[i] = ix[i];
}

// ---------------------------------------------------------------------------------------------------------------
// The plan rides in the forward pass's first three launches.  Its three steps depend only on each other, and the forward
// needs them late: the embeddings need no graph at all, conv v->c only the by-left offsets (step 1), conv c->v is the first
// consumer of the by-variable order (step 3).  So each step runs as extra blocks of the launch that precedes its consumer --
//   launch 1: embeddings            + count     launch 2: conv v->c edge pass + place     launch 3: conv v->c row program + order
// -- three launches and three dependent kernel boundaries fewer per call (the SCIP plugin's call is latency, not throughput).
// ---------------------------------------------------------------------------------------------------------------
// SPLIT: the four-waves-per-tile programs of k_rows_split.hpp (few tiles; NWAVES == 4)
template <int NWAVES, bool SPLIT = false>
__global__ __launch_bounds__(NWAVES * 64) void k_infer_s1(EmbGroupArgs m, IplanArgs ia) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;
    if (SPLIT) {
        if (b < m.blk0[1]) emb_split<14, 2>(m.v, smem, b, m.blk0[1]);
        else if (b < m.blk0[2]) emb_split<4, 1>(m.c, smem, b - m.blk0[1], m.blk0[2] - m.blk0[1]);
        else if (b < m.blk0[3]) emb_split<6, 1>(m.k, smem, b - m.blk0[2], m.blk0[3] - m.blk0[2]);
        else if (b < m.blk0[3] + 3) fuse_weights(m.fz, b - m.blk0[3], smem);
        else iplan_count_body(ia, b - m.blk0[3] - 3, NWAVES * 64);
        return;
    }
    if (b < m.blk0[1]) emb_program<14, 2, NWAVES * 64>(m.v, smem, b, m.blk0[1]);
    else if (b < m.blk0[2]) emb_program<4, 1, NWAVES * 64>(m.c, smem, b - m.blk0[1], m.blk0[2] - m.blk0[1]);
    else if (b < m.blk0[3]) emb_program<6, 1, NWAVES * 64>(m.k, smem, b - m.blk0[2], m.blk0[3] - m.blk0[2]);
    else if (b < m.blk0[3] + 3) fuse_weights(m.fz, b - m.blk0[3], smem);
    else iplan_count_body(ia, b - m.blk0[3] - 3, NWAVES * 64);
}
template <bool BLOCKSEG>
__global__ __launch_bounds__(256) void k_infer_s2(EdgeArgs e, IplanArgs ia, int edge_blocks) {
    extern __shared__ int pre[];   // place blocks: [n_vars + 1]
    const int b = blockIdx.x;
    if (b >= edge_blocks) { iplan_place_body<256>(ia, pre, b - edge_blocks, gridDim.x - edge_blocks); return; }
    if (BLOCKSEG) { edge_fwd_block_body<false>(e, b, edge_blocks); return; }
    const float s1 = *e.s1;
    if (s1 < 0.f) edge_fwd_impl<4, false, true>(e, s1, b, edge_blocks); else edge_fwd_impl<4, false, false>(e, s1, b, edge_blocks);
}
template <int NWAVES, bool SPLIT = false>
__global__ __launch_bounds__(NWAVES * 64) void k_infer_s3(ConvFArgs a, IplanArgs ia, int conv_blocks) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;
    if (b < conv_blocks) {
        if (SPLIT) convf_split<CF_PROJ>(a, smem, b, conv_blocks);   // (the folded form: inference never keeps A)
        else convf_program<CF_PROJ, NWAVES * 64>(a, smem, b, conv_blocks);
    }
    else iplan_order_body<NWAVES * 64>(ia, b - conv_blocks, gridDim.x - conv_blocks);
}
