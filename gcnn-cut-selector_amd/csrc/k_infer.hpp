#pragma once
#include "gcnn_common.hpp"

// ---------------------------------------------------------------------------------------------------------------
// Graph plan for ONE sampled state (the SCIP cut selector calls the model once per separation round on a single graph,
// model_evaluator.py:82-103): three launches instead of a device-wide radix sort.  Inference needs three segment
// structures: constraint edges by constraint (conv v->c), constraint edges by variable (conv c->v) and cut edges by cut
// (conv v->k).  The reference's get_state emits (row, col)-sorted COO (utils.py:102-104), so the two by-left structures are
// the input lists themselves plus segment offsets; only the by-variable order of the constraint edges has to be built:
//   1. k_iplan_count : one pass over both lists -- range check, sortedness check, by-left offsets, per-variable counts
//   2. k_iplan_place : every block scans the counts in LDS (no separate scan launch), then each edge takes a slot of its
//                      variable's segment with an integer atomic (arrival order)
//   3. k_iplan_order : per variable, ranks the segment's input positions (unique, so rank = stable order) and gathers
//                      (left id, coefficient) into place -- the result equals a stable sort by variable id
// flags[0] index out of range, flags[1] constraint list not sorted by row, flags[2] cut list not sorted by row, flags[3] a
// variable with more edges than IPLAN_MAX_DEG: in all four cases the caller takes the general path (gcnn_graph_build).
// ---------------------------------------------------------------------------------------------------------------
#define IPLAN_MAX_VARS 32768   // the per-block scan of the counts lives in LDS (128 KB)
#define IPLAN_MAX_DEG 2048

struct IplanSet { int* inds; int n_edges, n_left; int* l_ptr; };   // inds = [2,E]: left ids then variable ids
struct IplanArgs {
    IplanSet s[2];        // 0: constraint edges, 1: cut edges
    int n_vars;
    int* vcount;          // [n_vars]   zero on entry
    int* cursor;          // [n_vars]   zero on entry
    int* flags;           // [4]        zero on entry
    int* v_ptr; int* v_pos; int* v_oth; float* v_coef;   // by-variable structure of the constraint edges
    const float* cons_coef;
    int blocks0;          // blocks of set 0 in the count launch
};

// Robustness: the forward pass is launched right behind the plan without looking at the flags, so whatever the input, every
// structure the forward reads must stay inside its arrays.  Out-of-range ids are replaced by 0 in the uploaded copy (and
// counted as such), the by-left offset arrays arrive zero-filled with the upload (an unsorted list leaves gaps), and an
// over-long variable segment is filled with zeros.  The flags tell the host that the scores of such a call mean nothing.
__global__ __launch_bounds__(256) void k_iplan_count(IplanArgs a) {
    const int set = (int)blockIdx.x >= a.blocks0;
    const IplanSet s = a.s[set];
    const int i = ((int)blockIdx.x - (set ? a.blocks0 : 0)) * 256 + (int)threadIdx.x;
    if (i > s.n_edges) return;
    int* left = s.inds;
    int* var = s.inds + s.n_edges;
    int bad = 0, unsorted = 0;
    // by-left offsets: ptr[k] = first position whose left id >= k (ids clamped so that a bad index cannot write out of range)
    const int lo = i == 0 ? -1 : min(max(left[i - 1], -1), s.n_left);
    const int hi = i == s.n_edges ? s.n_left : min(max(left[i], -1), s.n_left);
    if (i < s.n_edges) {
        const int l = left[i];
        int v = var[i];
        const bool bad_l = l < 0 || l >= s.n_left, bad_v = v < 0 || v >= a.n_vars;
        bad = bad_l | bad_v;
        if (i + 1 < s.n_edges) unsorted = left[i + 1] < l;
        if (bad_v) { v = 0; var[i] = 0; }
        if (set == 0 && a.n_vars > 0) atomicAdd(&a.vcount[v], 1);
    }
    for (int k = lo + 1; k <= hi; ++k) s.l_ptr[k] = i;
    if (bad) atomicOr(&a.flags[0], 1);
    if (unsorted) atomicOr(&a.flags[1 + set], 1);
}
// (left ids out of range are clamped on use: they only ever index the offset array above; the lists keep them, the by-variable
// structure below stores them clamped.)

// exclusive scan of cnt[0..n) into LDS `pre` (n <= IPLAN_MAX_VARS), total in pre[n]; 1024 threads
__device__ __forceinline__ void iplan_scan(const int* __restrict__ cnt, int n, int* pre) {
    __shared__ int wsum[16];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int per = (n + 1023) / 1024, b = min(n, t * per), e = min(n, b + per);
    int sum = 0;
    for (int i = b; i < e; ++i) { const int c = cnt[i]; pre[i] = sum; sum += c; }
    int inc = sum;                                   // inclusive scan over the wave, then over the 16 waves
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(inc, off); if (lane >= off) inc += o; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int base = inc - sum;
    for (int w = 0; w < wv; ++w) base += wsum[w];
    for (int i = b; i < e; ++i) pre[i] += base;
    if (t == 1023) pre[n] = base + sum;
    __syncthreads();
}

__global__ __launch_bounds__(1024) void k_iplan_place(IplanArgs a) {
    extern __shared__ int pre[];   // [n_vars + 1]
    iplan_scan(a.vcount, a.n_vars, pre);
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i <= a.n_vars; i += 1024) a.v_ptr[i] = pre[i];
    const int n = a.s[0].n_edges;
    for (int e = blockIdx.x * 1024 + threadIdx.x; e < n; e += gridDim.x * 1024) {
        const int v = a.s[0].inds[n + e];           // sanitised by k_iplan_count
        a.v_pos[pre[v] + atomicAdd(&a.cursor[v], 1)] = e;
    }
}

// 16 lanes per variable.  Input positions are unique, so an element's rank among its segment is its place in the stable
// order.  The segment is staged in LDS (padded with INT_MAX) and every element compares itself with all of it, four at a
// time -- quadratic in the degree, which is what IPLAN_MAX_DEG bounds (segments beyond IPLAN_LDS_DEG re-read global memory).
#define IPLAN_LDS_DEG 256
__global__ __launch_bounds__(256) void k_iplan_order(IplanArgs a) {
    __shared__ __attribute__((aligned(16))) int seg[16][IPLAN_LDS_DEG];
    const int grp = threadIdx.x >> 4, v = blockIdx.x * 16 + grp, sub = threadIdx.x & 15;
    if (v >= a.n_vars) return;
    const int beg = a.v_ptr[v], n = a.v_ptr[v + 1] - beg;
    const int* left = a.s[0].inds;
    const int n_left = a.s[0].n_left;
    if (n > IPLAN_MAX_DEG) {
        if (sub == 0) atomicOr(&a.flags[3], 1);
        for (int k = sub; k < n; k += 16) { a.v_oth[beg + k] = 0; a.v_coef[beg + k] = 0.f; }
        return;
    }
    if (n <= IPLAN_LDS_DEG) {
        const int n4 = (n + 3) & ~3;
        for (int k = sub; k < n4; k += 16) seg[grp][k] = k < n ? a.v_pos[beg + k] : 0x7fffffff;
        // the 16 lanes of a group sit in one wave: LDS operations of a wave complete in order, no barrier needed
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

// Descending stable ranking of the scores (model_evaluator.py:110-111: sorted(range(n), key=quality, reverse=True) keeps equal
// scores in index order): the same bitonic network and total order as the ranking metric.  One block, n <= RK_MAX.
__global__ __launch_bounds__(256) void k_rank_scores(const float* __restrict__ scores, int n, int* __restrict__ order) {
    __shared__ float v[4096];
    __shared__ int ix[4096];
    int m = 1;
    while (m < n) m <<= 1;
    for (int i = threadIdx.x; i < m; i += 256) { v[i] = i < n ? scores[i] : -INFINITY; ix[i] = i < n ? i : 0x7fffffff; }
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
