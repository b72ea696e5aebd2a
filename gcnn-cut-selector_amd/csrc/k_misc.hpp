#pragma once
#include "gcnn_common.hpp"

// ---------------------------------------------------------------------------------------------------------------
// The loss head (model_trainer.py:271) and the gradient of the readout tail Dense(64->1) (model.py:208); the forward of
// that Dense closes the last forward row program (convf_program<READOUT> in k_rows.hpp).
// ---------------------------------------------------------------------------------------------------------------
// MSE head (model_trainer.py:271): loss = scale * sum_k (score_k - y_k)^2, d_score_k = 2*scale*(score_k - y_k).  One block.
__global__ __launch_bounds__(256) void k_mse(const float* __restrict__ score, const float* __restrict__ target, float scale,
                                             float* __restrict__ loss, float* __restrict__ d_score, int n) {
    __shared__ float red[256];
    float ls = 0.f;
    for (int k = threadIdx.x; k < n; k += 256) {
        const float d = score[k] - target[k];
        ls = fmaf(d, d, ls);
        if (d_score) d_score[k] = 2.f * d * scale;
    }
    red[threadIdx.x] = ls;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0 && loss) *loss = red[0] * scale;
}

// gradient of Dense(64->1): dO1pre[k][j] = ds_k*w2[j]*[O1[k][j] > 0]; dw2[j] = sum_k ds_k O1[k][j]; db2 = sum_k ds_k.
// One block per SB_ROWS cuts; per-block partial slab [2*64]: dw2 partial, then db2 partial in element 64.
#define SB_ROWS 64
__global__ __launch_bounds__(256) void k_score_bwd(const float* __restrict__ d_score, const float* __restrict__ o1,
                                                   const float* __restrict__ w2, float* __restrict__ d_o1,
                                                   float* __restrict__ partial, float* __restrict__ count_out, int n) {
    __shared__ float red[4][EMB];
    __shared__ float red2[256];
    const int col = threadIdx.x & 63, part = threadIdx.x >> 6;
    const float wj = w2[col];
    const int k0 = blockIdx.x * SB_ROWS, k1 = min(n, k0 + SB_ROWS);
    float gw = 0.f;
    for (int k = k0 + part; k < k1; k += 4) {
        const float ds = d_score[k];
        const float ov = o1[(size_t)k * EMB + col];
        gw = fmaf(ds, ov, gw);
        d_o1[(size_t)k * EMB + col] = ov > 0.f ? ds * wj : 0.f;  // gradient w.r.t. the pre-activation of out_1 (ReLU mask)
    }
    const int kk = k0 + threadIdx.x;
    red[part][col] = gw; red2[threadIdx.x] = kk < k1 ? d_score[kk] : 0.f;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red2[threadIdx.x] += red2[threadIdx.x + s];
        __syncthreads();
    }
    float* slab = partial + (size_t)blockIdx.x * 2 * EMB;
    if (threadIdx.x < EMB) slab[threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (threadIdx.x == 0) slab[EMB] = red2[0];
    if (count_out && blockIdx.x == 0 && threadIdx.x == 0) *count_out = (float)n;
}

// Keras-form Adam (model_trainer.py:131,273): eps outside the bias-corrected sqrt.
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, int n, float lr_t, float b1, float b2, float eps,
                                              const float* __restrict__ gscale, int recip) {
    // recip: *gscale is the all-reduced cut count of a data-parallel step.  A global batch without a single cut has no
    // loss (model_trainer.py:271 would average over nothing): that is "no step" -- weights and moments stay as they are.
    if (gscale && recip && !(*gscale > 0.f)) return;
    const float gs = gscale ? (recip ? 1.f / *gscale : *gscale) : 1.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float gi = g[i] * gs;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}
// The same update with every hyper-parameter and the step counter resident on the device, so that a captured hipGraph
// can be replayed step after step: opt = {lr, beta1, beta2, eps, t (step count), lr_t}.  k_adam_tick advances t and
// computes lr_t = lr*sqrt(1-b2^t)/(1-b1^t) in double; k_adam_dev applies it.
__global__ void k_adam_tick(float* __restrict__ opt, const float* __restrict__ gscale, int recip) {
    if (gscale && recip && !(*gscale > 0.f)) return;   // a global batch without cuts is no step (see k_adam)
    const double t = (double)opt[4] + 1.0;
    opt[4] = (float)t;
    opt[5] = (float)((double)opt[0] * sqrt(1.0 - pow((double)opt[2], t)) / (1.0 - pow((double)opt[1], t)));
}
__global__ __launch_bounds__(256) void k_adam_dev(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                  float* __restrict__ v, int n, const float* __restrict__ opt,
                                                  const float* __restrict__ gscale, int recip) {
    // recip: *gscale is the all-reduced cut count of a data-parallel step.  A global batch without a single cut has no
    // loss (model_trainer.py:271 would average over nothing): that is "no step" -- weights and moments stay as they are.
    if (gscale && recip && !(*gscale > 0.f)) return;
    const float gs = gscale ? (recip ? 1.f / *gscale : *gscale) : 1.f;
    const float b1 = opt[1], b2 = opt[2], eps = opt[3], lr_t = opt[5];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float gi = g[i] * gs;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Ranking-prefix accuracy (model_trainer.py:280-302, model_tester.py:205-224) on the device.  Per sample: rank the cuts
// by predicted and by true improvement -- descending, ties in index order, which is what Python's stable
// sorted(..., reverse=True) yields -- and take the first position where the two rankings differ, over the number of cuts.
// One block per sample; both (value, index) lists are sorted in LDS with a bitonic network under the total order
// "larger value first, then smaller index", whose result IS the stable order.  acc[f] += [frac >= fractions[f]]
// (integer-valued float adds: exact and order-independent), frac_out[s] = frac.
// ---------------------------------------------------------------------------------------------------------------
#define RK_MAX 4096
__device__ __forceinline__ bool rk_before(float va, int ia, float vb, int ib) { return va > vb || (va == vb && ia < ib); }

__global__ __launch_bounds__(256) void k_ranking(const float* __restrict__ pred, const float* __restrict__ truth,
                                                 const int* __restrict__ offsets, const float* __restrict__ fractions,
                                                 int nfrac, float* __restrict__ acc, float* __restrict__ frac_out,
                                                 const float* __restrict__ loss_in, float loss_weight, float* __restrict__ loss_acc) {
    __shared__ float v[2][RK_MAX];
    __shared__ int ix[2][RK_MAX];
    __shared__ int first_dev;
    const int s = blockIdx.x, beg = offsets[s], n = offsets[s + 1] - beg;
    if (s == 0 && threadIdx.x == 0 && loss_acc) *loss_acc += *loss_in * loss_weight;
    if (n <= 0) { if (threadIdx.x == 0 && frac_out) frac_out[s] = 0.f; return; }
    if (n <= 256) {   // the usual case (a sample has a few dozen cuts): thread i counts the entries ranked before its own -- its
        // position in the stable descending order -- in both lists; position r of the two orders differs exactly when the
        // entry the prediction puts there has another position by the truth.  One barrier instead of a sorting network's 21-36.
        const int i = threadIdx.x;
        const bool in = i < n;
        float a = in ? pred[beg + i] : -INFINITY, b = in ? truth[beg + i] : -INFINITY;
        a = a != a ? -INFINITY : a; b = b != b ? -INFINITY : b;   // NaN ranks as -inf, as below
        v[0][i] = a; v[1][i] = b;
        if (i == 0) first_dev = n;
        __syncthreads();
        if (in) {
            int ra = 0, rb = 0;
            for (int j = 0; j < n; ++j) { ra += rk_before(v[0][j], j, a, i); rb += rk_before(v[1][j], j, b, i); }
            if (ra != rb) atomicMin(&first_dev, ra);
        }
        __syncthreads();
        if (i == 0) {
            const float frac = (float)first_dev / (float)n;
            if (frac_out) frac_out[s] = frac;
            for (int f = 0; f < nfrac; ++f)
                if (frac >= fractions[f]) atomicAdd(&acc[f], 1.0f);
        }
        return;
    }
    int m = 1;
    while (m < n) m <<= 1;
    for (int i = threadIdx.x; i < m; i += 256) {
        const bool in = i < n;
        const float a = in ? pred[beg + i] : -INFINITY, b = in ? truth[beg + i] : -INFINITY;
        v[0][i] = a != a ? -INFINITY : a; v[1][i] = b != b ? -INFINITY : b;   // NaN ranks as -inf: both lists stay permutations
        ix[0][i] = ix[1][i] = in ? i : 0x7fffffff;   // padding sorts last in both lists
    }
    if (threadIdx.x == 0) first_dev = n;
    __syncthreads();
    for (int k = 2; k <= m; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < m; i += 256) {
                const int p = i ^ j;
                if (p > i) {
                    const bool up = (i & k) == 0;
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        const float vi = v[a][i], vp = v[a][p];
                        const int ii = ix[a][i], ip = ix[a][p];
                        const bool swap = up ? rk_before(vp, ip, vi, ii) : rk_before(vi, ii, vp, ip);
                        if (swap) { v[a][i] = vp; v[a][p] = vi; ix[a][i] = ip; ix[a][p] = ii; }
                    }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < n; i += 256)
        if (ix[0][i] != ix[1][i]) atomicMin(&first_dev, i);
    __syncthreads();
    if (threadIdx.x == 0) {
        const float frac = (float)first_dev / (float)n;
        if (frac_out) frac_out[s] = frac;
        for (int f = 0; f < nfrac; ++f)
            if (frac >= fractions[f]) atomicAdd(&acc[f], 1.0f);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// PreNorm fitting statistics (PreNormLayer.update_params, model.py:394-423): per batch, the population mean and the
// mean squared deviation of a layer's input, per unit.  Offline path (pretraining), so: two passes (mean, then centred
// second moment), double accumulators, per-block partials summed in a fixed order.  Three element sources:
//   ST_COLS  a dense [n, f] matrix, one unit per column            (the five input PreNorm layers)
//   ST_FLAT  a dense [n, 64] matrix, ONE unit over all elements     (post_conv_module, model.py:503, 570)
//   ST_EDGE  the joint edge pre-activations J_e[64] = PL[l_e] + c_e*w + PR[v_e], ONE unit (feature_module_final's
//            PreNorm, model.py:498, 563-565) -- never materialised
// ---------------------------------------------------------------------------------------------------------------
enum { ST_COLS = 0, ST_FLAT = 1, ST_EDGE = 2 };
#define ST_MAX_UNITS 16
#define ST_MAX_BLOCKS 1024
struct StatArgs {
    int src; int n; int f;                  // rows (or edges), columns/units
    const float* x;                         // ST_COLS / ST_FLAT
    const int* left; const int* right; const float* coef; const float* pl; const float* pr; const float* w_edge;
    const float* e_shift; const float* e_scale;   // ST_EDGE (by-left order arrays: left id via seg search is avoided: `left` is explicit)
    const double* mean;                     // pass 2: centre (device, [units]); nullptr in pass 1
    double* partial;                        // [gridDim.x][units]
};

__global__ __launch_bounds__(256) void k_stats(StatArgs a) {
    __shared__ double red[256];
    const int units = a.src == ST_COLS ? a.f : 1;
    double acc[ST_MAX_UNITS];
#pragma unroll
    for (int u = 0; u < ST_MAX_UNITS; ++u) acc[u] = 0.0;
    if (a.src == ST_COLS) {
        for (int r = blockIdx.x * 256 + threadIdx.x; r < a.n; r += gridDim.x * 256)
#pragma unroll
            for (int u = 0; u < ST_MAX_UNITS; ++u)
                if (u < a.f) {
                    const double v = (double)a.x[(size_t)r * a.f + u];
                    if (a.mean) { const double d = v - a.mean[u]; acc[u] += d * d; } else acc[u] += v;
                }
    } else if (a.src == ST_FLAT) {
        const double mu = a.mean ? a.mean[0] : 0.0;
        const size_t total = (size_t)a.n * EMB;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
            const double v = (double)a.x[i];
            if (a.mean) { const double d = v - mu; acc[0] += d * d; } else acc[0] += v;
        }
    } else {
        const double mu = a.mean ? a.mean[0] : 0.0;
        const float esh = *a.e_shift, esc = *a.e_scale;
        const int ch = (threadIdx.x & 15) * 4;
        const float4 w = *(const float4*)(a.w_edge + ch);
        for (int e = blockIdx.x * 16 + (threadIdx.x >> 4); e < a.n; e += gridDim.x * 16) {
            const float c = (a.coef[e] + esh) * esc;
            const float4 p = *(const float4*)(a.pl + (size_t)a.left[e] * EMB + ch);
            const float4 q = *(const float4*)(a.pr + (size_t)a.right[e] * EMB + ch);
            const float jv[4] = {jointf(p.x, __fmul_rn(c, w.x), q.x), jointf(p.y, __fmul_rn(c, w.y), q.y),
                                 jointf(p.z, __fmul_rn(c, w.z), q.z), jointf(p.w, __fmul_rn(c, w.w), q.w)};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double v = (double)jv[k];
                if (a.mean) { const double d = v - mu; acc[0] += d * d; } else acc[0] += v;
            }
        }
    }
    for (int u = 0; u < units; ++u) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < ST_MAX_UNITS; ++k) if (k == u) v = acc[k];
        red[threadIdx.x] = v;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) a.partial[(size_t)blockIdx.x * units + u] = red[0];
        __syncthreads();
    }
}
// out[u] = (sum over blocks of partial[b][u]) / count      (one block, fixed order)
__global__ __launch_bounds__(64) void k_stats_final(const double* __restrict__ partial, int nblocks, int units, double count,
                                                    double* __restrict__ out) {
    const int u = threadIdx.x;
    if (u >= units) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += partial[(size_t)b * units + u];
    out[u] = s / count;
}
// expand a by-left CSR pointer into explicit left ids (pretraining only)
__global__ void k_expand_ptr(const int* __restrict__ ptr, int n_seg, int* __restrict__ ids) {
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n_seg; r += gridDim.x * blockDim.x)
        for (int e = ptr[r]; e < ptr[r + 1]; ++e) ids[e] = r;
}

// ---------------------------------------------------------------------------------------------------------------
// device-side batch collation: gather the samples of one mini-batch out of a device-resident sample store
// (disjoint-union batching, utils.py:389-426).  Every store array keeps sample-local values; one launch copies all
// arrays of the batch (blockIdx.y = array), shifting indices by the batch offset of the sample they land in.
// ---------------------------------------------------------------------------------------------------------------
constexpr int COLLATE_MAX_JOBS = 24;
struct CollateJob {
    const int* src;
    int* dst;
    int unit_kind, width, add_kind, is_ptr;
};
struct CollateArgs {
    CollateJob job[COLLATE_MAX_JOBS];
    const long long* src_off;   // [n_kinds][batch]   first unit of each chosen sample in the store
    const long long* dst_off;   // [n_kinds][batch+1] first unit of each sample in the batch; last = batch total
    int batch;
};
// A sample's segment of an array is contiguous in the store AND in the batch, so the launch is a set of straight copies:
// block (x, y) takes chunk x % COLLATE_CHUNKS of sample x / COLLATE_CHUNKS of array y -- no per-word search for the sample a
// word belongs to, no per-word division by the row width (both were there at first: 116 us per setcov batch of 32; now ~10).
#define COLLATE_CHUNKS 8
__global__ __launch_bounds__(256) void k_collate(CollateArgs a) {
    const CollateJob j = a.job[blockIdx.y];
    const long long* so = a.src_off + (size_t)j.unit_kind * a.batch;
    const long long* dof = a.dst_off + (size_t)j.unit_kind * (a.batch + 1);
    const long long* aof = j.add_kind >= 0 ? a.dst_off + (size_t)j.add_kind * (a.batch + 1) : nullptr;
    if (j.is_ptr && blockIdx.x == 0 && threadIdx.x == 0)   // trailing entry of a segment-offset array: the batch's edge total
        j.dst[dof[a.batch] * j.width] = (int)aof[a.batch];
    for (int item = blockIdx.x; item < a.batch * COLLATE_CHUNKS; item += gridDim.x) {
        const int s = item / COLLATE_CHUNKS, c = item - s * COLLATE_CHUNKS;
        const long long words = (dof[s + 1] - dof[s]) * j.width;
        const long long per = (words + COLLATE_CHUNKS - 1) / COLLATE_CHUNKS;
        const long long w0 = min(words, c * per), w1 = min(words, w0 + per);
        const int* __restrict__ src = j.src + so[s] * j.width + w0;
        int* __restrict__ dst = j.dst + dof[s] * j.width + w0;
        const int add = aof ? (int)aof[s] : 0, n = (int)(w1 - w0);
        for (int i = threadIdx.x; i < n; i += 256) dst[i] = src[i] + add;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// graph plan kernels
// ---------------------------------------------------------------------------------------------------------------
// one pass over the COO list: flags[0] |= out-of-range index, flags[1] |= left ids not non-decreasing
__global__ void k_check_edges(const int* __restrict__ ei, int n, int n_left, int n_var, int* __restrict__ flags) {
    int bad = 0, unsorted = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int l = ei[i], v = ei[n + i];
        bad |= (l < 0) | (l >= n_left) | (v < 0) | (v >= n_var);
        if (i + 1 < n) unsorted |= ei[i + 1] < l;
    }
    if (bad) atomicOr(&flags[0], 1);
    if (unsorted) atomicOr(&flags[1], 1);
}
__global__ void k_iota(int* p, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = i;
}
// sorted keys -> segment offsets: ptr[k] = first position whose key >= k, ptr[n_seg] = n
__global__ void k_seg_offsets(const int* __restrict__ keys, int n, int n_seg, int* __restrict__ ptr) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += gridDim.x * blockDim.x) {
        const int lo = i == 0 ? -1 : max(keys[i - 1], -1);
        const int hi = i == n ? n_seg : min(keys[i], n_seg);
        for (int k = lo + 1; k <= hi; ++k) ptr[k] = i;
    }
}
// inv[perm[i]] = i
__global__ void k_invert_perm(const int* __restrict__ perm, int n, int* __restrict__ inv) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) inv[perm[i]] = i;
}
// out[i] = inv[perm[i]]
__global__ void k_compose_perm(const int* __restrict__ perm, const int* __restrict__ inv, int n, int* __restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = inv[perm[i]];
}
__global__ void k_gather_edges(const int* __restrict__ perm, const int* __restrict__ other, const float* __restrict__ coef,
                               int n, int* __restrict__ oth_out, float* __restrict__ coef_out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int e = perm[i];
        oth_out[i] = other[e]; coef_out[i] = coef[e];
    }
}

