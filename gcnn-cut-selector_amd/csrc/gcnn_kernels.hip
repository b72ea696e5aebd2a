// gcnn_kernels.hip -- hand-written gfx950 (MI355X / CDNA4) kernels + C ABI for the bipartite GCNN hot path.
//
// Reference semantics (all cites into /root/reference): GCNN.call model.py:257-300, PartialGraphConvolution.call
// model.py:533-575, PreNormLayer.call model.py:365-382, loss/step model_trainer.py:266-273.
//
// Design: every node tensor is a row-major [N,64] fp32 matrix (256-B rows).  The per-edge
// Dense(64->64) of the reference (model.py:499-500) is hoisted past the scatter-sum
//   sum_e (H_e W_f + b_f) = (sum_e H_e) W_f + deg_r b_f
// so the edge pass only gathers rows, applies ReLU and accumulates in registers (atomic-free segmented sum over
// receiver-sorted CSR); all 64x64 products run on the fp32 MFMA (v_mfma_f32_32x32x2_f32) with weights staged in LDS.
// Wavefront = 64 lanes everywhere.  No atomics on floats anywhere: every sum has a fixed order => bitwise
// reproducible results.

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "../../include/gcnn_hip.h"

#define EMB 64
#define LDW 68  // padded LDS row stride in floats: 272 B keeps 16-B alignment, b128 row reads conflict-free

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------------------------
// parameter layout (checkpoint order, model.py:53-56/215; shapes model.py:174-208, 486-508)
// ---------------------------------------------------------------------------------------------------------------
struct PInfo { int off, rows, cols, trainable; };
static PInfo g_pinfo[GCNN_N_PARAMS];
static int g_ptotal = 0;

enum {  // indices into the 62-array list
    P_CONS = 0, P_CONS_EDGE = 6, P_VAR = 8, P_CUT = 14, P_CUT_EDGE = 20, P_CONV0 = 22, P_CONV1 = 34, P_CONV2 = 46,
    P_OUT = 58
};
enum { E_SHIFT = 0, E_SCALE = 1, E_W1 = 2, E_B1 = 3, E_W2 = 4, E_B2 = 5 };  // embedding block
enum { C_WL = 0, C_BL = 1, C_WE = 2, C_WR = 3, C_S1 = 4, C_WF = 5, C_BF = 6, C_S2 = 7, C_W1 = 8, C_B1 = 9, C_W2 = 10,
       C_B2 = 11 };  // conv block

static void layout_init() {
    if (g_ptotal) return;
    int n = 0, off = 0;
    auto add = [&](int rows, int cols, int tr) {
        g_pinfo[n].off = off; g_pinfo[n].rows = rows; g_pinfo[n].cols = cols; g_pinfo[n].trainable = tr;
        off += (rows * cols + 3) & ~3; ++n;
    };
    auto emb = [&](int f) { add(1, f, 0); add(1, f, 0); add(f, EMB, 1); add(1, EMB, 1); add(EMB, EMB, 1); add(1, EMB, 1); };
    auto conv = [&]() {
        add(EMB, EMB, 1); add(1, EMB, 1); add(1, EMB, 1); add(EMB, EMB, 1); add(1, 1, 0); add(EMB, EMB, 1); add(1, EMB, 1);
        add(1, 1, 0); add(2 * EMB, EMB, 1); add(1, EMB, 1); add(EMB, EMB, 1); add(1, EMB, 1);
    };
    emb(4); add(1, 1, 0); add(1, 1, 0); emb(14); emb(6); add(1, 1, 0); add(1, 1, 0);
    conv(); conv(); conv();
    add(EMB, EMB, 1); add(1, EMB, 1); add(EMB, 1, 1); add(1, 1, 1);
    g_ptotal = off;
}
static inline int poff(int i) { return g_pinfo[i].off; }

// ---------------------------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// Blocks b and b+8 share an XCD (round-robin dispatch).  Give every XCD a contiguous range of work items so the
// rows a range gathers stay in that XCD's 4 MiB L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// joint pre-activation of one edge, in the reference's association order: (left + coef*w) + right, model.py:564-565
__device__ __forceinline__ float jointf(float pl, float cw, float pr) { return __fadd_rn(__fadd_rn(pl, cw), pr); }

// ---------------------------------------------------------------------------------------------------------------
// K2a: first embedding layer, relu(((x+shift)*scale) @ W[f,64] + b)      model.py:174-177 (and var/cut twins)
// 16 lanes per row, 4 channels per lane.  F <= 16.
// ---------------------------------------------------------------------------------------------------------------
template <int F>
__global__ __launch_bounds__(256) void k_embed1_fwd(const float* __restrict__ x, const float* __restrict__ shift,
                                                    const float* __restrict__ scale, const float* __restrict__ w,
                                                    const float* __restrict__ b, float* __restrict__ y, int n) {
    const int ch = (threadIdx.x & 15) * 4;
    float4 wr[F];
#pragma unroll
    for (int f = 0; f < F; ++f) wr[f] = *(const float4*)(w + f * EMB + ch);
    const float4 bb = *(const float4*)(b + ch);
    float sh[F], sc[F];
#pragma unroll
    for (int f = 0; f < F; ++f) { sh[f] = shift[f]; sc[f] = scale[f]; }
    for (int r = blockIdx.x * 16 + (threadIdx.x >> 4); r < n; r += gridDim.x * 16) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const float xv = (x[(size_t)r * F + f] + sh[f]) * sc[f];
            acc.x = fmaf(xv, wr[f].x, acc.x); acc.y = fmaf(xv, wr[f].y, acc.y);
            acc.z = fmaf(xv, wr[f].z, acc.z); acc.w = fmaf(xv, wr[f].w, acc.w);
        }
        acc.x = fmaxf(acc.x + bb.x, 0.f); acc.y = fmaxf(acc.y + bb.y, 0.f);
        acc.z = fmaxf(acc.z + bb.z, 0.f); acc.w = fmaxf(acc.w + bb.w, 0.f);
        *(float4*)(y + (size_t)r * EMB + ch) = acc;
    }
}

// gradient of the first embedding layer's weights: dW[f][j] = sum_r xn[r][f] * dPre[r][j], db[j] = sum_r dPre[r][j]
// with dPre = dY * (Y > 0).  One block per chunk of rows; per-block partial slab [(F+1)*64] (row F = bias).
template <int F>
__global__ __launch_bounds__(256) void k_embed1_wgrad(const float* __restrict__ x, const float* __restrict__ shift,
                                                      const float* __restrict__ scale, const float* __restrict__ dy,
                                                      const float* __restrict__ yact, float* __restrict__ partial,
                                                      int n, int rows_per_block) {
    __shared__ float red[4][(F + 1) * EMB];
    const int col = threadIdx.x & 63, part = threadIdx.x >> 6;
    float acc[F + 1];
#pragma unroll
    for (int f = 0; f <= F; ++f) acc[f] = 0.f;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(n, r0 + rows_per_block);
#pragma unroll 4
    for (int r = r0 + part; r < r1; r += 4) {
        float d = dy[(size_t)r * EMB + col];
        d = yact[(size_t)r * EMB + col] > 0.f ? d : 0.f;
#pragma unroll
        for (int f = 0; f < F; ++f) acc[f] = fmaf((x[(size_t)r * F + f] + shift[f]) * scale[f], d, acc[f]);
        acc[F] += d;
    }
#pragma unroll
    for (int f = 0; f <= F; ++f) red[part][f * EMB + col] = acc[f];
    __syncthreads();
    for (int i = threadIdx.x; i < (F + 1) * EMB; i += 256)
        partial[(size_t)blockIdx.x * (F + 1) * EMB + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

// ---------------------------------------------------------------------------------------------------------------
// K3/K4/K8'/K11/K12 and their input gradients: the generic 64-wide node GEMM on fp32 MFMA.
//
//   NN mode (forward):   y  = act( (sa*xa) @ wa  [+ xb @ wb] [+ bias] [+ deg (x) bd] )
//   TN mode (backward):  xa' = xa * (ymask > 0) (written back in place when write_back), then
//                        y  (=|+=) so * (xa' @ wa^T)      and optionally      y2 (=|+=) xa' @ wb^T
//
// Block = 4 waves; each wave owns a 32-row tile: stage rows in LDS (full 256-B lines, coalesced), read A fragments as
// b128 along k.  The k index is split between the two lane halves (half h takes k in [32h, 32h+32)): the order of the
// 64 products in each dot product is 0,32,1,33,... (exact fp32 FMA chain, different association than a CPU GEMM).
// Weights sit in LDS as [64][LDW]; B fragments are b32 reads along a row (NN) or b128 reads along k (TN), both
// conflict-free with LDW = 68.
// ---------------------------------------------------------------------------------------------------------------
struct LinArgs {
    const float* xa; const float* ymask; int write_back;
    const float* sa;
    const float* wa; const float* xb; const float* wb;
    const float* bias; const float* bd; const int* seg_ptr;
    const float* so;
    float* y; int beta_y; float* y2; int beta_y2;
    int relu; int n;
};

__device__ __forceinline__ void block_load_w(float* wl, const float* __restrict__ w) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = i * 256 + threadIdx.x;  // float4 index, 1024 of them
        const int r = idx >> 4, c = (idx & 15) * 4;
        *(float4*)(wl + r * LDW + c) = *(const float4*)(w + r * EMB + c);
    }
}

__device__ __forceinline__ void wave_load_tile(float* xs, const float* __restrict__ x, int row0, int n, float scale,
                                               const float* __restrict__ ymask, int write_back, float* xwb, int lane) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = i * 4 + (lane >> 4), c = (lane & 15) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + r < n) {
            const size_t o = (size_t)(row0 + r) * EMB + c;
            v = *(const float4*)(x + o);
            if (ymask) {
                const float4 m = *(const float4*)(ymask + o);
                v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f;
                v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
                if (write_back) *(float4*)(xwb + o) = v;
            }
            v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        }
        *(float4*)(xs + r * LDW + c) = v;
    }
}

template <bool TRANSB>
__device__ __forceinline__ void wave_gemm(const float* xs, const float* wl, f32x16 (&acc)[2], int lane) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float4 a = *(const float4*)(xs + r * LDW + h * 32 + q * 4);
        const float av[4] = {a.x, a.y, a.z, a.w};
        if (TRANSB) {
            const float4 b0 = *(const float4*)(wl + r * LDW + h * 32 + q * 4);
            const float4 b1 = *(const float4*)(wl + (32 + r) * LDW + h * 32 + q * 4);
            const float b0v[4] = {b0.x, b0.y, b0.z, b0.w}, b1v[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[0] = mfma32(av[t], b0v[t], acc[0]);
                acc[1] = mfma32(av[t], b1v[t], acc[1]);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int k = h * 32 + q * 4 + t;
                acc[0] = mfma32(av[t], wl[k * LDW + r], acc[0]);
                acc[1] = mfma32(av[t], wl[k * LDW + 32 + r], acc[1]);
            }
        }
    }
}

// accumulator (C/D layout: col = lane&31, row = (i&3) + 8*(i>>2) + 4*(lane>>5)) -> row-major LDS tile
__device__ __forceinline__ void wave_acc_to_tile(float* xs, const f32x16 (&acc)[2], int lane) {
    const int j = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int i = 0; i < 16; ++i) xs[((i & 3) + 8 * (i >> 2) + 4 * hh) * LDW + ct * 32 + j] = acc[ct][i];
}

__device__ __forceinline__ void wave_store_tile(const float* xs, float* __restrict__ y, int row0, int n, int beta,
                                                int lane) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = i * 4 + (lane >> 4), c = (lane & 15) * 4;
        if (row0 + r < n) {
            float4 v = *(const float4*)(xs + r * LDW + c);
            float* dst = y + (size_t)(row0 + r) * EMB + c;
            if (beta) {
                const float4 o = *(const float4*)dst;
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            *(float4*)dst = v;
        }
    }
}

template <bool TRANSB>
__global__ __launch_bounds__(256) void k_linear(LinArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wla = smem;                                  // [64][LDW]
    float* wlb = smem + 64 * LDW;                       // [64][LDW] (second weight, optional)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* xs = smem + 2 * 64 * LDW + wv * 32 * LDW;    // per-wave [32][LDW] tile

    block_load_w(wla, a.wa);
    if (a.wb) block_load_w(wlb, a.wb);
    __syncthreads();

    const float sa = a.sa ? *a.sa : 1.f;
    const float so = a.so ? *a.so : 1.f;
    const int j = lane & 31, hh = lane >> 5;
    const int ntile = (a.n + 31) >> 5;
    for (int tile = blockIdx.x * 4 + wv; tile < ntile; tile += gridDim.x * 4) {
        const int row0 = tile * 32;
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }

        wave_load_tile(xs, a.xa, row0, a.n, sa, a.ymask, a.write_back, const_cast<float*>(a.xa), lane);
        wave_gemm<TRANSB>(xs, wla, acc, lane);
        if (!TRANSB) {
            if (a.xb) {
                wave_load_tile(xs, a.xb, row0, a.n, 1.f, nullptr, 0, nullptr, lane);
                wave_gemm<false>(xs, wlb, acc, lane);
            }
            // epilogue: bias, degree-weighted bias (the hoisted b_f, model.py:499-500 + 568), activation
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const float bv = a.bias ? a.bias[ct * 32 + j] : 0.f;
                const float bdv = a.bd ? a.bd[ct * 32 + j] : 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float v = acc[ct][i] + bv;
                    if (a.bd) {
                        const int row = row0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                        const float deg = row < a.n ? (float)(a.seg_ptr[row + 1] - a.seg_ptr[row]) : 0.f;
                        v = fmaf(deg, bdv, v);
                    }
                    acc[ct][i] = a.relu ? fmaxf(v, 0.f) : v;
                }
            }
            wave_acc_to_tile(xs, acc, lane);
            wave_store_tile(xs, a.y, row0, a.n, a.beta_y, lane);
        } else {
            f32x16 acc2[2];
            if (a.y2) {
#pragma unroll
                for (int i = 0; i < 16; ++i) { acc2[0][i] = 0.f; acc2[1][i] = 0.f; }
                wave_gemm<true>(xs, wlb, acc2, lane);
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[ct][i] *= so;
            wave_acc_to_tile(xs, acc, lane);
            wave_store_tile(xs, a.y, row0, a.n, a.beta_y, lane);
            if (a.y2) {
                wave_acc_to_tile(xs, acc2, lane);
                wave_store_tile(xs, a.y2, row0, a.n, a.beta_y2, lane);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Fused row chains.  Every node-side layer of the model is row-local (a 16-row tile of a [N,64] matrix goes through
// a sequence of 64x64 products with element-wise epilogues), so a whole sequence -- e.g. S -> A -> Z1 -> X' -> PL'
// of one PartialGraphConvolution (model.py:498-508, 570-573) or its gradient -- runs in ONE launch: each wave owns
// 16-row tiles, reads every weight of the chain from LDS (staged once per block) and stores only the tensors the
// backward pass / the next edge pass need.  MFMA: v_mfma_f32_16x16x4_f32, 4 independent accumulators.
// ---------------------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

#define CH_MAX_STAGES 6
#define CH_MAX_W 5
#define CH_TILE (16 * LDW)
enum { CH_GEMM = 0, CH_EMBED1 = 1, CH_SCORE = 2 };
struct ChStage {
    int type;
    // A operand: global rows (loaded into LDS tile `ta`) or, when in_a == nullptr, whatever tile `ta` already holds
    const float* in_a; const float* sa; int ta; int wa;
    // optional second product accumulated into the same output: always from global, via tile `tb`
    const float* in_b; int tb; int wb;
    int transb;                       // 0: x @ W (forward), 1: x @ W^T (input gradients)
    // epilogue, in this order: *so, +bias, +deg*bd, +add, relu, *(mask > 0)
    const float* so; const float* bias; const float* bd; const int* seg_ptr; const float* add; const float* mask;
    int relu;
    float* out; int tout;             // global store (optional) and the LDS tile that keeps the result
    // optional element-wise side output of the result v:  em_out = *em_s * v * em_a
    const float* em_s; const float* em_a; float* em_out;
    // CH_EMBED1: x_raw [N,F], PreNorm shift/scale [F], kernel [F,64] (global), bias via `bias`
    const float* x_raw; const float* shift; const float* scale; const float* w1; int nfeat;
    // CH_SCORE: out[r] = tile(ta)[r] . w1[0:64] + *bias
};
struct ChArgs { int n; int nstage; int nw; const float* w[CH_MAX_W]; ChStage st[CH_MAX_STAGES]; };

// Register-resident chains.  Each stage computes the TRANSPOSED product  Y^T[64 x 16 rows] = Wop[64 x 64] . X^T  with the
// weights as the MFMA A operand (read from LDS, independent of the data, so the reads run ahead) and the activation
// tile as the B operand.  With that orientation the accumulator of one stage IS the B operand of the next one:
//   lane (j = lane&15, g = lane>>4) holds, for each 16-feature block mt and i = 0..3, the element
//   X[row0 + j][16*mt + 4*g + i]   -- as B operand of k-step (mt, i) (the instruction's k index is g), and as C/D
//   layout of the output block mo (rows of D = features 4*g + i of block mo, column = row j of the tile).
// So a whole chain runs without any LDS round trip for activations; global rows are read/written as float4 pieces
// X[row][16*mt + 4*g .. +3] straight from/to that layout.  The direction (x@W forward / x@W^T backward) is a
// compile-time parameter; the stage program itself is data (ChArgs).
struct RTile { float v[4][4]; };

__device__ __forceinline__ void rt_zero(RTile& t) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) t.v[m][i] = 0.f;
}
__device__ __forceinline__ void rt_load(RTile& t, const float* __restrict__ x, int row, bool ok, int g) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) f = *(const float4*)(x + (size_t)row * EMB + 16 * m + 4 * g);
        t.v[m][0] = f.x; t.v[m][1] = f.y; t.v[m][2] = f.z; t.v[m][3] = f.w;
    }
}
__device__ __forceinline__ void rt_store(const RTile& t, float* __restrict__ x, int row, bool ok, int g) {
    if (!ok) return;
#pragma unroll
    for (int m = 0; m < 4; ++m)
        *(float4*)(x + (size_t)row * EMB + 16 * m + 4 * g) = make_float4(t.v[m][0], t.v[m][1], t.v[m][2], t.v[m][3]);
}

// acc[mo] += Wop[16*mo + (lane&15)][kf] * T[kf], kf = 16*mt + 4*g + i;  NN: Wop[o][k] = W[k][o],  TN: Wop[o][k] = W[o][k]
template <bool TRANSB>
__device__ __forceinline__ void rt_gemm(const RTile& t, float scale, const float* wl, f32x4 (&acc)[4], int lane) {
    const int m = lane & 15, g = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        float av[4][4];  // [mo][i]
        if (TRANSB) {
#pragma unroll
            for (int mo = 0; mo < 4; ++mo) {
                const float4 w4 = *(const float4*)(wl + (16 * mo + m) * LDW + 16 * mt + 4 * g);
                av[mo][0] = w4.x; av[mo][1] = w4.y; av[mo][2] = w4.z; av[mo][3] = w4.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int mo = 0; mo < 4; ++mo) av[mo][i] = wl[(16 * mt + 4 * g + i) * LDW + 16 * mo + m];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float b = t.v[mt][i] * scale;
#pragma unroll
            for (int mo = 0; mo < 4; ++mo) acc[mo] = mfma16(av[mo][i], b, acc[mo]);
        }
    }
}

#define CH_PAR 144  // per-stage LDS parameter block: bias[64], bd[64], {sa, so, es, score bias}, padding
template <int NWAVES, bool TRANSB>
__global__ __launch_bounds__(NWAVES * 64) void k_chain(ChArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int ntile = (a.n + 15) >> 4;
    const int tile0 = blockIdx.x * NWAVES + wv;
    float* par = smem + a.nw * 64 * LDW;

    // per-stage bias vectors and scalars go to LDS next to the weights (one dependent global read, once per block)
    for (int s = 0; s < a.nstage; ++s) {
        const ChStage& st = a.st[s];
        if (threadIdx.x < 128) {
            const float* src = threadIdx.x < 64 ? st.bias : st.bd;
            const int col = threadIdx.x & 63;
            par[s * CH_PAR + threadIdx.x] = (src && st.type != CH_SCORE) ? src[col] : 0.f;
        } else if (threadIdx.x < 132) {
            const int k = threadIdx.x - 128;
            const float* src = k == 0 ? st.sa : (k == 1 ? st.so : (k == 2 ? st.em_s : (st.type == CH_SCORE ? st.bias : nullptr)));
            par[s * CH_PAR + threadIdx.x] = src ? *src : (k == 3 ? 0.f : 1.f);
        }
    }
    // first tile's stage-0 input rows: issue the loads before the weights so the latencies overlap
    RTile pre;
    const bool have_pre = a.st[0].type == CH_GEMM && a.st[0].in_a != nullptr;
    rt_load(pre, a.st[0].in_a, tile0 * 16 + j, have_pre && tile0 * 16 + j < a.n, g);
    // stage the chain's weights: [nw][64][LDW]; every load of every matrix is issued before the first LDS write, so the
    // block pays ONE global round trip (up to 20 float4 per thread in flight)
    {
        constexpr int PER = 1024 / (NWAVES * 64);  // float4 per thread per matrix: 4 (256 threads) or 2 (512)
        float4 tmp[CH_MAX_W][PER];
#pragma unroll
        for (int wi = 0; wi < CH_MAX_W; ++wi)
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                tmp[wi][i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (wi < a.nw) tmp[wi][i] = *(const float4*)(a.w[wi] + (size_t)(i * NWAVES * 64 + threadIdx.x) * 4);
            }
#pragma unroll
        for (int wi = 0; wi < CH_MAX_W; ++wi)
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int idx = i * NWAVES * 64 + threadIdx.x;
                if (wi < a.nw) *(float4*)(smem + wi * 64 * LDW + (idx >> 4) * LDW + (idx & 15) * 4) = tmp[wi][i];
            }
    }
    __syncthreads();

    for (int tile = tile0; tile < ntile; tile += gridDim.x * NWAVES) {
        const int row = tile * 16 + j;
        const bool ok = row < a.n;
        RTile t0, t1;   // the two register tiles stages read from / write to (ChStage.ta / tb / tout)
        rt_zero(t0); rt_zero(t1);
#pragma unroll 1
        for (int s = 0; s < a.nstage; ++s) {
            const ChStage& st = a.st[s];
            const float* ps = par + s * CH_PAR;
            if (!TRANSB && st.type == CH_SCORE) {   // closes a forward chain: Dense(64->1), model.py:208
                float sum = 0.f;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const float4 w = *(const float4*)(st.w1 + 16 * m + 4 * g);
                    const float* tv = st.ta ? t1.v[m] : t0.v[m];
                    sum = fmaf(tv[0], w.x, fmaf(tv[1], w.y, fmaf(tv[2], w.z, fmaf(tv[3], w.w, sum))));
                }
                sum += __shfl_xor(sum, 16); sum += __shfl_xor(sum, 32);
                if (g == 0 && ok) st.out[row] = sum + ps[131];
                continue;
            }
            // operands of this stage: issue the global loads now, consume them after the MFMAs
            RTile r_inb, r_add, r_mask, r_em;
            rt_load(r_inb, st.in_b, row, ok && st.in_b != nullptr, g);
            rt_load(r_add, st.add, row, ok && st.add != nullptr, g);
            rt_load(r_mask, st.mask, row, ok && st.mask != nullptr, g);
            rt_load(r_em, st.em_a, row, ok && st.em_a != nullptr, g);
            float deg = 0.f;
            if (st.bd && ok) deg = (float)(st.seg_ptr[row + 1] - st.seg_ptr[row]);
            const float sa = ps[128], so = ps[129], es = ps[130];

            RTile o;
            if (!TRANSB && st.type == CH_EMBED1) {   // opens a forward chain
                // ((x + shift) * scale) @ W1 on the VALU (K <= 16); bias and ReLU come with the common epilogue
                rt_zero(o);
                for (int f = 0; f < st.nfeat; ++f) {
                    const float xv = ok ? (st.x_raw[(size_t)row * st.nfeat + f] + st.shift[f]) * st.scale[f] : 0.f;
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const float4 w = *(const float4*)(st.w1 + f * EMB + 16 * m + 4 * g);
                        o.v[m][0] = fmaf(xv, w.x, o.v[m][0]); o.v[m][1] = fmaf(xv, w.y, o.v[m][1]);
                        o.v[m][2] = fmaf(xv, w.z, o.v[m][2]); o.v[m][3] = fmaf(xv, w.w, o.v[m][3]);
                    }
                }
            } else {
                f32x4 acc[4];
#pragma unroll
                for (int mo = 0; mo < 4; ++mo) acc[mo] = (f32x4){0.f, 0.f, 0.f, 0.f};
                RTile in;
                if (st.in_a) {
                    if (s == 0 && tile == tile0) in = pre; else rt_load(in, st.in_a, row, ok, g);
                } else {
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int i = 0; i < 4; ++i) in.v[m][i] = st.ta ? t1.v[m][i] : t0.v[m][i];
                }
                rt_gemm<TRANSB>(in, sa, smem + st.wa * 64 * LDW, acc, lane);
                if (st.in_b) rt_gemm<TRANSB>(r_inb, 1.f, smem + st.wb * 64 * LDW, acc, lane);
#pragma unroll
                for (int mo = 0; mo < 4; ++mo)
#pragma unroll
                    for (int i = 0; i < 4; ++i) o.v[mo][i] = acc[mo][i] * so;
            }
            // epilogue in registers: +bias, +deg*bd, +add, relu, *(mask > 0)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float4 bias = *(const float4*)(ps + 16 * m + 4 * g);
                const float4 bd = *(const float4*)(ps + 64 + 16 * m + 4 * g);
                const float bv[4] = {bias.x, bias.y, bias.z, bias.w}, dv[4] = {bd.x, bd.y, bd.z, bd.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = o.v[m][i] + bv[i];
                    v = fmaf(deg, dv[i], v);
                    v += r_add.v[m][i];
                    if (st.relu) v = fmaxf(v, 0.f);
                    if (st.mask) v = r_mask.v[m][i] > 0.f ? v : 0.f;
                    o.v[m][i] = ok ? v : 0.f;
                }
            }
            if (st.out) rt_store(o, st.out, row, ok, g);
            if (st.em_out) {
                RTile e;
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int i = 0; i < 4; ++i) e.v[m][i] = es * o.v[m][i] * r_em.v[m][i];
                rt_store(e, st.em_out, row, ok, g);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (st.tout) t1.v[m][i] = o.v[m][i]; else t0.v[m][i] = o.v[m][i];
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradients: G[64,64] = sum_r (sx*X[r])^T D[r], db = sum_r D[r], dbd = sum_r deg_r D[r]     (B3/B4/B8/B11)
// Grouped launch: one job per (X, D) pair, one block per 256-row chunk of a job.  Wave w owns the 32x32 quadrant
// (w>>1, w&1) of G; rows are the MFMA k dimension.  Per-block partial slab [64*64 + 64 + 64] floats; summed in a
// fixed order by k_reduce (no atomics).
// ---------------------------------------------------------------------------------------------------------------
#define WG_ROWS 256
#define WG_SLAB (EMB * EMB + 2 * EMB)
#define WG_MAX_JOBS 24
struct WgJob { const float* x; const float* sx; const float* d; const int* seg_ptr; const float* d2; int n; int blk0; int slab0; };
struct WgArgs { int njobs; int nblocks; float* partial; WgJob job[WG_MAX_JOBS]; };

__global__ __launch_bounds__(256) void k_wgrad(WgArgs a) {
    __shared__ __attribute__((aligned(16))) float xs[64 * LDW];
    __shared__ __attribute__((aligned(16))) float ds[64 * LDW];
    __shared__ float red[4][2 * EMB];
    int ji = 0;
    while (ji + 1 < a.njobs && (int)blockIdx.x >= a.job[ji + 1].blk0) ++ji;
    const WgJob jb = a.job[ji];
    const int lb = blockIdx.x - jb.blk0;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int qi = wv >> 1, qj = wv & 1, i32 = lane & 31, h = lane >> 5;
    const float sx = jb.sx ? *jb.sx : 1.f;
    const int col = threadIdx.x & 63, part = threadIdx.x >> 6;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float cs = 0.f, cds = 0.f;
    const int rbeg = lb * WG_ROWS, rend = min(jb.n, rbeg + WG_ROWS);
    for (int row0 = rbeg; row0 < rend; row0 += 64) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // 64 rows x 16 float4 = 1024 float4 per matrix
            const int idx = i * 256 + threadIdx.x;
            const int r = idx >> 4, c = (idx & 15) * 4;
            float4 xv = make_float4(0.f, 0.f, 0.f, 0.f), dv = xv;
            if (row0 + r < rend) {
                xv = *(const float4*)(jb.x + (size_t)(row0 + r) * EMB + c);
                dv = *(const float4*)(jb.d + (size_t)(row0 + r) * EMB + c);
                xv.x *= sx; xv.y *= sx; xv.z *= sx; xv.w *= sx;
            }
            *(float4*)(xs + r * LDW + c) = xv;
            *(float4*)(ds + r * LDW + c) = dv;
        }
        __syncthreads();
#pragma unroll 8
        for (int s = 0; s < 32; ++s) {
            const int r = h * 32 + s;
            acc = mfma32(xs[r * LDW + qi * 32 + i32], ds[r * LDW + qj * 32 + i32], acc);
        }
        // column sums of D (bias grads), 16 rows per thread
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
            const int r = part * 16 + s;
            const float dv = ds[r * LDW + col];
            cs += dv;
            const int gr = row0 + r;
            if (jb.seg_ptr) {        // second column sum: degree-weighted (gradient of the hoisted b_f)
                const float deg = gr < rend ? (float)(jb.seg_ptr[gr + 1] - jb.seg_ptr[gr]) : 0.f;
                cds = fmaf(deg, dv, cds);
            } else if (jb.d2) {      // ... or the plain column sum of a second matrix (Q -> d w_edge)
                cds += gr < rend ? jb.d2[(size_t)gr * EMB + col] : 0.f;
            }
        }
    }
    float* slab = a.partial + (size_t)(jb.slab0 + lb) * WG_SLAB;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        slab[(qi * 32 + row) * EMB + qj * 32 + i32] = acc[i];
    }
    red[part][col] = cs; red[part][EMB + col] = cds;
    __syncthreads();
    if (threadIdx.x < 2 * EMB)
        slab[EMB * EMB + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// Fixed-order sum of partial slabs into the flat gradient buffer.  One block per (job, 64-float chunk).
#define RD_MAX_JOBS 96
struct RdJob { const float* src; float* dst; int nparts; int stride; int len; int blk0; };
struct RdArgs { int njobs; RdJob job[RD_MAX_JOBS]; };

__global__ __launch_bounds__(256) void k_reduce(RdArgs a) {
    __shared__ float red[4][EMB];
    int ji = 0;
    while (ji + 1 < a.njobs && (int)blockIdx.x >= a.job[ji + 1].blk0) ++ji;
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
    if (part == 0 && e < jb.len) jb.dst[e] = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
}

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

// OR over the 16 lanes of a DPP row (= the 16-lane group that serves one edge): rotate-and-OR by 1, 2, 4, 8
__device__ __forceinline__ unsigned row_or16(unsigned x) {
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x121, 0xf, 0xf, false);  // row_ror:1
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x122, 0xf, 0xf, false);  // row_ror:2
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x124, 0xf, 0xf, false);  // row_ror:4
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xf, 0xf, false);  // row_ror:8
    return x;
}

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

// Forward edge pass.  relu(s1*J) = s1*max(J,0) for s1 >= 0 and s1*min(J,0) for s1 < 0, so the scale is applied once per
// receiver.  J_e = (c_e*w + P_oth[oth_e]) + P_own[r].
// SAVE also emits what the backward pass needs: per edge one 64-bit word (nibble c = the ReLU bits of channels
// 4c..4c+3, assembled across the edge's 16 lanes with DPP row rotations) and per receiver/channel the number N of
// active edges.  Because dS[r] is
// constant over a receiver's segment, dP_recv[r] = s1*dS[r]*N[r]: the receiver-ordered half of the backward pass is an
// element-wise epilogue (of the chain that produces dS), not an edge pass.
template <int SLOTS, bool SAVE, bool NEG>
__device__ __forceinline__ void edge_fwd_impl(const EdgeArgs& a, const float s1) {
    constexpr int G = 16 * SLOTS, RPW = 64 / G;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gl = lane % G, gbase = lane - gl, slot = gl >> 4, cl = gl & 15, ch = cl * 4;
    const float4 w = *(const float4*)(a.w_edge + ch);
    const float esh = *a.e_shift, esc = *a.e_scale;

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
                unsigned npk = 0;   // packed byte counters: at most 16 edges per slot per chunk, no overflow
                for (int i0 = 0; i0 < cnt; i0 += 4 * SLOTS) {
                    int oi[4]; float ci[4]; bool ok[4]; float4 p[4]; unsigned nib[4] = {0u, 0u, 0u, 0u};
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
                        if (ok[u]) {
                            float h0 = fmaf(ci[u], w.x, p[u].x) + pown.x, h1 = fmaf(ci[u], w.y, p[u].y) + pown.y;
                            float h2 = fmaf(ci[u], w.z, p[u].z) + pown.z, h3 = fmaf(ci[u], w.w, p[u].w) + pown.w;
                            h0 = NEG ? fminf(h0, 0.f) : fmaxf(h0, 0.f); h1 = NEG ? fminf(h1, 0.f) : fmaxf(h1, 0.f);
                            h2 = NEG ? fminf(h2, 0.f) : fmaxf(h2, 0.f); h3 = NEG ? fminf(h3, 0.f) : fmaxf(h3, 0.f);
                            acc.x += h0; acc.y += h1; acc.z += h2; acc.w += h3;
                            if (SAVE) {
                                // active bit: h > 0 read off the float's bit pattern (+0 -> 0, anything positive -> 1)
                                unsigned b0, b1, b2, b3;
                                if (NEG) { b0 = h0 < 0.f; b1 = h1 < 0.f; b2 = h2 < 0.f; b3 = h3 < 0.f; }
                                else {
                                    b0 = (__float_as_uint(h0) + 0x7fffffffu) >> 31; b1 = (__float_as_uint(h1) + 0x7fffffffu) >> 31;
                                    b2 = (__float_as_uint(h2) + 0x7fffffffu) >> 31; b3 = (__float_as_uint(h3) + 0x7fffffffu) >> 31;
                                }
                                nib[u] = b0 | (b1 << 1) | (b2 << 2) | (b3 << 3);
                                npk += (nib[u] * 0x00204081u) & 0x01010101u;   // four 8-bit counters, one per channel
                            }
                        }
                    }
                    if (SAVE) {
                        // each lane drops its nibble at bits 4*(c&7) of the low (c < 8) or high word; OR over the row
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const unsigned sh = nib[u] << (4 * (cl & 7));
                            const unsigned lo = row_or16(cl < 8 ? sh : 0u), hi = row_or16(cl < 8 ? 0u : sh);
                            if (ok[u] && cl == 0)
                                a.mask[base + i0 + u * SLOTS + slot] = (unsigned long long)lo | ((unsigned long long)hi << 32);
                        }
                    }
                }
                if (SAVE) { n0 += npk & 255u; n1 += (npk >> 8) & 255u; n2 += (npk >> 16) & 255u; n3 += npk >> 24; }
            }
            acc = slot_reduce<SLOTS>(acc);
            if (slot == 0) *(float4*)(a.out + (size_t)r * EMB + ch) = make_float4(s1 * acc.x, s1 * acc.y, s1 * acc.z, s1 * acc.w);
            if (SAVE) {
                const float4 nacc = slot_reduce<SLOTS>(make_float4((float)n0, (float)n1, (float)n2, (float)n3));
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
// the chain that produces dS; this kernel serves the per-op entry point.)
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
                    int oi[4]; float ci[4]; bool ok[4]; float4 d[4]; unsigned mb[4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int i = i0 + v * SLOTS + slot;
                        ok[v] = i < cnt;
                        const int src = gbase + (ok[v] ? i : 0);
                        oi[v] = __shfl(o, src); ci[v] = __shfl(c, src);
                        const unsigned wlo = __shfl(mlo, src), whi = __shfl(mhi, src);  // both by every lane: a shuffle
                        const unsigned word = cl < 8 ? wlo : whi;                        // must not sit under a lane mask
                        mb[v] = ok[v] ? (word >> (4 * (cl & 7))) & 15u : 0u;
                    }
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        d[v] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (ok[v]) d[v] = *(const float4*)(a.d_s + (size_t)oi[v] * EMB + ch);
                    }
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const float t0 = (mb[v] & 1u) ? d[v].x : 0.f, t1 = (mb[v] & 2u) ? d[v].y : 0.f;
                        const float t2 = (mb[v] & 4u) ? d[v].z : 0.f, t3 = (mb[v] & 8u) ? d[v].w : 0.f;
                        acc.x += t0; acc.y += t1; acc.z += t2; acc.w += t3;
                        dw.x = fmaf(ci[v], t0, dw.x); dw.y = fmaf(ci[v], t1, dw.y); dw.z = fmaf(ci[v], t2, dw.z); dw.w = fmaf(ci[v], t3, dw.w);
                    }
                }
            }
            acc = slot_reduce<SLOTS>(acc); dw = slot_reduce<SLOTS>(dw);
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

// ---------------------------------------------------------------------------------------------------------------
// Readout tail Dense(64->1) (model.py:208, 299-300) and the loss head (model_trainer.py:271).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_score(const float* __restrict__ o1, const float* __restrict__ w2,
                                               const float* __restrict__ b2, float* __restrict__ score, int n) {
    const int ch = (threadIdx.x & 15) * 4;
    const float4 w = *(const float4*)(w2 + ch);
    const float b = *b2;
    for (int r = blockIdx.x * 16 + (threadIdx.x >> 4); r < n; r += gridDim.x * 16) {
        const float4 v = *(const float4*)(o1 + (size_t)r * EMB + ch);
        float s = fmaf(v.x, w.x, fmaf(v.y, w.y, fmaf(v.z, w.z, v.w * w.w)));
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
        if ((threadIdx.x & 15) == 0) score[r] = s + b;
    }
}

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
                                                   float* __restrict__ partial, int n) {
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
}

// Keras-form Adam (model_trainer.py:131,273): eps outside the bias-corrected sqrt.
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, int n, float lr_t, float b1, float b2, float eps,
                                              const float* __restrict__ gscale) {
    const float gs = gscale ? *gscale : 1.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float gi = g[i] * gs;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] -= lr_t * mi / (sqrtf(vi) + eps);
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
// graph plan kernels
// ---------------------------------------------------------------------------------------------------------------
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

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)
#define LAUNCHCHK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static const int LIN_SMEM = (2 * 64 * LDW + 4 * 32 * LDW) * (int)sizeof(float);  // 69,632 B
static const int MAX_GRID = 2048;

static int launch_linear(bool transb, const LinArgs& a, hipStream_t st) {
    if (a.n <= 0) return 0;
    static bool attr_set = false;
    if (!attr_set) {  // 68 KB of dynamic LDS per block (gfx950 has 160 KB per CU)
        HIPCHK(hipFuncSetAttribute((const void*)k_linear<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LIN_SMEM));
        HIPCHK(hipFuncSetAttribute((const void*)k_linear<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LIN_SMEM));
        attr_set = true;
    }
    const int grid = std::min(cdiv(a.n, 128), MAX_GRID);
    if (transb) hipLaunchKernelGGL(k_linear<true>, dim3(grid), dim3(256), LIN_SMEM, st, a);
    else hipLaunchKernelGGL(k_linear<false>, dim3(grid), dim3(256), LIN_SMEM, st, a);
    LAUNCHCHK();
    return 0;
}

static LinArgs lin_fwd(const float* xa, const float* wa, const float* bias, int relu, float* y, int n) {
    LinArgs a; memset(&a, 0, sizeof(a));
    a.xa = xa; a.wa = wa; a.bias = bias; a.relu = relu; a.y = y; a.n = n;
    return a;
}
static LinArgs lin_bwd(const float* dy, const float* ymask, const float* wa, float* dx, int beta, int n) {
    LinArgs a; memset(&a, 0, sizeof(a));
    a.xa = dy; a.ymask = ymask; a.write_back = ymask != nullptr; a.wa = wa; a.y = dx; a.beta_y = beta; a.n = n;
    return a;
}

static inline int edge_slots(int n_own, int n_edges) {
    const double avg = (double)n_edges / (double)std::max(n_own, 1);
    return avg >= 12.0 ? 4 : (avg >= 5.0 ? 2 : 1);
}
// forward (owner = receiver); `save` also emits the ReLU nibbles and the N rows for the backward pass
static int launch_edge_fwd(const EdgeArgs& a, int n_edges, bool save, hipStream_t st) {
    if (a.n_recv <= 0) return 0;
    if (save && (!a.cnt_rows || (n_edges > 0 && !a.mask))) return GCNN_E_BADARG;
    const int slots = edge_slots(a.n_recv, n_edges);
    const int grid = std::min(cdiv(cdiv(a.n_recv, 4 / slots), 4), MAX_GRID);
#define EDGE_LAUNCH(S, V) hipLaunchKernelGGL((k_edge_fwd<S, V>), dim3(grid), dim3(256), 0, st, a)
    if (save) { if (slots == 4) EDGE_LAUNCH(4, true); else if (slots == 2) EDGE_LAUNCH(2, true); else EDGE_LAUNCH(1, true); }
    else { if (slots == 4) EDGE_LAUNCH(4, false); else if (slots == 2) EDGE_LAUNCH(2, false); else EDGE_LAUNCH(1, false); }
#undef EDGE_LAUNCH
    LAUNCHCHK();
    return 0;
}
// backward, sender-ordered (owner = sender)
static int launch_edge_bwd_send(const EdgeArgs& a, int n_edges, hipStream_t st) {
    if (a.n_recv <= 0) return 0;
    const int slots = edge_slots(a.n_recv, n_edges);
    const int grid = std::min(cdiv(cdiv(a.n_recv, 4 / slots), 4), MAX_GRID);
    if (slots == 4) hipLaunchKernelGGL(k_edge_bwd_send<4>, dim3(grid), dim3(256), 0, st, a);
    else if (slots == 2) hipLaunchKernelGGL(k_edge_bwd_send<2>, dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_edge_bwd_send<1>, dim3(grid), dim3(256), 0, st, a);
    LAUNCHCHK();
    return 0;
}

// ---- workspace carving ------------------------------------------------------------------------------------------
struct Acts {
    float *E1c, *Xc, *PL1, *S1, *A1, *Z1c, *Xc2, *PL2;          // C rows
    float *E1v, *Xv, *PR1, *PR2, *S2, *A2, *Z1v, *Xv2, *PR3;    // V rows
    float *E1k, *Xk, *PL3, *S3, *A3, *Z1k, *Xk2, *O1;           // K rows
};
struct Work {
    Acts a, g;            // activations and their gradients
    float* partial;       // weight-gradient slabs
    float* q[3];          // per-sender shares of d w_edge, one [n_send,64] matrix per convolution
    unsigned long long* mask[3];  // ReLU bits of the three edge passes, 8 bytes per edge, receiver order
    float* nrow[3];          // per receiver and channel: number of active edges
    float* emb_partial[3];
    float* score_partial; int score_nblk;
    double* stats; int* stat_ids;   // pretraining: per-block partial sums; explicit left ids of an edge set
    int emb_nblk[3];
    size_t total;
};
static inline size_t al4(size_t x) { return (x + 3) & ~(size_t)3; }
#define EMB1_ROWS 128

static size_t wg_slabs(const gcnn_dims* d) {  // total number of wgrad slabs over all 22 jobs
    const int C = d->n_cons, V = d->n_vars, K = d->n_cuts;
    const int bc = cdiv(C, WG_ROWS), bv = cdiv(V, WG_ROWS), bk = cdiv(K, WG_ROWS);
    // jobs per row set (see gcnn_backward): cons 7, var 8, cut 7; 8 each leaves slack
    return (size_t)bc * 8 + (size_t)bv * 8 + (size_t)bk * 8;
}

static void carve(const gcnn_dims* d, float* base, Work* w) {
    const size_t C = d->n_cons, V = d->n_vars, K = d->n_cuts;
    size_t off = 0;
    auto take = [&](size_t n) { float* p = base ? base + off : nullptr; off += al4(n); return p; };
    for (int pass = 0; pass < 2; ++pass) {
        Acts* t = pass ? &w->g : &w->a;
        float** pc[] = {&t->E1c, &t->Xc, &t->PL1, &t->S1, &t->A1, &t->Z1c, &t->Xc2, &t->PL2};
        float** pv[] = {&t->E1v, &t->Xv, &t->PR1, &t->PR2, &t->S2, &t->A2, &t->Z1v, &t->Xv2, &t->PR3};
        float** pk[] = {&t->E1k, &t->Xk, &t->PL3, &t->S3, &t->A3, &t->Z1k, &t->Xk2, &t->O1};
        for (auto p : pc) *p = take(C * EMB);
        for (auto p : pv) *p = take(V * EMB);
        for (auto p : pk) *p = take(K * EMB);
    }
    w->partial = take(wg_slabs(d) * WG_SLAB);
    const size_t nrecv[3] = {C, V, K};
    const size_t nsend[3] = {V, C, V};
    for (int i = 0; i < 3; ++i) { w->q[i] = take(nsend[i] * EMB); w->nrow[i] = take(nrecv[i] * EMB); }
    const size_t nedge[3] = {(size_t)d->n_cons_edges, (size_t)d->n_cons_edges, (size_t)d->n_cut_edges};
    for (int i = 0; i < 3; ++i) w->mask[i] = (unsigned long long*)take(2 * nedge[i]);
    const int nemb[3] = {d->n_cons, d->n_vars, d->n_cuts};
    const int femb[3] = {4, 14, 6};
    for (int i = 0; i < 3; ++i) {
        w->emb_nblk[i] = cdiv(nemb[i], EMB1_ROWS);
        w->emb_partial[i] = take((size_t)w->emb_nblk[i] * (femb[i] + 1) * EMB);
    }
    w->stats = (double*)take(2 * (size_t)(ST_MAX_BLOCKS * ST_MAX_UNITS + 2 * ST_MAX_UNITS));
    w->stat_ids = (int*)take((size_t)std::max(d->n_cons_edges, d->n_cut_edges));
    w->score_nblk = cdiv(d->n_cuts, SB_ROWS);
    w->score_partial = take((size_t)w->score_nblk * 2 * EMB);
    w->total = off;
}

extern "C" {

int gcnn_abi_version(void) { return 1; }
int gcnn_param_count(void) { return GCNN_N_PARAMS; }
int gcnn_param_total_floats(void) { layout_init(); return g_ptotal; }
int gcnn_param_info(int index, int* offset, int* rows, int* cols, int* trainable) {
    layout_init();
    if (index < 0 || index >= GCNN_N_PARAMS) return GCNN_E_BADARG;
    if (offset) *offset = g_pinfo[index].off;
    if (rows) *rows = g_pinfo[index].rows;
    if (cols) *cols = g_pinfo[index].cols;
    if (trainable) *trainable = g_pinfo[index].trainable;
    return 0;
}

size_t gcnn_workspace_floats(const gcnn_dims* dims) {
    if (!dims) return 0;
    Work w; carve(dims, nullptr, &w);
    return w.total;
}

// ---- graph plan -------------------------------------------------------------------------------------------------
static size_t sort_temp_bytes(int n) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const int*)nullptr, (int*)nullptr, (const int*)nullptr,
                                       (int*)nullptr, n > 0 ? n : 1);
    return (bytes + 255) & ~(size_t)255;
}
size_t gcnn_graph_temp_bytes(int32_t n_edges) {
    const size_t e = ((size_t)(n_edges > 0 ? n_edges : 1) * sizeof(int) + 255) & ~(size_t)255;
    return sort_temp_bytes(n_edges) + 5 * e;  // cub temp + iota + sorted keys + two permutations + one inverse
}

int gcnn_graph_build(const int32_t* edge_inds, const float* edge_feats, int32_t n_edges, int32_t n_left, int32_t n_var,
                     int32_t* l_ptr, int32_t* l_oth, float* l_coef, int32_t* v_ptr, int32_t* v_oth, float* v_coef,
                     int32_t* l2v, int32_t* v2l, int32_t* l_perm, void* temp, size_t temp_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (n_edges < 0 || n_left < 0 || n_var < 0 || !l_ptr || !v_ptr) return GCNN_E_BADARG;
    if (temp_bytes < gcnn_graph_temp_bytes(n_edges)) return GCNN_E_WORKSPACE;
    if (n_edges == 0) {
        HIPCHK(hipMemsetAsync(l_ptr, 0, (size_t)(n_left + 1) * sizeof(int), st));
        HIPCHK(hipMemsetAsync(v_ptr, 0, (size_t)(n_var + 1) * sizeof(int), st));
        return 0;
    }
    if (!edge_inds || !edge_feats || !l_oth || !l_coef || !v_oth || !v_coef || !temp) return GCNN_E_BADARG;
    const size_t e = ((size_t)n_edges * sizeof(int) + 255) & ~(size_t)255;
    size_t cub_bytes = sort_temp_bytes(n_edges);
    char* t = (char*)temp;
    void* cub_tmp = t;
    int* iota = (int*)(t + cub_bytes);
    int* keys = (int*)(t + cub_bytes + e);
    int* perm[2] = {(int*)(t + cub_bytes + 2 * e), (int*)(t + cub_bytes + 3 * e)};
    int* inv = (int*)(t + cub_bytes + 4 * e);
    const int grid = std::min(cdiv(n_edges + 1, 256), 4096);
    const int* left = edge_inds;
    const int* var = edge_inds + n_edges;
    hipLaunchKernelGGL(k_iota, dim3(grid), dim3(256), 0, st, iota, n_edges);
    LAUNCHCHK();
    for (int side = 0; side < 2; ++side) {
        const int* key_in = side == 0 ? left : var;
        const int nseg = side == 0 ? n_left : n_var;
        int bits = 1;
        while ((1ll << bits) < (long long)nseg + 1 && bits < 31) ++bits;
        HIPCHK(hipcub::DeviceRadixSort::SortPairs(cub_tmp, cub_bytes, key_in, keys, (const int*)iota, perm[side], n_edges,
                                                  0, bits, st));
        hipLaunchKernelGGL(k_seg_offsets, dim3(grid), dim3(256), 0, st, keys, n_edges, nseg, side == 0 ? l_ptr : v_ptr);
        LAUNCHCHK();
        hipLaunchKernelGGL(k_gather_edges, dim3(grid), dim3(256), 0, st, perm[side], side == 0 ? var : left, edge_feats,
                           n_edges, side == 0 ? l_oth : v_oth, side == 0 ? l_coef : v_coef);
        LAUNCHCHK();
    }
    if (l_perm) HIPCHK(hipMemcpyAsync(l_perm, perm[0], (size_t)n_edges * sizeof(int), hipMemcpyDeviceToDevice, st));
    if (v2l) {  // by-variable position -> by-left position of the same edge
        hipLaunchKernelGGL(k_invert_perm, dim3(grid), dim3(256), 0, st, perm[0], n_edges, inv); LAUNCHCHK();
        hipLaunchKernelGGL(k_compose_perm, dim3(grid), dim3(256), 0, st, perm[1], inv, n_edges, v2l); LAUNCHCHK();
    }
    if (l2v) {
        hipLaunchKernelGGL(k_invert_perm, dim3(grid), dim3(256), 0, st, perm[1], n_edges, inv); LAUNCHCHK();
        hipLaunchKernelGGL(k_compose_perm, dim3(grid), dim3(256), 0, st, perm[0], inv, n_edges, l2v); LAUNCHCHK();
    }
    return 0;
}

// ---- standalone scatter-sum pass --------------------------------------------------------------------------------
int gcnn_seg_sum_f32(const float* msg, const int32_t* seg_ptr, const int32_t* perm, int32_t n_recv, float* out,
                     void* stream) {
    if (n_recv < 0 || (n_recv > 0 && (!seg_ptr || !out))) return GCNN_E_BADARG;  // msg may be NULL when E == 0
    if (n_recv == 0) return 0;
    const int grid = std::min(cdiv(n_recv, 4), 8192);
    if (perm) hipLaunchKernelGGL((k_seg_sum<4, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, msg, seg_ptr, perm, n_recv, out);
    else hipLaunchKernelGGL((k_seg_sum<4, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, msg, seg_ptr, perm, n_recv, out);
    LAUNCHCHK();
    return 0;
}
int gcnn_seg_bcast_f32(const float* d_out, const int32_t* seg_ptr, const int32_t* perm, int32_t n_recv, float* d_msg,
                       void* stream) {
    if (n_recv < 0 || (n_recv > 0 && (!d_out || !seg_ptr))) return GCNN_E_BADARG;  // d_msg may be NULL when E == 0
    if (n_recv == 0) return 0;
    const int grid = std::min(cdiv(n_recv, 4), 8192);
    if (perm) hipLaunchKernelGGL((k_seg_bcast<true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, d_out, seg_ptr, perm, n_recv, d_msg);
    else hipLaunchKernelGGL((k_seg_bcast<false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, d_out, seg_ptr, perm, n_recv, d_msg);
    LAUNCHCHK();
    return 0;
}

// ---- per-op entry points (also the unit-test surface) -------------------------------------------------------------
int gcnn_linear_fwd(const float* xa, const float* sa, const float* wa, const float* xb, const float* wb, const float* bias,
                    const float* bd, const int32_t* seg_ptr, int32_t relu, float* y, int32_t n, void* stream) {
    if (n < 0 || (n > 0 && (!xa || !wa || !y)) || (xb && !wb) || (bd && !seg_ptr)) return GCNN_E_BADARG;
    LinArgs a = lin_fwd(xa, wa, bias, relu, y, n);
    a.sa = sa; a.xb = xb; a.wb = xb ? wb : nullptr; a.bd = bd; a.seg_ptr = seg_ptr;
    return launch_linear(false, a, (hipStream_t)stream);
}
int gcnn_linear_bwd(float* dy, const float* ymask, const float* wa, const float* so, float* dx, int32_t beta,
                    const float* wb, float* dx2, int32_t beta2, int32_t n, void* stream) {
    if (n < 0 || (n > 0 && (!dy || !wa || !dx)) || (dx2 && !wb)) return GCNN_E_BADARG;
    LinArgs a = lin_bwd(dy, ymask, wa, dx, beta, n);
    a.so = so; a.wb = dx2 ? wb : nullptr; a.y2 = dx2; a.beta_y2 = beta2;
    return launch_linear(true, a, (hipStream_t)stream);
}
int gcnn_conv_edge_fwd(const int32_t* seg_ptr, const int32_t* oth, const float* coef, int32_t n_recv, int32_t n_edges,
                       const float* p_recv, const float* p_oth, const float* w_edge, const float* e_shift,
                       const float* e_scale, const float* s1, float* s_out, uint64_t* mask_out, float* n_rows, void* stream) {
    if (n_recv < 0 || n_edges < 0) return GCNN_E_BADARG;
    if (n_recv > 0 && (!seg_ptr || !p_recv || !w_edge || !e_shift || !e_scale || !s1 || !s_out)) return GCNN_E_BADARG;
    if (n_edges > 0 && (!oth || !coef || !p_oth)) return GCNN_E_BADARG;
    const bool save = mask_out || n_rows;
    EdgeArgs e; memset(&e, 0, sizeof(e));
    e.seg_ptr = seg_ptr; e.oth = oth; e.coef = coef; e.p_recv = p_recv; e.p_oth = p_oth; e.w_edge = w_edge;
    e.e_shift = e_shift; e.e_scale = e_scale; e.s1 = s1; e.out = s_out; e.mask = (unsigned long long*)mask_out; e.cnt_rows = n_rows; e.n_recv = n_recv;
    return launch_edge_fwd(e, n_edges, save, (hipStream_t)stream);
}
int gcnn_conv_edge_bwd_recv(const float* d_s, const float* n_rows, const float* s1, int32_t n_recv, float* d_p_recv,
                            void* stream) {
    if (n_recv < 0 || (n_recv > 0 && (!d_s || !n_rows || !s1 || !d_p_recv))) return GCNN_E_BADARG;
    if (n_recv == 0) return 0;
    hipLaunchKernelGGL(k_edge_bwd_recv, dim3(std::min(cdiv(n_recv * 16, 256), MAX_GRID)), dim3(256), 0, (hipStream_t)stream,
                       d_s, n_rows, s1, d_p_recv, n_recv * 16);
    LAUNCHCHK();
    return 0;
}
int gcnn_conv_edge_bwd_send(const int32_t* seg_ptr, const int32_t* oth, const float* coef, const int32_t* xpos,
                            const uint64_t* mask, int32_t n_send, int32_t n_edges, const float* e_shift,
                            const float* e_scale, const float* s1, const float* d_s, float* d_p_send, float* dw_rows,
                            void* stream) {
    if (n_send < 0 || n_edges < 0) return GCNN_E_BADARG;
    if (n_send > 0 && (!seg_ptr || !s1 || !e_shift || !e_scale || !d_p_send || !dw_rows)) return GCNN_E_BADARG;
    if (n_edges > 0 && (!oth || !coef || !xpos || !mask || !d_s)) return GCNN_E_BADARG;
    EdgeArgs e; memset(&e, 0, sizeof(e));
    e.seg_ptr = seg_ptr; e.oth = oth; e.coef = coef; e.xpos = xpos; e.mask = (unsigned long long*)mask; e.s1 = s1;
    e.e_shift = e_shift; e.e_scale = e_scale; e.d_s = d_s; e.out = d_p_send; e.dw_rows = dw_rows; e.n_recv = n_send;
    return launch_edge_bwd_send(e, n_edges, (hipStream_t)stream);
}

}  // extern "C"

// ---- fused row chains: host-side builders --------------------------------------------------------------------------
struct Chain {
    ChArgs a;
    Chain(int n) { memset(&a, 0, sizeof(a)); a.n = n; }
    int weight(const float* w) {  // returns the LDS slot of a 64x64 weight, staging each distinct matrix once
        for (int i = 0; i < a.nw; ++i)
            if (a.w[i] == w) return i;
        a.w[a.nw] = w;
        return a.nw++;
    }
    ChStage& gemm(const float* in_a, int ta, const float* w, int transb, float* out, int tout) {
        ChStage& s = a.st[a.nstage++];
        s.type = CH_GEMM; s.in_a = in_a; s.ta = ta; s.wa = weight(w); s.transb = transb; s.out = out; s.tout = tout;
        return s;
    }
    ChStage& embed1(const float* x, int f, const float* p, int pb, float* out) {
        ChStage& s = a.st[a.nstage++];
        s.type = CH_EMBED1; s.x_raw = x; s.nfeat = f; s.shift = p + poff(pb + E_SHIFT); s.scale = p + poff(pb + E_SCALE);
        s.w1 = p + poff(pb + E_W1); s.bias = p + poff(pb + E_B1); s.relu = 1; s.out = out; s.tout = 0;
        return s;
    }
    ChStage& score(int ta, const float* w, const float* b, float* out) {
        ChStage& s = a.st[a.nstage++];
        s.type = CH_SCORE; s.ta = ta; s.w1 = w; s.bias = b; s.out = out;
        return s;
    }
};

static int launch_chain(const Chain& c, hipStream_t st) {
    const ChArgs& a = c.a;
    if (a.n <= 0 || a.nstage == 0) return 0;
    if (a.nstage > CH_MAX_STAGES || a.nw > CH_MAX_W) return GCNN_E_BADARG;
    // every stage of a chain multiplies in the same direction; CH_EMBED1 / CH_SCORE only open / close forward chains
    const bool transb = a.st[a.nstage - 1].type == CH_GEMM ? a.st[a.nstage - 1].transb != 0 : false;
    for (int i = 0; i < a.nstage; ++i) {
        if (a.st[i].type == CH_GEMM && (a.st[i].transb != 0) != transb) return GCNN_E_BADARG;
        if (a.st[i].type == CH_EMBED1 && (i != 0 || transb)) return GCNN_E_BADARG;
        if (a.st[i].type == CH_SCORE && (i != a.nstage - 1 || transb)) return GCNN_E_BADARG;
    }
    const int ntile = cdiv(a.n, 16);
    // one block per CU (the staged weights fill most of the LDS); 8 waves per block once there is more than one tile
    // per wave so two waves share each SIMD's MFMA pipe and hide each other's loads
    const bool big = ntile > 1024;
    const int nwaves = big ? 8 : 4;
    const int smem = (a.nw * 64 * LDW + CH_MAX_STAGES * CH_PAR) * (int)sizeof(float);  // weights + per-stage parameters
    const dim3 grid(std::min(cdiv(ntile, nwaves), 256)), block(nwaves * 64);
#define CHAIN_CASE(NW, TB)                                                                                              \
    do {                                                                                                                \
        static bool attr_set = false;                                                                                   \
        if (!attr_set) { HIPCHK(hipFuncSetAttribute((const void*)k_chain<NW, TB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_set = true; } \
        hipLaunchKernelGGL((k_chain<NW, TB>), grid, block, smem, st, a);                                                \
    } while (0)
    if (big) { if (transb) CHAIN_CASE(8, true); else CHAIN_CASE(8, false); }
    else { if (transb) CHAIN_CASE(4, true); else CHAIN_CASE(4, false); }
#undef CHAIN_CASE
    LAUNCHCHK();
    return 0;
}

// ---- side streams ---------------------------------------------------------------------------------------------
// The step is a chain of latency-bound launches that leave most of the chip idle, so independent work runs beside it on
// two auxiliary HIP streams, forked from / joined back into the caller's stream with events (the pattern stream capture
// turns into parallel graph branches): the three embedding chains in the forward pass; in the backward pass the weight
// gradient jobs of each convolution (as soon as its edge pass has finished), the cut-/constraint-row tail chains and the
// final reduction.  Results do not depend on the interleaving (no atomics).  GCNN_STREAMS=0 runs everything in order.
struct Side { hipStream_t s[2]; hipEvent_t ev[16]; int next; int state; };  // state: 0 = not tried, 1 = on, -1 = off
static Side g_side = {{nullptr, nullptr}, {}, 0, 0};
static bool side_on() {
    if (g_side.state == 0) {
        const char* env = getenv("GCNN_STREAMS");
        g_side.state = -1;
        if (!(env && env[0] == '0')) {
            bool ok = hipStreamCreateWithFlags(&g_side.s[0], hipStreamNonBlocking) == hipSuccess &&
                      hipStreamCreateWithFlags(&g_side.s[1], hipStreamNonBlocking) == hipSuccess;
            for (int i = 0; ok && i < 16; ++i) ok = hipEventCreateWithFlags(&g_side.ev[i], hipEventDisableTiming) == hipSuccess;
            if (ok) g_side.state = 1;
        }
    }
    return g_side.state == 1;
}
// An event record costs the recording stream ~7 us on this platform, so fork points record ONCE and let every
// dependent stream wait on the same event.
static int ev_record(hipStream_t on, hipEvent_t* out) {
    hipEvent_t e = g_side.ev[g_side.next];
    g_side.next = (g_side.next + 1) & 15;
    HIPCHK(hipEventRecord(e, on));
    *out = e;
    return 0;
}
static int ev_wait(hipStream_t st, hipEvent_t e, hipStream_t recorded_on) {
    if (st == recorded_on) return 0;
    HIPCHK(hipStreamWaitEvent(st, e, 0));
    return 0;
}

// ---- forward ----------------------------------------------------------------------------------------------------
struct ConvIO {           // one PartialGraphConvolution instance (model.py:201-203, 294-296)
    int pbase;            // first parameter index of the block
    const float* xl; const float* xv; int nl, nv, ne;
    bool recv_left;
    const gcnn_graph* g; int pedge;  // edge PreNorm parameter index (shift; scale = +1)
    float *PL, *PR, *S, *A, *Z1, *OUT;
    float *gPL, *gPR, *gS, *gA, *gZ1, *gOUT, *gXL, *gXV, *Q;
    unsigned long long* mask; float* N;
};

static EdgeArgs conv_edge_args(const float* p, const ConvIO& c, bool by_left) {
    EdgeArgs e; memset(&e, 0, sizeof(e));
    e.seg_ptr = by_left ? c.g->l_ptr : c.g->v_ptr; e.oth = by_left ? c.g->l_oth : c.g->v_oth;
    e.coef = by_left ? c.g->l_coef : c.g->v_coef;
    e.p_recv = by_left ? c.PL : c.PR; e.p_oth = by_left ? c.PR : c.PL;   // segment owner's table / gathered table
    e.w_edge = p + poff(c.pbase + C_WE); e.e_shift = p + poff(c.pedge); e.e_scale = p + poff(c.pedge + 1);
    e.s1 = p + poff(c.pbase + C_S1); e.n_recv = by_left ? c.nl : c.nv;
    return e;
}

// edge pass + the receiver-side update chain S -> A -> Z1 -> X' (model.py:498-508, 568-573); `tail` appends the stages
// that consume X' (the next convolution's projection or the readout) to the same launch
template <class Tail>
static int conv_forward(const float* p, const ConvIO& c, bool save, hipStream_t st, Tail tail) {
    int rc;
    const int nr = c.recv_left ? c.nl : c.nv;
    const float* xrecv = c.recv_left ? c.xl : c.xv;
    EdgeArgs e = conv_edge_args(p, c, c.recv_left);
    e.out = c.S; e.mask = c.mask; e.cnt_rows = c.N;
    if ((rc = launch_edge_fwd(e, c.ne, save, st))) return rc;
    Chain ch(nr);
    ChStage& s0 = ch.gemm(c.S, 0, p + poff(c.pbase + C_WF), 0, save ? c.A : nullptr, 0);   // A = S Wf + deg*bf (K8 hoisted)
    s0.bd = p + poff(c.pbase + C_BF); s0.seg_ptr = e.seg_ptr;
    ChStage& s1 = ch.gemm(nullptr, 0, p + poff(c.pbase + C_W1), 0, save ? c.Z1 : nullptr, 0);   // Z1 = relu([s2*A | x_recv] W1 + b1)
    s1.sa = p + poff(c.pbase + C_S2); s1.in_b = xrecv; s1.tb = 1; s1.wb = ch.weight(p + poff(c.pbase + C_W1) + EMB * EMB);
    s1.bias = p + poff(c.pbase + C_B1); s1.relu = 1;
    ChStage& s2 = ch.gemm(nullptr, 0, p + poff(c.pbase + C_W2), 0, c.OUT, 0);     // X' = relu(Z1 W2 + b2)
    s2.bias = p + poff(c.pbase + C_B2); s2.relu = 1;
    tail(ch);
    return launch_chain(ch, st);
}

static void conv_setup(ConvIO cv[3], const gcnn_dims* d, const Work& w, const gcnn_graph* cg, const gcnn_graph* kg) {
    const Acts &A = w.a, &G = w.g;
    cv[0] = ConvIO{P_CONV0, A.Xc, A.Xv, d->n_cons, d->n_vars, d->n_cons_edges, true, cg, P_CONS_EDGE,
                   A.PL1, A.PR1, A.S1, A.A1, A.Z1c, A.Xc2, G.PL1, G.PR1, G.S1, G.A1, G.Z1c, G.Xc2, G.Xc, G.Xv, w.q[0], w.mask[0], w.nrow[0]};
    cv[1] = ConvIO{P_CONV1, A.Xc2, A.Xv, d->n_cons, d->n_vars, d->n_cons_edges, false, cg, P_CONS_EDGE,
                   A.PL2, A.PR2, A.S2, A.A2, A.Z1v, A.Xv2, G.PL2, G.PR2, G.S2, G.A2, G.Z1v, G.Xv2, G.Xc2, G.Xv, w.q[1], w.mask[1], w.nrow[1]};
    cv[2] = ConvIO{P_CONV2, A.Xk, A.Xv2, d->n_cuts, d->n_vars, d->n_cut_edges, true, kg, P_CUT_EDGE,
                   A.PL3, A.PR3, A.S3, A.A3, A.Z1k, A.Xk2, G.PL3, G.PR3, G.S3, G.A3, G.Z1k, G.Xk2, G.Xk, G.Xv2, w.q[2], w.mask[2], w.nrow[2]};
}

static int check_common(const gcnn_dims* d, const float* params, const gcnn_graph* cg, const gcnn_graph* kg,
                        float* workspace, size_t workspace_floats) {
    if (!d || !params || !cg || !kg) return GCNN_E_BADARG;
    if (d->n_cons < 0 || d->n_vars < 0 || d->n_cuts < 0 || d->n_cons_edges < 0 || d->n_cut_edges < 0) return GCNN_E_BADARG;
    if (!workspace || workspace_floats < gcnn_workspace_floats(d)) return GCNN_E_WORKSPACE;
    if (((uintptr_t)workspace & 15) || ((uintptr_t)params & 15)) return GCNN_E_BADARG;
    return 0;
}

extern "C" int gcnn_forward(const gcnn_dims* d, const float* p, const float* cons_feats, const float* var_feats,
                 const float* cut_feats, const gcnn_graph* cg, const gcnn_graph* kg, float* workspace,
                 size_t workspace_floats, float* scores, int32_t save_for_backward, void* stream) {
    layout_init();
    int rc = check_common(d, p, cg, kg, workspace, workspace_floats);
    if (rc) return rc;
    if (d->n_cuts > 0 && !scores) return GCNN_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const bool save = save_for_backward != 0;
    Work w; carve(d, workspace, &w);
    const Acts& A = w.a;
    // embeddings (model.py:287-291) fused with the projections of the raw embeddings they feed (model.py:486-496):
    // three independent chains -> three streams
    const bool side = side_on();
    hipStream_t s1 = side ? g_side.s[0] : st, s2 = side ? g_side.s[1] : st;
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
    if (side && ((rc = ev_record(st, &e0)) || (rc = ev_wait(s1, e0, st)) || (rc = ev_wait(s2, e0, st)))) return rc;
    {
        Chain ch(d->n_cons);  // constraints: E1 -> Xc -> PL1
        ch.embed1(cons_feats, 4, p, P_CONS, save ? A.E1c : nullptr);
        ChStage& s = ch.gemm(nullptr, 0, p + poff(P_CONS + E_W2), 0, A.Xc, 0); s.bias = p + poff(P_CONS + E_B2); s.relu = 1;
        ChStage& t = ch.gemm(nullptr, 0, p + poff(P_CONV0 + C_WL), 0, A.PL1, 1); t.bias = p + poff(P_CONV0 + C_BL);
        if ((rc = launch_chain(ch, s1))) return rc;
    }
    {
        Chain ch(d->n_vars);  // variables: E1 -> Xv -> PR1, PR2 (both read the raw variable embedding, model.py:294-295)
        ch.embed1(var_feats, 14, p, P_VAR, save ? A.E1v : nullptr);
        ChStage& s = ch.gemm(nullptr, 0, p + poff(P_VAR + E_W2), 0, A.Xv, 0); s.bias = p + poff(P_VAR + E_B2); s.relu = 1;
        ch.gemm(nullptr, 0, p + poff(P_CONV0 + C_WR), 0, A.PR1, 1);
        ch.gemm(nullptr, 0, p + poff(P_CONV1 + C_WR), 0, A.PR2, 1);
        if ((rc = launch_chain(ch, st))) return rc;
    }
    {
        Chain ch(d->n_cuts);  // cuts: E1 -> Xk -> PL3
        ch.embed1(cut_feats, 6, p, P_CUT, save ? A.E1k : nullptr);
        ChStage& s = ch.gemm(nullptr, 0, p + poff(P_CUT + E_W2), 0, A.Xk, 0); s.bias = p + poff(P_CUT + E_B2); s.relu = 1;
        ChStage& t = ch.gemm(nullptr, 0, p + poff(P_CONV2 + C_WL), 0, A.PL3, 1); t.bias = p + poff(P_CONV2 + C_BL);
        if ((rc = launch_chain(ch, s2))) return rc;
    }
    if (side && ((rc = ev_record(s1, &e1)) || (rc = ev_record(s2, &e2)) || (rc = ev_wait(st, e1, s1)) || (rc = ev_wait(st, e2, s2)))) return rc;
    // convolutions (model.py:294-296), each followed in the same launch by what consumes its output
    ConvIO cv[3]; conv_setup(cv, d, w, cg, kg);
    if ((rc = conv_forward(p, cv[0], save, st, [&](Chain& ch) {   // updated constraints -> left projection of conv c->v
            ChStage& t = ch.gemm(nullptr, 0, p + poff(P_CONV1 + C_WL), 0, A.PL2, 1); t.bias = p + poff(P_CONV1 + C_BL);
        }))) return rc;
    if ((rc = conv_forward(p, cv[1], save, st, [&](Chain& ch) {   // updated variables -> right projection of conv v->k
            ch.gemm(nullptr, 0, p + poff(P_CONV2 + C_WR), 0, A.PR3, 1);
        }))) return rc;
    if ((rc = conv_forward(p, cv[2], save, st, [&](Chain& ch) {   // updated cuts -> readout (model.py:206-208, 299-300)
            ChStage& t = ch.gemm(nullptr, 0, p + poff(P_OUT), 0, save ? A.O1 : nullptr, 0); t.bias = p + poff(P_OUT + 1); t.relu = 1;
            ch.score(0, p + poff(P_OUT + 2), p + poff(P_OUT + 3), scores);
        }))) return rc;
    return 0;
}

// ---- backward ---------------------------------------------------------------------------------------------------
struct JobList {
    WgArgs wg; RdArgs rd; int nslab;
    int rdblk;
};
static void add_wg(JobList& jl, const float* x, const float* sx, const float* dmat, const int* seg_ptr, const float* d2,
                   int n, float* gw, float* gb, float* g2, float* partial) {
    if (n <= 0) return;  // empty input: gradients are exactly zero
    WgJob& j = jl.wg.job[jl.wg.njobs++];
    const int nb = cdiv(n, WG_ROWS);
    j.x = x; j.sx = sx; j.d = dmat; j.seg_ptr = seg_ptr; j.d2 = d2; j.n = n; j.blk0 = jl.wg.nblocks; j.slab0 = jl.nslab;
    const float* src = partial + (size_t)jl.nslab * WG_SLAB;
    jl.wg.nblocks += nb; jl.nslab += nb;
    auto rd = [&](const float* s, float* dst, int len) {
        RdJob& r = jl.rd.job[jl.rd.njobs++];
        r.src = s; r.dst = dst; r.nparts = nb; r.stride = WG_SLAB; r.len = len; r.blk0 = jl.rdblk;
        jl.rdblk += cdiv(len, EMB);
    };
    rd(src, gw, EMB * EMB);
    if (gb) rd(src + EMB * EMB, gb, EMB);
    if (g2) rd(src + EMB * EMB + EMB, g2, EMB);
}
// launch the weight-gradient jobs collected so far as one grouped kernel on `st`; slabs keep accumulating
static int flush_wg(JobList& jl, hipStream_t st) {
    if (jl.wg.nblocks > 0) {
        hipLaunchKernelGGL(k_wgrad, dim3(jl.wg.nblocks), dim3(256), 0, st, jl.wg);
        LAUNCHCHK();
    }
    jl.wg.njobs = 0; jl.wg.nblocks = 0;
    return 0;
}
static void add_rd(JobList& jl, const float* src, float* dst, int nparts, int stride, int len) {
    if (nparts <= 0) return;
    RdJob& r = jl.rd.job[jl.rd.njobs++];
    r.src = src; r.dst = dst; r.nparts = nparts; r.stride = stride; r.len = len; r.blk0 = jl.rdblk;
    jl.rdblk += cdiv(len, EMB);
}

// Receiver-side gradient chain of one convolution, appended to `ch` whose tile 0 already holds dX' (masked):
//   dZ1 = dX'pre W2^T (mask Z1) ; d x_recv = dZ1pre W1b^T ; dA = s2 * dZ1pre W1a^T ; dS = dA Wf^T
static void conv_bwd_chain(Chain& ch, const float* p, const ConvIO& c) {
    float* gxrecv = c.recv_left ? c.gXL : c.gXV;
    ChStage& s1 = ch.gemm(nullptr, 0, p + poff(c.pbase + C_W2), 1, c.gZ1, 0); s1.mask = c.Z1;
    ch.gemm(nullptr, 0, p + poff(c.pbase + C_W1) + EMB * EMB, 1, gxrecv, 1);
    ChStage& s3 = ch.gemm(nullptr, 0, p + poff(c.pbase + C_W1), 1, c.gA, 0); s3.so = p + poff(c.pbase + C_S2);
    // dS = dA Wf^T, and element-wise from it the receiver-ordered half of the edge gradient (see k_edge_fwd):
    //   dP_recv = s1*dS*N
    ChStage& s4 = ch.gemm(nullptr, 0, p + poff(c.pbase + C_WF), 1, c.gS, 0);
    s4.em_s = p + poff(c.pbase + C_S1); s4.em_a = c.N; s4.em_out = c.recv_left ? c.gPL : c.gPR;
}

// sender-ordered half of the edge gradient, and the weight-gradient jobs of the whole convolution
static int conv_backward_edges(const float* p, float* grads, const ConvIO& c, const Work& w, JobList& jl, hipStream_t st) {
    int rc;
    const int nr = c.recv_left ? c.nl : c.nv;
    const float* xrecv = c.recv_left ? c.xl : c.xv;
    // the receiver-ordered half (dP_recv) came out of the chain's epilogue; sender-ordered half: dS rows + 16-B masks,
    // which also yields Q, the per-sender share of d w_edge
    EdgeArgs e = conv_edge_args(p, c, !c.recv_left);
    e.d_s = c.gS; e.out = c.recv_left ? c.gPR : c.gPL; e.dw_rows = c.Q; e.mask = c.mask; e.xpos = c.recv_left ? c.g->v2l : c.g->l2v;
    if ((rc = launch_edge_bwd_send(e, c.ne, st))) return rc;
    const int* seg = c.recv_left ? c.g->l_ptr : c.g->v_ptr;
    float* gwe = grads + poff(c.pbase + C_WE);
    add_wg(jl, c.Z1, nullptr, c.gOUT, nullptr, nullptr, nr, grads + poff(c.pbase + C_W2), grads + poff(c.pbase + C_B2), nullptr, w.partial);
    add_wg(jl, c.A, p + poff(c.pbase + C_S2), c.gZ1, nullptr, nullptr, nr, grads + poff(c.pbase + C_W1), grads + poff(c.pbase + C_B1), nullptr, w.partial);
    add_wg(jl, xrecv, nullptr, c.gZ1, nullptr, nullptr, nr, grads + poff(c.pbase + C_W1) + EMB * EMB, nullptr, nullptr, w.partial);
    add_wg(jl, c.S, nullptr, c.gA, seg, nullptr, nr, grads + poff(c.pbase + C_WF), nullptr, grads + poff(c.pbase + C_BF), w.partial);
    // Q lives on the sender side: column-summed together with the sender-side projection's job
    add_wg(jl, c.xl, nullptr, c.gPL, nullptr, c.recv_left ? nullptr : c.Q, c.nl, grads + poff(c.pbase + C_WL),
           grads + poff(c.pbase + C_BL), c.recv_left ? nullptr : gwe, w.partial);
    add_wg(jl, c.xv, nullptr, c.gPR, nullptr, c.recv_left ? c.Q : nullptr, c.nv, grads + poff(c.pbase + C_WR), nullptr,
           c.recv_left ? gwe : nullptr, w.partial);
    return 0;
}

static int embed1_wgrad(int f, const float* x, const float* p, int pb, const float* dy, const float* yact, float* partial,
                        int n, int nblk, hipStream_t st) {
    if (n <= 0) return 0;
    const float *sh = p + poff(pb + E_SHIFT), *sc = p + poff(pb + E_SCALE);
    if (f == 4) hipLaunchKernelGGL(k_embed1_wgrad<4>, dim3(nblk), dim3(256), 0, st, x, sh, sc, dy, yact, partial, n, EMB1_ROWS);
    else if (f == 14) hipLaunchKernelGGL(k_embed1_wgrad<14>, dim3(nblk), dim3(256), 0, st, x, sh, sc, dy, yact, partial, n, EMB1_ROWS);
    else hipLaunchKernelGGL(k_embed1_wgrad<6>, dim3(nblk), dim3(256), 0, st, x, sh, sc, dy, yact, partial, n, EMB1_ROWS);
    LAUNCHCHK();
    return 0;
}

extern "C" int gcnn_mse_loss(const float* scores, const float* targets, int32_t n, float scale, float* loss_out, float* d_scores,
                  void* stream) {
    if (n < 0 || (n > 0 && (!scores || !targets))) return GCNN_E_BADARG;
    if (n == 0) {
        if (loss_out) HIPCHK(hipMemsetAsync(loss_out, 0, sizeof(float), (hipStream_t)stream));
        return 0;
    }
    hipLaunchKernelGGL(k_mse, dim3(1), dim3(256), 0, (hipStream_t)stream, scores, targets, scale, loss_out, d_scores, n);
    LAUNCHCHK();
    return 0;
}

extern "C" int gcnn_backward(const gcnn_dims* d, const float* p, const float* cons_feats, const float* var_feats,
                  const float* cut_feats, const gcnn_graph* cg, const gcnn_graph* kg, float* workspace,
                  size_t workspace_floats, const float* d_scores, float* grads, void* stream) {
    layout_init();
    int rc = check_common(d, p, cg, kg, workspace, workspace_floats);
    if (rc) return rc;
    if (!grads || (d->n_cuts > 0 && !d_scores)) return GCNN_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    Work w; carve(d, workspace, &w);
    const Acts &A = w.a, &G = w.g;
    JobList jl; memset(&jl, 0, sizeof(jl)); jl.wg.partial = w.partial;

    // side streams: sw runs the weight-gradient groups and the final reduction, sc the cut-/constraint-row tail chains
    const bool side = side_on();
    hipStream_t sw = side ? g_side.s[0] : st, sc = side ? g_side.s[1] : st;
    hipEvent_t ev = nullptr, ev_sc = nullptr, ev_sw = nullptr;
    // the reduction (re)writes every trainable gradient whenever all three node sets are non-empty; otherwise start from 0
    if (d->n_cons <= 0 || d->n_vars <= 0 || d->n_cuts <= 0) HIPCHK(hipMemsetAsync(grads, 0, (size_t)g_ptotal * sizeof(float), st));
    if (d->n_cuts <= 0) return 0;  // no cut => every gradient is 0
    if ((size_t)cdiv(d->n_cons, WG_ROWS) * 8 + (size_t)cdiv(d->n_vars, WG_ROWS) * 8 + (size_t)cdiv(d->n_cuts, WG_ROWS) * 8 > wg_slabs(d))
        return GCNN_E_WORKSPACE;
    ConvIO cv[3]; conv_setup(cv, d, w, cg, kg);
    struct { const float* x; const float* e1; float* gx; float* ge1; int n; int pb; int f; } em[3] = {
        {cons_feats, A.E1c, G.Xc, G.E1c, d->n_cons, P_CONS, 4},
        {var_feats, A.E1v, G.Xv, G.E1v, d->n_vars, P_VAR, 14},
        {cut_feats, A.E1k, G.Xk, G.E1k, d->n_cuts, P_CUT, 6}};
    // first embedding layer's weight gradient (VALU) + the reduction jobs of its slab
    auto embed_first_layer = [&](int i, hipStream_t s) -> int {
        int r = embed1_wgrad(em[i].f, em[i].x, p, em[i].pb, em[i].ge1, em[i].e1, w.emb_partial[i], em[i].n, w.emb_nblk[i], s);
        // kernel [f,64] and bias [64] are adjacent rows of the slab but separate (4-float aligned) tensors in the layout
        add_rd(jl, w.emb_partial[i], grads + poff(em[i].pb + E_W1), w.emb_nblk[i], (em[i].f + 1) * EMB, em[i].f * EMB);
        add_rd(jl, w.emb_partial[i] + em[i].f * EMB, grads + poff(em[i].pb + E_B1), w.emb_nblk[i], (em[i].f + 1) * EMB, EMB);
        return r;
    };

    // Dense(64->1) gradient (model.py:208): G.O1 = dscore (x) w2 masked by O1 > 0; dw2/db2 partials
    hipLaunchKernelGGL(k_score_bwd, dim3(w.score_nblk), dim3(256), 0, st, d_scores, A.O1, p + poff(P_OUT + 2), G.O1,
                       w.score_partial, d->n_cuts);
    LAUNCHCHK();
    add_rd(jl, w.score_partial, grads + poff(P_OUT + 2), w.score_nblk, 2 * EMB, EMB);
    add_rd(jl, w.score_partial + EMB, grads + poff(P_OUT + 3), w.score_nblk, 2 * EMB, 1);
    add_wg(jl, A.Xk2, nullptr, G.O1, nullptr, nullptr, d->n_cuts, grads + poff(P_OUT), grads + poff(P_OUT + 1), nullptr, w.partial);
    {   // cut rows: readout -> conv v->k receiver chain
        Chain ch(d->n_cuts);
        ChStage& s0 = ch.gemm(G.O1, 0, p + poff(P_OUT), 1, G.Xk2, 0); s0.mask = A.Xk2;
        conv_bwd_chain(ch, p, cv[2]);
        if ((rc = launch_chain(ch, st))) return rc;
    }
    if ((rc = conv_backward_edges(p, grads, cv[2], w, jl, st))) return rc;
    if (side && ((rc = ev_record(st, &ev)) || (rc = ev_wait(sw, ev, st)) || (rc = ev_wait(sc, ev, st)))) return rc;
    if ((rc = flush_wg(jl, sw))) return rc;   // readout + conv v->k weight gradients
    {   // cut rows: dXk = dXk(W1b part) + dPL3 Wl3^T, masked by Xk; dE1k; then the cut embedding's first layer
        Chain ch(d->n_cuts);
        ChStage& s0 = ch.gemm(G.PL3, 0, p + poff(P_CONV2 + C_WL), 1, G.Xk, 0); s0.add = G.Xk; s0.mask = A.Xk;
        ch.gemm(nullptr, 0, p + poff(P_CUT + E_W2), 1, G.E1k, 0);
        if ((rc = launch_chain(ch, sc)) || (rc = embed_first_layer(2, sc))) return rc;
    }
    {   // variable rows: dXv2 = dPR3 Wr3^T (mask Xv2) -> conv c->v receiver chain
        Chain ch(d->n_vars);
        ChStage& s0 = ch.gemm(G.PR3, 0, p + poff(P_CONV2 + C_WR), 1, G.Xv2, 0); s0.mask = A.Xv2;
        conv_bwd_chain(ch, p, cv[1]);
        if ((rc = launch_chain(ch, st))) return rc;
    }
    if ((rc = conv_backward_edges(p, grads, cv[1], w, jl, st))) return rc;
    if (side && ((rc = ev_record(st, &ev)) || (rc = ev_wait(sw, ev, st)))) return rc;
    if ((rc = flush_wg(jl, sw))) return rc;   // conv c->v weight gradients
    {   // constraint rows: dXc2 = dPL2 Wl2^T (mask Xc2) -> conv v->c receiver chain
        Chain ch(d->n_cons);
        ChStage& s0 = ch.gemm(G.PL2, 0, p + poff(P_CONV1 + C_WL), 1, G.Xc2, 0); s0.mask = A.Xc2;
        conv_bwd_chain(ch, p, cv[0]);
        if ((rc = launch_chain(ch, st))) return rc;
    }
    if ((rc = conv_backward_edges(p, grads, cv[0], w, jl, st))) return rc;
    if (side && ((rc = ev_record(st, &ev)) || (rc = ev_wait(sw, ev, st)) || (rc = ev_wait(sc, ev, st)))) return rc;
    if ((rc = flush_wg(jl, sw))) return rc;   // conv v->c weight gradients
    {   // constraint rows: dXc = dXc(W1b part) + dPL1 Wl1^T, masked by Xc; dE1c; then the constraint embedding's first layer
        Chain ch(d->n_cons);
        ChStage& s0 = ch.gemm(G.PL1, 0, p + poff(P_CONV0 + C_WL), 1, G.Xc, 0); s0.add = G.Xc; s0.mask = A.Xc;
        ch.gemm(nullptr, 0, p + poff(P_CONS + E_W2), 1, G.E1c, 0);
        if ((rc = launch_chain(ch, sc)) || (rc = embed_first_layer(0, sc))) return rc;
        if (side && (rc = ev_record(sc, &ev_sc))) return rc;
    }
    {   // variable rows: dXv = dXv(W1b part) + dPR2 Wr2^T + dPR1 Wr1^T, masked by Xv; dE1v
        Chain ch(d->n_vars);
        ChStage& s0 = ch.gemm(G.PR2, 0, p + poff(P_CONV1 + C_WR), 1, G.Xv, 0);
        s0.in_b = G.PR1; s0.tb = 1; s0.wb = ch.weight(p + poff(P_CONV0 + C_WR)); s0.add = G.Xv; s0.mask = A.Xv;
        ch.gemm(nullptr, 0, p + poff(P_VAR + E_W2), 1, G.E1v, 0);
        if ((rc = launch_chain(ch, st))) return rc;
    }
    // tail on sw (needs all three tail chains): the embeddings' weight gradients, then the reduction of every slab
    if (side && ((rc = ev_record(st, &ev)) || (rc = ev_wait(sw, ev, st)) || (rc = ev_wait(sw, ev_sc, sc)))) return rc;
    if ((rc = embed_first_layer(1, sw))) return rc;
    for (int i = 0; i < 3; ++i)
        add_wg(jl, em[i].e1, nullptr, em[i].gx, nullptr, nullptr, em[i].n, grads + poff(em[i].pb + E_W2), grads + poff(em[i].pb + E_B2), nullptr, w.partial);
    if ((rc = flush_wg(jl, sw))) return rc;
    if ((size_t)jl.nslab > wg_slabs(d)) return GCNN_E_WORKSPACE;
    if (jl.rdblk > 0) {
        hipLaunchKernelGGL(k_reduce, dim3(jl.rdblk), dim3(256), 0, sw, jl.rd);
        LAUNCHCHK();
    }
    if (side && ((rc = ev_record(sw, &ev_sw)) || (rc = ev_wait(st, ev_sw, sw)))) return rc;
    return 0;
}

// ---- PreNorm fitting statistics (model.py:394-423) ------------------------------------------------------------------
extern "C" int gcnn_prenorm_stats(const gcnn_dims* d, const float* p, const float* cons_feats, const float* var_feats,
                                  const float* cut_feats, const gcnn_graph* cg, const gcnn_graph* kg, float* workspace,
                                  size_t workspace_floats, int32_t layer, double* out_mean_var, void* stream) {
    layout_init();
    int rc = check_common(d, p, cg, kg, workspace, workspace_floats);
    if (rc) return rc;
    if (layer < 0 || layer > 10 || !out_mean_var) return GCNN_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    Work w; carve(d, workspace, &w);
    StatArgs a; memset(&a, 0, sizeof(a));
    double count = 0.0;
    int units = 1;
    if (layer <= 4) {   // input layers: raw features, one unit per column (edge features: a single column)
        const float* xs[5] = {cons_feats, cg->l_coef, var_feats, cut_feats, kg->l_coef};
        const int ns[5] = {d->n_cons, d->n_cons_edges, d->n_vars, d->n_cuts, d->n_cut_edges};
        const int fs[5] = {4, 1, 14, 6, 1};
        a.src = ST_COLS; a.x = xs[layer]; a.n = ns[layer]; a.f = fs[layer]; units = a.f; count = (double)a.n;
    } else {
        ConvIO cv[3]; conv_setup(cv, d, w, cg, kg);
        const ConvIO& c = cv[(layer - 5) >> 1];
        if (((layer - 5) & 1) == 0) {   // feature_module_final's PreNorm: all E*64 joint pre-activations, one unit
            a.src = ST_EDGE; a.n = c.ne; a.right = c.g->l_oth; a.coef = c.g->l_coef; a.pl = c.PL; a.pr = c.PR;
            a.w_edge = p + poff(c.pbase + C_WE); a.e_shift = p + poff(c.pedge); a.e_scale = p + poff(c.pedge + 1);
            if (c.ne > 0) {
                hipLaunchKernelGGL(k_expand_ptr, dim3(std::min(cdiv(c.nl, 256), 1024)), dim3(256), 0, st, c.g->l_ptr, c.nl, w.stat_ids);
                LAUNCHCHK();
            }
            a.left = w.stat_ids; count = (double)c.ne * EMB;
        } else {                        // post_conv_module's PreNorm: all R*64 elements of the scatter-sum output, one unit
            a.src = ST_FLAT; a.x = c.A; a.n = c.recv_left ? c.nl : c.nv; count = (double)a.n * EMB;
        }
    }
    if (count <= 0.0) {   // nothing to absorb: mean 0, variance 0 (the caller skips empty batches)
        HIPCHK(hipMemsetAsync(out_mean_var, 0, 2 * (size_t)units * sizeof(double), st));
        return 0;
    }
    const int work = a.src == ST_EDGE ? cdiv(a.n, 16) : (a.src == ST_FLAT ? cdiv(a.n, 4) : cdiv(a.n, 256));
    const int grid = std::max(1, std::min(work, ST_MAX_BLOCKS));
    a.partial = w.stats;
    a.mean = nullptr;
    hipLaunchKernelGGL(k_stats, dim3(grid), dim3(256), 0, st, a); LAUNCHCHK();
    hipLaunchKernelGGL(k_stats_final, dim3(1), dim3(64), 0, st, w.stats, grid, units, count, out_mean_var); LAUNCHCHK();
    a.mean = out_mean_var;
    hipLaunchKernelGGL(k_stats, dim3(grid), dim3(256), 0, st, a); LAUNCHCHK();
    hipLaunchKernelGGL(k_stats_final, dim3(1), dim3(64), 0, st, w.stats, grid, units, count, out_mean_var + units); LAUNCHCHK();
    return 0;
}

extern "C" int gcnn_adam_step(float* params, const float* grads, float* m, float* v, int32_t n, float lr_t, float beta1, float beta2,
                   float eps, const float* grad_scale, void* stream) {
    if (n < 0 || (n > 0 && (!params || !grads || !m || !v))) return GCNN_E_BADARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_adam, dim3(std::min(cdiv(n, 256), 1024)), dim3(256), 0, (hipStream_t)stream, params, grads, m, v, n,
                       lr_t, beta1, beta2, eps, grad_scale);
    LAUNCHCHK();
    return 0;
}
