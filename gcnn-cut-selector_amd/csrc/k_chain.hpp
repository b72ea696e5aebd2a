#pragma once
#include "gcnn_common.hpp"

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
// Independent chains over different row sets (the three embeddings; a tail chain beside a receiver chain) share ONE
// launch: blocks [blk0[i], blk0[i+1]) run chain i.  Overlap without streams or events, one launch ramp instead of three.
#define CH_MAX_GROUPS 3
struct ChMulti { int ngroups; int blk0[CH_MAX_GROUPS + 1]; ChArgs g[CH_MAX_GROUPS]; };
static_assert(sizeof(ChMulti) <= 4096, "kernel arguments are limited to 4 KB");

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
__global__ __launch_bounds__(NWAVES * 64) void k_chain(ChMulti m) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int gi = 0;
    while (gi + 1 < m.ngroups && (int)blockIdx.x >= m.blk0[gi + 1]) ++gi;
    const ChArgs& a = m.g[gi];
    const int bid = blockIdx.x - m.blk0[gi], nblk = m.blk0[gi + 1] - m.blk0[gi];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int ntile = (a.n + 15) >> 4;
    const int tile0 = bid * NWAVES + wv;
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

    for (int tile = tile0; tile < ntile; tile += nblk * NWAVES) {
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

