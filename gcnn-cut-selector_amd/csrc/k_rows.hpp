#pragma once
#include "gcnn_common.hpp"

// ---------------------------------------------------------------------------------------------------------------
// Row programs.  Every node-side layer of the model is row-local: a 16-row tile of a [N,64] matrix goes through a
// sequence of 64x64 products with element-wise epilogues.  Each such sequence -- an embedding with the projections it
// feeds (model.py:174-198, 486-496), the receiver-side update S -> A -> Z1 -> X' -> next projection of a
// PartialGraphConvolution (model.py:498-508, 570-573), or the gradient of either -- is ONE statically typed program:
// a wave owns a tile, reads every weight from LDS (staged once per block), issues ALL global operand loads of the tile
// before its first MFMA (one memory round trip per tile, whatever the number of stages) and stores only the tensors the
// backward pass / the next edge pass need.  Independent programs over different row sets share a launch (block ranges).
//
// Register-resident transposed products.  Each stage computes  Y^T[64 x 16 rows] = Wop[64 x 64] . X^T  with the weights as
// the MFMA A operand (v_mfma_f32_16x16x4_f32; read from LDS, independent of the data, so the reads run ahead) and the
// activation tile as the B operand.  With that orientation the accumulator of one stage IS the B operand of the next:
//   lane (j = lane&15, g = lane>>4) holds, for each 16-feature block m and i = 0..3, the element X[row0 + j][16*m + 4*g + i]
//   -- as B operand of k-step (m, i) (the instruction's k index is g), and as C/D layout of output block m (rows of D =
//   features 4*g + i of block m, column = row j of the tile).
// So a whole program runs without any LDS round trip for activations; global rows are read/written as float4 pieces
// X[row][16*m + 4*g .. +3] straight from/to that layout.
// ---------------------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

struct RTile { float v[4][4]; };

// Weights and bias vectors in LDS do not change from tile to tile, so the optimiser would hoist their reads out of the
// tile loop -- 64 VGPRs per matrix, i.e. spills.  Adding an opaque zero to the address keeps each read where it is used.
__device__ __forceinline__ const float* lds_here(const float* p) {
    int zero = 0;
    asm volatile("" : "+v"(zero));
    return p + zero;
}

__device__ __forceinline__ void rt_zero(RTile& t) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) t.v[m][i] = 0.f;
}
__device__ __forceinline__ void rt_load(RTile& t, const float* x, int row, bool ok, int g) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) f = *(const float4*)(x + (size_t)row * EMB + 16 * m + 4 * g);
        t.v[m][0] = f.x; t.v[m][1] = f.y; t.v[m][2] = f.z; t.v[m][3] = f.w;
    }
}
__device__ __forceinline__ void rt_store(const RTile& t, float* x, int row, bool ok, int g) {
    if (!ok || !x) return;
#pragma unroll
    for (int m = 0; m < 4; ++m)
        *(float4*)(x + (size_t)row * EMB + 16 * m + 4 * g) = make_float4(t.v[m][0], t.v[m][1], t.v[m][2], t.v[m][3]);
}

// acc[mo] += Wop[16*mo + (lane&15)][kf] * (scale * T[kf]), kf = 16*mt + 4*g + i;
// forward (x @ W): Wop[o][k] = W[k][o];  backward (x @ W^T): Wop[o][k] = W[o][k].  wl: the matrix in LDS.
// MODE (a bool converts): GEMM_FWD  = forward, matrix staged k-interleaved (stage_lds<..., true>): one float4 per four MFMAs
//                         GEMM_BWD  = backward, matrix staged row-major [64][LDW]: one float4 per four MFMAs
//                         GEMM_FWD_RM = forward from the ROW-MAJOR staging (four scalar reads per four MFMAs): for a program that
//                                     needs a matrix in both directions and has no LDS for two copies (convturn_program)
enum { GEMM_FWD = 0, GEMM_BWD = 1, GEMM_FWD_RM = 2 };
template <int MODE>
__device__ __forceinline__ void rt_gemm(const RTile& t, float scale, const float* wl, f32x4 (&acc)[4], int lane) {
    const int m = lane & 15, g = lane >> 4;
    wl = lds_here(wl);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        float av[4][4];  // [mo][i]
        if (MODE == GEMM_BWD) {
#pragma unroll
            for (int mo = 0; mo < 4; ++mo) {
                const float4 w4 = *(const float4*)(wl + (16 * mo + m) * LDW + 16 * mt + 4 * g);
                av[mo][0] = w4.x; av[mo][1] = w4.y; av[mo][2] = w4.z; av[mo][3] = w4.w;
            }
        } else if (MODE == GEMM_FWD) {   // W[4q..4q+3][o] are four consecutive floats
#pragma unroll
            for (int mo = 0; mo < 4; ++mo) {
                const float4 w4 = *(const float4*)(wl + ((4 * mt + g) * 64 + 16 * mo + m) * 4);
                av[mo][0] = w4.x; av[mo][1] = w4.y; av[mo][2] = w4.z; av[mo][3] = w4.w;
            }
        } else {
#pragma unroll
            for (int mo = 0; mo < 4; ++mo)
#pragma unroll
                for (int i = 0; i < 4; ++i) av[mo][i] = wl[(16 * mt + 4 * g + i) * LDW + 16 * mo + m];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float b = t.v[mt][i] * scale;
#pragma unroll
            for (int mo = 0; mo < 4; ++mo) acc[mo] = mfma16(av[mo][i], b, acc[mo]);
        }
    }
}
// o = (scale * in) x W
template <int TRANSB>
__device__ __forceinline__ void rt_mm(RTile& o, const RTile& in, float scale, const float* wl, int lane) {
    f32x4 acc[4];
#pragma unroll
    for (int mo = 0; mo < 4; ++mo) acc[mo] = (f32x4){0.f, 0.f, 0.f, 0.f};
    rt_gemm<TRANSB>(in, scale, wl, acc, lane);
#pragma unroll
    for (int mo = 0; mo < 4; ++mo)
#pragma unroll
        for (int i = 0; i < 4; ++i) o.v[mo][i] = acc[mo][i];
}
// o = (sa * a) x Wa + b x Wb   (same accumulators, a first)
template <int TRANSB>
__device__ __forceinline__ void rt_mm2(RTile& o, const RTile& a, float sa, const float* wa, const RTile& b, const float* wb, int lane) {
    f32x4 acc[4];
#pragma unroll
    for (int mo = 0; mo < 4; ++mo) acc[mo] = (f32x4){0.f, 0.f, 0.f, 0.f};
    rt_gemm<TRANSB>(a, sa, wa, acc, lane);
    rt_gemm<TRANSB>(b, 1.f, wb, acc, lane);
#pragma unroll
    for (int mo = 0; mo < 4; ++mo)
#pragma unroll
        for (int i = 0; i < 4; ++i) o.v[mo][i] = acc[mo][i];
}
// o += vec (a [64] vector in LDS), then optional ReLU
template <bool RELU>
__device__ __forceinline__ void rt_bias(RTile& o, const float* vec, int g) {
    vec = lds_here(vec);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float4 b = *(const float4*)(vec + 16 * m + 4 * g);
        const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = o.v[m][i] + bv[i];
            o.v[m][i] = RELU ? fmaxf(v, 0.f) : v;
        }
    }
}
__device__ __forceinline__ void rt_mask(RTile& o, const RTile& act) {   // o *= (act > 0): gradient of a ReLU whose output is act
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) o.v[m][i] = act.v[m][i] > 0.f ? o.v[m][i] : 0.f;
}
// ReLU patterns as bits.  A ReLU output that the backward pass needs only as a mask (X', Z1 of a convolution in its gradient
// program, a raw embedding X in its tail, E1 in the first layer's weight gradient) is not re-read as a [N,64] fp32 matrix
// (256 B per row) but as 64 bits per row: lane (j, g) of a tile keeps the 16 bits of ITS 16 values -- bit 4m+i <-> feature
// 16m+4g+i -- as one 16-bit word at mask[row][g] (8 B per row).  Writer and reader are the same lane of the same tile layout, so
// no bit crosses a lane; a 16-row tile's masks are 128 contiguous bytes.
typedef unsigned short mask16;
__device__ __forceinline__ unsigned rt_mask_bits(const RTile& t) {
    unsigned b = 0;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) b |= (t.v[m][i] > 0.f ? 1u : 0u) << (4 * m + i);
    return b;
}
__device__ __forceinline__ void rt_mask_store(const RTile& t, mask16* dst, int row, bool ok, int g) {
    if (ok && dst) dst[(size_t)row * 4 + g] = (mask16)rt_mask_bits(t);
}
__device__ __forceinline__ unsigned rt_mask_load(const mask16* src, int row, bool ok, int g) {
    return ok ? (unsigned)src[(size_t)row * 4 + g] : 0u;
}
__device__ __forceinline__ void rt_mask_apply(RTile& o, unsigned bits) {   // o *= pattern: gradient of a ReLU whose output had this pattern
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) o.v[m][i] = (bits >> (4 * m + i)) & 1u ? o.v[m][i] : 0.f;
}
__device__ __forceinline__ void rt_clear_unless(RTile& o, bool ok) {    // rows past the end of the matrix stay exactly zero
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) o.v[m][i] = ok ? o.v[m][i] : 0.f;
}

// Stage NM 64x64 matrices (one [64][LDW]-sized slot each) and NV 64-float vectors (nullptr -> zeros) into LDS: every global
// load is issued before the first LDS write, so a block pays ONE round trip for all of them.
// K4 = false: row-major with padded rows [64][LDW] -- what the backward products (x @ W^T: rt_gemm<true>) read as float4.
// K4 = true : k-interleaved [16][64][4] -- element W[k][o] at ((k>>2)*64 + o)*4 + (k&3) -- so that the forward products
//             (x @ W: rt_gemm<false>), whose A operand needs W[4q..4q+3][o] per lane, also read one float4 per four MFMAs
//             (row-major would take four scalar LDS reads, each waited for; 16 consecutive lanes read 256 contiguous bytes).
template <int NM, int NV, int NT, bool K4 = false>
__device__ __forceinline__ void stage_lds(float* smem, const float* const (&w)[NM], const float* const (&v)[NV]) {
    constexpr int PER = 1024 / NT;   // float4 per thread per matrix: 4 (256 threads) or 2 (512)
    float4 tmp[NM][PER];
    float vec[NV];
    if (K4) {   // thread -> column o = t & 63 and row quads q = (t >> 6) + (NT/64)*i: four coalesced scalar loads per quad
#pragma unroll
        for (int wi = 0; wi < NM; ++wi)
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int q = (threadIdx.x >> 6) + (NT / 64) * i, o = threadIdx.x & 63;
                const float* src = w[wi] + (size_t)(4 * q) * 64 + o;
                tmp[wi][i] = make_float4(src[0], src[64], src[128], src[192]);
            }
    } else {
#pragma unroll
        for (int wi = 0; wi < NM; ++wi)
#pragma unroll
            for (int i = 0; i < PER; ++i) tmp[wi][i] = *(const float4*)(w[wi] + (size_t)(i * NT + threadIdx.x) * 4);
    }
#pragma unroll
    for (int vi = 0; vi < NV; ++vi) vec[vi] = (threadIdx.x < 64 && v[vi]) ? v[vi][threadIdx.x] : 0.f;
#pragma unroll
    for (int wi = 0; wi < NM; ++wi)
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            if (K4) {
                const int q = (threadIdx.x >> 6) + (NT / 64) * i, o = threadIdx.x & 63;
                *(float4*)(smem + wi * 64 * LDW + (q * 64 + o) * 4) = tmp[wi][i];
            } else {
                const int idx = i * NT + threadIdx.x;
                *(float4*)(smem + wi * 64 * LDW + (idx >> 4) * LDW + (idx & 15) * 4) = tmp[wi][i];
            }
        }
    if (threadIdx.x < 64) {
#pragma unroll
        for (int vi = 0; vi < NV; ++vi) smem[NM * 64 * LDW + vi * 64 + threadIdx.x] = vec[vi];
    }
}
#define ROWS_LDS_FLOATS(NM, NV) ((NM) * 64 * LDW + (NV) * 64)

// ---------------------------------------------------------------------------------------------------------------
// Program 1 (forward): embedding + the projections of the raw embedding (model.py:174-198 applied :287-291; :486-496)
//   E1 = relu(((x + shift) * scale) W1 + b1)   [VALU, K = F <= 14]      (its ReLU pattern -> m_e1)
//   X  = relu(E1 W2 + b2)                                               -> xo
//   P_k = X Wp_k (+ bp_k)                        k < NPROJ               -> po[k]
// ---------------------------------------------------------------------------------------------------------------
struct EmbArgs {
    const float *x, *shift, *scale, *w1, *b1; mask16* m_e1;    // m_e1, m_x: ReLU patterns for the backward pass (optional); E1 itself is never stored (k_wgrad.hpp rebuilds it)
    const float *w2, *b2; float* xo; mask16* m_x;
    const float* wp[2]; const float* bp[2]; float* po[2];
    int n;
};
template <int F, int NPROJ, int NT>
__device__ __forceinline__ void emb_program(const EmbArgs& a, float* smem, int bid, int nblk) {
    constexpr int NWAVES = NT / 64, NM = 1 + NPROJ, NV = 2 + NPROJ;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
    const int ntile = (a.n + 15) >> 4;
    float* vecs = smem + NM * 64 * LDW;
    float* w1s = smem + ROWS_LDS_FLOATS(NM, NV);   // first-layer kernel [F][64], bias b1 in vecs[0]
    int tile = bid + nblk * wv;   // tiles are dealt round-robin over the blocks, then over a block's waves
    float xv[F], xn_[F];   // operands of the current tile / of the wave's next tile (requested before the current one is computed)
    auto load_ops = [&](float (&dst)[F], int t) {
        const int row = t * 16 + j;
#pragma unroll
        for (int f = 0; f < F; ++f) dst[f] = row < a.n ? a.x[(size_t)row * F + f] : 0.f;
    };
    load_ops(xv, tile);
    // the first-layer kernel [F][64] is requested BEFORE the staging of the matrices (whose LDS writes wait for their loads): one
    // memory round trip for the whole prologue instead of two
    constexpr int W1N = (F * 64 + NT - 1) / NT;
    float w1r[W1N];
#pragma unroll
    for (int q = 0; q < W1N; ++q) { const int i = threadIdx.x + q * NT; w1r[q] = i < F * 64 ? a.w1[i] : 0.f; }
    if (NPROJ == 2) {
        const float* const w[3] = {a.w2, a.wp[0], a.wp[1]};
        const float* const v[4] = {a.b1, a.b2, a.bp[0], a.bp[1]};
        stage_lds<3, 4, NT, true>((float*)smem, w, v);
    } else {
        const float* const w[2] = {a.w2, a.wp[0]};
        const float* const v[3] = {a.b1, a.b2, a.bp[0]};
        stage_lds<2, 3, NT, true>((float*)smem, w, v);
    }
#pragma unroll
    for (int q = 0; q < W1N; ++q) { const int i = threadIdx.x + q * NT; if (i < F * 64) w1s[i] = w1r[q]; }
    float shift[F], scale[F];
#pragma unroll
    for (int f = 0; f < F; ++f) { shift[f] = a.shift[f]; scale[f] = a.scale[f]; }
    __syncthreads();
    for (; tile < ntile; tile += nblk * NWAVES) {
        load_ops(xn_, tile + nblk * NWAVES);   // past the last tile: no loads
        const int row = tile * 16 + j;
        const bool ok = row < a.n;
        RTile o, t;
        rt_zero(o);
        const float* w1h = lds_here(w1s);
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const float xn = ok ? (xv[f] + shift[f]) * scale[f] : 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float4 w = *(const float4*)(w1h + f * EMB + 16 * m + 4 * g);
                o.v[m][0] = fmaf(xn, w.x, o.v[m][0]); o.v[m][1] = fmaf(xn, w.y, o.v[m][1]);
                o.v[m][2] = fmaf(xn, w.z, o.v[m][2]); o.v[m][3] = fmaf(xn, w.w, o.v[m][3]);
            }
        }
        rt_bias<true>(o, vecs, g);
        rt_clear_unless(o, ok);
        rt_mask_store(o, a.m_e1, row, ok, g);
        rt_mm<false>(t, o, 1.f, smem, lane);
        rt_bias<true>(t, vecs + 64, g);
        rt_clear_unless(t, ok);
        rt_store(t, a.xo, row, ok, g);
        rt_mask_store(t, a.m_x, row, ok, g);
#pragma unroll
        for (int k = 0; k < NPROJ; ++k) {
            rt_mm<false>(o, t, 1.f, smem + (1 + k) * 64 * LDW, lane);
            rt_bias<false>(o, vecs + (2 + k) * 64, g);
            rt_store(o, a.po[k], row, ok, g);
        }
#pragma unroll
        for (int f = 0; f < F; ++f) xv[f] = xn_[f];
    }
}
// The folded weights of the three convolutions (Program 2): M = s2 * Wf * W1a [64,64] followed by u = s2 * bf * W1a [64], FUSE_FLOATS
// floats per convolution.  One 256-thread block each, riding in the embedding launch (the first consumer is the third launch
// of a forward pass); exact fp32 FMA chains in k order.  smem: 2 * 64 * LDW floats.  Behind M | u the block leaves copies of Wf, W1a
// and bf as they were in this forward pass: the blocks that turn the folded gradients into those of Wf, W1a and bf (fold_block,
// k_wgrad.hpp) ride in the launch whose Adam updates overwrite these parameters, so they read the copies.
#define FUSE_WF (EMB * EMB + EMB)
#define FUSE_W1A (FUSE_WF + EMB * EMB)
#define FUSE_BF (FUSE_W1A + EMB * EMB)
#define FUSE_FLOATS (FUSE_BF + EMB)
struct FuseArgs { const float *wf[3], *bf[3], *s2[3], *w1a[3]; float* out[3]; };
__device__ __forceinline__ void fuse_weights(const FuseArgs& f, const int k, float* smem) {
    float* wfs = smem;                  // Wf  [i][LDW]
    float* was = smem + 64 * LDW;       // W1a [j][LDW]
    const int t = threadIdx.x;
    if (t < 256) {
        for (int q = t; q < 1024; q += 256) {
            const float4 a = *(const float4*)(f.wf[k] + q * 4), b = *(const float4*)(f.w1a[k] + q * 4);
            *(float4*)(wfs + (q >> 4) * LDW + (q & 15) * 4) = a;
            *(float4*)(was + (q >> 4) * LDW + (q & 15) * 4) = b;
            *(float4*)(f.out[k] + FUSE_WF + q * 4) = a;
            *(float4*)(f.out[k] + FUSE_W1A + q * 4) = b;
        }
        if (t < EMB) f.out[k][FUSE_BF + t] = f.bf[k][t];
    }
    __syncthreads();
    if (t >= 256) return;
    const float s2 = *f.s2[k];
    const int i0 = (t >> 4) * 4, o0 = (t & 15) * 4;   // a 4 x 4 block of M per thread
    float acc[4][4] = {};
    for (int j = 0; j < EMB; ++j) {
        const float4 w = *(const float4*)(was + j * LDW + o0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = wfs[(i0 + r) * LDW + j];
            acc[r][0] = fmaf(a, w.x, acc[r][0]); acc[r][1] = fmaf(a, w.y, acc[r][1]);
            acc[r][2] = fmaf(a, w.z, acc[r][2]); acc[r][3] = fmaf(a, w.w, acc[r][3]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        *(float4*)(f.out[k] + (i0 + r) * EMB + o0) = make_float4(s2 * acc[r][0], s2 * acc[r][1], s2 * acc[r][2], s2 * acc[r][3]);
    if (t < EMB) {
        float u = 0.f;
        for (int j = 0; j < EMB; ++j) u = fmaf(f.bf[k][j], was[j * LDW + t], u);
        f.out[k][EMB * EMB + t] = s2 * u;
    }
}
struct EmbGroupArgs { int blk0[4]; EmbArgs v, c, k; FuseArgs fz; };   // variables (F=14, two projections), constraints (4), cuts (6); blocks blk0[3] .. +2: fuse_weights
#define EMB_LDS_FLOATS (ROWS_LDS_FLOATS(3, 4) + 14 * 64)
static_assert(EMB_LDS_FLOATS >= 2 * 64 * LDW, "fuse_weights stages two matrices in the embedding launch's LDS");
// Two blocks of this launch share a CU when the row sets are large (launch_embed_fwd: 52 KB of LDS each, four waves per SIMD with
// 8-wave blocks), which takes at most 128 registers per lane: pinned, since two registers more silently halve the residency.
template <int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) __attribute__((amdgpu_waves_per_eu(NWAVES / 2, NWAVES / 2))) void k_embed_fwd(EmbGroupArgs m) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;
    if (b < m.blk0[1]) emb_program<14, 2, NWAVES * 64>(m.v, smem, b, m.blk0[1]);
    else if (b < m.blk0[2]) emb_program<4, 1, NWAVES * 64>(m.c, smem, b - m.blk0[1], m.blk0[2] - m.blk0[1]);
    else if (b < m.blk0[3]) emb_program<6, 1, NWAVES * 64>(m.k, smem, b - m.blk0[2], m.blk0[3] - m.blk0[2]);
    else fuse_weights(m.fz, b - m.blk0[3], smem);
}

// ---------------------------------------------------------------------------------------------------------------
// Program 2 (forward): receiver-side update of a PartialGraphConvolution after the edge pass (model.py:498-508, 568-573)
//   A  = S Wf + deg_r bf                 (the per-edge Dense hoisted past the scatter-sum)
//   Z1 = relu([s2*A | x_recv] W1 + b1)                                                             -> z1 (optional)
//        Nothing lies between the two layers but the PreNorm scale s2, so they are ONE product with the folded matrix
//        M = s2*Wf*W1a and vector u = s2*bf*W1a (made once per forward by fuse_weights, riding in the embedding launch):
//          Z1 = relu(S M + deg_r u + x_recv W1b + b1)
//        -- a 64x64 product per receiver row less, and A is neither written nor read again (the backward pass folds the same
//        way: dS = dZ1 M^T; the gradients of Wf, bf and W1a come out of S^T dZ1, see fold_block, k_wgrad.hpp).  KEEP_A = true is the
//        two-layer form that materialises A: PreNorm fitting needs its statistics (model.py:503, 570).
//   X' = relu(Z1 W2 + b2)                                                                          -> out
//   TAIL = CF_PROJ   :  T = X' Wt (+ bt): the next convolution's projection                        -> t_out
//   TAIL = CF_READOUT:  O1 = relu(X' Wt + bt) -> o1 (optional);  score = O1 . ws + bs (model.py:206-208) -> scores
//   TAIL = CF_LOSS   :  the readout and, in the same pass, the MSE head (model_trainer.py:271) and the gradient of the
//                       readout's Dense(64->1): with d_k = score_k - y_k,  ds_k = 2*scale*d_k,
//                         dO1pre[k] = ds_k * ws * (O1[k] > 0)                                       -> g_o1
//                         per-tile partials {dws = sum_k ds_k O1[k], dbs = sum_k ds_k, loss = scale*sum_k d_k^2} -> head_partial
// ---------------------------------------------------------------------------------------------------------------
struct ConvFArgs {
    const float* s; const int* seg_ptr; const float *wf, *bf; float* a_out;     // wf, bf, a_out, s2, w1a: the KEEP_A form only
    const float *mfuse, *ufuse;                                                    // M = s2*Wf*W1a [64,64], u = s2*bf*W1a [64] (k_fuse)
    const float *s2, *xrecv, *w1a, *w1b, *b1; float* z1; mask16* m_z1;     // m_z1, m_out: ReLU patterns for the gradient program (optional)
    const float *w2, *b2; float* out; mask16* m_out;
    const float *wt, *bt; float* t_out;      // readout: wt/bt = readout Dense(64,relu), t_out = O1
    const float *ws, *bs; float* scores;      // readout only
    const float* targets; float loss_scale; float* g_o1; float* head_partial;   // CF_LOSS only; partial: [tiles][HEAD_SLAB]
    int n;
};
enum { CF_PROJ = 0, CF_READOUT = 1, CF_LOSS = 2 };
#define HEAD_SLAB (2 * EMB)   // per-tile partial of the loss head: dws[64], dbs at 64, loss at 65
// sum over the 16 lanes of a DPP row (= the 16 rows of a tile), result in every lane of the row
__device__ __forceinline__ float row_sum16(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x121, 0xf, 0xf, false));  // row_ror:1
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x122, 0xf, 0xf, false));  // row_ror:2
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xf, 0xf, false));  // row_ror:4
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false));  // row_ror:8
    return x;
}
// score = O1 . ws + bs (model.py:208) for the tile's 16 rows (every lane of a row's four lane groups gets the row's score)
__device__ __forceinline__ float readout_score(const RTile& o1, const float* ws_lds, float bs, int g) {
    float sum = 0.f;
    const float* wsv = lds_here(ws_lds);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float4 w = *(const float4*)(wsv + 16 * m + 4 * g);
        sum = fmaf(o1.v[m][0], w.x, fmaf(o1.v[m][1], w.y, fmaf(o1.v[m][2], w.z, fmaf(o1.v[m][3], w.w, sum))));
    }
    sum += __shfl_xor(sum, 16); sum += __shfl_xor(sum, 32);
    return sum + bs;
}
// the MSE head of one tile (CF_LOSS above): go = dO1pre, and the tile's partial slab {dws, dbs, loss}
__device__ __forceinline__ void loss_head_tile(RTile& go, const RTile& o1, float score, const ConvFArgs& a, const float* ws_lds,
                                               int tile, int row, bool ok, int lane) {
    const int j = lane & 15, g = lane >> 4;
    const float dlt = ok ? score - a.targets[row] : 0.f;
    const float ds = 2.f * dlt * a.loss_scale;
    const float* wsv = lds_here(ws_lds);
    float* slab = a.head_partial + (size_t)tile * HEAD_SLAB;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float4 w = *(const float4*)(wsv + 16 * m + 4 * g);
        const float wv4[4] = {w.x, w.y, w.z, w.w};
        float cs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            go.v[m][i] = o1.v[m][i] > 0.f ? ds * wv4[i] : 0.f;
            cs[i] = row_sum16(ds * o1.v[m][i]);
        }
        if (j == 0) *(float4*)(slab + 16 * m + 4 * g) = make_float4(cs[0], cs[1], cs[2], cs[3]);
    }
    const float dbs = row_sum16(ds), ls = row_sum16(a.loss_scale * dlt * dlt);
    if (lane == 0) { slab[EMB] = dbs; slab[EMB + 1] = ls; }
}
template <int TAIL, int NT, bool KEEP_A = false>
__device__ __forceinline__ void convf_program(const ConvFArgs& a, float* smem, int bid, int nblk) {
    constexpr int NWAVES = NT / 64, NM = KEEP_A ? 5 : 4, NV = 5;   // [Wf W1a | M] W1b W2 Wt | [bf | u] b1 b2 bt ws
    constexpr int iW1B = NM - 3, iW2 = NM - 2, iWT = NM - 1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
    const int ntile = (a.n + 15) >> 4;
    float* vecs = smem + NM * 64 * LDW;
    int tile = bid + nblk * wv;   // tiles are dealt round-robin over the blocks, then over a block's waves
    struct Ops { RTile s_in, xr; int seg0, seg1; } cur, nxt;   // current tile / the wave's next tile (requested one tile ahead)
    auto load_ops = [&](Ops& d, int t) {
        const int row = t * 16 + j;
        const bool ok = row < a.n;
        rt_load(d.s_in, a.s, row, ok, g);
        rt_load(d.xr, a.xrecv, row, ok, g);
        d.seg0 = ok ? a.seg_ptr[row] : 0; d.seg1 = ok ? a.seg_ptr[row + 1] : 0;
    };
    load_ops(cur, tile);
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
    for (; tile < ntile; tile += nblk * NWAVES) {
        load_ops(nxt, tile + nblk * NWAVES);   // past the last tile: no loads
        const RTile &s_in = cur.s_in, &xr = cur.xr;
        const int row = tile * 16 + j;
        const bool ok = row < a.n;
        RTile t0, t1;
        auto add_deg = [&](RTile& t) {   // + deg * (bf | u)
            const float deg = (float)(cur.seg1 - cur.seg0);
            const float* bfv = lds_here(vecs);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float4 b = *(const float4*)(bfv + 16 * m + 4 * g);
                t.v[m][0] = fmaf(deg, b.x, t.v[m][0]); t.v[m][1] = fmaf(deg, b.y, t.v[m][1]);
                t.v[m][2] = fmaf(deg, b.z, t.v[m][2]); t.v[m][3] = fmaf(deg, b.w, t.v[m][3]);
            }
        };
        if (KEEP_A) {
            rt_mm<false>(t0, s_in, 1.f, smem, lane);
            add_deg(t0);
            rt_clear_unless(t0, ok);
            rt_store(t0, a.a_out, row, ok, g);
            rt_mm2<false>(t1, t0, s2, smem + 64 * LDW, xr, smem + 2 * 64 * LDW, lane);
        } else {
            rt_mm2<false>(t1, s_in, 1.f, smem, xr, smem + iW1B * 64 * LDW, lane);
            add_deg(t1);
        }
        rt_bias<true>(t1, vecs + 64, g);
        rt_clear_unless(t1, ok);
        rt_store(t1, a.z1, row, ok, g);
        rt_mask_store(t1, a.m_z1, row, ok, g);
        rt_mm<false>(t0, t1, 1.f, smem + iW2 * 64 * LDW, lane);
        rt_bias<true>(t0, vecs + 2 * 64, g);
        rt_clear_unless(t0, ok);
        rt_store(t0, a.out, row, ok, g);
        rt_mask_store(t0, a.m_out, row, ok, g);
        rt_mm<false>(t1, t0, 1.f, smem + iWT * 64 * LDW, lane);
        if (TAIL == CF_PROJ) {
            rt_bias<false>(t1, vecs + 3 * 64, g);
            rt_store(t1, a.t_out, row, ok, g);
        } else {
            rt_bias<true>(t1, vecs + 3 * 64, g);
            rt_clear_unless(t1, ok);
            rt_store(t1, a.t_out, row, ok, g);
            const float score = readout_score(t1, vecs + 4 * 64, bs, g);
            if (g == 0 && ok) a.scores[row] = score;
            if (TAIL == CF_LOSS) {
                RTile go;
                loss_head_tile(go, t1, score, a, vecs + 4 * 64, tile, row, ok, lane);
                rt_store(go, a.g_o1, row, ok, g);
            }
        }
        cur = nxt;
    }
}
template <int NWAVES, int TAIL, bool KEEP_A = false>
__global__ __launch_bounds__(NWAVES * 64) void k_conv_fwd(ConvFArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    convf_program<TAIL, NWAVES * 64, KEEP_A>(a, smem, blockIdx.x, gridDim.x);
}

// ---------------------------------------------------------------------------------------------------------------
// Program 3 (backward): receiver-side gradient of a PartialGraphConvolution, entered through the layer that consumed
// its output X' (the next projection, or the readout):
//   dX'  = (in W0^T) * (X' > 0)                                              -> g_out
//   dZ1  = (dX' W2^T) * (Z1 > 0)                                             -> g_z1
//   dx_r = dZ1 W1b^T                                                         -> g_xrecv
//   dS   = dZ1 M^T          (= s2 * (dZ1 W1a^T) Wf^T with the folded matrix of Program 2)   -> g_s
//   dP_recv = s1 * dS * N   (receiver-ordered half of the edge gradient, see k_edge_fwd)   -> g_precv
// ---------------------------------------------------------------------------------------------------------------
struct ConvBArgs {
    const float *in, *w0; const mask16* m_out; float* g_out;     // m_out, m_z1: the ReLU patterns of X' and Z1 (Program 2)
    const float* w2; const mask16* m_z1; float* g_z1;
    const float* w1b; float* g_xrecv;
    const float* mfuse; float* g_s;
    const float *s1, *nrows; float* g_precv;
    int n;
};
template <int NT>
__device__ __forceinline__ void convb_program(const ConvBArgs& a, float* smem, int bid, int nblk) {
    constexpr int NWAVES = NT / 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
    const int ntile = (a.n + 15) >> 4;
    int tile = bid + nblk * wv;   // tiles are dealt round-robin over the blocks, then over a block's waves
    struct Ops { RTile in, nr; unsigned m0, m1; } cur, nxt;   // current tile / the wave's next tile (requested one tile ahead)
    auto load_ops = [&](Ops& d, int t) {
        const int row = t * 16 + j;
        const bool ok = row < a.n;
        rt_load(d.in, a.in, row, ok, g);
        d.m0 = rt_mask_load(a.m_out, row, ok, g);
        d.m1 = rt_mask_load(a.m_z1, row, ok, g);
        rt_load(d.nr, a.nrows, row, ok, g);
    };
    load_ops(cur, tile);
    {
        const float* const w[4] = {a.w0, a.w2, a.w1b, a.mfuse};
        const float* const v[1] = {nullptr};
        stage_lds<4, 1, NT>(smem, w, v);
    }
    const float s1 = *a.s1;
    __syncthreads();
    for (; tile < ntile; tile += nblk * NWAVES) {
        load_ops(nxt, tile + nblk * NWAVES);   // past the last tile: no loads
        const RTile &in = cur.in, &nr = cur.nr;
        const int row = tile * 16 + j;
        const bool ok = row < a.n;
        RTile t0, t1;
        rt_mm<true>(t0, in, 1.f, smem, lane);
        rt_mask_apply(t0, cur.m0);
        rt_store(t0, a.g_out, row, ok, g);
        rt_mm<true>(t1, t0, 1.f, smem + 64 * LDW, lane);
        rt_mask_apply(t1, cur.m1);
        rt_store(t1, a.g_z1, row, ok, g);
        rt_mm<true>(t0, t1, 1.f, smem + 2 * 64 * LDW, lane);
        rt_store(t0, a.g_xrecv, row, ok, g);
        rt_mm<true>(t0, t1, 1.f, smem + 3 * 64 * LDW, lane);
        rt_store(t0, a.g_s, row, ok, g);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) t0.v[m][i] = s1 * t0.v[m][i] * nr.v[m][i];
        rt_store(t0, a.g_precv, row, ok, g);
        cur = nxt;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Program 2+3 (training turnaround): the LAST forward program (cut rows: receiver update of conv v->k, readout, MSE head)
// and the FIRST backward program (the same rows' receiver gradient, entered through the readout's Dense(64,relu)) are both
// row-local on the same tiles with nothing between them, so a training step runs them as ONE program: the activations the
// backward half masks with (X', Z1, O1) and its input gradient dO1pre never leave the registers, and the step has one launch
// (and one weight staging) less.  Every tensor the weight-gradient launch or the edge passes read is still stored.
// The four matrices serve both directions, so they are staged once, row-major: the forward products read them with
// rt_gemm<GEMM_FWD_RM>, the backward products as float4.
//   f: as convf_program<CF_LOSS>;  b: as convb_program with in = f.g_o1, w0 = f.wt, x_out = f.out, z1 = f.z1 (not re-read)
// ---------------------------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void convturn_program(const ConvFArgs& a, const ConvBArgs& b, float* smem, int bid, int nblk) {
    constexpr int NWAVES = NT / 64, NM = 4;   // M W1b W2 Wt | u b1 b2 bt ws
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
    const int ntile = (a.n + 15) >> 4;
    float* vecs = smem + NM * 64 * LDW;
    int tile = bid + nblk * wv;
    struct Ops { RTile s_in, xr, nr; int seg0, seg1; } cur, nxt;
    auto load_ops = [&](Ops& d, int t) {
        const int row = t * 16 + j;
        const bool ok = row < a.n;
        rt_load(d.s_in, a.s, row, ok, g);
        rt_load(d.xr, a.xrecv, row, ok, g);
        rt_load(d.nr, b.nrows, row, ok, g);
        d.seg0 = ok ? a.seg_ptr[row] : 0; d.seg1 = ok ? a.seg_ptr[row + 1] : 0;
    };
    load_ops(cur, tile);
    {
        const float* const w[4] = {a.mfuse, a.w1b, a.w2, a.wt};
        const float* const v[5] = {a.ufuse, a.b1, a.b2, a.bt, a.ws};
        stage_lds<4, 5, NT>(smem, w, v);
    }
    const float s1 = *b.s1, bs = *a.bs;
    float* const MF = smem; float* const W1B = smem + 64 * LDW;
    float* const W2 = smem + 2 * 64 * LDW; float* const WT = smem + 3 * 64 * LDW;
    __syncthreads();
    for (; tile < ntile; tile += nblk * NWAVES) {
        load_ops(nxt, tile + nblk * NWAVES);   // past the last tile: no loads
        const int row = tile * 16 + j;
        const bool ok = row < a.n;
        RTile t0, z1, xo, t1;
        // ---- forward half (convf_program<CF_LOSS>)
        rt_mm2<GEMM_FWD_RM>(z1, cur.s_in, 1.f, MF, cur.xr, W1B, lane);
        {   // + deg * u
            const float deg = (float)(cur.seg1 - cur.seg0);
            const float* bfv = lds_here(vecs);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float4 bb = *(const float4*)(bfv + 16 * m + 4 * g);
                z1.v[m][0] = fmaf(deg, bb.x, z1.v[m][0]); z1.v[m][1] = fmaf(deg, bb.y, z1.v[m][1]);
                z1.v[m][2] = fmaf(deg, bb.z, z1.v[m][2]); z1.v[m][3] = fmaf(deg, bb.w, z1.v[m][3]);
            }
        }
        rt_bias<true>(z1, vecs + 64, g);
        rt_clear_unless(z1, ok);
        rt_store(z1, a.z1, row, ok, g);
        rt_mm<GEMM_FWD_RM>(xo, z1, 1.f, W2, lane);
        rt_bias<true>(xo, vecs + 2 * 64, g);
        rt_clear_unless(xo, ok);
        rt_store(xo, a.out, row, ok, g);
        rt_mm<GEMM_FWD_RM>(t1, xo, 1.f, WT, lane);
        rt_bias<true>(t1, vecs + 3 * 64, g);
        rt_clear_unless(t1, ok);
        rt_store(t1, a.t_out, row, ok, g);
        const float score = readout_score(t1, vecs + 4 * 64, bs, g);
        if (g == 0 && ok) a.scores[row] = score;
        loss_head_tile(t0, t1, score, a, vecs + 4 * 64, tile, row, ok, lane);   // t0 = dO1pre
        rt_store(t0, a.g_o1, row, ok, g);
        // ---- backward half (convb_program)
        rt_mm<GEMM_BWD>(t1, t0, 1.f, WT, lane);
        rt_mask(t1, xo);
        rt_store(t1, b.g_out, row, ok, g);
        rt_mm<GEMM_BWD>(t0, t1, 1.f, W2, lane);
        rt_mask(t0, z1);
        rt_store(t0, b.g_z1, row, ok, g);
        rt_mm<GEMM_BWD>(t1, t0, 1.f, W1B, lane);
        rt_store(t1, b.g_xrecv, row, ok, g);
        rt_mm<GEMM_BWD>(t1, t0, 1.f, MF, lane);
        rt_store(t1, b.g_s, row, ok, g);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) t1.v[m][i] = s1 * t1.v[m][i] * cur.nr.v[m][i];
        rt_store(t1, b.g_precv, row, ok, g);
        cur = nxt;
    }
}
template <int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void k_conv_turn(ConvFArgs f, ConvBArgs b) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    convturn_program<NWAVES * 64>(f, b, smem, blockIdx.x, gridDim.x);
}

// ---------------------------------------------------------------------------------------------------------------
// Program 4 (backward): gradient reaching a raw embedding X (it fed one or two projections, and the concat of a
// convolution whose share `add` is already in place), then through the embedding's second layer:
//   dX  = (in_a Wa^T [+ in_b Wb^T] + add) * (X > 0)                          -> g_x   (may alias add)
//   dE1 = dX W2^T                      (masked later, inside the first layer's weight-gradient kernel)   -> g_e1
// ---------------------------------------------------------------------------------------------------------------
struct TailBArgs {
    const float *in_a, *wa, *in_b, *wb, *add; const mask16* m_x; float* g_x;    // m_x: the ReLU pattern of the raw embedding X
    const float* w2; float* g_e1;
    int n;
};
template <bool HAS_INB, int NT>
__device__ __forceinline__ void tailb_program(const TailBArgs& a, float* smem, int bid, int nblk) {
    constexpr int NWAVES = NT / 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
    const int ntile = (a.n + 15) >> 4;
    int tile = bid + nblk * wv;   // tiles are dealt round-robin over the blocks, then over a block's waves
    struct Ops { RTile ia, ib, ad; unsigned mk; } cur, nxt;   // current tile / the wave's next tile (requested one tile ahead)
    auto load_ops = [&](Ops& d, int t) {
        const int row = t * 16 + j;
        const bool ok = row < a.n;
        rt_load(d.ia, a.in_a, row, ok, g);
        if (HAS_INB) rt_load(d.ib, a.in_b, row, ok, g);
        rt_load(d.ad, a.add, row, ok, g);
        d.mk = rt_mask_load(a.m_x, row, ok, g);
    };
    load_ops(cur, tile);
    const float* const v[1] = {nullptr};
    if (HAS_INB) { const float* const w[3] = {a.wa, a.w2, a.wb}; stage_lds<3, 1, NT>(smem, w, v); }
    else { const float* const w[2] = {a.wa, a.w2}; stage_lds<2, 1, NT>(smem, w, v); }
    __syncthreads();
    for (; tile < ntile; tile += nblk * NWAVES) {
        load_ops(nxt, tile + nblk * NWAVES);   // past the last tile: no loads
        const RTile &ia = cur.ia, &ib = cur.ib, &ad = cur.ad;
        const int row = tile * 16 + j;
        const bool ok = row < a.n;
        RTile t0, t1;
        if (HAS_INB) rt_mm2<true>(t0, ia, 1.f, smem, ib, smem + 2 * 64 * LDW, lane);
        else rt_mm<true>(t0, ia, 1.f, smem, lane);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) t0.v[m][i] += ad.v[m][i];
        rt_mask_apply(t0, cur.mk);
        rt_store(t0, a.g_x, row, ok, g);
        rt_mm<true>(t1, t0, 1.f, smem + 64 * LDW, lane);
        rt_store(t1, a.g_e1, row, ok, g);
        cur = nxt;
    }
}

// backward launches: a receiver-gradient program, optionally with a tail program over another row set beside it ...
struct ConvBGroupArgs { int blk0[3]; ConvBArgs cb; TailBArgs tail; };
template <int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void k_conv_bwd(ConvBGroupArgs m) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;
    if (b < m.blk0[1]) convb_program<NWAVES * 64>(m.cb, smem, b, m.blk0[1]);
    else tailb_program<false, NWAVES * 64>(m.tail, smem, b - m.blk0[1], m.blk0[2] - m.blk0[1]);
}
// ... and the two last tails together: `a` sums two projections (the raw variable embedding fed two convolutions)
struct TailGroupArgs { int blk0[3]; TailBArgs a, b; };
template <int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void k_tail_bwd(TailGroupArgs m) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;
    if (b < m.blk0[1]) tailb_program<true, NWAVES * 64>(m.a, smem, b, m.blk0[1]);
    else tailb_program<false, NWAVES * 64>(m.b, smem, b - m.blk0[1], m.blk0[2] - m.blk0[1]);
}
