// gcnn_capi.hip -- C ABI (include/gcnn_hip.h) and host-side orchestration of the gfx950 (MI355X / CDNA4) kernels for the
// bipartite GCNN hot path.  Kernels: k_rows.hpp (fused row programs), k_edge.hpp (edge passes, scatter-sum), k_wgrad.hpp
// (weight gradients), k_linear.hpp (standalone GEMM), k_misc.hpp (heads, Adam, PreNorm statistics, graph plan).
//
// Reference semantics (all cites into /root/reference): GCNN.call model.py:257-300, PartialGraphConvolution.call
// model.py:533-575, PreNormLayer.call model.py:365-382, loss/step model_trainer.py:266-273.
//
// Design: every node tensor is a row-major [N,64] fp32 matrix (256-B rows).  The per-edge
// Dense(64->64) of the reference (model.py:499-500) is hoisted past the scatter-sum
//   sum_e (H_e W_f + b_f) = (sum_e H_e) W_f + deg_r b_f
// so the edge pass only gathers rows, applies ReLU and accumulates in registers (atomic-free segmented sum over
// receiver-sorted CSR); all 64x64 products run on the fp32 MFMA (v_mfma_f32_16x16x4_f32; weights in LDS for the row
// programs, operands straight from global memory for the weight gradients).  A training step is 15 launches on one stream.
// Wavefront = 64 lanes everywhere.  No atomics on floats anywhere: every sum has a fixed order => bitwise
// reproducible results.

#include "gcnn_common.hpp"
#include "k_rows.hpp"
#include "k_edge.hpp"
#include "k_linear.hpp"
#include "k_misc.hpp"
#include "k_wgrad.hpp"
#include "k_infer.hpp"

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)
#define LAUNCHCHK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
// Tuning knobs (tools/README.md): an experiment build (-DGCNN_TUNING, tools/mklib.sh) reads them from the environment once per
// process; the product library is built without and every knob is its compile-time default -- no getenv on the launch path.
#ifdef GCNN_TUNING
#define GCNN_KNOB(name, dflt) ([] { static const int v_ = getenv(name) ? atoi(getenv(name)) : (dflt); return v_; }())
#else
#define GCNN_KNOB(name, dflt) (dflt)
#endif
// hipFuncSetAttribute acts on the current device's copy of a kernel: remember it per device, not per process
#define GCNN_MAX_DEVICES 64
struct PerDeviceOnce {
    bool done[GCNN_MAX_DEVICES] = {};
    bool first() {   // true the first time on the current device (always true for a device id outside the table)
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= GCNN_MAX_DEVICES) return true;
        if (done[d]) return false;
        done[d] = true;
        return true;
    }
};
static int device_cus() {   // compute units of the current device (MI355X: 256); cached per device
    static int cus[GCNN_MAX_DEVICES] = {};
    int d = 0, v = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= GCNN_MAX_DEVICES) return 256;
    if (cus[d] == 0) cus[d] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d) == hipSuccess && v > 0) ? v : 256;
    return cus[d];
}
// ---- per-launch timing (gcnn_profile_begin / gcnn_profile_end): bench.py's roofline_step block -----------------------------
// While enabled every kernel launch of the library is bracketed by two HIP events on the launch stream.  A measuring aid for
// one thread; off by default and then a single predictable branch per launch.
#define GCNN_PROF_MAX 512
struct ProfRec { const char* name; hipEvent_t e0, e1; };
// `dev`: the device the cached events belong to (events are per device; gcnn_profile_begin drops them when the current device
// is another one).  One profiling session at a time, on one device.
static struct { bool on; int n; int dev; ProfRec rec[GCNN_PROF_MAX]; } g_prof;
struct ProfScope {
    hipStream_t st; bool live;
    ProfScope(const char* name, hipStream_t s) : st(s), live(g_prof.on && g_prof.n < GCNN_PROF_MAX) {
        if (!live) return;
        ProfRec& r = g_prof.rec[g_prof.n];
        r.name = name;
        if (!r.e0) {   // both events or none: a half-made pair would be reused for ever
            hipEvent_t a = nullptr, b = nullptr;
            if (hipEventCreate(&a) != hipSuccess) { live = false; return; }
            if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); live = false; return; }
            r.e0 = a; r.e1 = b;
        }
        (void)hipEventRecord(r.e0, st);
    }
    ~ProfScope() { if (live) { (void)hipEventRecord(g_prof.rec[g_prof.n].e1, st); ++g_prof.n; } }
};

static const int LIN_SMEM = (2 * 64 * LDW + 4 * 32 * LDW) * (int)sizeof(float);  // 69,632 B
static const int MAX_GRID = 2048;
// Edge passes: fine-grained blocks balance better than a grid sized to what is resident (segments differ in length; measured
// 2048 / 4096 / 8192 blocks: 0.308 / 0.301 / 0.298 ms per setcov-500 x 32 step)
static const int EDGE_MAX_GRID = 8192;

static int launch_linear(bool transb, const LinArgs& a, hipStream_t st) {
    if (a.n <= 0) return 0;
    static PerDeviceOnce attr;
    if (attr.first()) {  // 68 KB of dynamic LDS per block (gfx950 has 160 KB per CU)
        HIPCHK(hipFuncSetAttribute((const void*)k_linear<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LIN_SMEM));
        HIPCHK(hipFuncSetAttribute((const void*)k_linear<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LIN_SMEM));
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

// 16*SLOTS lanes serve one segment.  Measured on setcov-500 x 32 (tools/micro/bench_edge.hip): mean degree 50 runs best with a
// whole wave per segment, mean degree 25 with two to four segments per wave.
static inline int edge_slots(int n_own, int n_edges) {
    const double avg = (double)n_edges / (double)std::max(n_own, 1);
    return avg >= GCNN_KNOB("GCNN_SLOTS4_DEG", 40) ? 4 : (avg >= GCNN_KNOB("GCNN_SLOTS2_DEG", 12) ? 2 : 1);
}
// Segments longer than edge_long_threshold(slots) are left to a second launch that gives each a whole wave (k_edge.hpp).
// max_deg = longest segment of the list if the caller knows it (gcnn_graph.*_max_deg), 0 = unknown: then the finder always runs.
static inline bool edge_needs_long_pass(int slots, int max_deg) {
    return slots < 4 && (max_deg <= 0 || max_deg > edge_long_threshold(slots));
}
// blocks that find and serve them, riding at the front of the pass's launch: a multiple of 8 (the other blocks' XCD remap)
static inline int edge_long_grid(int n_own) { return (std::max(1, std::min(cdiv(n_own, 4), MAX_GRID)) + 7) & ~7; }
static_assert(GCNN_EDGE_DW_PARTS >= EDGE_MAX_GRID + MAX_GRID + 8, "dw partial rows: main blocks + long-segment blocks");
// forward (owner = receiver); `count` also emits the N rows (active edges per receiver and channel) for the backward pass
static int launch_plan_place(const IplanArgs& ia, hipStream_t st);
static int launch_edge_fwd(const EdgeArgs& a, int n_edges, int max_deg, bool count, hipStream_t st, const IplanArgs* plan = nullptr) {
    if (plan) {   // single-state inference: the plan's place step rides in this launch as extra blocks (k_infer_s2)
        const bool blockseg = a.n_own <= 1024 && n_edges >= 48ll * a.n_own;
        if (!count && a.n_own <= 16384 && plan->n_vars <= IPLAN_FUSE_MAX_VARS) {
            const int edge_blocks = a.n_own <= 0 ? 0 : (blockseg ? a.n_own : std::min(cdiv(a.n_own, 4), MAX_GRID));
            const int place_blocks = std::max(1, std::min(cdiv(plan->s[0].n_edges, 256), 256));   // one returning atomic per thread
            static PerDeviceOnce attr;
            if (attr.first()) {
                HIPCHK(hipFuncSetAttribute((const void*)k_infer_s2<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (IPLAN_MAX_VARS + 1)));
                HIPCHK(hipFuncSetAttribute((const void*)k_infer_s2<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (IPLAN_MAX_VARS + 1)));
            }
            ProfScope prof("k_infer_s2 (conv v->c edge pass + plan: place)", st);
            const size_t smem = 4 * (size_t)(plan->n_vars + 1);
            if (blockseg) hipLaunchKernelGGL(k_infer_s2<true>, dim3(edge_blocks + place_blocks), dim3(256), smem, st, a, *plan, edge_blocks);
            else hipLaunchKernelGGL(k_infer_s2<false>, dim3(edge_blocks + place_blocks), dim3(256), smem, st, a, *plan, edge_blocks);
            LAUNCHCHK();
            return 0;
        }
        int rc = launch_plan_place(*plan, st);   // a graph too large for the fused form: the step as a launch of its own
        if (rc) return rc;
    }
    if (a.n_own <= 0) return 0;
    if (count && !a.cnt_rows) return GCNN_E_BADARG;
    // inference on a small graph (one sampled state): latency, not lane efficiency, decides -- a whole wave per segment needs no
    // long-segment launch and finishes a hub row in a quarter of the gather rounds
    const int slots = (!count && a.n_own <= 16384) ? 4 : edge_slots(a.n_own, n_edges);
    // a few long segments: a block each (k_edge_fwd_block) -- the cut rows of one sampled state or of a stacked batch (1,893
    // segments of ~105 edges at setcov x 32: a wave per segment walks seven dependent gather rounds on under two waves per SIMD,
    // four waves per segment two).  The same rule with and without counts: training and inference forward add in the same order.
    if (a.n_own <= 4096 && n_edges >= 48ll * a.n_own) {
        ProfScope prof(count ? "k_edge_fwd_block<count>" : "k_edge_fwd_block", st);
        if (count) hipLaunchKernelGGL(k_edge_fwd_block<true>, dim3(a.n_own), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(k_edge_fwd_block<false>, dim3(a.n_own), dim3(256), 0, st, a);
        LAUNCHCHK();
        return 0;
    }
    const int grid = std::min(cdiv(cdiv(a.n_own, 4 / slots), 4), EDGE_MAX_GRID);
    const int lb = edge_needs_long_pass(slots, max_deg) ? edge_long_grid(a.n_own) : 0;   // long-segment blocks, first in the grid
    {
        ProfScope prof(count ? (lb ? "k_edge_fwd<count> + long segments" : "k_edge_fwd<count>") : "k_edge_fwd", st);
#define EDGE_LAUNCH(S, V)                                                                                              \
        do {                                                                                                            \
            if (lb) hipLaunchKernelGGL((k_edge_fwd<S, V, (S < 4)>), dim3(grid + lb), dim3(256), 0, st, a, lb);          \
            else hipLaunchKernelGGL((k_edge_fwd<S, V, false>), dim3(grid), dim3(256), 0, st, a, 0);                     \
        } while (0)
        if (count) { if (slots == 4) EDGE_LAUNCH(4, true); else if (slots == 2) EDGE_LAUNCH(2, true); else EDGE_LAUNCH(1, true); }
        else { if (slots == 4) EDGE_LAUNCH(4, false); else if (slots == 2) EDGE_LAUNCH(2, false); else EDGE_LAUNCH(1, false); }
#undef EDGE_LAUNCH
        LAUNCHCHK();
    }
    return 0;
}
// backward, sender-ordered (owner = sender)
// *n_parts = rows of a.dw_partial written (one per block of the launch(es)): what k_reduce has to sum into d w_edge
static int launch_edge_bwd_send(EdgeArgs a, int n_edges, int max_deg, int* n_parts, hipStream_t st) {
    *n_parts = 0;
    if (a.n_own <= 0) return 0;
    const int slots = edge_slots(a.n_own, n_edges);
    const int grid = std::min(cdiv(cdiv(a.n_own, 4 / slots), 4), EDGE_MAX_GRID);
    const int lb = edge_needs_long_pass(slots, max_deg) ? edge_long_grid(a.n_own) : 0;
    *n_parts = grid + lb;
    ProfScope prof(lb ? "k_edge_bwd_send + long segments" : "k_edge_bwd_send", st);
    if (slots == 4) hipLaunchKernelGGL((k_edge_bwd_send<4, false>), dim3(grid), dim3(256), 0, st, a, 0);
    else if (slots == 2 && lb) hipLaunchKernelGGL((k_edge_bwd_send<2, true>), dim3(grid + lb), dim3(256), 0, st, a, lb);
    else if (slots == 2) hipLaunchKernelGGL((k_edge_bwd_send<2, false>), dim3(grid), dim3(256), 0, st, a, 0);
    else if (lb) hipLaunchKernelGGL((k_edge_bwd_send<1, true>), dim3(grid + lb), dim3(256), 0, st, a, lb);
    else hipLaunchKernelGGL((k_edge_bwd_send<1, false>), dim3(grid), dim3(256), 0, st, a, 0);
    LAUNCHCHK();
    return 0;
}

static int launch_plan_place(const IplanArgs& ia, hipStream_t st) {
    static PerDeviceOnce attr;
    if (attr.first()) HIPCHK(hipFuncSetAttribute((const void*)k_iplan_place, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (IPLAN_MAX_VARS + 1)));
    ProfScope prof("k_iplan_place", st);
    hipLaunchKernelGGL(k_iplan_place, dim3(std::max(1, std::min(cdiv(ia.s[0].n_edges, 1024), 64))), dim3(1024), 4 * (size_t)(ia.n_vars + 1), st, ia);
    LAUNCHCHK();
    return 0;
}

// ---- workspace carving ------------------------------------------------------------------------------------------
struct Acts {
    float *E1c, *Xc, *PL1, *S1, *A1, *Z1c, *Xc2, *PL2;          // C rows
    float *E1v, *Xv, *PR1, *PR2, *S2, *A2, *Z1v, *Xv2, *PR3;    // V rows
    float *E1k, *Xk, *PL3, *S3, *A3, *Z1k, *Xk2, *O1;           // K rows
};
struct Masks {   // ReLU patterns, 8 B per row (k_rows.hpp, mask16): of the embedding layers and of each convolution's Z1 and X'
    mask16 *E1c, *Xc, *Z1c, *Xc2, *E1v, *Xv, *Z1v, *Xv2, *E1k, *Xk, *Z1k, *Xk2;
};
struct Work {
    Acts a, g;            // activations and their gradients
    Masks m;
    float* partial;       // weight-gradient slabs
    float* dwp[3];        // per-block partials of d w_edge, one [GCNN_EDGE_DW_PARTS,64] array per convolution
    float* dwp2[3];       // ... and their pre-reduction to [GCNN_EDGE_DW_PARTS / DW_CHUNK, 64] (k_wgrad's third block type)
    float* nrow[3];          // per receiver and channel: number of active edges
    float* fuse[3];          // folded weights of each convolution: M [64,64] | u [64], and copies of Wf, W1a, bf  (fuse_weights)
    float* score_partial; int score_nblk;
    double* stats; int* stat_ids;   // pretraining: per-block partial sums; explicit left ids of an edge set
    size_t total;
};
static inline size_t al4(size_t x) { return (x + 3) & ~(size_t)3; }

// weight-gradient slabs: one per block of the k_wgrad launch, which is sized to one resident round (place_wg) -- a constant bound
#define WG_MAX_SLABS 1024
static size_t wg_slabs(const gcnn_dims*) { return WG_MAX_SLABS; }

static void carve(const gcnn_dims* d, float* base, Work* w) {
    const size_t C = d->n_cons, V = d->n_vars, K = d->n_cuts;
    size_t off = 0;
    auto take = [&](size_t n) { float* p = base ? base + off : nullptr; off += al4(n); return p; };
    for (int pass = 0; pass < 2; ++pass) {
        Acts* t = pass ? &w->g : &w->a;
        float** pc[] = {&t->E1c, &t->Xc, &t->PL1, &t->S1, &t->A1, &t->Z1c, &t->Xc2, &t->PL2};
        float** pv[] = {&t->E1v, &t->Xv, &t->PR1, &t->PR2, &t->S2, &t->A2, &t->Z1v, &t->Xv2, &t->PR3};
        float** pk[] = {&t->E1k, &t->Xk, &t->PL3, &t->S3, &t->A3, &t->Z1k, &t->Xk2, &t->O1};
        // (no gradient of A exists since the layers around it are folded, k_rows.hpp Program 2; A itself stays: PreNorm fitting
        // materialises it.  No E1 activation either: the weight-gradient launch recomputes it, k_wgrad.hpp EXTRA == 3; its gradient stays)
        for (auto p : pc) *p = ((pass && p == &t->A1) || (!pass && p == &t->E1c)) ? nullptr : take(C * EMB);
        for (auto p : pv) *p = ((pass && p == &t->A2) || (!pass && p == &t->E1v)) ? nullptr : take(V * EMB);
        for (auto p : pk) *p = ((pass && p == &t->A3) || (!pass && p == &t->E1k)) ? nullptr : take(K * EMB);
    }
    {
        mask16** mc[] = {&w->m.E1c, &w->m.Xc, &w->m.Z1c, &w->m.Xc2};
        mask16** mv[] = {&w->m.E1v, &w->m.Xv, &w->m.Z1v, &w->m.Xv2};
        mask16** mk[] = {&w->m.E1k, &w->m.Xk, &w->m.Z1k, &w->m.Xk2};
        for (auto q : mc) *q = (mask16*)take(2 * C);      // 8 B = two floats per row
        for (auto q : mv) *q = (mask16*)take(2 * V);
        for (auto q : mk) *q = (mask16*)take(2 * K);
    }
    w->partial = take(wg_slabs(d) * WG_SLAB);
    const size_t nrecv[3] = {C, V, K};
    for (int i = 0; i < 3; ++i) {
        w->dwp[i] = take((size_t)GCNN_EDGE_DW_PARTS * EMB); w->dwp2[i] = take((size_t)(GCNN_EDGE_DW_PARTS / DW_CHUNK) * EMB);
        w->nrow[i] = take(nrecv[i] * EMB);
        w->fuse[i] = take(FUSE_FLOATS);
    }
    w->stats = (double*)take(2 * (size_t)(ST_MAX_BLOCKS * ST_MAX_UNITS + 2 * ST_MAX_UNITS));
    w->stat_ids = (int*)take((size_t)std::max(d->n_cons_edges, d->n_cut_edges));
    w->score_nblk = cdiv(d->n_cuts, SB_ROWS);
    w->score_partial = take((size_t)cdiv(d->n_cuts, 16) * HEAD_SLAB);   // per 64-cut block (k_score_bwd) or per tile (CF_LOSS)
    w->total = off;
}

extern "C" {

int gcnn_abi_version(void) { return 11; }

int gcnn_profile_begin(void) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) return GCNN_E_HIP;
    if (d != g_prof.dev)   // the cached events were made on another device: they cannot be recorded on this one's streams
        for (ProfRec& r : g_prof.rec) {
            if (r.e0) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
            r.e0 = r.e1 = nullptr;
        }
    g_prof.dev = d; g_prof.n = 0; g_prof.on = true;
    return 0;
}
int gcnn_profile_end(int32_t capacity, const char** names, float* ms) {
    g_prof.on = false;
    const int n = g_prof.n;
    for (int i = 0; i < n; ++i) {
        float t = 0.f;   // the return value is a count, so a HIP failure is reported as GCNN_E_HIP
        if (hipEventSynchronize(g_prof.rec[i].e1) != hipSuccess || hipEventElapsedTime(&t, g_prof.rec[i].e0, g_prof.rec[i].e1) != hipSuccess) {
            g_prof.n = 0;
            return GCNN_E_HIP;
        }
        if (i < capacity) { if (names) names[i] = g_prof.rec[i].name; if (ms) ms[i] = t; }
    }
    g_prof.n = 0;
    return n;
}
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
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const int*)nullptr, (int*)nullptr, (const int*)nullptr, (int*)nullptr,
                                    (unsigned)(n > 0 ? n : 1), 0u, 32u, (hipStream_t)0);
    return (bytes + 255) & ~(size_t)255;
}
size_t gcnn_graph_temp_bytes(int32_t n_edges) {
    const size_t e = ((size_t)(n_edges > 0 ? n_edges : 1) * sizeof(int) + 255) & ~(size_t)255;
    return sort_temp_bytes(n_edges) + 3 * e;  // radix-sort temp + iota + sorted keys + one permutation
}

// One launch gathers every array of a mini-batch out of a device-resident sample store (see k_collate)
int gcnn_collate(const gcnn_collate_job* jobs, int32_t n_jobs, const int64_t* src_off, const int64_t* dst_off,
                 int32_t batch, int64_t max_words, void* stream) {
    if (n_jobs < 0 || n_jobs > COLLATE_MAX_JOBS || batch < 0 || max_words < 0) return GCNN_E_BADARG;
    if (n_jobs == 0 || batch == 0) return 0;
    if (!jobs || !src_off || !dst_off) return GCNN_E_BADARG;
    CollateArgs a;
    for (int i = 0; i < n_jobs; ++i) {
        const gcnn_collate_job& j = jobs[i];
        if (j.width < 1 || j.unit_kind < 0 || (j.is_ptr && (j.width != 1 || j.add_kind < 0))) return GCNN_E_BADARG;
        a.job[i] = CollateJob{(const int*)j.src, (int*)j.dst, j.unit_kind, j.width, j.add_kind, j.is_ptr};
    }
    a.src_off = (const long long*)src_off;
    a.dst_off = (const long long*)dst_off;
    a.batch = batch;
    const int bx = std::min(batch * COLLATE_CHUNKS, 2048);   // one block per (sample, chunk); larger batches loop
    hipLaunchKernelGGL(k_collate, dim3(bx, n_jobs), dim3(256), 0, (hipStream_t)stream, a);
    LAUNCHCHK();
    return 0;
}

// flags[0] != 0: an index is out of range; flags[1] != 0: the list is NOT sorted by left id (ties in any order)
int gcnn_graph_check(const int32_t* edge_inds, int32_t n_edges, int32_t n_left, int32_t n_var, int32_t* flags,
                     void* stream) {
    if (n_edges < 0 || !flags || (n_edges > 0 && !edge_inds)) return GCNN_E_BADARG;
    HIPCHK(hipMemsetAsync(flags, 0, 2 * sizeof(int), (hipStream_t)stream));
    if (n_edges == 0) return 0;
    hipLaunchKernelGGL(k_check_edges, dim3(std::min(cdiv(n_edges, 256), 1024)), dim3(256), 0, (hipStream_t)stream,
                       edge_inds, n_edges, n_left, n_var, flags);
    LAUNCHCHK();
    return 0;
}

int gcnn_graph_build(const int32_t* edge_inds, const float* edge_feats, int32_t n_edges, int32_t n_left, int32_t n_var,
                     int32_t left_sorted, int32_t* l_ptr, int32_t* l_oth, float* l_coef, int32_t* v_ptr, int32_t* v_oth,
                     float* v_coef, int32_t* l_perm, void* temp, size_t temp_bytes, void* stream) {
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
    size_t sort_bytes = sort_temp_bytes(n_edges);
    char* t = (char*)temp;
    void* sort_tmp = t;
    int* iota = (int*)(t + sort_bytes);
    int* keys = (int*)(t + sort_bytes + e);
    int* perm = (int*)(t + sort_bytes + 2 * e);
    const int grid = std::min(cdiv(n_edges + 1, 256), 4096);
    const int* left = edge_inds;
    const int* var = edge_inds + n_edges;
    hipLaunchKernelGGL(k_iota, dim3(grid), dim3(256), 0, st, iota, n_edges);
    LAUNCHCHK();
    for (int side = 0; side < 2; ++side) {
        const int* key_in = side == 0 ? left : var;
        const int nseg = side == 0 ? n_left : n_var;
        if (side == 0 && left_sorted) {
            // the reference emits (row, col)-sorted COO (utils.py:102-104): the by-left order is the input order
            hipLaunchKernelGGL(k_seg_offsets, dim3(grid), dim3(256), 0, st, left, n_edges, nseg, l_ptr);
            LAUNCHCHK();
            HIPCHK(hipMemcpyAsync(l_oth, var, (size_t)n_edges * sizeof(int), hipMemcpyDeviceToDevice, st));
            HIPCHK(hipMemcpyAsync(l_coef, edge_feats, (size_t)n_edges * sizeof(float), hipMemcpyDeviceToDevice, st));
            if (l_perm) HIPCHK(hipMemcpyAsync(l_perm, iota, (size_t)n_edges * sizeof(int), hipMemcpyDeviceToDevice, st));
            continue;
        }
        unsigned bits = 1;
        while ((1ll << bits) < (long long)nseg + 1 && bits < 31) ++bits;
        // stable LSD radix sort of (node id, input position): ties keep the input order => every sum has a fixed order
        HIPCHK(rocprim::radix_sort_pairs(sort_tmp, sort_bytes, key_in, keys, (const int*)iota, perm, (unsigned)n_edges, 0u, bits, st));
        hipLaunchKernelGGL(k_seg_offsets, dim3(grid), dim3(256), 0, st, keys, n_edges, nseg, side == 0 ? l_ptr : v_ptr);
        LAUNCHCHK();
        hipLaunchKernelGGL(k_gather_edges, dim3(grid), dim3(256), 0, st, perm, side == 0 ? var : left, edge_feats,
                           n_edges, side == 0 ? l_oth : v_oth, side == 0 ? l_coef : v_coef);
        LAUNCHCHK();
        if (side == 0 && l_perm) HIPCHK(hipMemcpyAsync(l_perm, perm, (size_t)n_edges * sizeof(int), hipMemcpyDeviceToDevice, st));
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
                       const float* e_scale, const float* s1, float* s_out, float* n_rows, int32_t max_degree, void* stream) {
    if (n_recv < 0 || n_edges < 0) return GCNN_E_BADARG;
    if (n_recv > 0 && (!seg_ptr || !p_recv || !w_edge || !e_shift || !e_scale || !s1 || !s_out)) return GCNN_E_BADARG;
    if (n_edges > 0 && (!oth || !coef || !p_oth)) return GCNN_E_BADARG;
    EdgeArgs e; memset(&e, 0, sizeof(e));
    e.seg_ptr = seg_ptr; e.oth = oth; e.coef = coef; e.p_own = p_recv; e.p_oth = p_oth; e.w_edge = w_edge;
    e.e_shift = e_shift; e.e_scale = e_scale; e.s1 = s1; e.out = s_out; e.cnt_rows = n_rows; e.n_own = n_recv;
    return launch_edge_fwd(e, n_edges, max_degree, n_rows != nullptr, (hipStream_t)stream);
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
int gcnn_conv_edge_bwd_send(const int32_t* seg_ptr, const int32_t* oth, const float* coef, int32_t n_send, int32_t n_edges,
                            const float* p_send, const float* p_recv, const float* w_edge, const float* e_shift,
                            const float* e_scale, const float* s1, const float* d_s, float* d_p_send, float* dw_partial,
                            int32_t* n_parts, int32_t max_degree, void* stream) {
    if (n_send < 0 || n_edges < 0 || !n_parts) return GCNN_E_BADARG;
    if (n_send > 0 && (!seg_ptr || !p_send || !w_edge || !s1 || !e_shift || !e_scale || !d_p_send || !dw_partial)) return GCNN_E_BADARG;
    if (n_edges > 0 && (!oth || !coef || !p_recv || !d_s)) return GCNN_E_BADARG;
    EdgeArgs e; memset(&e, 0, sizeof(e));
    e.seg_ptr = seg_ptr; e.oth = oth; e.coef = coef; e.p_own = p_send; e.p_oth = p_recv; e.w_edge = w_edge; e.s1 = s1;
    e.e_shift = e_shift; e.e_scale = e_scale; e.d_s = d_s; e.out = d_p_send; e.dw_partial = dw_partial; e.n_own = n_send;
    int parts = 0;
    const int rc = launch_edge_bwd_send(e, n_edges, max_degree, &parts, (hipStream_t)stream);
    *n_parts = parts;
    return rc;
}

}  // extern "C"

// ---- row programs: host-side launchers ---------------------------------------------------------------------------------
// Blocks per program of a grouped launch.  One block per CU (the staged weights fill most of the LDS); its 4 or 8 waves share
// them (8 = two per SIMD, covering each other's loads and epilogues, once there is more than one tile per SIMD).  Tiles are
// dealt round-robin over a program's blocks and then over a block's waves (k_rows.hpp), so every SIMD of every block gets
// the same number of tiles +-1.  Row sets too small for 256 full blocks still spread over all CUs; otherwise 256 blocks
// split by work (tiles x stages).  (16 waves per block, four per SIMD, were measured and rejected: profiles/README.md.)
static int rows_blocks(const int* n, const int* nstage, int ngroups, int* blk0, int cap = 256) {
    int ntile[3], total = 0;
    for (int i = 0; i < ngroups; ++i) { ntile[i] = n[i] > 0 ? cdiv(n[i], 16) : 0; total += ntile[i]; }
    const int forced = GCNN_KNOB("GCNN_ROWS_WAVES", 0);
    const int nwaves = (forced == 4 || forced == 8) ? forced : (total > 1024 ? 8 : 4);
    int want[3], sum_want = 0;
    long long work[3], sum_work = 0;
    for (int i = 0; i < ngroups; ++i) {
        want[i] = cdiv(ntile[i], nwaves); sum_want += want[i];
        work[i] = (long long)ntile[i] * nstage[i]; sum_work += work[i];
    }
    // fewer than 256 blocks of this size: spread the tiles over all CUs instead (fewer tiles per block, idle waves are free)
    const bool spread = sum_want < cap && total > 1024;
    int nb[3];
    for (int i = 0; i < ngroups; ++i) {
        nb[i] = want[i];
        if ((sum_want > cap || spread) && nb[i] > 0)
            nb[i] = std::max(1, std::min(spread ? ntile[i] : want[i], (int)(((long long)cap * work[i] + sum_work - 1) / sum_work)));
    }
    // rounding each share up can leave 257 or 258 blocks for 256 CUs, and a block of a one-block-per-CU launch that has to wait
    // for a free CU adds its whole run time to the launch: take the excess from the largest group
    if (sum_want > cap || spread)
        for (int total = nb[0] + (ngroups > 1 ? nb[1] : 0) + (ngroups > 2 ? nb[2] : 0); total > cap; --total) {
            int big = 0;
            for (int i = 1; i < ngroups; ++i) if (nb[i] > nb[big]) big = i;
            if (nb[big] <= 1) break;
            --nb[big];
        }
    blk0[0] = 0;
    for (int i = 0; i < ngroups; ++i) blk0[i + 1] = blk0[i] + nb[i];
    return nwaves;
}
// Few tiles (the cut rows of a training batch, every row set of a single-state inference call): one block of four waves per
// tile, the waves sharing the tile's products (k_rows_split.hpp).  Blocks = tiles.
static bool rows_split(const int* n, int ngroups, int* blk0) {
    const int max_tiles = GCNN_KNOB("GCNN_SPLIT_MAX_TILES", 256);
    int total = 0;
    blk0[0] = 0;
    for (int i = 0; i < ngroups; ++i) { const int t = n[i] > 0 ? cdiv(n[i], 16) : 0; total += t; blk0[i + 1] = blk0[i] + t; }
    return total <= max_tiles;
}
#define SPLIT_LAUNCH(NAME, KERNEL, GRID, SMEM, ST, ...)                                                                \
    do {                                                                                                                \
        ProfScope prof(NAME, ST);                                                                                       \
        static PerDeviceOnce attr;                                                                                      \
        if (attr.first()) HIPCHK(hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024)); \
        hipLaunchKernelGGL(KERNEL, dim3(GRID), dim3(256), SMEM, ST, __VA_ARGS__);                                       \
        LAUNCHCHK();                                                                                                    \
    } while (0)
#define ROWS_LAUNCH(NAME, KERNEL8, KERNEL4, NWAVES, GRID, SMEM, ST, ...)                                                \
    do {                                                                                                                \
        ProfScope prof(NAME, ST);                                                                                       \
        static PerDeviceOnce attr;                                                                                      \
        if (attr.first()) {                                                                                             \
            HIPCHK(hipFuncSetAttribute((const void*)KERNEL8, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));  \
            HIPCHK(hipFuncSetAttribute((const void*)KERNEL4, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));  \
        }                                                                                                               \
        if ((NWAVES) == 8) hipLaunchKernelGGL(KERNEL8, dim3(GRID), dim3(512), SMEM, ST, __VA_ARGS__);                   \
        else hipLaunchKernelGGL(KERNEL4, dim3(GRID), dim3(256), SMEM, ST, __VA_ARGS__);                                 \
        LAUNCHCHK();                                                                                                    \
    } while (0)

// `plan` (single-state inference, gcnn_infer): the plan's count step rides in this launch as extra blocks
static int launch_embed_fwd(EmbGroupArgs& m, IplanArgs* plan, hipStream_t st) {
    const int n[3] = {m.v.n, m.c.n, m.k.n}, ns[3] = {4, 3, 3};
    if (rows_split(n, 3, m.blk0)) {
        const size_t smem = EMB_SPLIT_LDS_FLOATS * sizeof(float);
        if (plan) {
            plan->blocks0 = std::min(cdiv(plan->s[0].n_edges + 1, 256), 32);
            plan->blocks1 = std::min(cdiv(plan->s[1].n_edges + 1, 256), 8);
            SPLIT_LAUNCH("k_infer_s1 (embeddings + plan: count)", (k_infer_s1<4, true>), m.blk0[3] + 3 + plan->blocks0 + plan->blocks1, smem, st, m, *plan);
        } else if (m.blk0[3] > 0) SPLIT_LAUNCH("k_embed_fwd_split", k_embed_fwd_split, m.blk0[3] + 3, smem, st, m);   // + fuse_weights
        return 0;
    }
    // the embedding programs stage three matrices (52 KB): two blocks fit a CU, and with many tiles per wave four waves per SIMD
    // overlap the store-heavy epilogues with the MFMAs better than two (capfac x 32, indset x 64)
    const int cap_knob = GCNN_KNOB("GCNN_EMB_CAP", 0);
    const int tiles = cdiv(std::max(m.v.n, 0), 16) + cdiv(std::max(m.c.n, 0), 16) + cdiv(std::max(m.k.n, 0), 16);
    const int cap = cap_knob > 0 ? cap_knob : (tiles >= 8192 ? 2 * device_cus() - 3 : 256);   // (- 3: the fuse_weights blocks are resident too)
    const int nwaves = rows_blocks(n, ns, 3, m.blk0, cap);
    if (plan) {
        const int nt = nwaves * 64;
        plan->blocks0 = std::min(cdiv(plan->s[0].n_edges + 1, nt), 32);   // few blocks, looping: they hold a CU slot of this launch's size
        plan->blocks1 = std::min(cdiv(plan->s[1].n_edges + 1, nt), 8);
        const int grid = m.blk0[3] + 3 + plan->blocks0 + plan->blocks1;
        ROWS_LAUNCH("k_infer_s1 (embeddings + plan: count)", k_infer_s1<8>, k_infer_s1<4>, nwaves, grid, EMB_LDS_FLOATS * sizeof(float), st, m, *plan);
        return 0;
    }
    if (m.blk0[3] == 0) return 0;
    // + 3 blocks: fuse_weights, the folded matrices of the three convolutions (two blocks of this launch fit a CU, so they do not
    // queue behind the embedding blocks)
    ROWS_LAUNCH("k_embed_fwd", k_embed_fwd<8>, k_embed_fwd<4>, nwaves, m.blk0[3] + 3, EMB_LDS_FLOATS * sizeof(float), st, m);
    return 0;
}
// keep_a: the two-layer form that materialises A (PreNorm fitting); never together with a plan
static int launch_conv_fwd(const ConvFArgs& a, int tail, const IplanArgs* plan, hipStream_t st, bool keep_a = false) {
    int blk0[2];
    const int ns = 4;
    if (rows_split(&a.n, 1, blk0)) {
        const size_t smem = CONV_SPLIT_LDS_FLOATS * sizeof(float);
        if (plan && tail == CF_PROJ && plan->n_vars <= IPLAN_FUSE_MAX_VARS) {   // the plan's order step rides in this launch
            const int grid = blk0[1] + std::min(cdiv(plan->n_vars, 16), 48);
            if (grid > 0) SPLIT_LAUNCH("k_infer_s3 (conv row program + plan: order)", (k_infer_s3<4, true>), grid, smem, st, a, *plan, blk0[1]);
            return 0;
        }
        if (plan && tail == CF_PROJ && plan->n_vars > 0) {
            ProfScope prof("k_iplan_order", st);
            hipLaunchKernelGGL(k_iplan_order, dim3(std::min(cdiv(plan->n_vars, 16), 2048)), dim3(256), 0, st, *plan);
            LAUNCHCHK();
        }
        if (blk0[1] == 0) return 0;
        if (keep_a) {
            if (tail == CF_READOUT) SPLIT_LAUNCH("k_conv_fwd<readout, keep A>", (k_conv_fwd_split<CF_READOUT, true>), blk0[1], smem, st, a);
            else SPLIT_LAUNCH("k_conv_fwd<proj, keep A>", (k_conv_fwd_split<CF_PROJ, true>), blk0[1], smem, st, a);
        } else if (tail == CF_READOUT) SPLIT_LAUNCH("k_conv_fwd<readout>", k_conv_fwd_split<CF_READOUT>, blk0[1], smem, st, a);
        else SPLIT_LAUNCH("k_conv_fwd<proj>", k_conv_fwd_split<CF_PROJ>, blk0[1], smem, st, a);
        return 0;
    }
    const int nwaves = rows_blocks(&a.n, &ns, 1, blk0);
    const size_t smem = ROWS_LDS_FLOATS(5, 5) * sizeof(float);
    if (plan && tail == CF_PROJ && plan->n_vars <= IPLAN_FUSE_MAX_VARS) {   // the plan's order step rides in this launch
        const int grid = blk0[1] + std::min(cdiv(plan->n_vars, nwaves * 4), 48);
        if (grid == 0) return 0;
        ROWS_LAUNCH("k_infer_s3 (conv row program + plan: order)", k_infer_s3<8>, k_infer_s3<4>, nwaves, grid, smem, st, a, *plan, blk0[1]);
        return 0;
    }
    if (plan && tail == CF_PROJ && plan->n_vars > 0) {   // many variables: the order step wants every lane group resident at once
        ProfScope prof("k_iplan_order", st);
        hipLaunchKernelGGL(k_iplan_order, dim3(std::min(cdiv(plan->n_vars, 16), 2048)), dim3(256), 0, st, *plan);
        LAUNCHCHK();
    }
    if (blk0[1] == 0) return 0;
    if (keep_a) {
        if (tail == CF_READOUT) ROWS_LAUNCH("k_conv_fwd<readout, keep A>", (k_conv_fwd<8, CF_READOUT, true>), (k_conv_fwd<4, CF_READOUT, true>), nwaves, blk0[1], smem, st, a);
        else ROWS_LAUNCH("k_conv_fwd<proj, keep A>", (k_conv_fwd<8, CF_PROJ, true>), (k_conv_fwd<4, CF_PROJ, true>), nwaves, blk0[1], smem, st, a);
    } else if (tail == CF_READOUT) ROWS_LAUNCH("k_conv_fwd<readout>", (k_conv_fwd<8, CF_READOUT>), (k_conv_fwd<4, CF_READOUT>), nwaves, blk0[1], smem, st, a);
    else ROWS_LAUNCH("k_conv_fwd<proj>", (k_conv_fwd<8, CF_PROJ>), (k_conv_fwd<4, CF_PROJ>), nwaves, blk0[1], smem, st, a);
    return 0;
}
// training turnaround: the last forward program and the first backward program of the cut rows in one launch (k_rows.hpp)
static int launch_conv_turn(const ConvFArgs& a, const ConvBArgs& b, hipStream_t st) {
    int blk0[2];
    const int ns = 10;
    if (rows_split(&a.n, 1, blk0)) {
        if (blk0[1] > 0)
            SPLIT_LAUNCH("k_conv_turn (readout + loss head + cut-row gradients)", k_conv_turn_split, blk0[1], CONV_SPLIT_LDS_FLOATS * sizeof(float), st, a, b);
        return 0;
    }
    const int nwaves = rows_blocks(&a.n, &ns, 1, blk0);
    if (blk0[1] == 0) return 0;
    ROWS_LAUNCH("k_conv_turn (readout + loss head + cut-row gradients)", k_conv_turn<8>, k_conv_turn<4>, nwaves, blk0[1],
                ROWS_LDS_FLOATS(5, 5) * sizeof(float), st, a, b);
    return 0;
}
static int launch_conv_bwd(ConvBGroupArgs& m, hipStream_t st) {
    const int n[2] = {m.cb.n, m.tail.n}, ns[2] = {5, 2};
    const int nwaves = rows_blocks(n, ns, 2, m.blk0);
    if (m.blk0[2] == 0) return 0;
    ROWS_LAUNCH("k_conv_bwd", k_conv_bwd<8>, k_conv_bwd<4>, nwaves, m.blk0[2], ROWS_LDS_FLOATS(5, 1) * sizeof(float), st, m);
    return 0;
}
static int launch_tail_bwd(TailGroupArgs& m, hipStream_t st) {
    const int n[2] = {m.a.n, m.b.n}, ns[2] = {3, 2};
    const int nwaves = rows_blocks(n, ns, 2, m.blk0);
    if (m.blk0[2] == 0) return 0;
    ROWS_LAUNCH("k_tail_bwd", k_tail_bwd<8>, k_tail_bwd<4>, nwaves, m.blk0[2], ROWS_LDS_FLOATS(3, 1) * sizeof(float), st, m);
    return 0;
}

// ---- forward ----------------------------------------------------------------------------------------------------
struct ConvIO {           // one PartialGraphConvolution instance (model.py:201-203, 294-296)
    int pbase;            // first parameter index of the block
    const float* xl; const float* xv; int nl, nv, ne;
    bool recv_left;
    const gcnn_graph* g; int pedge;  // edge PreNorm parameter index (shift; scale = +1)
    float *PL, *PR, *S, *A, *Z1, *OUT;
    float *gPL, *gPR, *gS, *gA, *gZ1, *gOUT, *gXL, *gXV, *DWP, *DWP2;
    float* N;
    float* FZ;             // folded weights M | u
    mask16 *mZ1, *mOUT;    // ReLU patterns of Z1 and OUT
};

static EdgeArgs conv_edge_args(const float* p, const ConvIO& c, bool by_left) {
    EdgeArgs e; memset(&e, 0, sizeof(e));
    e.seg_ptr = by_left ? c.g->l_ptr : c.g->v_ptr; e.oth = by_left ? c.g->l_oth : c.g->v_oth;
    e.coef = by_left ? c.g->l_coef : c.g->v_coef;
    e.p_own = by_left ? c.PL : c.PR; e.p_oth = by_left ? c.PR : c.PL;   // segment owner's table / gathered table
    e.w_edge = p + poff(c.pbase + C_WE); e.e_shift = p + poff(c.pedge); e.e_scale = p + poff(c.pedge + 1);
    e.s1 = p + poff(c.pbase + C_S1); e.n_own = by_left ? c.nl : c.nv;
    return e;
}

// edge pass + the receiver-side update program S -> A -> Z1 -> X' (model.py:498-508, 568-573) and, in the same launch, what
// consumes X': the next convolution's projection (wt, bt -> t_out) or the readout
struct LossHead { const float* targets; float scale; float* g_o1; float* partial; };   // CF_LOSS extras
static ConvBArgs conv_bwd_args(const float* p, const ConvIO& c, const float* in, const float* w0);
static int conv_forward(const float* p, const ConvIO& c, bool save, hipStream_t st, const float* wt, const float* bt,
                        float* t_out, int tail, float* scores, const LossHead* head, const IplanArgs* plan = nullptr, bool keep_a = false) {
    int rc;
    EdgeArgs e = conv_edge_args(p, c, c.recv_left);
    e.out = c.S; e.cnt_rows = c.N;
    if ((rc = launch_edge_fwd(e, c.ne, c.recv_left ? c.g->l_max_deg : c.g->v_max_deg, save, st, plan))) return rc;
    ConvFArgs a; memset(&a, 0, sizeof(a));
    a.n = c.recv_left ? c.nl : c.nv;
    a.s = c.S; a.seg_ptr = e.seg_ptr; a.wf = p + poff(c.pbase + C_WF); a.bf = p + poff(c.pbase + C_BF); a.a_out = keep_a ? c.A : nullptr;
    a.mfuse = c.FZ; a.ufuse = c.FZ + EMB * EMB;
    a.s2 = p + poff(c.pbase + C_S2); a.xrecv = c.recv_left ? c.xl : c.xv;
    a.w1a = p + poff(c.pbase + C_W1); a.w1b = p + poff(c.pbase + C_W1) + EMB * EMB; a.b1 = p + poff(c.pbase + C_B1);
    a.z1 = save ? c.Z1 : nullptr; a.m_z1 = save ? c.mZ1 : nullptr; a.m_out = save ? c.mOUT : nullptr;
    a.w2 = p + poff(c.pbase + C_W2); a.b2 = p + poff(c.pbase + C_B2); a.out = c.OUT;
    a.wt = wt; a.bt = bt; a.t_out = t_out;
    if (tail != CF_PROJ) { a.ws = p + poff(P_OUT + 2); a.bs = p + poff(P_OUT + 3); a.scores = scores; }
    if (tail == CF_LOSS) {   // training: the cut rows turn around in this launch (their receiver gradients land in the workspace)
        a.targets = head->targets; a.loss_scale = head->scale; a.g_o1 = head->g_o1; a.head_partial = head->partial;
        return launch_conv_turn(a, conv_bwd_args(p, c, head->g_o1, wt), st);
    }
    return launch_conv_fwd(a, tail, plan, st, keep_a);
}

static void conv_setup(ConvIO cv[3], const gcnn_dims* d, const Work& w, const gcnn_graph* cg, const gcnn_graph* kg) {
    const Acts &A = w.a, &G = w.g;
    cv[0] = ConvIO{P_CONV0, A.Xc, A.Xv, d->n_cons, d->n_vars, d->n_cons_edges, true, cg, P_CONS_EDGE,
                   A.PL1, A.PR1, A.S1, A.A1, A.Z1c, A.Xc2, G.PL1, G.PR1, G.S1, G.A1, G.Z1c, G.Xc2, G.Xc, G.Xv, w.dwp[0], w.dwp2[0], w.nrow[0], w.fuse[0], w.m.Z1c, w.m.Xc2};
    cv[1] = ConvIO{P_CONV1, A.Xc2, A.Xv, d->n_cons, d->n_vars, d->n_cons_edges, false, cg, P_CONS_EDGE,
                   A.PL2, A.PR2, A.S2, A.A2, A.Z1v, A.Xv2, G.PL2, G.PR2, G.S2, G.A2, G.Z1v, G.Xv2, G.Xc2, G.Xv, w.dwp[1], w.dwp2[1], w.nrow[1], w.fuse[1], w.m.Z1v, w.m.Xv2};
    cv[2] = ConvIO{P_CONV2, A.Xk, A.Xv2, d->n_cuts, d->n_vars, d->n_cut_edges, true, kg, P_CUT_EDGE,
                   A.PL3, A.PR3, A.S3, A.A3, A.Z1k, A.Xk2, G.PL3, G.PR3, G.S3, G.A3, G.Z1k, G.Xk2, G.Xk, G.Xv2, w.dwp[2], w.dwp2[2], w.nrow[2], w.fuse[2], w.m.Z1k, w.m.Xk2};
}

static int check_common(const gcnn_dims* d, const float* params, const gcnn_graph* cg, const gcnn_graph* kg,
                        float* workspace, size_t workspace_floats) {
    if (!d || !params || !cg || !kg) return GCNN_E_BADARG;
    if (d->n_cons < 0 || d->n_vars < 0 || d->n_cuts < 0 || d->n_cons_edges < 0 || d->n_cut_edges < 0) return GCNN_E_BADARG;
    if (!workspace || workspace_floats < gcnn_workspace_floats(d)) return GCNN_E_WORKSPACE;
    if (((uintptr_t)workspace & 15) || ((uintptr_t)params & 15)) return GCNN_E_BADARG;
    // the edge passes address a gathered table as base + 32-bit byte offset (row index << 8)
    if (d->n_cons > (1 << 24) || d->n_vars > (1 << 24) || d->n_cuts > (1 << 24)) return GCNN_E_UNSUPPORTED;
    return 0;
}

// `targets` != nullptr: the last launch also evaluates the MSE head and the readout's Dense(64->1) gradient (CF_LOSS)
static int forward_impl(const gcnn_dims* d, const float* p, const float* cons_feats, const float* var_feats,
                        const float* cut_feats, const gcnn_graph* cg, const gcnn_graph* kg, float* workspace,
                        size_t workspace_floats, float* scores, int save_mode, const float* targets, float loss_scale,
                        hipStream_t st, IplanArgs* plan = nullptr) {
    const bool save = save_mode != 0, keep_a = save_mode == 2;   // 2: also materialise A of every convolution (PreNorm fitting)
    layout_init();
    int rc = check_common(d, p, cg, kg, workspace, workspace_floats);
    if (rc) return rc;
    if (d->n_cuts > 0 && !scores) return GCNN_E_BADARG;
    Work w; carve(d, workspace, &w);
    const Acts& A = w.a;
    // embeddings (model.py:287-291) fused with the projections of the raw embeddings they feed (model.py:486-496): three
    // independent programs, one grouped launch
    {
        EmbGroupArgs m; memset(&m, 0, sizeof(m));
        auto emb = [&](EmbArgs& e, const float* x, int pb, float* xo, int n, mask16* me1, mask16* mx) {
            e.x = x; e.shift = p + poff(pb + E_SHIFT); e.scale = p + poff(pb + E_SCALE); e.w1 = p + poff(pb + E_W1);
            e.b1 = p + poff(pb + E_B1); e.w2 = p + poff(pb + E_W2); e.b2 = p + poff(pb + E_B2);
            e.xo = xo; e.n = n; e.m_e1 = save ? me1 : nullptr; e.m_x = save ? mx : nullptr;
        };
        emb(m.v, var_feats, P_VAR, A.Xv, d->n_vars, w.m.E1v, w.m.Xv);    // variables: E1 -> Xv -> PR1, PR2 (model.py:294-295)
        m.v.wp[0] = p + poff(P_CONV0 + C_WR); m.v.po[0] = A.PR1; m.v.wp[1] = p + poff(P_CONV1 + C_WR); m.v.po[1] = A.PR2;
        emb(m.c, cons_feats, P_CONS, A.Xc, d->n_cons, w.m.E1c, w.m.Xc);  // constraints: E1 -> Xc -> PL1
        m.c.wp[0] = p + poff(P_CONV0 + C_WL); m.c.bp[0] = p + poff(P_CONV0 + C_BL); m.c.po[0] = A.PL1;
        emb(m.k, cut_feats, P_CUT, A.Xk, d->n_cuts, w.m.E1k, w.m.Xk);    // cuts: E1 -> Xk -> PL3
        m.k.wp[0] = p + poff(P_CONV2 + C_WL); m.k.bp[0] = p + poff(P_CONV2 + C_BL); m.k.po[0] = A.PL3;
        const int convs[3] = {P_CONV0, P_CONV1, P_CONV2};     // the folded weights of the three convolutions ride in this launch
        for (int k = 0; k < 3; ++k) {
            m.fz.wf[k] = p + poff(convs[k] + C_WF); m.fz.bf[k] = p + poff(convs[k] + C_BF); m.fz.s2[k] = p + poff(convs[k] + C_S2);
            m.fz.w1a[k] = p + poff(convs[k] + C_W1); m.fz.out[k] = w.fuse[k];
        }
        if ((rc = launch_embed_fwd(m, plan, st))) return rc;
    }
    // convolutions (model.py:294-296), each followed in the same launch by what consumes its output
    ConvIO cv[3]; conv_setup(cv, d, w, cg, kg);
    // updated constraints -> left projection of conv c->v
    if ((rc = conv_forward(p, cv[0], save, st, p + poff(P_CONV1 + C_WL), p + poff(P_CONV1 + C_BL), A.PL2, CF_PROJ, nullptr, nullptr, plan, keep_a))) return rc;
    // updated variables -> right projection of conv v->k
    if ((rc = conv_forward(p, cv[1], save, st, p + poff(P_CONV2 + C_WR), nullptr, A.PR3, CF_PROJ, nullptr, nullptr, nullptr, keep_a))) return rc;
    // updated cuts -> readout (model.py:206-208, 299-300)
    if (targets) {   // O1 itself is not needed afterwards: its ReLU mask is folded into dO1pre, its values into the dws partials
        const LossHead head = {targets, loss_scale, w.g.O1, w.score_partial};
        return conv_forward(p, cv[2], save, st, p + poff(P_OUT), p + poff(P_OUT + 1), nullptr, CF_LOSS, scores, &head);
    }
    return conv_forward(p, cv[2], save, st, p + poff(P_OUT), p + poff(P_OUT + 1), save ? A.O1 : nullptr, CF_READOUT, scores, nullptr, nullptr, keep_a);
}
extern "C" int gcnn_forward(const gcnn_dims* d, const float* p, const float* cons_feats, const float* var_feats,
                 const float* cut_feats, const gcnn_graph* cg, const gcnn_graph* kg, float* workspace,
                 size_t workspace_floats, float* scores, int32_t save_for_backward, void* stream) {
    if (save_for_backward < 0 || save_for_backward > 2) return GCNN_E_BADARG;
    return forward_impl(d, p, cons_feats, var_feats, cut_feats, cg, kg, workspace, workspace_floats, scores,
                        save_for_backward, nullptr, 0.f, (hipStream_t)stream);
}
// ---- single-state inference: the SCIP cut selector's call (model_evaluator.py:82-111) as ONE entry point --------------------
static inline size_t al16(size_t x) { return (x + 15) & ~(size_t)15; }
extern "C" int gcnn_infer_layout_for(const gcnn_dims* d, gcnn_infer_layout* L) {
    if (!d || !L || d->n_cons < 0 || d->n_vars < 0 || d->n_cuts < 0 || d->n_cons_edges < 0 || d->n_cut_edges < 0) return GCNN_E_BADARG;
    if (d->n_vars > IPLAN_MAX_VARS) return GCNN_E_UNSUPPORTED;
    const size_t C = d->n_cons, V = d->n_vars, K = d->n_cuts, E1 = d->n_cons_edges, E2 = d->n_cut_edges;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += al16(bytes); return o; };
    L->in_off[0] = take(4 * (2 * V + 8 + (C + 1) + (K + 1)));   // zero block: vcount[V] | cursor[V] | flags[8] | l_ptr cons | l_ptr cut
    L->in_off[1] = take(16 * C);                     // cons_feats [C,4]
    L->in_off[2] = take(8 * E1);                     // cons_edge_inds [2,E1]
    L->in_off[3] = take(4 * E1);                     // cons_edge_feats [E1]
    L->in_off[4] = take(56 * V);                     // var_feats [V,14]
    L->in_off[5] = take(24 * K);                     // cut_feats [K,6]
    L->in_off[6] = take(8 * E2);                     // cut_edge_inds [2,E2]
    L->in_off[7] = take(4 * E2);                     // cut_edge_feats [E2]
    L->in_bytes = off;
    L->out_off[0] = 0; L->out_off[1] = al16(4 * K); L->out_off[2] = L->out_off[1] + al16(4 * K);
    L->out_bytes = L->out_off[2] + 16;               // scores[K] | order[K] | flags[4]
    // device arena: the uploaded block, the plan, the output block, the forward workspace
    size_t a = off;
    auto dev = [&](size_t bytes) { const size_t o = a; a += (bytes + 255) & ~(size_t)255; return o; };
    L->dev_off[0] = L->in_off[0] + 4 * (2 * V + 8); L->dev_off[1] = L->dev_off[0] + 4 * (C + 1);            // l_ptr cons, l_ptr cut: inside the zero block
    L->dev_off[2] = dev(4 * (V + 1));                                                                          // v_ptr
    L->dev_off[3] = dev(4 * E1); L->dev_off[4] = dev(4 * E1); L->dev_off[5] = dev(4 * E1);                    // v_pos, v_oth, v_coef
    L->dev_off[6] = dev(L->out_bytes);
    L->dev_off[7] = dev(sizeof(float) * gcnn_workspace_floats(d));
    L->arena_bytes = a;
    return 0;
}

extern "C" int gcnn_infer(const gcnn_dims* d, const float* params, const void* host_in, void* host_out, void* arena,
                          size_t arena_bytes, int32_t want_order, void* stream) {
    gcnn_infer_layout L;
    int rc = gcnn_infer_layout_for(d, &L);
    if (rc) return rc;
    if (!params || !host_in || !host_out || !arena || arena_bytes < L.arena_bytes || ((uintptr_t)arena & 255)) return GCNN_E_BADARG;
    if (want_order && d->n_cuts > 4096) return GCNN_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    char* A = (char*)arena;
    HIPCHK(hipMemcpyAsync(A, host_in, L.in_bytes, hipMemcpyHostToDevice, st));      // ONE upload: zero block + the seven arrays
    const int C = d->n_cons, V = d->n_vars, K = d->n_cuts, E1 = d->n_cons_edges, E2 = d->n_cut_edges;
    int* zero = (int*)(A + L.in_off[0]);
    IplanArgs ia; memset(&ia, 0, sizeof(ia));
    ia.s[0] = IplanSet{(int*)(A + L.in_off[2]), E1, C, (int*)(A + L.dev_off[0])};
    ia.s[1] = IplanSet{(int*)(A + L.in_off[6]), E2, K, (int*)(A + L.dev_off[1])};
    ia.n_vars = V; ia.vcount = zero; ia.cursor = zero + V; ia.flags = zero + 2 * V;
    ia.v_ptr = (int*)(A + L.dev_off[2]); ia.v_pos = (int*)(A + L.dev_off[3]); ia.v_oth = (int*)(A + L.dev_off[4]);
    ia.v_coef = (float*)(A + L.dev_off[5]); ia.cons_coef = (const float*)(A + L.in_off[3]);
    gcnn_graph cg, kg; memset(&cg, 0, sizeof(cg)); memset(&kg, 0, sizeof(kg));
    cg.l_ptr = ia.s[0].l_ptr; cg.l_oth = ia.s[0].inds + E1; cg.l_coef = ia.cons_coef;       // by-left order = the input lists
    cg.v_ptr = ia.v_ptr; cg.v_oth = ia.v_oth; cg.v_coef = ia.v_coef;
    kg.l_ptr = ia.s[1].l_ptr; kg.l_oth = ia.s[1].inds + E2; kg.l_coef = (const float*)(A + L.in_off[7]);
    kg.v_ptr = ia.v_ptr;   // never read: conv v->k gathers by cut only and nothing is differentiated
    float* out = (float*)(A + L.dev_off[6]);
    // the plan's three steps ride in the forward pass's first three launches (k_infer.hpp)
    rc = forward_impl(d, params, (const float*)(A + L.in_off[1]), (const float*)(A + L.in_off[4]), (const float*)(A + L.in_off[5]),
                      &cg, &kg, (float*)(A + L.dev_off[7]), gcnn_workspace_floats(d), out, 0, nullptr, 0.f, st, &ia);
    if (rc) return rc;
    if (want_order && K > 0) {
        ProfScope prof("k_rank_scores", st);
        hipLaunchKernelGGL(k_rank_scores, dim3(1), dim3(256), 0, st, out, K, (int*)((char*)out + L.out_off[1]));
        LAUNCHCHK();
    }
    HIPCHK(hipMemcpyAsync((char*)out + L.out_off[2], ia.flags, 16, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(host_out, out, L.out_bytes, hipMemcpyDeviceToHost, st));  // ONE download: scores | order | flags
    return 0;
}

// host side of the single-state call: stable counting sort of an unsorted edge list by row, straight into the staging buffer
extern "C" int gcnn_host_sort_edges_by_row(const int32_t* rows, const int32_t* cols, const float* vals, int32_t n_edges, int32_t n_left,
                                           int32_t* out_inds, float* out_vals, int32_t* scratch) {
    if (n_edges < 0 || n_left < 0 || (n_edges > 0 && (!rows || !cols || !vals || !out_inds || !out_vals || !scratch))) return GCNN_E_BADARG;
    for (int i = 0; i <= n_left; ++i) scratch[i] = 0;
    for (int e = 0; e < n_edges; ++e) {
        const int r = rows[e];
        if (r < 0 || r >= n_left) return GCNN_E_BADARG;
        ++scratch[r + 1];
    }
    for (int i = 0; i < n_left; ++i) scratch[i + 1] += scratch[i];      // scratch[r] = first output slot of row r
    for (int e = 0; e < n_edges; ++e) {
        const int at = scratch[rows[e]]++;
        out_inds[at] = rows[e]; out_inds[n_edges + at] = cols[e]; out_vals[at] = vals[e];
    }
    return 0;
}

// the usual case in one call: copy the list into the staging buffer and look at its order on the way; only a list that turns out
// not to be sorted by row goes through the counting sort above
extern "C" int gcnn_host_pack_edges(const int32_t* rows, const int32_t* cols, const float* vals, int32_t n_edges, int32_t n_left,
                                    int32_t* out_inds, float* out_vals, int32_t* scratch) {
    if (n_edges < 0 || n_left < 0 || (n_edges > 0 && (!rows || !cols || !vals || !out_inds || !out_vals || !scratch))) return GCNN_E_BADARG;
    int unsorted = 0;
    for (int e = 1; e < n_edges; ++e) unsorted |= rows[e] < rows[e - 1];
    if (unsorted && gcnn_host_sort_edges_by_row(rows, cols, vals, n_edges, n_left, out_inds, out_vals, scratch) == 0) return 1;
    memcpy(out_inds, rows, sizeof(int32_t) * (size_t)n_edges);            // sorted -- or a row id out of range: packed as it is,
    memcpy(out_inds + n_edges, cols, sizeof(int32_t) * (size_t)n_edges);   // the device check reports it
    memcpy(out_vals, vals, sizeof(float) * (size_t)n_edges);
    return unsorted ? 2 : 0;
}

extern "C" int gcnn_forward_loss(const gcnn_dims* d, const float* p, const float* cons_feats, const float* var_feats,
                      const float* cut_feats, const gcnn_graph* cg, const gcnn_graph* kg, float* workspace,
                      size_t workspace_floats, float* scores, const float* targets, float loss_scale, void* stream) {
    if (d && d->n_cuts > 0 && !targets) return GCNN_E_BADARG;
    return forward_impl(d, p, cons_feats, var_feats, cut_feats, cg, kg, workspace, workspace_floats, scores, 1,
                        d && d->n_cuts > 0 ? targets : nullptr, loss_scale, (hipStream_t)stream);
}


// ---- backward ---------------------------------------------------------------------------------------------------
struct PendWg { const float *x, *sx, *d; const int* seg_ptr; int n; float *gw, *gb, *g2; const mask16* mask; const float *shift, *scale; int f;
                const float *w1, *b1; int f2;
                int fold; };   // > 0: the job's G and degree-weighted column sum stay in its slabs for fold entry (fold - 1) (fold_block, k_wgrad.hpp)
struct JobList {
    WgArgs wg; RdArgs rd; int nslab;
    int rdblk;
    DwRedArgs dw; int ndw;   // d w_edge pre-reductions (one per convolution with edge-gradient partials)
    FoldArgs fold;           // convolutions whose folded-layer gradients the front blocks of the k_reduce launch unfold
    PendWg pend[WG_MAX_JOBS]; int npend;   // weight-gradient jobs as collected; ordered and placed by place_wg
};
static void add_wg(JobList& jl, const float* x, const float* sx, const float* dmat, const int* seg_ptr,
                   int n, float* gw, float* gb, float* g2, float* /*partial*/) {
    if (n <= 0) return;  // empty input: gradients are exactly zero
    jl.pend[jl.npend++] = PendWg{x, sx, dmat, seg_ptr, n, gw, gb, g2, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, 0};
}
// first layer of an embedding: x = raw features [n][f], dmat = dE1 (unmasked), e1 = the ReLU pattern of the layer's output
// (k_wgrad.hpp, EXTRA == 2)
static void add_wg_emb1(JobList& jl, const float* x, const float* shift, const float* scale, const float* dmat, const mask16* e1,
                        int n, int f, float* gw, float* gb) {
    if (n <= 0) return;
    jl.pend[jl.npend++] = PendWg{x, nullptr, dmat, nullptr, n, gw, gb, nullptr, e1, shift, scale, f, nullptr, nullptr, 0, 0};
}
// second layer of an embedding with its input E1 recomputed from the raw features (k_wgrad.hpp, EXTRA == 3): dmat = dX
static void add_wg_emb2(JobList& jl, const float* x, const float* shift, const float* scale, const float* w1, const float* b1,
                        const float* dmat, int n, int f, float* gw, float* gb) {
    if (n <= 0) return;
    jl.pend[jl.npend++] = PendWg{x, nullptr, dmat, nullptr, n, gw, gb, nullptr, nullptr, shift, scale, 0, w1, b1, f, 0};
}
// Order the collected jobs and give them their block ranges.  Several jobs read the same matrix (dZ1 feeds the gradients of
// both halves of W1; a raw embedding X is the operand of up to three products): such jobs are placed next to each other and
// k_wgrad rotates each job's block -> row-block mapping so that (row block) = (block index) mod 8: blocks are dealt to the eight
// XCDs round-robin, so the same rows of two adjacent jobs are read on the same XCD at about the same time and the second read
// hits that XCD's L2 instead of going to memory.  Values do not depend on the order.
static void place_wg(JobList& jl, float* partial) {
    bool used[WG_MAX_JOBS] = {};
    int order[WG_MAX_JOBS], last = -1;
    for (int k = 0; k < jl.npend; ++k) {
        int pick = -1;
        const int share = GCNN_KNOB("GCNN_WG_SHARE", 1);
        if (last >= 0 && share)
            for (int i = 0; i < jl.npend && pick < 0; ++i)
                if (!used[i] && jl.pend[i].n == jl.pend[last].n && (jl.pend[i].x == jl.pend[last].x || jl.pend[i].d == jl.pend[last].d)) pick = i;
        for (int i = 0; i < jl.npend && pick < 0; ++i)   // otherwise the longest job not yet placed (ties: collection order)
            if (!used[i]) { pick = i; for (int q = i + 1; q < jl.npend; ++q) if (!used[q] && jl.pend[q].n > jl.pend[pick].n) pick = q; }
        used[pick] = true; order[k] = last = pick;
    }
    // Chunk size: the launch is ONE resident round -- two blocks (8 waves) per CU, every SIMD holding two waves of (nearly) the
    // same length from start to end, so the MFMA pipes stay shared evenly and there is no second, partly filled round.
    // Smallest chunk (a multiple of 16 rows, at least WG_ROWS) whose block count fits; each job then spreads its rows evenly.
    // (every job needs at least one block, so fewer slots than jobs -- a device with a handful of CUs -- can never be met)
    // A job's rows count with its cost per row in sixteenths of the plain product's (64 MFMAs and 512 B per 16 rows): the first
    // embedding layer's job issues 16 MFMAs and reads 320 B, the recomputing job 64 + 4 * ceil(f/4) MFMAs and <= 312 B
    // (10/16 and (16 + ceil(f/4))/16 measured best on capfac, setcov and indset batches: profiles/README.md, round 3).
    const int slots = std::max(std::min(2 * device_cus(), WG_MAX_SLABS - WG_MAX_JOBS), jl.npend);
    const int cost2 = GCNN_KNOB("GCNN_WG_COST2", 10), cost3 = GCNN_KNOB("GCNN_WG_COST3", 16);
    auto nblocks = [&](const PendWg& q, int r) {
        const long long c = q.f ? cost2 : q.f2 ? cost3 + (q.f2 + 3) / 4 * GCNN_KNOB("GCNN_WG_COST3K", 1) : 16;
        return (int)std::max<long long>(1, ((long long)q.n * c + 16LL * r * WG_WAVES - 1) / (16LL * r * WG_WAVES));
    };
    auto blocks_at = [&](int r) { long long t = 0; for (int k = 0; k < jl.npend; ++k) t += nblocks(jl.pend[k], r); return t; };
    int rows = WG_ROWS;
    if (blocks_at(rows) > slots) {   // bisect on multiples of 16
        int lo = rows / 16, hi = lo;
        while (blocks_at(hi * 16) > slots && hi < (1 << 24)) hi *= 2;
        while (hi - lo > 1) { const int mid = (lo + hi) / 2; if (blocks_at(mid * 16) > slots) lo = mid; else hi = mid; }
        rows = hi * 16;
    }
    const int forced = GCNN_KNOB("GCNN_WG_ROWS", 0);
    if (forced >= 16 && forced % 16 == 0 && blocks_at(forced) <= WG_MAX_SLABS) rows = forced;
    for (int k = 0; k < jl.npend; ++k) {
        const PendWg& q = jl.pend[order[k]];
        WgJob& j = jl.wg.job[jl.wg.njobs++];
        const int nb = std::min(nblocks(q, rows), cdiv(q.n, 16 * WG_WAVES));   // blocks = slabs: four chunks each
        j.nb = nb; j.rows = (cdiv(q.n, nb * WG_WAVES) + 15) & ~15;
        j.x = q.x; j.sx = q.sx; j.d = q.d; j.seg_ptr = q.seg_ptr; j.n = q.n; j.blk0 = jl.wg.nblocks; j.slab0 = jl.nslab;
        j.mask = q.mask; j.shift = q.shift; j.scale = q.scale; j.f = q.f; j.w1 = q.w1; j.b1 = q.b1; j.f2 = q.f2;
        const float* src = partial + (size_t)jl.nslab * WG_SLAB;
        jl.wg.nblocks += nb; jl.nslab += nb;
        auto rd = [&](const float* s, float* dst, int len) {
            RdJob& r = jl.rd.job[jl.rd.njobs++];
            r.src = s; r.dst = dst; r.nparts = nb; r.stride = WG_SLAB; r.len = len; r.blk0 = jl.rdblk;
            jl.rdblk += cdiv(len, EMB);
        };
        if (q.fold) { jl.fold.slab[q.fold - 1] = src; jl.fold.nparts[q.fold - 1] = nb; }   // summed by the blocks that consume it
        else rd(src, q.gw, (q.f ? q.f : EMB) * EMB);   // first-layer kernel [f,64]: the first f rows of the slab
        if (q.gb) rd(src + EMB * EMB, q.gb, EMB);
        if (q.g2) rd(src + EMB * EMB + EMB, q.g2, EMB);
    }
    jl.npend = 0;
}
static void add_rd(JobList& jl, const float* src, float* dst, int nparts, int stride, int len) {
    if (nparts <= 0) return;
    RdJob& r = jl.rd.job[jl.rd.njobs++];
    r.src = src; r.dst = dst; r.nparts = nparts; r.stride = stride; r.len = len; r.blk0 = jl.rdblk;
    jl.rdblk += cdiv(len, EMB);
}

// Receiver-side gradient program of one convolution, entered through the layer (w0) that consumed its output:
// `in` is that layer's output gradient
static ConvBArgs conv_bwd_args(const float* p, const ConvIO& c, const float* in, const float* w0) {
    ConvBArgs a; memset(&a, 0, sizeof(a));
    a.n = c.recv_left ? c.nl : c.nv;
    a.in = in; a.w0 = w0; a.m_out = c.mOUT; a.g_out = c.gOUT;
    a.w2 = p + poff(c.pbase + C_W2); a.m_z1 = c.mZ1; a.g_z1 = c.gZ1;
    a.w1b = p + poff(c.pbase + C_W1) + EMB * EMB; a.g_xrecv = c.recv_left ? c.gXL : c.gXV;
    a.mfuse = c.FZ; a.g_s = c.gS;
    a.s1 = p + poff(c.pbase + C_S1); a.nrows = c.N; a.g_precv = c.recv_left ? c.gPL : c.gPR;
    return a;
}

// sender-ordered half of the edge gradient, and the weight-gradient jobs of the whole convolution
static int conv_backward_edges(const float* p, float* grads, const ConvIO& c, const Work& w, JobList& jl, hipStream_t st) {
    int rc;
    const int nr = c.recv_left ? c.nl : c.nv;
    const float* xrecv = c.recv_left ? c.xl : c.xv;
    // the receiver-ordered half (dP_recv) came out of the row program's epilogue; sender-ordered half: gathers dS and P_recv rows
    // (the ReLU pattern is recomputed) and leaves d w_edge as one 64-float partial per block
    EdgeArgs e = conv_edge_args(p, c, !c.recv_left);
    e.d_s = c.gS; e.out = c.recv_left ? c.gPR : c.gPL; e.dw_partial = c.DWP;
    int dw_parts = 0;
    if ((rc = launch_edge_bwd_send(e, c.ne, c.recv_left ? c.g->v_max_deg : c.g->l_max_deg, &dw_parts, st))) return rc;
    if (dw_parts > 0) {   // DW_CHUNK partial rows per block of the k_wgrad launch, then the usual fixed-order reduction
        const int k = jl.ndw++;
        jl.dw.src[k] = c.DWP; jl.dw.dst[k] = c.DWP2; jl.dw.nparts[k] = dw_parts;
        jl.dw.blk0[k + 1] = jl.dw.blk0[k] + cdiv(dw_parts, DW_CHUNK);
        add_rd(jl, c.DWP2, grads + poff(c.pbase + C_WE), cdiv(dw_parts, DW_CHUNK), EMB, EMB);
    }
    const int* seg = c.recv_left ? c.g->l_ptr : c.g->v_ptr;
    add_wg(jl, c.Z1, nullptr, c.gOUT, nullptr, nr, grads + poff(c.pbase + C_W2), grads + poff(c.pbase + C_B2), nullptr, w.partial);
    // the folded layers: ONE product S^T dZ1 (-> G1) with the column sum (-> d b1) and the degree-weighted column sum (-> g2) of
    // dZ1; fold_block turns G1 | g2 into the gradients of Wf, bf and the upper half of W1 (the two-layer form needed A^T dZ1 and
    // S^T dA: two products, and A and dA in memory)
    if (nr > 0) {
        add_wg(jl, c.S, nullptr, c.gZ1, seg, nr, nullptr, grads + poff(c.pbase + C_B1), nullptr, w.partial);
        const int k = jl.fold.n++;
        jl.pend[jl.npend - 1].fold = k + 1;
        // (the forward's copies of Wf, bf, W1a: the launch that reads them also applies Adam to the originals)
        jl.fold.wf[k] = c.FZ + FUSE_WF; jl.fold.bf[k] = c.FZ + FUSE_BF;
        jl.fold.w1a[k] = c.FZ + FUSE_W1A; jl.fold.s2[k] = p + poff(c.pbase + C_S2);
        jl.fold.gwf[k] = grads + poff(c.pbase + C_WF); jl.fold.gbf[k] = grads + poff(c.pbase + C_BF); jl.fold.gw1a[k] = grads + poff(c.pbase + C_W1);
    }
    add_wg(jl, xrecv, nullptr, c.gZ1, nullptr, nr, grads + poff(c.pbase + C_W1) + EMB * EMB, nullptr, nullptr, w.partial);
    add_wg(jl, c.xl, nullptr, c.gPL, nullptr, c.nl, grads + poff(c.pbase + C_WL), grads + poff(c.pbase + C_BL), nullptr, w.partial);
    add_wg(jl, c.xv, nullptr, c.gPR, nullptr, c.nv, grads + poff(c.pbase + C_WR), nullptr, nullptr, w.partial);
    return 0;
}

extern "C" int gcnn_mse_loss(const float* scores, const float* targets, int32_t n, float scale, float* loss_out, float* d_scores,
                  void* stream) {
    if (n < 0 || (n > 0 && (!scores || !targets))) return GCNN_E_BADARG;
    if (n == 0) {
        if (loss_out) HIPCHK(hipMemsetAsync(loss_out, 0, sizeof(float), (hipStream_t)stream));
        return 0;
    }
    ProfScope prof("k_mse", (hipStream_t)stream);
    hipLaunchKernelGGL(k_mse, dim3(1), dim3(256), 0, (hipStream_t)stream, scores, targets, scale, loss_out, d_scores, n);
    LAUNCHCHK();
    return 0;
}

extern "C" int gcnn_backward(const gcnn_dims* d, const float* p, const float* cons_feats, const float* var_feats,
                  const float* cut_feats, const gcnn_graph* cg, const gcnn_graph* kg, float* workspace,
                  size_t workspace_floats, const float* d_scores, float* grads, float* cut_count_out, float* loss_out,
                  const gcnn_adam_args* adam, void* stream) {
    layout_init();
    int rc = check_common(d, p, cg, kg, workspace, workspace_floats);
    if (rc) return rc;
    if (!grads || (adam && (!adam->params || !adam->m || !adam->v))) return GCNN_E_BADARG;
    // The optimizer step rides in the reduction launch when that launch (re)writes every trainable gradient; otherwise
    // (degenerate batches) it runs as its own launch at the end -- either way the caller gets backward + Adam.
    auto adam_after = [&]() -> int {
        return adam ? gcnn_adam_step(adam->params, grads, adam->m, adam->v, g_ptotal, adam->lr_t, adam->beta1, adam->beta2, adam->eps,
                                     nullptr, 0, stream) : 0;
    };
    // d_scores == NULL: the loss head already ran inside gcnn_forward_loss (dO1pre and its partials are in the workspace)
    const bool fused_head = d_scores == nullptr;
    if (loss_out && (!fused_head || d->n_cuts <= 0)) HIPCHK(hipMemsetAsync(loss_out, 0, sizeof(float), (hipStream_t)stream));
    // data-parallel callers all-reduce [gradients | cut count]: the count is stored by a backward kernel
    if (cut_count_out && d->n_cuts <= 0) HIPCHK(hipMemsetAsync(cut_count_out, 0, sizeof(float), (hipStream_t)stream));
    hipStream_t st = (hipStream_t)stream;
    Work w; carve(d, workspace, &w);
    const Acts &A = w.a, &G = w.g;
    JobList jl; memset(&jl, 0, sizeof(jl)); jl.wg.partial = w.partial;

    // the reduction (re)writes every trainable gradient whenever all three node sets are non-empty; otherwise start from 0
    if (d->n_cons <= 0 || d->n_vars <= 0 || d->n_cuts <= 0) HIPCHK(hipMemsetAsync(grads, 0, (size_t)g_ptotal * sizeof(float), st));
    if (d->n_cuts <= 0) return adam_after();  // no cut => every gradient is 0
    ConvIO cv[3]; conv_setup(cv, d, w, cg, kg);
    struct { const float* x; const mask16* me1; float* gx; float* ge1; int n; int pb; int f; } em[3] = {
        {cons_feats, w.m.E1c, G.Xc, G.E1c, d->n_cons, P_CONS, 4},
        {var_feats, w.m.E1v, G.Xv, G.E1v, d->n_vars, P_VAR, 14},
        {cut_feats, w.m.E1k, G.Xk, G.E1k, d->n_cuts, P_CUT, 6}};
    // Dense(64->1) gradient (model.py:208): G.O1 = dscore (x) w2 masked by O1 > 0; dw2/db2 partials
    int head_parts = w.score_nblk;
    if (fused_head) {
        head_parts = cdiv(d->n_cuts, 16);   // one partial per tile of the readout program
        if (loss_out) add_rd(jl, w.score_partial + EMB + 1, loss_out, head_parts, HEAD_SLAB, 1);
        jl.rd.cdst = cut_count_out; jl.rd.cval = (float)d->n_cuts;
    } else {
        ProfScope prof("k_score_bwd", st);
        hipLaunchKernelGGL(k_score_bwd, dim3(w.score_nblk), dim3(256), 0, st, d_scores, A.O1, p + poff(P_OUT + 2), G.O1,
                           w.score_partial, cut_count_out, d->n_cuts);
        LAUNCHCHK();
    }
    add_rd(jl, w.score_partial, grads + poff(P_OUT + 2), head_parts, HEAD_SLAB, EMB);
    add_rd(jl, w.score_partial + EMB, grads + poff(P_OUT + 3), head_parts, HEAD_SLAB, 1);
    add_wg(jl, A.Xk2, nullptr, G.O1, nullptr, d->n_cuts, grads + poff(P_OUT), grads + poff(P_OUT + 1), nullptr, w.partial);
    auto tail = [&](TailBArgs& t, const float* in_a, const float* wa, const float* in_b, const float* wb, float* gx, const mask16* mx,
                    int pb, float* ge1, int n) {
        t.in_a = in_a; t.wa = wa; t.in_b = in_b; t.wb = wb; t.add = gx; t.m_x = mx; t.g_x = gx; t.w2 = p + poff(pb + E_W2);
        t.g_e1 = ge1; t.n = n;
    };
    if (!fused_head) {   // cut rows: readout -> conv v->k receiver gradients (fused head: gcnn_forward_loss's last launch did it)
        ConvBGroupArgs m; memset(&m, 0, sizeof(m));
        m.cb = conv_bwd_args(p, cv[2], G.O1, p + poff(P_OUT));
        if ((rc = launch_conv_bwd(m, st))) return rc;
    }
    if ((rc = conv_backward_edges(p, grads, cv[2], w, jl, st))) return rc;
    {   // variable rows: dXv2 = dPR3 Wr3^T (mask Xv2) -> conv c->v receiver gradients; in the same launch the cut rows' tail:
        // dXk = dXk(W1b part) + dPL3 Wl3^T, masked by Xk; dE1k
        ConvBGroupArgs m; memset(&m, 0, sizeof(m));
        m.cb = conv_bwd_args(p, cv[1], G.PR3, p + poff(P_CONV2 + C_WR));
        tail(m.tail, G.PL3, p + poff(P_CONV2 + C_WL), nullptr, nullptr, G.Xk, w.m.Xk, P_CUT, G.E1k, d->n_cuts);
        if ((rc = launch_conv_bwd(m, st))) return rc;
    }
    if ((rc = conv_backward_edges(p, grads, cv[1], w, jl, st))) return rc;
    {   // constraint rows: dXc2 = dPL2 Wl2^T (mask Xc2) -> conv v->c receiver gradients
        ConvBGroupArgs m; memset(&m, 0, sizeof(m));
        m.cb = conv_bwd_args(p, cv[0], G.PL2, p + poff(P_CONV1 + C_WL));
        if ((rc = launch_conv_bwd(m, st))) return rc;
    }
    if ((rc = conv_backward_edges(p, grads, cv[0], w, jl, st))) return rc;
    {   // variable rows: dXv = dXv(W1b part) + dPR2 Wr2^T + dPR1 Wr1^T, masked by Xv; dE1v; in the same launch the
        // constraint rows' tail: dXc = dXc(W1b part) + dPL1 Wl1^T, masked by Xc; dE1c
        TailGroupArgs m; memset(&m, 0, sizeof(m));
        tail(m.a, G.PR2, p + poff(P_CONV1 + C_WR), G.PR1, p + poff(P_CONV0 + C_WR), G.Xv, w.m.Xv, P_VAR, G.E1v, d->n_vars);
        tail(m.b, G.PL1, p + poff(P_CONV0 + C_WL), nullptr, nullptr, G.Xc, w.m.Xc, P_CONS, G.E1c, d->n_cons);
        if ((rc = launch_tail_bwd(m, st))) return rc;
    }
    // Weight gradients: every operand pair now exists, so ALL of them go out as two grouped launches -- the 22 [64,64]
    // products and the three first embedding layers ([f,64], f <= 14: a quarter of the MFMAs), then the fixed-order reduction of
    // the slabs.  (Running them beside the critical path on side streams was slower: a cross-stream event edge costs
    // 7-14 us here.)
    for (int i = 0; i < 3; ++i)   // kernel [f,64] and bias [64] are separate (4-float aligned) tensors in the layout
        add_wg_emb1(jl, em[i].x, p + poff(em[i].pb + E_SHIFT), p + poff(em[i].pb + E_SCALE), em[i].ge1, em[i].me1, em[i].n, em[i].f,
                    grads + poff(em[i].pb + E_W1), grads + poff(em[i].pb + E_B1));
    for (int i = 0; i < 3; ++i)   // E1^T dX with E1 recomputed from the raw features: the forward pass does not store it
        add_wg_emb2(jl, em[i].x, p + poff(em[i].pb + E_SHIFT), p + poff(em[i].pb + E_SCALE), p + poff(em[i].pb + E_W1), p + poff(em[i].pb + E_B1),
                    em[i].gx, em[i].n, em[i].f, grads + poff(em[i].pb + E_W2), grads + poff(em[i].pb + E_B2));
    place_wg(jl, w.partial);
    if ((size_t)jl.nslab > wg_slabs(d)) return GCNN_E_WORKSPACE;
    for (int k = jl.ndw; k < 3; ++k) jl.dw.blk0[k + 1] = jl.dw.blk0[k];
    if (jl.wg.nblocks + jl.dw.blk0[3] > 0) {
        static PerDeviceOnce attr;
        const size_t smem = (size_t)WG_WAVES * WG_SLAB * sizeof(float);   // 67.6 KB: above the 64 KB default (+ 3.8 KB static, k_wgrad.hpp: the attribute bounds the sum by 160 KB)
        if (attr.first()) HIPCHK(hipFuncSetAttribute((const void*)k_wgrad, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        ProfScope prof("k_wgrad", st);
        hipLaunchKernelGGL(k_wgrad, dim3(jl.wg.nblocks + jl.dw.blk0[3]), dim3(64 * WG_WAVES), smem, st, jl.wg, jl.dw);
        LAUNCHCHK();
    }
    const bool fuse_adam = adam && d->n_cons > 0 && d->n_vars > 0 && jl.rdblk > 0;
    if (fuse_adam) jl.rd.adam = RdAdam{adam->params, adam->m, adam->v, grads, g_ptotal, adam->lr_t, adam->beta1, adam->beta2, adam->eps};
    if (jl.rdblk + jl.fold.n > 0) {   // front blocks: G1 | g2 -> gradients of Wf, bf, W1a (fold_block), with the same Adam update
        ProfScope prof(fuse_adam ? "k_reduce<adam>" : "k_reduce", st);
        hipLaunchKernelGGL(k_reduce, dim3(FOLD_BLOCKS * jl.fold.n + jl.rdblk), dim3(256), 0, st, jl.rd, jl.fold);
        LAUNCHCHK();
    }
    return fuse_adam ? 0 : adam_after();
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
                   float eps, const float* grad_scale, int32_t scale_is_divisor, void* stream) {
    if (n < 0 || (n > 0 && (!params || !grads || !m || !v))) return GCNN_E_BADARG;
    if (n == 0) return 0;
    ProfScope prof("k_adam", (hipStream_t)stream);
    hipLaunchKernelGGL(k_adam, dim3(std::min(cdiv(n, 256), 1024)), dim3(256), 0, (hipStream_t)stream, params, grads, m, v, n,
                       lr_t, beta1, beta2, eps, grad_scale, scale_is_divisor);
    LAUNCHCHK();
    return 0;
}

extern "C" int gcnn_adam_step_dev(float* params, const float* grads, float* m, float* v, int32_t n, float* opt_state,
                                  const float* grad_scale, int32_t scale_is_divisor, void* stream) {
    if (n < 0 || !opt_state || (n > 0 && (!params || !grads || !m || !v))) return GCNN_E_BADARG;
    hipLaunchKernelGGL(k_adam_tick, dim3(1), dim3(1), 0, (hipStream_t)stream, opt_state, grad_scale, scale_is_divisor);
    LAUNCHCHK();
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_adam_dev, dim3(std::min(cdiv(n, 256), 1024)), dim3(256), 0, (hipStream_t)stream, params, grads, m, v, n,
                       opt_state, grad_scale, scale_is_divisor);
    LAUNCHCHK();
    return 0;
}

// ---- ranking-prefix accuracy (model_trainer.py:280-302) on the device ------------------------------------------------
extern "C" int gcnn_ranking_metric(const float* pred, const float* truth, const int32_t* offsets, int32_t n_samples,
                                   int32_t max_cuts, const float* fractions, int32_t n_fractions, float* acc,
                                   float* frac_out, const float* loss_in, float loss_weight, float* loss_acc, void* stream) {
    if (n_samples < 0 || n_fractions < 0 || (n_samples > 0 && (!pred || !truth || !offsets))) return GCNN_E_BADARG;
    if (n_fractions > 0 && (!fractions || !acc)) return GCNN_E_BADARG;
    if (loss_acc && !loss_in) return GCNN_E_BADARG;
    if (max_cuts > RK_MAX) return GCNN_E_WORKSPACE;   // a sample with more cuts than the LDS sort holds: use the host metric
    if (n_samples == 0) return 0;
    hipLaunchKernelGGL(k_ranking, dim3(n_samples), dim3(256), 0, (hipStream_t)stream, pred, truth, offsets, fractions,
                       n_fractions, acc, frac_out, loss_in, loss_weight, loss_acc);
    LAUNCHCHK();
    return 0;
}
