// micro-benchmark: inner-loop variants of the forward / sender-backward edge passes on a setcov-500 x 32 shaped graph
// (developer tool).  Build it twice for an A/B on one box: once against a saved copy of the kernel header, once against the library's:
//   hipcc -O3 --offload-arch=gfx950 -DKHDR='"../ab/k_edge_base.hpp"' -o tools/micro/bench_edge2_base tools/micro/bench_edge2.hip
//   hipcc -O3 --offload-arch=gfx950 -o tools/micro/bench_edge2 tools/micro/bench_edge2.hip
// Both print times and checksums of every output (equal checksums = equal results up to summation order).
#ifndef KHDR
#define KHDR "../../gcnn-cut-selector_amd/csrc/k_edge.hpp"
#endif
#include KHDR
#ifndef LAUNCH_EXTRA   // trailing kernel arguments: the library's kernels take the number of long-segment blocks (-DLAUNCH_EXTRA= for a header without)
#define LAUNCH_EXTRA , 0
#endif
#include <cstdio>
#include <random>
#include <vector>
#include <numeric>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); return 1; } } while (0)
template <class T> static T* up(const std::vector<T>& v) { T* d; hipMalloc(&d, v.size() * sizeof(T) + 16); hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice); return d; }
static int cdiv(int a, int b) { return (a + b - 1) / b; }
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32, C = 500, V = 1000;
    std::mt19937 rng(1);
    std::vector<int> l_ptr{0}, l_oth, rows;
    for (int s = 0; s < B; ++s)
        for (int r = 0; r < C; ++r) {
            std::vector<int> pick(V); std::iota(pick.begin(), pick.end(), 0); std::shuffle(pick.begin(), pick.end(), rng);
            const int nnz = 25 + (int)(rng() % 52);
            std::vector<int> cs(pick.begin(), pick.begin() + nnz); std::sort(cs.begin(), cs.end());
            for (int c : cs) { l_oth.push_back(s * V + c); rows.push_back(s * C + r); }
            l_ptr.push_back((int)l_oth.size());
        }
    const int E = (int)l_oth.size(), NL = B * C, NV = B * V;
    std::vector<int> order(E); std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return l_oth[a] < l_oth[b]; });
    std::vector<int> v_ptr(NV + 1, 0), v_oth(E);
    for (int i = 0; i < E; ++i) { v_oth[i] = rows[order[i]]; v_ptr[l_oth[order[i]] + 1]++; }
    for (int i = 0; i < NV; ++i) v_ptr[i + 1] += v_ptr[i];
    std::vector<float> coef(E); for (auto& x : coef) x = (float)((int)(rng() % 2001) - 1000) / 1000.f;
    int *dl_ptr = up(l_ptr), *dl_oth = up(l_oth), *dv_ptr = up(v_ptr), *dv_oth = up(v_oth);
    float* dcoef = up(coef);
    float *PL, *PR, *DS, *S, *N, *S2, *N2, *Q, *Q2, *par;
    CK(hipMalloc(&PL, (size_t)NV * 256)); CK(hipMalloc(&PR, (size_t)NV * 256)); CK(hipMalloc(&DS, (size_t)NV * 256)); CK(hipMalloc(&S, (size_t)NV * 256)); CK(hipMalloc(&N, (size_t)NV * 256));
    CK(hipMalloc(&S2, (size_t)NV * 256)); CK(hipMalloc(&N2, (size_t)NV * 256)); CK(hipMalloc(&Q, (size_t)16384 * 256)); CK(hipMalloc(&Q2, (size_t)16384 * 256)); CK(hipMalloc(&par, 1024));
    std::vector<float> rnd((size_t)NV * 64);
    for (float* dst : {PL, PR, DS}) { for (auto& x : rnd) x = (float)((int)(rng() % 2001) - 1000) / 1000.f; CK(hipMemcpy(dst, rnd.data(), rnd.size() * 4, hipMemcpyHostToDevice)); }
    std::vector<float> pv(256, 0.f); for (int i = 0; i < 64; ++i) pv[i] = (float)((int)(rng() % 2001) - 1000) / 1000.f;
    pv[64] = 0.1f; pv[65] = 1.5f; pv[66] = 0.7f;   // w_edge[0..63], e_shift, e_scale, s1
    CK(hipMemcpy(par, pv.data(), 1024, hipMemcpyHostToDevice));
    printf("B=%d E=%d\n", B, E);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < 5; ++i) launch();
        hipEventRecord(e0, 0);
        for (int i = 0; i < 100; ++i) launch();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %.2f us%s\n", name, ms * 10, hipGetLastError() == hipSuccess ? "" : "  LAUNCH ERROR");
    };
    std::vector<float> ha((size_t)NV * 64), hb((size_t)NV * 64), hc((size_t)NV * 64), hd((size_t)NV * 64);
    auto snap = [&](float* d, std::vector<float>& h, int rows) { hipDeviceSynchronize(); hipMemcpy(h.data(), d, (size_t)rows * 256, hipMemcpyDeviceToHost); };
    auto cmp = [&](const char* what, const std::vector<float>& x, const std::vector<float>& y, int rows) {
        double md = 0, mx = 0; for (size_t i = 0; i < (size_t)rows * 64; ++i) { md = std::max(md, (double)fabsf(x[i] - y[i])); mx = std::max(mx, (double)fabsf(x[i])); }
        printf("  check %-28s max|diff| %.3g (max|ref| %.3g)\n", what, md, mx);
    };
    auto csum = [&](const char* what, float* d, int rows) {
        snap(d, ha, rows); double t = 0, q = 0; for (size_t i = 0; i < (size_t)rows * 64; ++i) { t += ha[i]; q += fabs((double)ha[i]); }
        printf("  %-14s sum %.6e  sum|.| %.6e\n", what, t, q);
    };
    for (int dir = 0; dir < 2; ++dir) {   // 0: owners = constraints (deg ~50); 1: owners = variables (deg ~25)
        char nm[96];
        EdgeArgs a; memset(&a, 0, sizeof(a));
        a.seg_ptr = dir ? dv_ptr : dl_ptr; a.oth = dir ? dv_oth : dl_oth; a.coef = dcoef; a.p_own = dir ? PR : PL; a.p_oth = dir ? PL : PR;
        a.w_edge = par; a.e_shift = par + 64; a.e_scale = par + 65; a.s1 = par + 66; a.out = S; a.cnt_rows = N; a.d_s = DS; a.dw_partial = Q;
        a.n_own = dir ? NV : NL;
#define AB(SL)                                                                                                                       \
        {                                                                                                                            \
            const int grid = std::min(cdiv(cdiv(a.n_own, 4 / SL), 4), 8192);                                                         \
            snprintf(nm, 96, "fwd<count> dir=%d slots=%d", dir, SL); timeit(nm, [&] { hipLaunchKernelGGL((k_edge_fwd<SL, true>), dim3(grid), dim3(256), 0, 0, a LAUNCH_EXTRA); }); \
            csum("S", S, a.n_own); csum("N", N, a.n_own);                                                                            \
            snprintf(nm, 96, "fwd dir=%d slots=%d", dir, SL); timeit(nm, [&] { hipLaunchKernelGGL((k_edge_fwd<SL, false>), dim3(grid), dim3(256), 0, 0, a LAUNCH_EXTRA); }); \
            csum("S (no count)", S, a.n_own);                                                                                        \
            snprintf(nm, 96, "bwd_send dir=%d slots=%d", dir, SL); timeit(nm, [&] { hipLaunchKernelGGL((k_edge_bwd_send<SL>), dim3(grid), dim3(256), 0, 0, a LAUNCH_EXTRA); }); \
            csum("dP_send", S, a.n_own); csum("dw partials", Q, grid);                                                               \
        }
        AB(4) AB(2) AB(1)
    }
    return 0;
}
