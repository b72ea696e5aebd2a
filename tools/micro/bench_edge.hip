// micro-benchmark of the edge passes on a setcov-500 x 32 shaped graph (developer tool; not part of the library)
#include "../../gcnn-cut-selector_amd/csrc/k_edge.hpp"
namespace r01 {
#include "old/k_edge_r01.hpp"
}
#include <cstdio>
#include <random>
#include <vector>
#include <numeric>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); return 1; } } while (0)
template <class T> static T* up(const std::vector<T>& v) { T* d; hipMalloc(&d, v.size() * sizeof(T) + 16); hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice); return d; }
static int cdiv(int a, int b) { return (a + b - 1) / b; }
int main() {
    const int B = 32, C = 500, V = 1000, NNZ = 50;
    std::mt19937 rng(1);
    std::vector<int> l_ptr{0}, l_oth, rows, cols;
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
    std::vector<int> v_ptr(NV + 1, 0), v_oth(E), v2l(E), l2v(E);
    for (int i = 0; i < E; ++i) { v_oth[i] = rows[order[i]]; v2l[i] = order[i]; l2v[order[i]] = i; v_ptr[l_oth[order[i]] + 1]++; }
    for (int i = 0; i < NV; ++i) v_ptr[i + 1] += v_ptr[i];
    std::vector<float> coef(E, 0.1f);
    int *dl_ptr = up(l_ptr), *dl_oth = up(l_oth), *dv_ptr = up(v_ptr), *dv_oth = up(v_oth), *dv2l = up(v2l), *dl2v = up(l2v);
    float* dcoef = up(coef);
    float *PL, *PR, *S, *S2, *N, *Q, *par; unsigned long long* mask;
    CK(hipMalloc(&PL, (size_t)NV * 256)); CK(hipMalloc(&PR, (size_t)NV * 256)); CK(hipMalloc(&S, (size_t)NV * 256)); CK(hipMalloc(&S2, (size_t)NV * 256)); CK(hipMemset(S2, 0, (size_t)NV * 256));
    CK(hipMalloc(&N, (size_t)NV * 256)); CK(hipMalloc(&Q, (size_t)NV * 256)); CK(hipMalloc(&mask, (size_t)E * 8)); CK(hipMalloc(&par, 1024));
    std::vector<float> rnd((size_t)NV * 64); for (auto& x : rnd) x = (float)((int)(rng() % 2001) - 1000) / 1000.f;
    CK(hipMemcpy(PL, rnd.data(), rnd.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(PR, rnd.data(), rnd.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> pv(256, 0.f); pv[64] = 0.f; pv[65] = 1.f; pv[66] = 1.f;   // w_edge[0..63], e_shift, e_scale, s1
    CK(hipMemcpy(par, pv.data(), 1024, hipMemcpyHostToDevice));
    printf("E=%d\n", E);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < 5; ++i) launch();
        hipEventRecord(e0, 0);
        for (int i = 0; i < 50; ++i) launch();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-40s %.2f us\n", name, ms * 20);
    };
    std::vector<float> ha((size_t)NV * 64), hb((size_t)NV * 64);
    auto snap = [&](float* d, std::vector<float>& h, int rows) { hipDeviceSynchronize(); hipMemcpy(h.data(), d, (size_t)rows * 256, hipMemcpyDeviceToHost); };
    auto cmp = [&](const char* what, const std::vector<float>& x, const std::vector<float>& y, int rows) {
        double md = 0, mx = 0; for (size_t i = 0; i < (size_t)rows * 64; ++i) { md = std::max(md, (double)fabsf(x[i] - y[i])); mx = std::max(mx, (double)fabsf(x[i])); }
        printf("  check %-28s max|diff| %.3g (max|ref| %.3g)\n", what, md, mx);
    };
    std::vector<float> rS, rN, rO, rQ, nS, nN, nO, nQ;
    for (int dir = 0; dir < 2; ++dir) {   // 0: receivers = constraints (deg ~50); 1: receivers = variables (deg ~25)
        char nm[64];
        {   // round-1 kernels: per-edge 64-bit ReLU masks written by the forward, gathered through xpos by the send pass
            r01::EdgeArgs a; memset(&a, 0, sizeof(a));
            a.seg_ptr = dir ? dv_ptr : dl_ptr; a.oth = dir ? dv_oth : dl_oth; a.coef = dcoef; a.p_recv = dir ? PR : PL; a.p_oth = dir ? PL : PR;
            a.w_edge = par; a.e_shift = par + 64; a.e_scale = par + 65; a.s1 = par + 66; a.out = S; a.mask = mask; a.cnt_rows = N;
            a.n_recv = dir ? NV : NL;
            const int grid = std::min(cdiv(a.n_recv, 4), 2048);
            snprintf(nm, 64, "r01 fwd dir=%d slots=4 save(mask+N)", dir); timeit(nm, [&] { hipLaunchKernelGGL((r01::k_edge_fwd<4, true>), dim3(grid), dim3(256), 0, 0, a); });
            snprintf(nm, 64, "r01 fwd dir=%d slots=4 nosave", dir); timeit(nm, [&] { hipLaunchKernelGGL((r01::k_edge_fwd<4, false>), dim3(grid), dim3(256), 0, 0, a); });
            r01::EdgeArgs b; memset(&b, 0, sizeof(b));
            b.seg_ptr = dir ? dl_ptr : dv_ptr; b.oth = dir ? dl_oth : dv_oth; b.coef = dcoef; b.e_shift = par + 64; b.e_scale = par + 65; b.s1 = par + 66;
            b.d_s = PL; b.xpos = dir ? dl2v : dv2l; b.mask = mask; b.out = S; b.dw_rows = Q; b.n_recv = dir ? NL : NV;
            const int gridb = std::min(cdiv(b.n_recv, 4), 2048);
            hipLaunchKernelGGL((r01::k_edge_fwd<4, true>), dim3(grid), dim3(256), 0, 0, a);
            rS = ha; rN = ha; snap(S, rS, a.n_recv); snap(N, rN, a.n_recv);
            snprintf(nm, 64, "r01 bwd_send dir=%d slots=4 (mask)", dir); timeit(nm, [&] { hipLaunchKernelGGL((r01::k_edge_bwd_send<4>), dim3(gridb), dim3(256), 0, 0, b); });
            rO = ha; rQ = ha; snap(S, rO, b.n_recv); snap(Q, rQ, b.n_recv);
        }
        EdgeArgs a; memset(&a, 0, sizeof(a));
        a.seg_ptr = dir ? dv_ptr : dl_ptr; a.oth = dir ? dv_oth : dl_oth; a.coef = dcoef; a.p_own = dir ? PR : PL; a.p_oth = dir ? PL : PR;
        a.w_edge = par; a.e_shift = par + 64; a.e_scale = par + 65; a.s1 = par + 66; a.out = S; a.cnt_rows = N;
        a.n_own = dir ? NV : NL;
        const int grid = std::min(cdiv(a.n_own, 4), 2048), grid2 = std::min(cdiv(cdiv(a.n_own, 2), 4), 2048);
        snprintf(nm, 64, "fwd dir=%d slots=4 count", dir); timeit(nm, [&] { hipLaunchKernelGGL((k_edge_fwd<4, true>), dim3(grid), dim3(256), 0, 0, a); });
        snprintf(nm, 64, "fwd dir=%d slots=4 nocount", dir); timeit(nm, [&] { hipLaunchKernelGGL((k_edge_fwd<4, false>), dim3(grid), dim3(256), 0, 0, a); });
        snprintf(nm, 64, "fwd dir=%d slots=2 count", dir); timeit(nm, [&] { hipLaunchKernelGGL((k_edge_fwd<2, true>), dim3(grid2), dim3(256), 0, 0, a); });
        snprintf(nm, 64, "fwd dir=%d slots=2 nocount", dir); timeit(nm, [&] { hipLaunchKernelGGL((k_edge_fwd<2, false>), dim3(grid2), dim3(256), 0, 0, a); });
        { const int grid1 = std::min(cdiv(cdiv(a.n_own, 4), 4), 2048);
          snprintf(nm, 64, "fwd dir=%d slots=1 count", dir); timeit(nm, [&] { hipLaunchKernelGGL((k_edge_fwd<1, true>), dim3(grid1), dim3(256), 0, 0, a); });
          snprintf(nm, 64, "fwd dir=%d slots=1 nocount", dir); timeit(nm, [&] { hipLaunchKernelGGL((k_edge_fwd<1, false>), dim3(grid1), dim3(256), 0, 0, a); }); }
        hipLaunchKernelGGL((k_edge_fwd<4, true>), dim3(grid), dim3(256), 0, 0, a);
        nS = ha; nN = ha; snap(S, nS, a.n_own); snap(N, nN, a.n_own); cmp("fwd S slots=4", rS, nS, a.n_own); cmp("fwd N slots=4", rN, nN, a.n_own);
        hipLaunchKernelGGL((k_edge_fwd<2, true>), dim3(grid2), dim3(256), 0, 0, a);
        snap(S, nS, a.n_own); snap(N, nN, a.n_own); cmp("fwd S slots=2", rS, nS, a.n_own); cmp("fwd N slots=2", rN, nN, a.n_own);
        hipLaunchKernelGGL((k_edge_fwd<1, true>), dim3(std::min(cdiv(cdiv(a.n_own, 4), 4), 2048)), dim3(256), 0, 0, a);
        snap(S, nS, a.n_own); snap(N, nN, a.n_own); cmp("fwd S slots=1", rS, nS, a.n_own); cmp("fwd N slots=1", rN, nN, a.n_own);
        // send pass: owner = the other side; gathers dS and P_recv rows of the forward's receivers
        EdgeArgs b; memset(&b, 0, sizeof(b));
        b.seg_ptr = dir ? dl_ptr : dv_ptr; b.oth = dir ? dl_oth : dv_oth; b.coef = dcoef; b.w_edge = par; b.e_shift = par + 64; b.e_scale = par + 65; b.s1 = par + 66;
        b.d_s = PL; b.p_own = dir ? PL : PR; b.p_oth = dir ? PR : PL; b.out = S; b.dw_partial = Q; b.n_own = dir ? NL : NV;
        const int gridb = std::min(cdiv(b.n_own, 4), 2048), gridb2 = std::min(cdiv(cdiv(b.n_own, 2), 4), 2048);
        snprintf(nm, 64, "bwd_send dir=%d slots=4 (recompute)", dir); timeit(nm, [&] { hipLaunchKernelGGL((k_edge_bwd_send<4>), dim3(gridb), dim3(256), 0, 0, b); });
        snprintf(nm, 64, "bwd_send dir=%d slots=2 (recompute)", dir); timeit(nm, [&] { hipLaunchKernelGGL((k_edge_bwd_send<2>), dim3(gridb2), dim3(256), 0, 0, b); });
        snprintf(nm, 64, "bwd_send dir=%d slots=1 (recompute)", dir); timeit(nm, [&] { hipLaunchKernelGGL((k_edge_bwd_send<1>), dim3(std::min(cdiv(cdiv(b.n_own, 4), 4), 2048)), dim3(256), 0, 0, b); });
        hipLaunchKernelGGL((k_edge_bwd_send<4>), dim3(gridb), dim3(256), 0, 0, b);
        nO = ha; nQ = ha; snap(S, nO, b.n_own); cmp("bwd dP slots=4", rO, nO, b.n_own);
        hipLaunchKernelGGL((k_edge_bwd_send<1>), dim3(std::min(cdiv(cdiv(b.n_own, 4), 4), 2048)), dim3(256), 0, 0, b);
        snap(S, nO, b.n_own); cmp("bwd dP slots=1", rO, nO, b.n_own);
    }
    return 0;
}
