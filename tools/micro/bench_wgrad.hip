// micro-benchmark of k_wgrad job shapes (developer tool; not part of the library)
#include "../../gcnn-cut-selector_amd/csrc/k_wgrad.hpp"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); return 1; } } while (0)
static int cdiv(int a, int b) { return (a + b - 1) / b; }
int main() {
    const int NMAX = 32000;
    float *x, *d, *q, *partial, *sx; int* seg;
    CK(hipMalloc(&x, (size_t)6 * NMAX * 64 * 4)); CK(hipMalloc(&d, (size_t)6 * NMAX * 64 * 4)); CK(hipMalloc(&q, (size_t)NMAX * 64 * 4));
    CK(hipMalloc(&partial, (size_t)4096 * WG_SLAB * 4)); CK(hipMalloc(&sx, 4)); CK(hipMalloc(&seg, (NMAX + 1) * 4));
    CK(hipMemset(x, 0, (size_t)6 * NMAX * 64 * 4)); CK(hipMemset(d, 0, (size_t)6 * NMAX * 64 * 4)); CK(hipMemset(q, 0, (size_t)NMAX * 64 * 4));
    CK(hipMemset(seg, 0, (NMAX + 1) * 4)); CK(hipMemset(sx, 0, 4));
    struct Cfg { const char* name; int njobs; int n[6]; int extra[6]; };
    Cfg cfgs[] = {
        {"1 plain job 32k", 1, {32000}, {0}},
        {"1 deg job 32k", 1, {32000}, {1}},
        {"1 q job 32k", 1, {32000}, {2}},
        {"1 plain job 16k", 1, {16000}, {0}},
        {"1 plain job 2k", 1, {1893}, {0}},
        {"5 plain jobs 32k", 5, {32000, 32000, 32000, 32000, 32000}, {0, 0, 0, 0, 0}},
        {"conv2-like", 6, {32000, 32000, 32000, 32000, 32000, 16000}, {0, 0, 0, 1, 0, 2}},
        {"conv3-like", 6, {1893, 1893, 1893, 1893, 1893, 32000}, {0, 0, 0, 1, 0, 2}},
        {"embed-like", 3, {16000, 32000, 1893}, {0, 0, 0}},
        {"conv3 no extras", 6, {1893, 1893, 1893, 1893, 1893, 32000}, {0, 0, 0, 0, 0, 0}},
        {"conv3 only q", 6, {1893, 1893, 1893, 1893, 1893, 32000}, {0, 0, 0, 0, 0, 2}},
        {"conv3 only deg", 6, {1893, 1893, 1893, 1893, 1893, 32000}, {0, 0, 0, 1, 0, 0}},
        {"1 deg job 2k", 1, {1893}, {1}},
        {"1 q job 2k", 1, {1893}, {2}},
        {"2 plain 32k", 2, {32000, 32000}, {0, 0}},
        {"3 plain 32k", 3, {32000, 32000, 32000}, {0, 0, 0}},
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)k_wgrad, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (auto& c : cfgs) {
        WgArgs a; a.njobs = c.njobs; a.partial = partial; a.rows_per_wave = WG_ROWS; int blk = 0;
        for (int j = 0; j < c.njobs; ++j) {
            a.job[j] = WgJob{x + (size_t)j * NMAX * 64, j == 1 ? sx : nullptr, d + (size_t)j * NMAX * 64, c.extra[j] == 1 ? seg : nullptr,
                             c.n[j], blk, blk};
            blk += cdiv(c.n[j], WG_ROWS * WG_WAVES);
        }
        a.nblocks = blk;
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_wgrad, dim3(blk), dim3(64 * WG_WAVES), WG_WAVES * WG_SLAB * sizeof(float), 0, a, Emb1Args{}, DwRedArgs{});
        CK(hipEventRecord(e0, 0));
        const int R = 50;
        for (int i = 0; i < R; ++i) hipLaunchKernelGGL(k_wgrad, dim3(blk), dim3(64 * WG_WAVES), WG_WAVES * WG_SLAB * sizeof(float), 0, a, Emb1Args{}, DwRedArgs{});
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-20s blocks %4d  %.2f us/launch\n", c.name, blk, ms * 1000 / R);
    }
    return 0;
}
