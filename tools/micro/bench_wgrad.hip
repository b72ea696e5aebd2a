// micro-benchmark of k_wgrad job shapes (developer tool; not part of the library)
#include "../../gcnn-cut-selector_amd/csrc/k_wgrad.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); return 1; } } while (0)
static int cdiv(int a, int b) { return (a + b - 1) / b; }
int main(int argc, char** argv) {
    const int NMAX = 32000, NJ = 24;
    const bool rnd = argc > 1 && atoi(argv[1]) != 0;   // random operands (default: zeros)
    float *x, *d, *partial, *sx, *feat, *epart; int* seg;
    const size_t nel = (size_t)NJ * NMAX * 64;
    CK(hipMalloc(&x, nel * 4)); CK(hipMalloc(&d, nel * 4));
    CK(hipMalloc(&partial, (size_t)8192 * WG_SLAB * 4)); CK(hipMalloc(&sx, 64)); CK(hipMalloc(&seg, (NMAX + 1) * 4));
    CK(hipMalloc(&feat, (size_t)NMAX * 16 * 4)); CK(hipMalloc(&epart, (size_t)1024 * 15 * 64 * 4));
    CK(hipMemset(x, 0, nel * 4)); CK(hipMemset(d, 0, nel * 4)); CK(hipMemset(feat, 0, (size_t)NMAX * 16 * 4));
    CK(hipMemset(seg, 0, (NMAX + 1) * 4)); CK(hipMemset(sx, 0, 64));
    if (rnd) {
        std::vector<float> h(nel);
        for (size_t i = 0; i < nel; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
        CK(hipMemcpy(x, h.data(), nel * 4, hipMemcpyHostToDevice));
        for (size_t i = 0; i < nel; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
        CK(hipMemcpy(d, h.data(), nel * 4, hipMemcpyHostToDevice));
    }
    struct Cfg { const char* name; int njobs; int n[NJ]; int rows; int pad8; int emb; };
    const int C = 16000, V = 32000, K = 1893;
#define STEP {V, V, V, V, V, V, V, V, C, C, C, C, C, C, C, K, K, K, K, K, K, K}
    Cfg cfgs[] = {
        {"1 job 32k", 1, {V}, 128, 0, 0},
        {"1 job 2k", 1, {K}, 128, 0, 0},
        {"5 jobs 32k", 5, {V, V, V, V, V}, 128, 0, 0},
        {"step r=128", 22, STEP, 128, 0, 0},
        {"step r=128 pad8", 22, STEP, 128, 1, 0},
        {"step r=128 +emb1", 22, STEP, 128, 0, 1},
        {"step r=192", 22, STEP, 192, 0, 0},
        {"step r=192 +emb1", 22, STEP, 192, 0, 1},
        {"step r=256", 22, STEP, 256, 0, 0},
        {"step r=96", 22, STEP, 96, 0, 0},
        {"step r=160 +emb1", 22, STEP, 160, 0, 1},
        {"step r=176 +emb1", 22, STEP, 176, 0, 1},
        {"step r=208 +emb1", 22, STEP, 208, 0, 1},
        {"step r=224 +emb1", 22, STEP, 224, 0, 1},
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)k_wgrad, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (auto& c : cfgs) {
        WgArgs a; a.njobs = c.njobs; a.partial = partial; int blk = 0, slab = 0;
        for (int j = 0; j < c.njobs; ++j) {
            const int nb = cdiv(c.n[j], c.rows * WG_WAVES);
            a.job[j] = WgJob{x + (size_t)j * NMAX * 64, nullptr, d + (size_t)j * NMAX * 64, nullptr, nullptr, nullptr, nullptr, c.n[j], blk, slab, 0,
                             nb, (cdiv(c.n[j], nb * WG_WAVES) + 15) & ~15};
            blk += c.pad8 ? (nb + 7) & ~7 : nb; slab += nb;
        }
        a.nblocks = blk;
        int nemb = 0;
        if (c.emb) {   // the three first embedding layers as jobs of their own (EXTRA == 2)
            const int ns[3] = {V, C, K}, fs[3] = {14, 4, 6};
            for (int i = 0; i < 3; ++i) {
                const int j = a.njobs++;
                const int nb = cdiv(ns[i], c.rows * WG_WAVES);
                a.job[j] = WgJob{feat, nullptr, d + (size_t)i * NMAX * 64, nullptr, (const unsigned short*)(x + (size_t)i * NMAX * 64), sx, sx, ns[i], blk, slab, fs[i],   // mask: 8 B of pattern bits per row
                                 nb, (cdiv(ns[i], nb * WG_WAVES) + 15) & ~15};
                blk += c.pad8 ? (nb + 7) & ~7 : nb; slab += nb; nemb += nb;
            }
            a.nblocks = blk;
        }
        const int grid = blk;
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_wgrad, dim3(grid), dim3(64 * WG_WAVES), WG_WAVES * WG_SLAB * sizeof(float), 0, a, DwRedArgs{});
        CK(hipEventRecord(e0, 0));
        const int R = 50;
        for (int i = 0; i < R; ++i) hipLaunchKernelGGL(k_wgrad, dim3(grid), dim3(64 * WG_WAVES), WG_WAVES * WG_SLAB * sizeof(float), 0, a, DwRedArgs{});
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-20s blocks %4d (%d emb1)  %.2f us/launch\n", c.name, blk, nemb, ms * 1000 / R);
    }
    return 0;
}
