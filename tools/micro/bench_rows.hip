// Does a row program pay for instruction fetch when other kernels ran since its last launch?  (developer probe)
// k_conv_fwd<4, CF_PROJ> on 119 tiles (the K-phase shape): launched back to back vs in rotation with seven other row-program
// instantiations (~100 KB of code between two launches of the same kernel, as in a training step).
#include "../../gcnn-cut-selector_amd/csrc/k_rows.hpp"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); return 1; } } while (0)
int main() {
    const int n = 1904;
    float* buf; CK(hipMalloc(&buf, (size_t)64 << 20)); CK(hipMemset(buf, 0, (size_t)64 << 20));
    int* seg; CK(hipMalloc(&seg, (n + 1) * 4)); CK(hipMemset(seg, 0, (n + 1) * 4));
    auto mat = [&](int i) { return buf + (size_t)i * n * 64; };
    float* w = buf + (size_t)40 * n * 64;
    ConvFArgs a; memset(&a, 0, sizeof(a));
    a.s = mat(0); a.seg_ptr = seg; a.wf = w; a.bf = w + 4096; a.a_out = mat(1); a.s2 = w + 5000; a.xrecv = mat(2); a.w1a = w + 8192; a.w1b = w + 12288;
    a.b1 = w + 4200; a.z1 = mat(3); a.w2 = w + 16384; a.b2 = w + 4300; a.out = mat(4); a.wt = w + 20480; a.bt = w + 4400; a.t_out = mat(5);
    a.ws = w + 4500; a.bs = w + 4600; a.scores = mat(6); a.targets = mat(7); a.loss_scale = 1.f; a.g_o1 = mat(8); a.head_partial = mat(9); a.n = n;
    EmbGroupArgs m; memset(&m, 0, sizeof(m));
    auto emb = [&](EmbArgs& e, int f, int k) { e.x = mat(10 + k); e.shift = w + 4700; e.scale = w + 4800; e.w1 = w + 24576; e.b1 = w + 4200; e.e1 = mat(13 + k); e.w2 = w; e.b2 = w + 4300; e.xo = mat(16 + k); e.wp[0] = w + 8192; e.po[0] = mat(19 + k); e.wp[1] = w + 12288; e.po[1] = mat(22 + k); e.n = n; (void)f; };
    emb(m.v, 14, 0); emb(m.c, 4, 1); emb(m.k, 6, 2);
    m.blk0[0] = 0; m.blk0[1] = 30; m.blk0[2] = 60; m.blk0[3] = 90;
    const size_t smem = ROWS_LDS_FLOATS(5, 5) * sizeof(float), esmem = EMB_LDS_FLOATS * sizeof(float);
#define ATTR(K) CK(hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024))
    ATTR((k_conv_fwd<4, CF_PROJ>)); ATTR((k_conv_fwd<8, CF_PROJ>)); ATTR((k_conv_fwd<4, CF_READOUT>)); ATTR((k_conv_fwd<8, CF_READOUT>));
    ATTR((k_conv_fwd<4, CF_LOSS>)); ATTR((k_conv_fwd<8, CF_LOSS>)); ATTR(k_embed_fwd<4>); ATTR(k_embed_fwd<8>);
    auto target = [&] { hipLaunchKernelGGL((k_conv_fwd<4, CF_PROJ>), dim3(30), dim3(256), smem, 0, a); };
    auto others = [&] {
        hipLaunchKernelGGL((k_conv_fwd<8, CF_PROJ>), dim3(15), dim3(512), smem, 0, a);
        hipLaunchKernelGGL((k_conv_fwd<4, CF_READOUT>), dim3(30), dim3(256), smem, 0, a);
        hipLaunchKernelGGL((k_conv_fwd<8, CF_READOUT>), dim3(15), dim3(512), smem, 0, a);
        hipLaunchKernelGGL((k_conv_fwd<4, CF_LOSS>), dim3(30), dim3(256), smem, 0, a);
        hipLaunchKernelGGL((k_conv_fwd<8, CF_LOSS>), dim3(15), dim3(512), smem, 0, a);
        hipLaunchKernelGGL(k_embed_fwd<4>, dim3(90), dim3(256), esmem, 0, m);
        hipLaunchKernelGGL(k_embed_fwd<8>, dim3(90), dim3(512), esmem, 0, m);
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int i = 0; i < 5; ++i) { target(); others(); }
    CK(hipEventRecord(e0, 0)); for (int i = 0; i < 100; ++i) target(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); const float back = ms * 10;
    CK(hipEventRecord(e0, 0)); for (int i = 0; i < 100; ++i) others(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); const float oth = ms * 10;
    CK(hipEventRecord(e0, 0)); for (int i = 0; i < 100; ++i) { target(); others(); } CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); const float rot = ms * 10;
    printf("k_conv_fwd<4,PROJ> back to back: %.2f us per launch\n", back);
    {   // the same launch with no rows: kernel boundary + weight staging + barrier only
        ConvFArgs z = a; z.n = 0;
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_conv_fwd<4, CF_PROJ>), dim3(30), dim3(256), smem, 0, z);
        CK(hipEventRecord(e0, 0)); for (int i = 0; i < 100; ++i) hipLaunchKernelGGL((k_conv_fwd<4, CF_PROJ>), dim3(30), dim3(256), smem, 0, z);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  ... with n = 0 (launch + weight staging only): %.2f us per launch\n", ms * 10);
        CK(hipEventRecord(e0, 0)); for (int i = 0; i < 100; ++i) hipLaunchKernelGGL((k_conv_fwd<4, CF_PROJ>), dim3(1), dim3(256), smem, 0, z);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  ... n = 0, one block: %.2f us per launch\n", ms * 10);
        z = a; z.a_out = nullptr; z.z1 = nullptr;
        CK(hipEventRecord(e0, 0)); for (int i = 0; i < 100; ++i) hipLaunchKernelGGL((k_conv_fwd<4, CF_PROJ>), dim3(30), dim3(256), smem, 0, z);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  ... inference form (A and Z1 not stored): %.2f us per launch\n", ms * 10);
    }
    printf("seven other row kernels in rotation: %.2f us per round; with the target in the rotation: %.2f us  => target costs %.2f us there\n", oth, rot, rot - oth);
    return 0;
}
