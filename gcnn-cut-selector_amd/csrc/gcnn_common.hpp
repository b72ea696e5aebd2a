// gcnn_common.hpp -- shared definitions: parameter layout, MFMA wrappers, small device helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
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

