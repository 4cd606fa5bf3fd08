// The conv_gemm kernel template (see k_conv.hip for the design notes).  Its instantiations are spread over three translation units so that
// the library builds in parallel (one unit with all ~70 of them took 5 minutes of a 2-minute build): k_conv.hip -- 32-wide tiles, K-split,
// fp32 weights (+ the host side of every launch); k_conv16.hip -- 32 x 16 tiles; k_convw.hip -- the M-split forms and the bfloat16-weight
// kernels.  The host side reaches the other units' kernels through the accessor functions of kernels.h (conv_kernel16 / conv_kernel_wide /
// conv_kernel32_w16).
#pragma once
#include "conv_body.h"

#ifndef MUGD_KARG_WARM
#define MUGD_KARG_WARM 1
#endif

namespace {

// One workgroup = one tile: decodes blockIdx into (row tile, batch row, column tile) and runs the shared tile body (conv_body.h).
// TN = 16: the 32 x 16 tiles of conv_body.h (ConvGeo) -- the same body, half the columns per workgroup.
// MS: the M-split ("wide") form -- the grid's row axis counts GROUPS of WK row tiles (conv_body.h).
template <int WK, bool DUAL, int KIND, int NITG, class WT = float, int TN = CONV_TN, int MS = 0>
__global__ __launch_bounds__(WK * 64) MUGD_WAVES_PER_EU(2) void conv_gemm_kernel(const ConvArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[conv_lds_bytes<WK, DUAL, TN, MS>()];
    TL_BEGIN();
#if MUGD_KARG_WARM
    KARG_WARM(sizeof(ConvArgs));
#endif

    // ---- kernel arguments of the prologue in one batch (common.h: KARG_PIN)
    const int gx = a.gx, gy = a.gy, gz = a.gz;
    KARG_PIN4(gx, gy, gz, a.xcd_cols);
    KARG_PIN4(a.mgx, a.mgy, a.mgxz, a.nseg);
    KARG_PIN4(a.gn_groups, a.gn_cg, a.gn_nseg, a.Mout);
    KARG_PIN4(a.wpk, a.w_mt_stride, a.Tout, a.nchunk);
    KARG_PIN4(a.seg[0].x, a.seg[0].C, a.seg[0].Tin, a.seg[0].xf);
    KARG_PIN4(a.seg[0].xf_a, a.seg[0].xf_stride, a.seg[0].bmod, a.seg[0].xf_np);
    KARG_PIN4(a.bias, a.rowadd, a.resid, a.rowadd_stride);

    // ---- XCD-aware renumbering: hardware deals consecutive workgroup ids round-robin to the 8 XCDs;
    // give each XCD a contiguous slab of the (row tile major) tile order so a weight tile is pulled
    // into ONE private L2 and reused there by all sample tiles / batch rows.
    const int nblk = gx * gy * gz;
    int lid = blockIdx.x;
    if ((nblk & 7) == 0) lid = (lid & 7) * (nblk >> 3) + (lid >> 3);
    int mt, rem;
    if (a.xcd_cols) { rem = fastdiv(lid, a.mgy, gy); mt = lid - rem * gy; }       // row tile fastest: an XCD's slab is a range of column tiles
    else { mt = fastdiv(lid, a.mgxz, gx * gz); rem = lid - mt * (gx * gz); }
    const int b = fastdiv(rem, a.mgx, gx);
    const int t0 = (rem - b * gx) * TN;
    conv_tile<WK, DUAL, KIND, NITG, WT, ConvArgs, TN, MS>(a, MS ? (WK / MS) * mt : mt, b, t0, rem, (int)threadIdx.x, lds, true);      // MS: the grid's row axis counts GROUPS of row tiles
    TL_END(a.tl, WK);
}

}  // namespace
