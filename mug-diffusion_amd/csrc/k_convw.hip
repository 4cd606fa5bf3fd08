// conv_gemm: the M-split ("wide") forms of the 32 x 32 tiles (conv_body.h: MS -- the waves of a workgroup own different row tiles and
// share every staged window) and the bfloat16-weight kernels of the reduced-precision sampling mode.  A translation unit of its own so that
// the library's three conv_gemm units compile in parallel (conv_kernel.h); the host side of a launch -- which form, when -- is k_conv.hip's.
#include "conv_kernel.h"

// NW waves per workgroup = NW / KS row tiles x KS K-slices.  Forms that exist: KS = 1 with 2 | 4 | 8 row tiles; 4 x 2 and 2 x 2 (round 5).
// (Round 6 measured a 2 x 4 form for the tall gated projections at batch 4 -- 12 - 90 % SLOWER than the K-split form on every shape,
// profiles/r6_forms_sweep.txt -- and did not keep it.)  null: no such form.
const void* conv_kernel_wide(int nw, bool dual, int ks) {
#define MUGD_KW(N, K)                                                                                                             \
    if (nw == N && ks == K)                                                                                                       \
        return dual ? reinterpret_cast<const void*>(static_cast<ConvKernel>(conv_gemm_kernel<N, true, 0, 1, float, 32, K>))       \
                    : reinterpret_cast<const void*>(static_cast<ConvKernel>(conv_gemm_kernel<N, false, 0, 1, float, 32, K>));
    MUGD_KW(2, 1) MUGD_KW(4, 1) MUGD_KW(8, 1) MUGD_KW(4, 2) MUGD_KW(8, 2)
#undef MUGD_KW
    return nullptr;
}

const void* conv_kernel32_w16(int wk, bool dual) {
#define MUGD_KB(W)                                                                                                                \
    case W:                                                                                                                       \
        return dual ? reinterpret_cast<const void*>(static_cast<ConvKernel>(conv_gemm_kernel<W, true, 0, 1, unsigned short>))     \
                    : reinterpret_cast<const void*>(static_cast<ConvKernel>(conv_gemm_kernel<W, false, 0, 1, unsigned short>));
    switch (wk) { MUGD_KB(1) MUGD_KB(2) MUGD_KB(4) MUGD_KB(8) }
#undef MUGD_KB
    return nullptr;
}
