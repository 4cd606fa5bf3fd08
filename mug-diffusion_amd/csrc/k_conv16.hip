// conv_gemm on 32 x 16 tiles (conv_body.h: ConvGeo<16>): the instantiations of the kernel template for the layers whose 32-wide tiling
// leaves CUs without a workgroup (k_conv.hip: conv_pick_tn), fp32 and bfloat16 weight fragments.  A translation unit of its own so that the
// library's three conv_gemm units compile in parallel (conv_kernel.h); the host side of a launch is k_conv.hip's.
#include "conv_kernel.h"

const void* conv_kernel16(int wk, bool dual, bool w16) {
#define MUGD_K16(W)                                                                                                               \
    case W:                                                                                                                       \
        if (w16) return dual ? reinterpret_cast<const void*>(static_cast<ConvKernel>(conv_gemm_kernel<W, true, 0, 1, unsigned short, 16>))   \
                             : reinterpret_cast<const void*>(static_cast<ConvKernel>(conv_gemm_kernel<W, false, 0, 1, unsigned short, 16>)); \
        return dual ? reinterpret_cast<const void*>(static_cast<ConvKernel>(conv_gemm_kernel<W, true, 0, 1, float, 16>))          \
                    : reinterpret_cast<const void*>(static_cast<ConvKernel>(conv_gemm_kernel<W, false, 0, 1, float, 16>));
    switch (wk) { MUGD_K16(1) MUGD_K16(2) MUGD_K16(4) MUGD_K16(8) }
#undef MUGD_K16
    return nullptr;
}
