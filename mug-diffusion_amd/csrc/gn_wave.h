// GroupNorm statistics inside a conv_gemm workgroup (ConvSeg::xf == 4), shared by k_conv.hip and k_conv16.hip.
//
// The producers of the normalised tensors accumulated fp64 {sum, sum of squares} per row (ConvArgs::rowstat).  A wave only
// needs the groups its own K-slice [g0, g1) touches -- with the K axis split over 8 waves that is 4..5 of the 32 groups --
// so every wave reduces exactly those, 8 lanes per group, and keeps {mean, rstd} in its own 32-entry LDS table.  No
// cross-wave synchronisation, one memory round trip, instead of a 5.7 us statistics launch.
#pragma once
#include "kernels.h"

__device__ __forceinline__ double shfl_xor_d(double v, int o) {
    return __hiloint2double(__shfl_xor(__double2hiint(v), o), __shfl_xor(__double2loint(v), o));
}

// gst: this wave's table [32] of {mean, rstd}; g0/g1: the wave's chunk range in the K axis (16 channels per chunk)
__device__ __forceinline__ void wave_gn_stats(const ConvArgs& a, int b, int lane, int g0, int g1, float2* gst) {
    int cdom = 0;
    for (int i = 0; i < a.gn_nseg; ++i) cdom += a.seg[i].C;
    const int c_lo = g0 * CONV_CK;
    int c_hi = g1 * CONV_CK;
    c_hi = c_hi < cdom ? c_hi : cdom;
    if (c_lo < c_hi) {
        const int cg = a.gn_cg;
        const int gfirst = c_lo / cg, glast = (c_hi - 1) / cg;
        const int team = lane >> 3, j = lane & 7;
        for (int gb = gfirst; gb <= glast; gb += 8) {
            const int g = gb + team;
            const bool active = g <= glast;
            double s1 = 0.0, s2 = 0.0;
            if (active) {
                for (int cc = j; cc < cg; cc += 8) {
                    int cl = g * cg + cc, si = 0;
                    while (si + 1 < a.gn_nseg && cl >= a.seg[si].C) { cl -= a.seg[si].C; ++si; }
                    const ConvSeg& s = a.seg[si];
                    const int bb = s.bmod > 0 ? b % s.bmod : b;
                    const double* p = reinterpret_cast<const double*>(s.xf_a) + (size_t)bb * s.xf_stride + 2 * (size_t)cl;
                    s1 += p[0]; s2 += p[1];
                }
            }
            s1 += shfl_xor_d(s1, 1); s2 += shfl_xor_d(s2, 1);
            s1 += shfl_xor_d(s1, 2); s2 += shfl_xor_d(s2, 2);
            s1 += shfl_xor_d(s1, 4); s2 += shfl_xor_d(s2, 4);
            if (active && j == 0) {
                const double mean = s1 / (double)a.gn_count;
                double var = s2 / (double)a.gn_count - mean * mean;
                var = var > 0.0 ? var : 0.0;
                gst[g] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)a.gn_eps)));
            }
        }
    }
    wave_sync();
}

