// The XCD-resident executor: see xexec.h for the mapping and the memory model.
//
// One launch = ops [lo, hi) of a program's op table.  Per op every workgroup walks the work items of ITS XCD's batch rows
// (b = xcd, xcd + 8, ...): a workgroup of 8 waves is split into 8 / wk "virtual workgroups" of wk waves that each take one item per
// round -- a conv tile (wk = its K-split), an attention query tile or an S4 row (wk = 4) -- through the SAME device code the stand-alone
// kernels run (conv_body.h, attn_body.h, s4_body.h); barriers inside those bodies are whole-workgroup barriers, which is fine because
// the virtual workgroups of one workgroup execute the same op in lockstep (an idle slot of the last round runs on a clamped item with
// its stores masked).  Between two ops: the XCD barrier (32 arrivals on the XCD's own counter).
#include "xexec.h"

#include <algorithm>

#include "attn_body.h"
#include "conv_body.h"
#include "s4_body.h"

namespace {

constexpr int XW = 8;                   // waves per workgroup
#ifdef MUGD_EMULATED
constexpr int XCD_WGS = 4;              // the emulated build walks the same work lists with 4 workgroups per "XCD" (8x fewer fibres per phase)
#else
constexpr int XCD_WGS = 32;             // workgroups (CUs) per XCD
#endif
constexpr int NXCD = 8;

constexpr int cmax(int a, int b) { return a > b ? a : b; }
template <int WK, bool DUAL> constexpr int conv_wg_bytes() { return (XW / WK) * conv_lds_bytes<WK, DUAL>(); }
constexpr int XLDS_CONV = cmax(cmax(cmax(conv_wg_bytes<1, true>(), conv_wg_bytes<2, true>()), conv_wg_bytes<4, true>()), conv_wg_bytes<8, true>());
constexpr int XLDS_ATTN = 2 * AttnLds<64>::BYTES;
constexpr int XLDS_S4 = 2 * S4Lds<8>::BYTES;
constexpr int XLDS = cmax(cmax(XLDS_CONV, XLDS_ATTN), XLDS_S4);
static_assert(XLDS > 80 * 1024, "the executor relies on ONE workgroup per CU");
static_assert((XCD_WGS & (XCD_WGS - 1)) == 0, "rank = ticket mod XCD_WGS");
static_assert(XLDS + 64 <= 160 * 1024, "LDS block exceeds a CU's 160 KB");

typedef const MUGD_CONST_AS XOp XOpC;

// The workgroup's LDS block.  The per-op bodies below are INLINED into the kernel: as real functions every call saved and restored
// ~110 callee-saved VGPRs per lane (0.46 MB per workgroup and op through scratch: measured 10.8 ms per step against 3.9 for per-op
// launches, profiles/r4_xexec_ab.txt) and their argument block arrived in VGPRs (vector loads, no scalar registers).  Inlined, every
// arm of the dispatch switch reads its argument block through a pointer the optimiser cannot identify with the other arms' (X_OPAQUE):
// otherwise the loads common to all arms are hoisted above the switch and ~150 scalar values stay live across it (2335 SGPR spills).
__shared__ __attribute__((aligned(16))) char g_xlds[XLDS];
#define X_NOINLINE __forceinline__
#ifdef MUGD_EMULATED
#define X_OPAQUE(p) (p)
#define X_OPAQUE_TID(t) do {} while (0)
#else
// ... and its thread index through a value the optimiser cannot hoist: everything derived from threadIdx is invariant in the op loop, and
// loop-invariant code motion otherwise computes the index arithmetic of ALL arms in front of the loop and parks it in scratch
#define X_OPAQUE_TID(t) asm volatile("" : "+v"(t))
template <class T>
__device__ __forceinline__ T* x_opaque(T* p) {
    asm volatile("" : "+s"(p));
    return p;
}
#define X_OPAQUE(p) x_opaque(p)
#endif

#ifndef MUGD_EMULATED
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf; }      // HW_REG_XCC_ID
#endif

// Phase barrier among the 32 workgroups of one XCD.  Stores of this workgroup are in the XCD's L2 once vmcnt drains; the counter is the
// XCD's own (no other XCD ever touches the line).  Returns false when the wait gave up.
__device__ __forceinline__ bool xcd_barrier(XSync* sync, unsigned xcd, unsigned target) {
#ifdef MUGD_EMULATED
    (void)sync; (void)xcd; (void)target;
    __syncthreads();
    return true;
#else
    __shared__ int ok_s;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned* c = &sync->arrive[xcd][0];
        __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1, spins = 0;
        while ((int)(__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 24)) { ok = 0; break; }                    // ~1 s: placement was not 32 per XCD, or a peer died
            if ((spins & 1023) == 0 && __hip_atomic_load(&sync->err[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = 0; break; }
        }
        if (!ok) __hip_atomic_fetch_add(&sync->err[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok_s = ok;
    }
    __syncthreads();
    return ok_s != 0;
#endif
}

// Inside the item loops the thread index and the argument pointer are made opaque ONCE PER ITEM: both are invariant in the loop, and
// the optimiser otherwise hoists the tile body's index arithmetic and all of its argument loads in front of the loop, where they stay
// live (VGPRs parked in scratch, SGPRs in spill lanes) for the whole tile -- the stand-alone kernels compute them at their point of use.
template <int WK, bool DUAL, int KIND, int NITG, class CA>
__device__ X_NOINLINE void x_conv(const CA* ap, const int tiles, const int gx, const int xcd, const int rank, const int B) {
    char* lds = g_xlds;
    constexpr int PAR = XW / WK;
    const int nb = (B - xcd + NXCD - 1) / NXCD;               // batch rows of this XCD: b = xcd + 8 j
    const int total = nb * tiles;
    for (int base = 0; base < total; base += XCD_WGS * PAR) {
        int tid = (int)threadIdx.x;
        X_OPAQUE_TID(tid);
        const CA& a = *X_OPAQUE(ap);
        const int v = __builtin_amdgcn_readfirstlane(tid / (WK * 64));
        const int vtid = tid - v * (WK * 64);
        char* vlds = lds + v * conv_lds_bytes<WK, DUAL>();
        int item = base + rank * PAR + v;                     // a workgroup's virtual workgroups take neighbouring column tiles of one weight row tile
        const bool live = item < total;
        item = live ? item : total - 1;
        const int bl = item / tiles, t = item - bl * tiles;
        const int mt = t / gx, ct = t - mt * gx;
        __syncthreads();                                      // the previous item's LDS (exchange / epilogue scratch) is free
        conv_tile<WK, DUAL, KIND, NITG, float>(a, mt, xcd + NXCD * bl, ct * CONV_TN, ct, vtid, vlds, live);
    }
}

template <int D, class AA>
__device__ X_NOINLINE void x_attn(const AA* ap, const int items, const int gx, const int xcd, const int rank, const int B) {
    char* lds = g_xlds;
    const int nb = (B - xcd + NXCD - 1) / NXCD;
    const int total = nb * items;
    for (int base = 0; base < total; base += XCD_WGS * 2) {
        int tid = (int)threadIdx.x;
        X_OPAQUE_TID(tid);
        const AA& a = *X_OPAQUE(ap);
        const int v = __builtin_amdgcn_readfirstlane(tid >> 8), vtid = tid & 255;
        char* vlds = lds + v * AttnLds<D>::BYTES;
        int item = base + rank * 2 + v;
        const bool live = item < total;
        item = live ? item : total - 1;
        const int bl = item / items, t = item - bl * items;
        const int head = t / gx, qt = t - head * gx;
        __syncthreads();
        attention_tile_d<D>(a, qt * 32, head, xcd + NXCD * bl, vtid, vlds, live);
    }
}

template <int R, class SA>
__device__ X_NOINLINE void x_s4(const SA* ap, const int H, const int xcd, const int rank, const int B) {
    char* lds = g_xlds;
    const int nb = (B - xcd + NXCD - 1) / NXCD;
    const int total = nb * H;
    for (int base = 0; base < total; base += XCD_WGS * 2) {
        int tid = (int)threadIdx.x;
        X_OPAQUE_TID(tid);
        const SA& a = *X_OPAQUE(ap);
        const int v = __builtin_amdgcn_readfirstlane(tid >> 8), vtid = tid & 255;
        char* vlds = lds + v * S4Lds<R>::BYTES;
        int item = base + rank * 2 + v;
        const bool live = item < total;
        item = live ? item : total - 1;
        const int bl = item / H, h = item - bl * H;
        __syncthreads();
        s4_conv_fast_row<R>(a, h, xcd + NXCD * bl, vtid, vlds, live);
    }
}

__global__ __launch_bounds__(XW * 64) MUGD_WAVES_PER_EU(2) void xexec_kernel(const XOp* ops_g, int lo, int hi, XSync* sync, unsigned barriers_done, int B,
                                                                             unsigned long long* tl) {
    (void)0;
    __shared__ unsigned s_xcd, s_rank;
#ifdef MUGD_EMULATED
    const unsigned xcd = blockIdx.x & 7, rank = blockIdx.x >> 3;
    (void)s_xcd; (void)s_rank;
#else
    if (threadIdx.x == 0) {
        const unsigned x = xcc_id() & 7;
        s_xcd = x;
        s_rank = __hip_atomic_fetch_add(&sync->ticket[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (XCD_WGS - 1);
    }
    __syncthreads();
    const unsigned xcd = __builtin_amdgcn_readfirstlane(s_xcd), rank = __builtin_amdgcn_readfirstlane(s_rank);
#endif
    XOpC* ops = to_const_as(ops_g);
#ifndef MUGD_EMULATED
    // development aid (launch_xexec's `tl`): rank 0 of every XCD stamps the 100 MHz clock at the start of each phase and at the end
    if (tl && rank == 0 && threadIdx.x == 0) tl[xcd] = __builtin_amdgcn_s_memrealtime();
#endif
    // development experiment (tl set and bit 16 of B): every phase runs TWICE back to back (the second pass finds its weights, activations and
    // code hot in the XCD's L2) with a barrier and a clock stamp in between -- an upper bound on what a perfect prefetch of phase p + 1's
    // operands during phase p could buy.  Results of such a launch are NOT valid (row sums are added twice).
    const bool twice = tl != nullptr && (B & 0x10000) != 0;
    B &= 0xffff;
    unsigned extra = 0;                     // barriers this launch executed beyond one per phase boundary
    for (int i = lo; i < hi; ++i) {
      for (int rep = 0; rep < (twice ? 2 : 1); ++rep) {
        if (rep == 1) {
            if (!xcd_barrier(sync, xcd, (barriers_done + (unsigned)(i - lo) + extra + 1u) * XCD_WGS)) return;
            ++extra;
#ifndef MUGD_EMULATED
            if (rank == 0 && threadIdx.x == 0) tl[(size_t)(hi - lo + 1 + (i - lo)) * NXCD + xcd] = __builtin_amdgcn_s_memrealtime();
#endif
        }
        XOpC& op = ops[i];
        const int items = op.items, gx = op.gx;
        if (op.type == XOP_CONV) {
#define X_CONV(WK, DUAL, KIND, NITG) x_conv<WK, DUAL, KIND, NITG>(&op.u.conv, items, gx, (int)xcd, (int)rank, B)
            switch (op.wk * 16 + op.variant) {
                case 8 * 16 + 0: X_CONV(8, false, 0, 1); break;
                case 4 * 16 + 0: X_CONV(4, false, 0, 1); break;
                case 2 * 16 + 0: X_CONV(2, false, 0, 1); break;
                case 1 * 16 + 0: X_CONV(1, false, 0, 1); break;
                case 8 * 16 + 1: X_CONV(8, true, 0, 1); break;
                case 4 * 16 + 1: X_CONV(4, true, 0, 1); break;
                case 2 * 16 + 1: X_CONV(2, true, 0, 1); break;
                case 1 * 16 + 1: X_CONV(1, true, 0, 1); break;
                case 8 * 16 + 4: X_CONV(8, false, 2, 9); break;
                case 4 * 16 + 4: X_CONV(4, false, 2, 9); break;
                case 8 * 16 + 12: X_CONV(8, false, 2, 17); break;
                case 4 * 16 + 12: X_CONV(4, false, 2, 17); break;
                default: break;
            }
#undef X_CONV
        } else if (op.type == XOP_ATTN) {
            switch (op.variant) {
                case 32: x_attn<32>(&op.u.attn, items, gx, (int)xcd, (int)rank, B); break;
                case 48: x_attn<48>(&op.u.attn, items, gx, (int)xcd, (int)rank, B); break;
                case 64: x_attn<64>(&op.u.attn, items, gx, (int)xcd, (int)rank, B); break;
                default: break;
            }
        } else {
            switch (op.variant) {
                case 1: x_s4<1>(&op.u.s4, items, (int)xcd, (int)rank, B); break;
                case 2: x_s4<2>(&op.u.s4, items, (int)xcd, (int)rank, B); break;
                case 4: x_s4<4>(&op.u.s4, items, (int)xcd, (int)rank, B); break;
                case 8: x_s4<8>(&op.u.s4, items, (int)xcd, (int)rank, B); break;
                default: break;
            }
        }
      }
        if (i + 1 < hi) {
            if (!xcd_barrier(sync, xcd, (barriers_done + (unsigned)(i - lo) + extra + 1u) * XCD_WGS)) return;
        }
#ifndef MUGD_EMULATED
        if (tl && rank == 0 && threadIdx.x == 0) tl[(size_t)(i - lo + 1) * NXCD + xcd] = __builtin_amdgcn_s_memrealtime();
#endif
    }
}

int pick_conv_wk(const ConvArgs& a, int items_total) {
    // estimated rounds x (K-chunks per wave + a fixed per-tile cost of ~3 chunk times); a 3-tap chunk counts 2 (kernels.h: conv_split_k)
    long long cost = 0;
    for (int i = 0; i < a.nseg; ++i) cost += (long long)(a.seg[i].C / CONV_CK) * (a.seg[i].taps == 3 ? 2 : 1);
    int best = 8;
    long long best_t = -1;
    for (int wk = 8; wk >= 1; wk >>= 1) {
        if (wk > 1 && a.nchunk < wk) continue;
        const int slots = XCD_WGS * (XW / wk);
        const long long rounds = (items_total + slots - 1) / slots;
        const long long t = rounds * ((cost + wk - 1) / wk + 3);
        if (best_t < 0 || t < best_t) { best = wk; best_t = t; }
    }
    return best;
}

}  // namespace

bool xexec_conv_supported(const ConvArgs& a) {
    if (a.w16 || a.tn != 32 || a.nseg < 1 || a.nseg > CONV_MAXSEG) return false;
    bool all_vec = true, lean = true;
    int nitg = 0;
    for (int i = 0; i < a.nseg; ++i) {
        const ConvSeg& s = a.seg[i];
        if (!(s.stride == 1 && !s.ups && (s.Tin & 3) == 0 && s.pad <= 8)) all_vec = false;
    }
    for (int i = 0; i < a.nseg; ++i) {
        const ConvSeg& s = a.seg[i];
        if (all_vec) { if (s.taps == 3 && s.dil != 1) lean = false; }
        else nitg = std::max(nitg, cdiv(CONV_CK * (31 * s.stride + (s.taps - 1) * s.dil + 1), 64));
    }
    const bool dual = a.epi == EPI_GLU || a.epi == EPI_GEGLU;
    if (all_vec) return lean;                        // KIND 0 (dilated KIND 1 windows only occur in the wave encoder / VAE: not run here)
    if (dual || nitg > 17) return false;
    for (int i = 0; i < a.nseg; ++i) if (nitg > 9 && a.seg[i].xf) return false;
    return true;                                     // KIND 2
}
bool xexec_attn_supported(const AttnArgs& a) { return (a.d == 32 || a.d == 48 || a.d == 64) && a.pmax <= ATT_PMAX; }
bool xexec_s4_supported(const S4ConvArgs& a) { return a.L >= 1 && a.L <= 512; }

XOp xexec_make_conv(const ConvArgs& a0, int B) {
    XOp op;
    op.type = XOP_CONV;
    ConvArgs a = a0;
    const int gx = cdiv(a.Tout, CONV_TN), gy = cdiv(a.Mout, 32);
    bool all_vec = true;
    int nitg = 0;
    for (int i = 0; i < a.nseg; ++i) {
        const ConvSeg& s = a.seg[i];
        if (!(s.stride == 1 && !s.ups && (s.Tin & 3) == 0 && s.pad <= 8)) all_vec = false;
    }
    if (!all_vec)
        for (int i = 0; i < a.nseg; ++i) nitg = std::max(nitg, cdiv(CONV_CK * (31 * a.seg[i].stride + (a.seg[i].taps - 1) * a.seg[i].dil + 1), 64));
    const bool dual = a.epi == EPI_GLU || a.epi == EPI_GEGLU;
    const int nb = (B + NXCD - 1) / NXCD;
    int wk = pick_conv_wk(a, nb * gx * gy);
    if (!all_vec && wk < 4) wk = 4;                  // the generic-window bodies are instantiated for 4 and 8 waves
    conv_split_k(a, wk);
    conv_set_grid(a, gx, gy, B);
    a.tl = nullptr;
    op.wk = wk;
    op.variant = (dual ? 1 : 0) | (all_vec ? 0 : 4) | (nitg > 9 ? 8 : 0);
    op.items = gx * gy;
    op.gx = gx;
    op.u.conv = a;
    return op;
}
XOp xexec_make_attn(const AttnArgs& a) {
    XOp op;
    op.type = XOP_ATTN; op.wk = 4; op.variant = a.d;
    op.gx = cdiv(a.Tq, 32); op.items = op.gx * a.heads;
    op.u.attn = a;
    return op;
}
XOp xexec_make_s4(const S4ConvArgs& a) {
    XOp op;
    op.type = XOP_S4; op.wk = 4;
    op.variant = a.L <= 64 ? 1 : a.L <= 128 ? 2 : a.L <= 256 ? 4 : 8;
    op.items = a.H; op.gx = 1;
    op.u.s4 = a;
    return op;
}

void launch_xexec(hipStream_t st, const XOp* dev_ops, int lo, int hi, XSync* sync, unsigned barriers_done, int B, unsigned long long* tl) {
    if (hi <= lo) return;
#ifdef MUGD_EMULATED
    // the emulation runs workgroups one after another: a barrier BETWEEN workgroups cannot be waited for -- one launch per phase
    for (int i = lo; i < hi; ++i) hipLaunchKernelGGL(xexec_kernel, dim3(NXCD * XCD_WGS), dim3(XW * 64), 0, st, dev_ops, i, i + 1, sync, barriers_done, B, tl);
#else
    hipLaunchKernelGGL(xexec_kernel, dim3(NXCD * XCD_WGS), dim3(XW * 64), 0, st, dev_ops, lo, hi, sync, barriers_done, B, tl);
#endif
}

bool xexec_device_ok(int device) {
#ifdef MUGD_EMULATED
    (void)device;
    return true;
#else
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return false;
    return cus == NXCD * XCD_WGS;
#endif
}
