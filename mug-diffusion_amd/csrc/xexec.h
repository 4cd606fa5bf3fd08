// XCD-resident executor (xexec.hip): a run of consecutive ops of a network program -- conv_gemm tiles, self-attention tiles, S4 rows --
// executed by ONE persistent kernel instead of one launch per op.
//
// Mapping: MI355X = 8 XCDs x 32 CUs, one private L2 per XCD.  256 workgroups of 8 waves (one per CU; the LDS block forces that) read
// their XCC_ID and take a rank among the 32 workgroups of their XCD.  Batch row b belongs to XCD b % 8 for the whole run: every
// op of sample b is computed by that XCD's 32 CUs, its activations are written and read through ONE L2 and never cross XCDs, so the
// phase barrier between two ops is an XCD-LOCAL barrier (profiles/r3_xcd_barrier.txt: ~1.1 us; a kernel boundary costs 4-5 us of a
// dependent chain, a device-wide barrier 19 us).  Weights are read by every XCD (profiles/r4_xcd_exec_probe.txt: 0.9 TB/s per XCD when
// all eight stream the same 400 MB, 7.2 TB/s in total -- the Infinity Cache absorbs the replication).
//
// Memory model used (no cache maintenance instruction anywhere in the kernel):
//   * producer side: plain stores, `s_waitcnt vmcnt(0)` (stores have reached the XCD's L2), agent-scope atomic on the XCD's counter;
//   * consumer side: PLAIN loads.  That is only correct because the program's workspace is SINGLE-ASSIGNMENT inside one launch
//     (Arena::monotonic: no buffer address is reused within a U-Net evaluation) and the CU's L1 starts a kernel empty: a line a CU reads
//     was written before the read in program order, or not at all in this launch -- it cannot be in the reader's L1 in a stale state.
//     (`buffer_inv sc0` does NOT invalidate the L1 for this purpose on gfx950, `buffer_inv sc1` costs +1.9 us per phase, sc1 loads
//     would need every activation load of the tile bodies rewritten: profiles/r4_xcd_exec_probe.txt.)
#pragma once
#include "kernels.h"

enum { XOP_CONV = 0, XOP_ATTN = 1, XOP_S4 = 2 };

struct XOp {
    int type;
    int wk;         // conv: waves per tile (K-split) 1 | 2 | 4 | 8; attention / S4: 4
    int variant;    // conv: dual | kind << 1 | (nitg > 9 ? 8 : 0);  attention: head dim D;  S4: R = ceil(L / 64) rounded to {1, 2, 4, 8}
    int items;      // work items per batch row: conv gx * gy tiles; attention ceil(Tq / 32) * heads; S4: H rows
    int gx;         // conv: column tiles per row tile; attention: query tiles per head
    int pad_[3];
    union U {
        ConvArgs conv;
        AttnArgs attn;
        S4ConvArgs s4;
        U() {}
    } u;
    XOp() : type(0), wk(0), variant(0), items(0), gx(0), pad_{0, 0, 0} {}
};

struct XSync {
    unsigned ticket[8][32];     // [xcd][0]: rank tickets (a 128-byte line per XCD); every launch takes exactly 32 per XCD
    unsigned arrive[8][32];     // [xcd][0]: monotonic barrier arrivals
    unsigned err[32];           // [0]: a workgroup gave up waiting (placement was not 32 per XCD, or a peer died)
};

// host side: whether / how an op can run inside the executor.  `B` = batch rows of the program.
bool xexec_conv_supported(const ConvArgs& a);
bool xexec_attn_supported(const AttnArgs& a);
bool xexec_s4_supported(const S4ConvArgs& a);
XOp xexec_make_conv(const ConvArgs& a, int B);
XOp xexec_make_attn(const AttnArgs& a);
XOp xexec_make_s4(const S4ConvArgs& a);
// ops[lo, hi) of a device-resident table as one persistent launch.  `barriers_done`: XCD barriers executed by earlier launches on
// this XSync block (the caller adds hi - lo - 1 after the call).
// tl (development aid, nullable): (hi - lo + 1) x 8 words -- rank 0 of XCD x writes the 100 MHz s_memrealtime at the start of phase k to tl[8 k + x]
// (the last row: the end of the launch); the last phase's end is NOT behind a barrier (it is that XCD's rank 0 finishing its own share).
void launch_xexec(hipStream_t st, const XOp* dev_ops, int lo, int hi, XSync* sync, unsigned barriers_done, int B, unsigned long long* tl = nullptr);
bool xexec_device_ok(int device);      // 256 CUs in 8 XCDs
