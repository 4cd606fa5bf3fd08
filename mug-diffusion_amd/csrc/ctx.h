// The opaque context of the C ABI (include/mugd.h), shared by api.hip and train.hip.
#pragma once
#include "net.h"

#include <algorithm>
#include <cstring>
#include <map>
#include <tuple>
#include <vector>

// Scratch blocks of the training entry points, kept across calls (a training step makes several hundred block calls of a few
// recurring sizes: hipMalloc / hipFree per call -- the latter synchronises the device -- were most of the step's wall time).
struct TrainPool {
    struct Block { void* p; size_t bytes; bool used; };
    std::vector<Block> blocks;
    void* take(size_t bytes) {
        int best = -1;
        for (int i = 0; i < (int)blocks.size(); ++i)
            if (!blocks[i].used && blocks[i].bytes >= bytes && (best < 0 || blocks[i].bytes < blocks[best].bytes)) best = i;
        if (best >= 0 && blocks[best].bytes <= 2 * bytes + (1 << 20)) { blocks[best].used = true; return blocks[best].p; }
        void* p = nullptr;
        HIP_CHECK(hipMalloc(&p, bytes));
        blocks.push_back(Block{p, bytes, true});
        return p;
    }
    void give(void* p) {
        for (auto& b : blocks)
            if (b.p == p) { b.used = false; return; }
    }
    void release() {
        for (auto& b : blocks) hipFree(b.p);
        blocks.clear();
    }
};

#include <unordered_map>

// mugd_train_profile: while enabled, every training GEMM launch (conv / Linear forward + data gradient: kind 0; weight gradient: kind 1)
// is bracketed by a HIP event pair on the context's stream; read() sums the elapsed times and the algorithmic FLOPs.
struct TrainProfile {
    struct Rec { hipEvent_t a, b; int kind; double flops; };
    bool on = false;
    std::vector<Rec> recs;
    std::vector<hipEvent_t> free_events;
    hipEvent_t get() {
        if (!free_events.empty()) { hipEvent_t e = free_events.back(); free_events.pop_back(); return e; }
        hipEvent_t e = nullptr;
        HIP_CHECK(hipEventCreate(&e));
        return e;
    }
};

// ---- the step bracket of the bf16 training path (mugd_train_step_begin / _flush / _end) --------------------------------------
// Inside the bracket the caller promises that no weight tensor changes.  That buys two things:
//  * packed weights: the bf16 A-fragment form of every conv / Linear weight (both orientations: forward and data gradient) lives in a
//    cache keyed by (tensor, geometry); step_begin refreshes ALL of it with ONE table-driven launch instead of ~930 small ones;
//  * deferred reductions: split-K partial tiles of the weight gradients and the per-batch-row bias sums stay in their pool blocks
//    until flush / step_end sums all of them with ONE table-driven launch (same fixed summation order: deterministic).
// Outside the bracket every call packs and reduces on its own, as before.
struct PackEntry {
    const float* src; unsigned short* dst; int rows, K, taps, flip; long long s_row, s_k, total; int MT, nkb; long long packed_epoch, used_epoch;
};
// a table handed to a kernel: pinned host staging + device copy, a ring of slots so that the host can be several launches ahead
struct UploadRing {
    static constexpr int SLOTS = 4;
    struct Slot { void* host = nullptr; void* dev = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool busy = false; };
    Slot slots[SLOTS];
    int next = 0;
    // copies `bytes` from src to a device table ordered on st; call mark(st) after the consuming launch
    void* upload(const void* src, size_t bytes, hipStream_t st) {
        Slot& s = slots[next];
        if (s.busy) { HIP_CHECK(hipEventSynchronize(s.done)); s.busy = false; }      // only when the device is SLOTS tables behind
        if (s.cap < bytes) {
            if (s.host) { HIP_CHECK(hipHostFree(s.host)); HIP_CHECK(hipFree(s.dev)); }
            s.cap = std::max<size_t>(2 * bytes, 1 << 16);
            HIP_CHECK(hipHostMalloc(&s.host, s.cap, 0));
            HIP_CHECK(hipMalloc(&s.dev, s.cap));
        }
        if (!s.done) HIP_CHECK(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
        memcpy(s.host, src, bytes);
        HIP_CHECK(hipMemcpyAsync(s.dev, s.host, bytes, hipMemcpyHostToDevice, st));
        return s.dev;
    }
    void mark(hipStream_t st) {
        Slot& s = slots[next];
        HIP_CHECK(hipEventRecord(s.done, st));
        s.busy = true;
        next = (next + 1) % SLOTS;
    }
    void release() {
        for (Slot& s : slots) {
            if (s.host) { hipHostFree(s.host); hipFree(s.dev); }
            if (s.done) hipEventDestroy(s.done);
            s = Slot{};
        }
    }
};
// A second stream for the weight-gradient GEMM of a Linear / conv while its data-gradient GEMM runs on the context's stream (both read
// the same upstream gradient and write disjoint outputs; the small layers' launches fill a fraction of the chip each): fork = the side
// stream waits for the main one, join = the main stream waits for the side one.  Events are reused round-robin (a wait captures the
// record that precedes it).
struct SideStream {
    static constexpr int NEV = 16;
    hipStream_t st = nullptr;
    hipEvent_t ev[NEV] = {};
    int next = 0;
    hipEvent_t event() {
        hipEvent_t& e = ev[next];
        next = (next + 1) % NEV;
        if (!e) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        return e;
    }
    void fork(hipStream_t main_st) {
        if (!st) HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        hipEvent_t e = event();
        HIP_CHECK(hipEventRecord(e, main_st));
        HIP_CHECK(hipStreamWaitEvent(st, e, 0));
    }
    void join(hipStream_t main_st) {
        hipEvent_t e = event();
        HIP_CHECK(hipEventRecord(e, st));
        HIP_CHECK(hipStreamWaitEvent(main_st, e, 0));
    }
    void release() {
        for (auto& e : ev) if (e) { hipEventDestroy(e); e = nullptr; }
        if (st) { hipStreamDestroy(st); st = nullptr; }
    }
};
struct TrainStep {
    bool on = false;
    int side_mode = 0;                     // 1: weight-gradient GEMMs on the side stream next to their data-gradient GEMM
    SideStream side;
    long long epoch = 0;
    std::map<std::tuple<const float*, int, int, int, int>, int> index;      // (tensor, flip, rows, K, taps) -> packs[]
    std::vector<PackEntry> packs;
    std::vector<TReduceDesc> jobs;
    std::vector<void*> held;               // pool blocks of the queued partial sums
    UploadRing ring;
};

struct mugd_ctx {
    Ctx c;
    TrainProfile tprof;
    TrainPool pool;
    TrainStep step;
    // forward intermediates a training block kept for its backward call (mugd_train_*'s `state` argument): pool blocks in the
    // block's allocation order
    std::unordered_map<long long, std::vector<void*>> saved;
    long long next_state = 1;
};
