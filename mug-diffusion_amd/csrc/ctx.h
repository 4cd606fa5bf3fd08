// The opaque context of the C ABI (include/mugd.h), shared by api.hip and train.hip.
#pragma once
#include "net.h"

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

struct mugd_ctx {
    Ctx c;
    TrainProfile tprof;
    TrainPool pool;
    // forward intermediates a training block kept for its backward call (mugd_train_*'s `state` argument): pool blocks in the
    // block's allocation order
    std::unordered_map<long long, std::vector<void*>> saved;
    long long next_state = 1;
};
