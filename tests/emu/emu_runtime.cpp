// TEST INFRASTRUCTURE ONLY (see tests/emu/include/hip/hip_runtime.h).
// Fiber-based functional emulation of a HIP workgroup on one OS thread.
#include <hip/hip_runtime.h>

#include <chrono>
#include <vector>

EmuIdx threadIdx, blockIdx, blockDim, gridDim;

// ---------------------------------------------------------------- context switch (x86-64 SysV)
extern "C" void emu_ctx_switch(void** save_sp, void* new_sp);
asm(R"(
.text
.globl emu_ctx_switch
.type emu_ctx_switch,@function
emu_ctx_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_ctx_switch,.-emu_ctx_switch
)");

namespace {

enum State { RUN, WAIT_WAVE, WAIT_BLOCK, DONE };
constexpr size_t STACK_BYTES = 256 * 1024;

struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    State st = DONE;
    EmuIdx tid{};
    int lin = 0;
};

struct WaveScratch {
    float a[64], b[64];
    float a8[64][8], b8[64][8];       // bf16 MFMA operands, widened
};

std::vector<Fiber> fibers;
std::vector<WaveScratch> waves;
void* sched_sp = nullptr;
Fiber* cur = nullptr;
const std::function<void()>* cur_body = nullptr;
int n_threads = 0;

void yield_to_sched() {
    Fiber* me = cur;
    emu_ctx_switch(&me->sp, sched_sp);
}

void fiber_entry() {
    (*cur_body)();
    cur->st = DONE;
    yield_to_sched();
    fprintf(stderr, "emu: resumed a finished fiber\n");
    abort();
}

void prepare(Fiber& f) {
    if (!f.stack) f.stack = (char*)aligned_alloc(64, STACK_BYTES);
    uintptr_t top = ((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)63;
    void** sp = (void**)top;
    *--sp = nullptr;                    // alignment slot: entry sees rsp % 16 == 8
    *--sp = (void*)&fiber_entry;        // 'ret' target
    for (int i = 0; i < 6; ++i) *--sp = nullptr;   // rbp rbx r12 r13 r14 r15
    f.sp = sp;
    f.st = RUN;
}

void resume(Fiber& f) {
    cur = &f;
    threadIdx = f.tid;
    emu_ctx_switch(&sched_sp, f.sp);
    cur = nullptr;
}

void run_block() {
    const int nw = (n_threads + 63) / 64;
    for (;;) {
        bool all_done = true;
        for (int w = 0; w < nw; ++w) {
            const int l0 = w * 64, l1 = std::min(n_threads, l0 + 64);
            for (;;) {                                   // run this wave up to its next block-level event
                for (int l = l0; l < l1; ++l)
                    while (fibers[l].st == RUN) resume(fibers[l]);
                int nwave = 0, nother = 0;
                for (int l = l0; l < l1; ++l) (fibers[l].st == WAIT_WAVE ? nwave : nother)++;
                if (nwave == 0) break;
                if (nother != 0) {
                    fprintf(stderr, "emu: divergent wave collective (block %u,%u,%u wave %d: %d lanes waiting, %d elsewhere)\n",
                            blockIdx.x, blockIdx.y, blockIdx.z, w, nwave, nother);
                    abort();
                }
                for (int l = l0; l < l1; ++l) fibers[l].st = RUN;
            }
        }
        int nb = 0;
        for (int l = 0; l < n_threads; ++l) {
            if (fibers[l].st == WAIT_BLOCK) { nb++; all_done = false; }
        }
        if (all_done) return;
        for (int l = 0; l < n_threads; ++l)
            if (fibers[l].st == WAIT_BLOCK) fibers[l].st = RUN;
        (void)nb;
    }
}

void run_grid(dim3 grid, dim3 block, const std::function<void()>& body) {
    n_threads = (int)(block.x * block.y * block.z);
    if (n_threads <= 0 || n_threads > 1024) { fprintf(stderr, "emu: bad block size %d\n", n_threads); abort(); }
    if ((int)fibers.size() < n_threads) fibers.resize(n_threads);
    waves.resize((n_threads + 63) / 64);
    gridDim = EmuIdx{grid.x, grid.y, grid.z};
    blockDim = EmuIdx{block.x, block.y, block.z};
    cur_body = &body;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                blockIdx = EmuIdx{bx, by, bz};
                int lin = 0;
                for (unsigned tz = 0; tz < block.z; ++tz)
                    for (unsigned ty = 0; ty < block.y; ++ty)
                        for (unsigned tx = 0; tx < block.x; ++tx, ++lin) {
                            fibers[lin].tid = EmuIdx{tx, ty, tz};
                            fibers[lin].lin = lin;
                            prepare(fibers[lin]);
                        }
                run_block();
            }
    cur_body = nullptr;
}

}  // namespace

// ---------------------------------------------------------------- streams / graphs
struct EmuStream {
    bool capturing = false;
    std::vector<std::function<void()>>* cap = nullptr;
};
struct EmuGraph {
    std::vector<std::function<void()>> nodes;
};
struct EmuEvent {
    std::chrono::steady_clock::time_point t;
};
static EmuStream default_stream;
static EmuStream* S(hipStream_t s) { return s ? s : &default_stream; }

namespace emu {

void enqueue(hipStream_t st, std::function<void()> fn) {
    EmuStream* s = S(st);
    if (s->capturing) s->cap->push_back(std::move(fn));
    else fn();
}

void launch(dim3 grid, dim3 block, hipStream_t st, std::function<void()> body) {
    if (cur) { fprintf(stderr, "emu: nested launch\n"); abort(); }
    enqueue(st, [grid, block, body]() { run_grid(grid, block, body); });
}

void block_barrier() {
    cur->st = WAIT_BLOCK;
    yield_to_sched();
}
void wave_barrier() {
    cur->st = WAIT_WAVE;
    yield_to_sched();
}
int lane_id() { return cur->lin & 63; }

float wave_xchg(float v, int src_lane) {
    WaveScratch& w = waves[cur->lin >> 6];
    w.a[cur->lin & 63] = v;
    wave_barrier();
    float r = w.a[src_lane & 63];
    wave_barrier();
    return r;
}

// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31];
// D reg r of lane l = D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31]; k-ordered fmaf chain.
void mfma_32x32x2(float a, float b, const float* c, float* d) {
    WaveScratch& w = waves[cur->lin >> 6];
    const int l = cur->lin & 63;
    w.a[l] = a;
    w.b[l] = b;
    wave_barrier();
    const int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) acc = fmaf(w.a[row + 32 * k], w.b[col + 32 * k], acc);
        d[r] = acc;
    }
    wave_barrier();
}

// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15];
// D reg r of lane l = D[row=(l>>4)*4+r][col=l&15].
void mfma_16x16x4(float a, float b, const float* c, float* d) {
    WaveScratch& w = waves[cur->lin >> 6];
    const int l = cur->lin & 63;
    w.a[l] = a;
    w.b[l] = b;
    wave_barrier();
    const int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(w.a[row + 16 * k], w.b[col + 16 * k], acc);
        d[r] = acc;
    }
    wave_barrier();
}

// v_mfma_f32_16x16x16f16: lane l holds 4 consecutive k (k = 4 (l >> 4) + e) of A row i = l & 15 and of B column j = l & 15; D map as the
// 16x16x4 f32 form (lane l: column l & 15, rows 4 (l >> 4) + r)
void mfma_16x16x16(const float* a4, const float* b4, const float* c, float* d) {
    WaveScratch& w = waves[cur->lin >> 6];
    const int l = cur->lin & 63;
    for (int e = 0; e < 4; ++e) { w.a8[l][e] = a4[e]; w.b8[l][e] = b4[e]; }
    wave_barrier();
    const int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 16; ++k) acc = fmaf(w.a8[row + 16 * (k >> 2)][k & 3], w.b8[col + 16 * (k >> 2)][k & 3], acc);
        d[r] = acc;
    }
    wave_barrier();
}

// v_mfma_f32_32x32x16_bf16: lane l holds 8 consecutive k of A row i = l & 31 and of B column j = l & 31, k = 8 (l >> 5) + e;
// D map as the f32 form.  The hardware's internal summation order is not specified: fp32 accumulation, k ascending here.
void mfma_32x32x16_bf16(const float* a8, const float* b8, const float* c, float* d) {
    WaveScratch& w = waves[cur->lin >> 6];
    const int l = cur->lin & 63;
    for (int e = 0; e < 8; ++e) { w.a8[l][e] = a8[e]; w.b8[l][e] = b8[e]; }
    wave_barrier();
    const int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 16; ++k) acc = fmaf(w.a8[row + 32 * (k >> 3)][k & 7], w.b8[col + 32 * (k >> 3)][k & 7], acc);
        d[r] = acc;
    }
    wave_barrier();
}

}  // namespace emu

// ---------------------------------------------------------------- API
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emu error"; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) {
    *p = aligned_alloc(256, (n + 255) / 256 * 256 + 256);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t st) {
    emu::enqueue(st, [=]() { memmove(d, s, n); });
    return hipSuccess;
}
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st) {
    emu::enqueue(st, [=]() { memset(d, v, n); });
    return hipSuccess;
}
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t* s) { *s = new EmuStream(); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new EmuEvent(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
hipError_t hipStreamBeginCapture(hipStream_t st, hipStreamCaptureMode) {
    EmuStream* s = S(st);
    s->capturing = true;
    s->cap = new std::vector<std::function<void()>>();
    return hipSuccess;
}
hipError_t hipStreamEndCapture(hipStream_t st, hipGraph_t* g) {
    EmuStream* s = S(st);
    *g = new EmuGraph();
    (*g)->nodes = std::move(*s->cap);
    delete s->cap;
    s->cap = nullptr;
    s->capturing = false;
    return hipSuccess;
}
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, void*, void*, size_t) {
    *e = new EmuGraph(*g);
    return hipSuccess;
}
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t) {
    for (auto& n : e->nodes) n();
    return hipSuccess;
}
hipError_t hipGraphExecDestroy(hipGraphExec_t e) { delete e; return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t g) { delete g; return hipSuccess; }
