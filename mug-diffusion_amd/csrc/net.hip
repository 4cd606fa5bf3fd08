#include "net.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>

// =======================================================================================
// Arena / Net plumbing
// =======================================================================================
static constexpr size_t ALIGN = 256;

float* Arena::alloc(size_t nfloats) {
    size_t bytes = (nfloats * sizeof(float) + ALIGN - 1) / ALIGN * ALIGN;
    size_t off = top_;
    top_ += bytes;
    if (top_ > peak_) peak_ = top_;
    if (dry_) return reinterpret_cast<float*>(ALIGN + off);      // never dereferenced
    MUGD_CHECK(top_ <= cap_, -4, "workspace arena overflow");
    return reinterpret_cast<float*>(base_ + off);
}
void Arena::reserve(size_t bytes) {
    if (bytes <= cap_) return;
    if (base_) HIP_CHECK(hipFree(base_));
    base_ = nullptr;
    HIP_CHECK(hipMalloc((void**)&base_, bytes));
    cap_ = bytes;
}
void Arena::free_all() {
    if (base_) hipFree(base_);
    base_ = nullptr;
    cap_ = 0;
}

Net::~Net() {
    for (void* p : owned) hipFree(p);
    arena.free_all();
}

void Net::set_param(const std::string& name, const void* ptr, int dtype, int ndim, const long long* shape) {
    Param p;
    p.ptr = ptr;
    p.dtype = dtype;
    p.shape.assign(shape, shape + ndim);
    params[name] = p;
}

const char* op_kind_name(int k) {
    static const char* n[OP_KINDS] = {"conv_gemm", "conv_gemm_gated", "group_norm", "layer_norm", "attention", "s4_conv", "small"};
    return (k >= 0 && k < OP_KINDS) ? n[k] : "?";
}

void Net::profile_program(ProfileRow* rows) {
    hipStream_t st = ctx->stream;
    std::vector<hipEvent_t> ev(ops.size() + 1);
    for (auto& e : ev) HIP_CHECK(hipEventCreate(&e));
    HIP_CHECK(hipStreamSynchronize(st));
    HIP_CHECK(hipEventRecord(ev[0], st));
    for (size_t i = 0; i < ops.size(); ++i) {
        ops[i].fn(st);
        HIP_CHECK(hipEventRecord(ev[i + 1], st));
    }
    HIP_CHECK(hipStreamSynchronize(st));
    for (size_t i = 0; i < ops.size(); ++i) {
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
        ProfileRow& r = rows[ops[i].kind];
        r.ms += ms; r.flops += ops[i].flops; r.launches += 1;
    }
    if (const char* path = getenv("MUGD_PROFILE_CSV")) {          // per-launch rows for offline analysis
        if (FILE* f = fopen(path, "a")) {
            fprintf(f, "idx,kind,us,gflop,label\n");
            for (size_t i = 0; i < ops.size(); ++i) {
                float ms = 0.f;
                hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
                fprintf(f, "%zu,%s,%.2f,%.4f,%s\n", i, op_kind_name(ops[i].kind), ms * 1e3, ops[i].flops / 1e9, ops[i].label.c_str());
            }
            fclose(f);
        }
    }
    for (auto& e : ev) hipEventDestroy(e);
}

void Net::host_enqueue(int passes, double* us_per_pass, int64_t* launches_per_pass) {
    hipStream_t st = ctx->stream;
    MUGD_CHECK(!ops.empty(), -2, "host_enqueue: no compiled program (run the network once first)");
    HIP_CHECK(hipStreamSynchronize(st));
    run_ops(st);                                       // (first pass after a synchronisation: module / queue warm-up outside the clock)
    HIP_CHECK(hipStreamSynchronize(st));
    const auto t0 = std::chrono::steady_clock::now();
    for (int p = 0; p < passes; ++p) run_ops(st);
    const auto t1 = std::chrono::steady_clock::now();
    HIP_CHECK(hipStreamSynchronize(st));
    *us_per_pass = std::chrono::duration<double, std::micro>(t1 - t0).count() / passes;
    *launches_per_pass = (int64_t)ops.size();
}

TlSink g_tl;

#ifdef MUGD_TL
// Development build only: runs the compiled program once (eagerly) with every conv_gemm launch writing its per-wave phase
// records, reduces them per launch and appends one CSV row per launch to `path`.  Columns: cycles are s_memtime ticks (shader
// clock); *_ns come from s_memrealtime (100 MHz, device-global), so they also order waves of different CUs / XCDs.
void Net::timeline_program(const char* path, const char* raw_path, int raw_op) {
    hipStream_t st = ctx->stream;
    const size_t cap = (size_t)48 << 20;                     // 48 M words = 384 MB: a U-Net evaluation at batch 4 needs ~ 15 M
    unsigned long long* buf = nullptr;
    HIP_CHECK(hipMalloc((void**)&buf, cap * 8));
    HIP_CHECK(hipMemsetAsync(buf, 0, cap * 8, st));
    HIP_CHECK(hipStreamSynchronize(st));
    g_tl.buf = buf; g_tl.cap = cap; g_tl.used = 0; g_tl.launches.clear();
    std::vector<size_t> first(ops.size() + 1, 0);
    for (size_t i = 0; i < ops.size(); ++i) {
        first[i] = g_tl.launches.size();
        ops[i].fn(st);
    }
    first[ops.size()] = g_tl.launches.size();
    HIP_CHECK(hipStreamSynchronize(st));
    g_tl.buf = nullptr;
    std::vector<unsigned long long> h(g_tl.used);
    HIP_CHECK(hipMemcpy(h.data(), buf, g_tl.used * 8, hipMemcpyDeviceToHost));
    hipFree(buf);
    FILE* f = fopen(path, "a");
    MUGD_CHECK(f != nullptr, -2, std::string("timeline: cannot open ") + path);
    FILE* fr = raw_path ? fopen(raw_path, "a") : nullptr;
    const char* raw_label = getenv("MUGD_TL_RAW_LABEL");          // raw per-wave records of the launches whose label contains this
    fprintf(f, "op,kind,gflop,tn,wk,blocks,waves,span_ns,start_skew_ns,first_end_ns,mhz,chunks_med,"
               "su_issue,su_side,su_wait,su_reduce,setup_med,setup_max,first_med,first_max,loop_med,loop_max,cyc_per_chunk,combine_med,combine_max,store_med,store_max,tail_med,tail_max,total_med,total_max,f_issue,f_arrive,f_park,label\n");
    auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto mx = [](const std::vector<double>& v) { double m = 0; for (double x : v) m = std::max(m, x); return m; };
    for (size_t i = 0; i < ops.size(); ++i) {
        for (size_t li = first[i]; li < first[i + 1]; ++li) {
            const TlLaunch& L = g_tl.launches[li];
            const size_t nw = (size_t)L.nblk * L.nwaves;
            unsigned long long r0 = ~0ull, r1 = 0, r0max = 0, r1min = ~0ull;
            std::vector<double> ph[7], chunks, mhz, su[4], fd[3];
            for (size_t w = 0; w < nw; ++w) {
                const unsigned long long* r = &h[L.off + w * TL_WORDS];
                if (r[0] == 0) continue;
                r0 = std::min(r0, r[7]); r0max = std::max(r0max, r[7]); r1 = std::max(r1, r[8]); r1min = std::min(r1min, r[8]);
                // stamp 2 (first chunk parked) is missing on the non-pipelined paths: fold it into the loop phase
                const unsigned long long t2 = r[2] ? r[2] : r[1];
                ph[0].push_back((double)(r[1] - r[0])); ph[1].push_back((double)(t2 - r[1])); ph[2].push_back((double)(r[3] - t2));
                ph[3].push_back((double)(r[4] - r[3])); ph[4].push_back((double)(r[5] - r[4])); ph[5].push_back((double)(r[6] - r[5]));
                ph[6].push_back((double)(r[6] - r[0]));
                chunks.push_back((double)r[10]);
                // round 6 order of a wave's prologue: statistics requested (11) -> side operands requested (12) -> the first segment's operand
                // ring issued (14) -> sums arrived (13) -> reduced + barrier (1) -> chunk 0's window arrived (15) -> parked (2)
                auto dpos = [](unsigned long long a, unsigned long long b) { return a > b ? (double)(a - b) : 0.0; };
                if (r[11] && r[12]) {
                    const unsigned long long t14 = r[14] ? r[14] : r[12];
                    const unsigned long long t13 = r[13] ? r[13] : t14;
                    su[0].push_back(dpos(r[11], r[0])); su[1].push_back(dpos(r[12], r[11]));
                    su[2].push_back(dpos(t13, t14)); su[3].push_back(dpos(r[1], t13));          // wait for the sums BEHIND the ring issue; reduce + barrier
                }
                if (r[14] && r[15] && r[2]) {      // ring issued (from the side operands) / chunk 0's window arrived (from statistics done) / transformed + parked
                    fd[0].push_back(dpos(r[14], r[12])); fd[1].push_back(dpos(r[15], r[1])); fd[2].push_back(dpos(r[2], r[15]));
                }
                if (r[8] > r[7]) mhz.push_back((double)(r[6] - r[0]) / ((double)(r[8] - r[7]) * 10.0) * 1e3);
                if (fr && ((int)i == raw_op || raw_op == -1 || (raw_label && ops[i].label.find(raw_label) != std::string::npos)))
                    fprintf(fr, "%zu,%zu,%zu,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu\n", i, w / L.nwaves, w % L.nwaves,
                            r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[10]);
            }
            const double cm = med(chunks), lm = med(ph[2]);
            fprintf(f, "%zu,%s,%.4f,%d,%d,%d,%zu,%llu,%llu,%llu,%.0f,%.0f", i, op_kind_name(ops[i].kind), ops[i].flops / 1e9, L.tn, L.nwaves, L.nblk,
                    nw, (r1 - r0) * 10ull, (r0max - r0) * 10ull, (r1min - r0) * 10ull, med(mhz), cm);
            fprintf(f, ",%.0f,%.0f,%.0f,%.0f", med(su[0]), med(su[1]), med(su[2]), med(su[3]));
            for (int k = 0; k < 6; ++k) {
                fprintf(f, ",%.0f,%.0f", med(ph[k]), mx(ph[k]));
                if (k == 2) fprintf(f, ",%.0f", cm > 0 ? lm / cm : 0.0);
            }
            fprintf(f, ",%.0f,%.0f,%.0f,%.0f,%.0f,%s\n", med(ph[6]), mx(ph[6]), med(fd[0]), med(fd[1]), med(fd[2]), ops[i].label.c_str());
        }
    }
    fclose(f);
    if (fr) fclose(fr);
}
#endif

void Net::invalidate() {
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (void* p : owned) hipFree(p);
    owned.clear();
    packed.clear();
    baked.clear();
    norm_absmax.clear();
    derived.clear();
    rs_base = nullptr; rs_cap = 0; rs_top = 0;
    ops.clear();
    pre_ops.clear();
    emb_ops.clear();
}

void Net::run_ops(hipStream_t st) {
    for (auto& o : ops) o.fn(st);
}

const Param& Net::P(const std::string& n) const {
    auto it = params.find(n);
    if (it == params.end()) {
        it = derived.find(n);
        MUGD_CHECK(it != derived.end(), -5, "missing parameter: " + n);
    }
    return it->second;
}

// W = A B for two weight matrices (trailing unit kernel-width axes ignored): A (M, K), B (K, N) -> (M, N)
std::string Net::derive_product(const std::string& a_name, const std::string& b_name) {
    const std::string key = a_name + "*" + b_name;
    if (derived.count(key)) return key;
    const Param& A = P(a_name);
    const Param& B = P(b_name);
    MUGD_CHECK(A.dtype == 0 && B.dtype == 0 && A.shape.size() >= 2 && B.shape.size() >= 2, -2, "derive_product: bad operands " + key);
    const int M = (int)A.shape[0], K = (int)(A.numel() / A.shape[0]), N = (int)(B.numel() / B.shape[0]);
    MUGD_CHECK((int)B.shape[0] == K, -2, "derive_product: inner dimensions differ for " + key);
    float* C = dev_alloc((size_t)M * N);
    launch_derive_matmul(ctx->stream, DeriveMatmulArgs{(const float*)A.ptr, K, (const float*)B.ptr, N, nullptr, 0, C, N, M, N, K});
    Param p; p.ptr = C; p.dtype = 0; p.shape = {M, N};
    derived[key] = p;
    return key;
}

// A b + c for a weight matrix A (M, K) and two bias vectors
std::string Net::derive_bias(const std::string& a_name, const std::string& b_name, const std::string& c_name) {
    const std::string key = a_name + "*" + b_name + "+" + c_name;
    if (derived.count(key)) return key;
    const Param& A = P(a_name);
    const Param& b = P(b_name);
    const Param& c = P(c_name);
    const int M = (int)A.shape[0], K = (int)(A.numel() / A.shape[0]);
    MUGD_CHECK((int)b.numel() == K && (int)c.numel() == M, -2, "derive_bias: shape mismatch for " + key);
    float* out = dev_alloc(M);
    launch_derive_matmul(ctx->stream, DeriveMatmulArgs{(const float*)A.ptr, K, (const float*)b.ptr, 1, (const float*)c.ptr, 1, out, 1, M, 1, K});
    Param p; p.ptr = out; p.dtype = 0; p.shape = {M};
    derived[key] = p;
    return key;
}

float* Net::dev_alloc(size_t nfloats, bool zero) {
    float* p = nullptr;
    size_t bytes = std::max<size_t>(nfloats, 1) * sizeof(float) + 8192;     // slack: A-fragment prefetch may run past the end
    HIP_CHECK(hipMalloc((void**)&p, bytes));
    owned.push_back(p);
    if (zero) HIP_CHECK(hipMemsetAsync(p, 0, bytes, ctx->stream));
    return p;
}

// =======================================================================================
// layer emitters
// =======================================================================================
Tensor Net::group_norm(const std::string& prefix, const std::vector<Tensor>& segs, int groups, bool silu) {
    MUGD_CHECK(!segs.empty() && (int)segs.size() <= CONV_MAXSEG, -2, "group_norm: bad segment count");
    GnArgs a{};
    a.nseg = (int)segs.size();
    int C = 0;
    for (int i = 0; i < a.nseg; ++i) {
        a.seg[i] = NormSeg{segs[i].p, segs[i].C, segs[i].bmod};
        C += segs[i].C;
        MUGD_CHECK(segs[i].T == segs[0].T, -2, "group_norm: segment lengths differ at " + prefix);
    }
    MUGD_CHECK(P(prefix + ".weight").numel() == C, -2, "group_norm: weight size mismatch at " + prefix);
    Tensor y = talloc(C, segs[0].T);
    a.Ctot = C; a.T = segs[0].T; a.groups = groups; a.B = Bn; a.silu = silu ? 1 : 0;
    a.gamma = PF(prefix + ".weight"); a.beta = PF(prefix + ".bias");
    a.eps = 1e-6f; a.y = y.p;
    emit([a](hipStream_t st) { launch_group_norm(st, a); }, OP_GROUP_NORM, 0,
         prefix + " C=" + std::to_string(C) + " T=" + std::to_string(a.T));
    return y;
}

Tensor Net::layer_norm(const std::string& prefix, const Tensor& x) {
    Tensor y = talloc(x.C, x.T);
    LnArgs a{x.p, y.p, PF(prefix + ".weight"), PF(prefix + ".bias"), Bn, x.C, x.T, 1e-5f};
    emit([a](hipStream_t st) { launch_layer_norm(st, a); }, OP_LAYER_NORM, 0, prefix + " C=" + std::to_string(x.C) + " T=" + std::to_string(x.T));
    return y;
}

std::vector<ConvIn> Net::gn_inputs(const std::string& prefix, const std::vector<Tensor>& segs, int groups, bool silu, int taps, int dil, int pad) {
    if (ctx->fuse_norm) {
        std::vector<ConvIn> r = normed(segs, gn_stats(prefix, segs, groups), silu, taps, dil, pad);
        int Ctot = 0;
        for (auto& t : segs) Ctot += t.C;
        const float sx0 = norm_scale(prefix, (double)std::max(1, Ctot / groups) * segs[0].T);
        for (auto& in : r) {
            in.xf.sx0 = sx0;
            if (ctx->fast_act && in.xf.act == 1) in.xf.act = 2;
        }
        return r;
    }
    return {ConvIn{group_norm(prefix, segs, groups, silu), taps, dil, 1, pad, 0}};
}

void Net::gn_inputs(ConvSpec& spec, const std::string& prefix, const std::vector<Tensor>& segs, int groups, bool silu, int taps, int dil, int pad) {
    bool have = ctx->fuse_norm && ctx->fuse_stats && groups <= 32 && segs[0].T % 4 == 0 && (taps == 1 || true);
    int Ctot = 0;
    for (auto& t : segs) { have = have && t.rowstat != nullptr; Ctot += t.C; }
    if (!have || Ctot % groups != 0) {
        spec.in = gn_inputs(prefix, segs, groups, silu, taps, dil, pad);
        return;
    }
    const float* tab = norm_table(prefix, Ctot);
    int off = 0;
    spec.in.clear();
    for (auto& t : segs) {
        ConvIn in{t, taps, dil, 1, pad, 0};
        in.xf.kind = 4; in.xf.act = silu ? (ctx->fast_act ? 2 : 1) : 0;
        in.xf.sx0 = norm_scale(prefix, (double)(Ctot / groups) * segs[0].T);
        in.xf.a = reinterpret_cast<const float*>(t.rowstat); in.xf.b = tab;
        in.xf.stride = 2 * t.C;                 // doubles per batch row
        in.xf.coff = off;
        spec.in.push_back(in);
        off += t.C;
    }
    spec.gn.nseg = (int)segs.size(); spec.gn.groups = groups; spec.gn.cg = Ctot / groups;
    spec.gn.count = (float)(Ctot / groups) * (float)segs[0].T; spec.gn.eps = 1e-6f;
    // GROUP tables (round 6; ConvArgs::gn_table / gsink): when every segment was produced by a conv launch of THIS program that still has a free
    // sink, those launches add their tiles' sums per group of this domain and the consumer's prologue shrinks to one load per lane.  Eligibility
    // depends on shapes and emission order only, so the dry pass and the real pass allocate alike (every kernel form -- K-split and M-split --
    // has the group combine in its epilogue).
    // Measured on one box, six alternating runs each (profiles/r6_gn_group_ab.txt): -2.9 % per DDIM step at batch 4, -1.8 % at batch 8, -1.2 % at
    // batch 16 (with the first form of the producers' combine -- a loop over the group's rows -- it was -1.5 % / -0.5 % / +0.3 %).
    // MUGD_GN_GROUP=0 switches the tables off (A/B arm).
    static const bool group_on = !(getenv("MUGD_GN_GROUP") && atoi(getenv("MUGD_GN_GROUP")) == 0);
    bool eligible = group_on;
    for (auto& t : segs) eligible = eligible && t.prod >= 0 && t.prod < (int)prods.size() && t.bmod == 0 && prods[t.prod].rows && prods[t.prod].nsink < 2;
    for (size_t i = 0; i < segs.size() && eligible; ++i)          // (one launch feeding two segments of the same domain would need both its sinks)
        for (size_t j = 0; j < i; ++j) eligible = eligible && segs[i].prod != segs[j].prod;
    if (eligible) {
        double* table = alloc_rowstat((size_t)Bn * 64);
        bool ok = true;
        for (auto& t : segs) { prods[t.prod].nsink++; ok = ok && (dry || prods[t.prod].L != nullptr); }
        if (ok && !dry) {
            off = 0;
            for (auto& t : segs) {
                ConvArgs& pa = prods[t.prod].L->a;
                const int k = pa.gsink[0].p ? 1 : 0;
                pa.gsink[k].p = table; pa.gsink[k].coff = off; pa.gsink[k].cg = Ctot / groups;
                off += t.C;
            }
            spec.gn.table = table;
            if (getenv("MUGD_GN_GROUP_LOG")) fprintf(stderr, "[mugd] group table: %s (%d segment(s), %d channels per group)\n", prefix.c_str(), (int)segs.size(), Ctot / groups);
        }
    }
    if (!spec.gn.table) for (auto& t : segs) use_rowstat(t);      // this consumer maps, fetches and reduces its producers' ROW sums
}

// Launches that feed group tables and whose row sums no consumer reads (the next block's GroupNorm and the skip concat both took the group
// form) stop accumulating rows: 32 fp64 atomic pairs per tile less.  MUGD_GN_KEEP_ROWS=1 keeps them (A/B arm).
void Net::finish_stats() {
    static const bool keep = getenv("MUGD_GN_KEEP_ROWS") && atoi(getenv("MUGD_GN_KEEP_ROWS")) != 0;
    int n = 0;
    for (auto& pr : prods)
        if (pr.L && pr.L->a.gsink[0].p && pr.row_users == 0 && !keep) { pr.L->a.rowstat = nullptr; ++n; }
    if (getenv("MUGD_GN_GROUP_LOG")) fprintf(stderr, "[mugd] %d of %d statistic-producing launches accumulate group sums only; accumulator block: %zu of %zu doubles, %zu cleared per evaluation\n", n, (int)prods.size(), rs_top, rs_cap, rs_zero_n);
}

ConvIn Net::ln_input(const std::string& prefix, const Tensor& x) {
    if (ctx->fuse_norm) {
        ConvIn in{x};
        in.xf = layer_norm_xf(prefix, x);
        return in;
    }
    return ConvIn{layer_norm(prefix, x)};
}

// weight blocks of one tensor laid over consecutive conv inputs [first, first + n)
static void span_weights(ConvSpec& s, const std::string& wname, size_t first, size_t n, int row_off = 0) {
    int ci = 0;
    for (size_t i = first; i < first + n; ++i) {
        s.w.push_back(WBlock{wname, (int)i, row_off, ci});
        ci += s.in[i].x.C;
    }
}

const float* Net::gn_stats(const std::string& prefix, const std::vector<Tensor>& segs, int groups) {
    MUGD_CHECK(!segs.empty() && (int)segs.size() <= CONV_MAXSEG, -2, "group_norm: bad segment count");
    GnStatArgs a{};
    a.nseg = (int)segs.size();
    int C = 0;
    for (int i = 0; i < a.nseg; ++i) {
        a.seg[i] = NormSeg{segs[i].p, segs[i].C, segs[i].bmod};
        C += segs[i].C;
        MUGD_CHECK(segs[i].T == segs[0].T, -2, "group_norm: segment lengths differ at " + prefix);
    }
    MUGD_CHECK(P(prefix + ".weight").numel() == C, -2, "group_norm: weight size mismatch at " + prefix);
    float* aff = arena.alloc((size_t)Bn * C * 2);
    a.Ctot = C; a.T = segs[0].T; a.groups = groups; a.B = Bn;
    a.gamma = PF(prefix + ".weight"); a.beta = PF(prefix + ".bias");
    a.eps = 1e-6f; a.aff = aff;
    emit([a](hipStream_t st) { launch_gn_stats(st, a); }, OP_GROUP_NORM, 0,
         prefix + " C=" + std::to_string(C) + " T=" + std::to_string(a.T));
    return aff;
}

std::vector<ConvIn> Net::normed(const std::vector<Tensor>& segs, const float* aff, bool silu, int taps, int dil, int pad) {
    int Ctot = 0;
    for (auto& t : segs) Ctot += t.C;
    std::vector<ConvIn> r;
    int off = 0;
    for (auto& t : segs) {
        ConvIn in{t, taps, dil, 1, pad, 0};
        in.xf.kind = 1; in.xf.act = silu ? 1 : 0; in.xf.a = aff + 2 * (size_t)off; in.xf.stride = 2 * Ctot;
        r.push_back(in);
        off += t.C;
    }
    return r;
}

float* Net::norm_table(const std::string& prefix, int C) {
    const std::string key = prefix + "#gb";
    auto it = baked.find(key);
    if (it != baked.end()) return it->second;
    MUGD_CHECK(P(prefix + ".weight").numel() == C, -2, "norm: weight size mismatch at " + prefix);
    float* gb = dev_alloc((size_t)C * 2);
    launch_interleave2(ctx->stream, PF(prefix + ".weight"), PF(prefix + ".bias"), gb, C);
    return baked[key] = gb;
}

// |GroupNorm(x)| <= max|gamma| sqrt(n - 1) + max|beta| over the n elements of a group (LayerNorm: the C channels of a sample), SiLU only shrinks it:
// the static H3 scale of the operand (kernels.h: h3_static_scale).  The two maxima are read back once per layer and parameter set.
float Net::norm_scale(const std::string& prefix, double n) {
    auto it = norm_absmax.find(prefix);
    if (it == norm_absmax.end()) {
        const Param& g = P(prefix + ".weight");
        const Param& b = P(prefix + ".bias");
        unsigned* w = reinterpret_cast<unsigned*>(dev_alloc(2, true));
        launch_absmax(ctx->stream, (const float*)g.ptr, (long long)g.numel(), w);
        launch_absmax(ctx->stream, (const float*)b.ptr, (long long)b.numel(), w + 1);
        unsigned bits[2] = {0, 0};
        HIP_CHECK(hipMemcpyAsync(bits, w, sizeof(bits), hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        float gm, bm;
        memcpy(&gm, &bits[0], 4); memcpy(&bm, &bits[1], 4);
        it = norm_absmax.emplace(prefix, std::make_pair(gm, bm)).first;
    }
    return h3_static_scale(it->second.first, it->second.second, n);
}

double* Net::alloc_rowstat(size_t nd) {
    const size_t off = rs_top;
    rs_top += (nd + 31) / 32 * 32;
    if (dry) return reinterpret_cast<double*>(256 + off * sizeof(double));      // never dereferenced
    MUGD_CHECK(rs_top <= rs_cap, -9, "internal: row-sum block overflow");
    return rs_base + off;
}

void Net::begin_rowstat() {
    prods.clear();
    launches.clear();                            // (the ops that pointed into it were cleared by build())
    if (!dry) {                                  // sized by the dry pass that just ran
        if (rs_top > rs_cap) {
            rs_cap = rs_top;
            rs_base = reinterpret_cast<double*>(dev_alloc(rs_cap * 2));
        }
        rs_zero_op = -1; rs_zero_n = 0;
        if (rs_cap) {
            double* p = rs_base; const size_t bytes = rs_top * sizeof(double);
            rs_zero_op = (int)ops.size(); rs_zero_n = rs_top;       // (the DDIM loop zeroes them from its per-step kernel instead: UNet::step_body)
            emit([p, bytes](hipStream_t st) { HIP_CHECK(hipMemsetAsync(p, 0, bytes, st)); }, OP_SMALL, 0, "zero row-sum accumulators");
        }
    }
    rs_top = 0;
}

Xf Net::layer_norm_xf(const std::string& prefix, const Tensor& x) {
    float* gb = norm_table(prefix, x.C);
    if (x.colstat && x.T % 4 == 0 && ctx->fuse_stats) {             // statistics come from the producer conv's epilogue: no launch
        Xf xf;
        xf.kind = 3; xf.a = x.colstat; xf.b = gb; xf.stride = 2 * x.colstat_np * x.T; xf.np = x.colstat_np; xf.eps = 1e-5f;
        xf.sx0 = norm_scale(prefix, (double)x.C);
        // Column SUMS (round 6; ConvArgs::colsum): the producer's tiles add their columns' sums to one fp64 pair per column (cleared with the
        // row-sum block) instead of storing a part per row tile; the consumer loads the finished pair -- no parts to sum through LDS, no
        // workgroup barrier in its prologue (conv_stats.h; xf.np == 0 marks the form).  Measured (profiles/r6_ln_sums_ab.txt, six alternating runs
        // each on one box): +0.1 % per DDIM step at batch 4 (noise), -0.2 % at batch 8, -0.6 % at batch 16: on at every batch size;
        // MUGD_LN_SUMS=0 keeps the per-row-tile parts (A/B arm).
        static const bool sums_on = !(getenv("MUGD_LN_SUMS") && atoi(getenv("MUGD_LN_SUMS")) == 0);
        const bool sums = sums_on && x.prod >= 0 && x.prod < (int)prods.size() && x.bmod == 0;
        if (sums) {
            Prod& pr = prods[x.prod];
            if (!pr.colsum) pr.colsum = alloc_rowstat((size_t)Bn * x.T * 2);
            if (!dry && pr.L) { pr.L->a.colsum = pr.colsum; pr.L->a.colstat = nullptr; }
            xf.a = reinterpret_cast<const float*>(pr.colsum); xf.np = 0; xf.stride = 0;
            if (!dry && getenv("MUGD_GN_GROUP_LOG")) fprintf(stderr, "[mugd] column sums: %s\n", prefix.c_str());
        }
        return xf;
    }
    float* stat = arena.alloc((size_t)Bn * x.T * 2);
    LnStatArgs a{x.p, stat, Bn, x.C, x.T, 1e-5f};
    emit([a](hipStream_t st) { launch_ln_stats(st, a); }, OP_LAYER_NORM, 0, prefix + " C=" + std::to_string(x.C) + " T=" + std::to_string(x.T));
    Xf xf;
    xf.kind = 2; xf.a = stat; xf.b = gb; xf.stride = 2 * x.T;
    xf.sx0 = norm_scale(prefix, (double)x.C);
    return xf;
}

const PackedW& Net::get_packed(const ConvSpec& s, int tn, bool w16) {
    const std::string pkey = (tn == 16 ? s.key + "#16" : s.key) + (w16 ? "#bf16" : "");
    auto it = packed.find(pkey);
    if (it != packed.end()) return it->second;
    PackedW pw;
    int chunk = 0, woff = 0;
    for (auto& in : s.in) {
        MUGD_CHECK(in.x.C % CONV_CK == 0, -2, "conv: channel count not a multiple of 16 at " + s.key);
        pw.chunk0.push_back(chunk);
        pw.woff.push_back(woff);
        chunk += in.x.C / CONV_CK;
        woff += (in.x.C / CONV_CK) * in.taps * 512;
    }
    pw.nchunk = chunk;
    pw.mt_stride = woff;
    const int MT = cdiv(s.Mrows, 32);
    pw.wpk = dev_alloc(w16 ? ((size_t)MT * pw.mt_stride + 1) / 2 : (size_t)MT * pw.mt_stride, true);       // bf16: half the bytes
    if (!w16) pw.wmax = reinterpret_cast<unsigned*>(dev_alloc(1, true));
    // pass 0: max |w| over every block of the set (the H3 weight scale is one power of two per set); pass 1: pack
    for (int pass = 0; pass < 2; ++pass) {
        for (auto& wb : s.w) {
            const Param& p = P(wb.name);
            MUGD_CHECK(p.dtype == 0 && p.shape.size() >= 2, -2, "conv: bad weight tensor " + wb.name);
            const ConvIn& in = s.in[wb.seg];
            const int rows = (int)p.shape[0];
            const int cw = (int)p.shape[1];
            const int tw = p.shape.size() > 2 ? (int)p.shape[2] : 1;
            MUGD_CHECK(tw == in.taps, -2, "conv: kernel width mismatch for " + wb.name);
            MUGD_CHECK(wb.ci_off + in.x.C <= cw && wb.row_off + rows <= s.Mrows, -2, "conv: weight block out of range: " + wb.name);
            PackArgs pa{pw.wpk, pw.mt_stride, pw.woff[wb.seg], in.x.C, in.taps, (const float*)p.ptr, cw * tw, wb.ci_off, rows, wb.row_off, w16 ? 1 : 0, pw.wmax};
            if (pass == 0) launch_weight_absmax(ctx->stream, pa);
            else if (tn == 16) launch_pack_weights16(ctx->stream, pa);
            else launch_pack_weights(ctx->stream, pa);
        }
        if (pass == 0 && pw.wmax) {          // once per set and parameter load: the kernels get 1 / S_w by value instead of a cold load per wave
            unsigned bits = 0;
            HIP_CHECK(hipMemcpyAsync(&bits, pw.wmax, sizeof(bits), hipMemcpyDeviceToHost, ctx->stream));
            HIP_CHECK(hipStreamSynchronize(ctx->stream));
            pw.winv = h3_pow2_recip(h3_wscale(bits));
        }
    }
    if (!s.bias.empty()) {
        pw.bias = dev_alloc(s.Mrows, true);
        for (auto& b : s.bias) {
            const Param& p = P(b.first);
            MUGD_CHECK(b.second + p.numel() <= s.Mrows, -2, "conv: bias out of range: " + b.first);
            launch_bias_sum(ctx->stream, pw.bias + b.second, (const float*)p.ptr, pw.bias + b.second, (int)p.numel());
        }
    }
    return packed[pkey] = pw;
}

// Per-call, per-batch-row weights (ConvSpec::ext_plain): the packed copy is library-owned and re-filled by a pre-op on every call.
const PackedW& Net::get_packed_ext(const ConvSpec& s, int tn) {
    MUGD_CHECK(s.in.size() == 1 && s.in[0].taps == 1, -2, "conv: external weights need a single 1x1 input at " + s.key);
    const std::string pkey = s.key + "#ext" + std::to_string(Bn) + (tn == 16 ? "#16" : "");
    auto it = packed.find(pkey);
    if (it == packed.end()) {
        PackedW pw;
        const int K = s.in[0].x.C;
        MUGD_CHECK(K % CONV_CK == 0, -2, "conv: channel count not a multiple of 16 at " + s.key);
        pw.chunk0 = {0}; pw.woff = {0};
        pw.nchunk = K / CONV_CK;
        pw.mt_stride = (long long)pw.nchunk * 512;
        pw.wpk = dev_alloc((size_t)Bn * cdiv(s.Mrows, 32) * pw.mt_stride, true);
        pw.wmax = reinterpret_cast<unsigned*>(dev_alloc(1, true));
        if (!s.bias.empty()) {
            pw.bias = dev_alloc(s.Mrows, true);
            for (auto& b : s.bias) {
                const Param& p = P(b.first);
                MUGD_CHECK(b.second + p.numel() <= s.Mrows, -2, "conv: bias out of range: " + b.first);
                launch_bias_sum(ctx->stream, pw.bias + b.second, (const float*)p.ptr, pw.bias + b.second, (int)p.numel());
            }
        }
        it = packed.emplace(pkey, pw).first;
    }
    const PackedW& pw = it->second;
    {   // per call: (re)pack every batch row's weight set
        const int K = s.in[0].x.C, M = s.Mrows, B = Bn;
        const float* src = s.ext_plain; float* dst = pw.wpk; const long long mts = pw.mt_stride;
        unsigned* wmax = pw.wmax;
        const long long bstride = (long long)cdiv(M, 32) * mts;
        const bool save = to_pre;
        to_pre = true;
        emit([=](hipStream_t st) {
            // one H3 weight scale for the B weight sets of the call: max |w| over all of them (the B sets are contiguous rows of src)
            HIP_CHECK(hipMemsetAsync(wmax, 0, sizeof(unsigned), st));
            launch_weight_absmax(st, PackArgs{nullptr, 0, 0, K, 1, src, K, 0, B * M, 0, 0, wmax});
            for (int b = 0; b < B; ++b) {
                PackArgs pa{dst + (size_t)b * bstride, mts, 0, K, 1, src + (size_t)b * M * K, K, 0, M, 0, 0, wmax};
                if (tn == 16) launch_pack_weights16(st, pa); else launch_pack_weights(st, pa);
            }
        }, OP_SMALL, 0, s.key + " pack per-call weights");
        to_pre = save;
    }
    return pw;
}

Tensor Net::conv(const ConvSpec& s) {
    MUGD_CHECK(!s.in.empty() && (int)s.in.size() <= CONV_MAXSEG, -2, "conv: bad segment count at " + s.key);
    Tensor y = s.out.p ? s.out : talloc(s.Mout, s.Tout);
    y.prod = -1;
    ConvArgs a{};
    a.nseg = (int)s.in.size();
    for (int i = 0; i < a.nseg; ++i) {
        const ConvIn& in = s.in[i];
        a.seg[i] = ConvSeg{in.x.p, in.x.C, in.x.T, in.taps, in.dil, in.stride, in.pad, in.ups, 0, 0, in.x.bmod,
                           in.xf.kind, in.xf.act, in.xf.a, in.xf.b, in.xf.stride, in.xf.np, in.xf.eps, in.xf.coff, in.xf.sx0};
    }
    a.gn_nseg = s.gn.nseg; a.gn_groups = s.gn.groups; a.gn_cg = s.gn.cg; a.gn_count = s.gn.count; a.gn_eps = s.gn.eps; a.gn_table = s.gn.table;
    a.B = Bn; a.Mrows = s.Mrows; a.Mout = s.Mout; a.Tout = s.Tout; a.epi = s.epi;
    a.wk = ctx->force_wk;
    a.tn = ctx->force_tn == 16 ? (conv16_supported(a) ? 16 : 32) : ctx->force_tn == 32 ? 32 : conv_pick_tn(a);
    a.xs_rel = s.xs_rel; a.xs_cemb = s.xs_cemb; a.xs_heads = s.xs_heads; a.xs_pmax = s.xs_pmax; a.xs_ntok = s.xs_ntok; a.xs_scale = s.xs_scale;
    a.w16 = (ctx->weights_bf16 && !s.ext_plain && conv_w16_supported(a)) ? 1 : 0;
    const PackedW& pw = s.ext_plain ? get_packed_ext(s, a.tn) : get_packed(s, a.tn, a.w16 != 0);       // the tile width decides the weight fragment order
    for (int i = 0; i < a.nseg; ++i) { a.seg[i].chunk0 = pw.chunk0[i]; a.seg[i].woff = pw.woff[i]; }
    a.wpk = pw.wpk; a.w_mt_stride = pw.mt_stride; a.bias = pw.bias;
    if (s.ext_plain) a.wmax = pw.wmax; else a.winv = pw.winv;        // H3 weight scale: device word for per-call sets, by value otherwise
    a.w_b_stride = s.ext_plain ? (long long)cdiv(s.Mrows, 32) * pw.mt_stride : 0;
    a.rowadd = s.rowadd; a.rowadd_stride = s.rowadd_stride;
    a.resid = s.resid.p;
    if (s.resid.p) MUGD_CHECK(s.resid.C == s.Mout && s.resid.T == s.Tout && s.resid.bmod == 0, -2, "conv: residual shape mismatch at " + s.key);
    a.y = y.p; a.nchunk = pw.nchunk;
    bool rowstat_pass = false;
    if (s.want_rowstat && ctx->fuse_norm && ctx->fuse_stats && s.epi == EPI_NONE) {
        y.rowstat = alloc_rowstat((size_t)Bn * s.Mout * 2);
        // Every tile ADDS its rows' sums with fp64 atomics: cdiv(T, tn) of them queue on each address (~0.2 us apiece across
        // the XCDs).  16 per address in the U-Net -- hidden under the other tiles; 128..1024 in the wave encoder and the VAE
        // decoder, where they WERE the layer's duration (T = 32768: 213 us for 21 us of MFMA work).  Long rows take one more
        // pass over the output instead.
        static const int max_tiles = getenv("MUGD_ROWSTAT_MAX_TILES") ? atoi(getenv("MUGD_ROWSTAT_MAX_TILES")) : CONV_ROWSTAT_MAX_TILES;
        rowstat_pass = cdiv(s.Tout, a.tn) > max_tiles;
        a.rowstat = rowstat_pass ? nullptr : y.rowstat;
    }
    if (s.want_colstat && ctx->fuse_stats && ctx->fuse_norm && s.epi == EPI_NONE && s.Tout % 4 == 0) {
        bool fast = true;                                   // only the fast-window kernels emit column sums
        for (auto& in : s.in) fast = fast && in.stride == 1 && !in.ups && in.x.T % 4 == 0 && (in.taps == 1 || in.dil == 1);
        if (fast) {
            y.colstat_np = cdiv(s.Mout, 32);
            y.colstat = arena.alloc((size_t)Bn * y.colstat_np * s.Tout * 2);
            a.colstat = y.colstat;
        }
    }
    double kdim = 0;
    for (auto& in : s.in) kdim += (double)in.x.C * in.taps;
    const std::string label = s.key + " M=" + std::to_string(s.Mrows) + " K=" + std::to_string((long long)kdim) + " T=" + std::to_string(s.Tout) +
                              " nseg=" + std::to_string(a.nseg) + " tn=" + std::to_string(a.tn);
    const int okind = (s.epi == EPI_GLU || s.epi == EPI_GEGLU) ? OP_CONV_GATED : OP_CONV;
    const double oflops = 2.0 * s.Mrows * kdim * s.Tout * Bn;
    ConvLaunch* Lp = nullptr;
    if (!dry) {
        // validated, its kernel form chosen and its K-slices / grid decode filled in NOW; a step only launches (kernels.h: ConvLaunch).  The block
        // lives in `launches`: a consumer compiled later may still attach a group table to it (gn_inputs)
        launches.push_back(conv_prepare(a));
        Lp = &launches.back();
        emit([Lp](hipStream_t st) { conv_launch(st, *Lp); }, okind, oflops, label);
    }
    if ((a.rowstat || a.colstat) && s.epi == EPI_NONE) { y.prod = (int)prods.size(); prods.push_back(Prod{Lp, 0, 0, a.rowstat != nullptr, nullptr}); }
    if (rowstat_pass) {
        const float* yp = y.p; double* rp = y.rowstat; const int rows = Bn * s.Mout, T = s.Tout;
        emit([=](hipStream_t st) { launch_row_sums_add(st, yp, rp, rows, T); }, OP_SMALL, 0, s.key + " row sums");
    }
    return y;
}

Tensor Net::conv_simple(const std::string& prefix, const Tensor& x, int taps, int dil, int stride, int pad, int ups,
                        int Tout, const Tensor& resid, const Tensor& out) {
    ConvSpec s;
    s.key = prefix;
    s.in.push_back(ConvIn{x, taps, dil, stride, pad, ups});
    s.w.push_back(WBlock{prefix + ".weight", 0, 0, 0});
    if (has(prefix + ".bias")) s.bias.push_back({prefix + ".bias", 0});
    s.Mrows = s.Mout = (int)P(prefix + ".weight").shape[0];
    s.Tout = Tout;
    s.resid = resid;
    s.out = out;
    s.want_rowstat = true;               // every plain conv on these paths feeds a GroupNorm (next block / skip connection)
    return conv(s);
}

Tensor Net::attention(const std::string& prefix, const Tensor& q, const Tensor& k, const Tensor& v, int C, int heads,
                      int q_off, int k_off, int v_off) {
    Tensor o = talloc(C, q.T);
    AttnArgs a{};
    a.q = q.p + (size_t)q_off * q.T; a.q_bstride = q.C * q.T;
    a.k = k.p + (size_t)k_off * k.T; a.k_bstride = k.C * k.T;
    a.v = v.p + (size_t)v_off * v.T; a.v_bstride = v.C * v.T;
    a.out = o.p; a.o_bstride = C * q.T;
    const Param& rel = P(prefix + ".relative_position_embedding");
    a.rel = (const float*)rel.ptr; a.cemb = PF(prefix + ".C_embedding");
    a.pmax = (int)(rel.shape[0] - 1) / 2;
    MUGD_CHECK((int)rel.shape[1] == heads, -2, "attention: head count mismatch at " + prefix);
    a.B = Bn; a.heads = heads; a.d = C / heads; a.Tq = q.T; a.Tk = k.T;
    a.scale = 1.0f / sqrtf((float)a.d);
    const std::string label = prefix + " d=" + std::to_string(a.d) + " Tq=" + std::to_string(q.T) + " Tk=" + std::to_string(k.T);
    emit([a](hipStream_t st) { launch_attention(st, a); }, OP_ATTENTION, 4.0 * Bn * C * (double)q.T * k.T, label);
    return o;
}

// mug/model/attention.py:154-199 (depth 1).  context == nullptr: attn2 is a second self-attention.
// GroupNorm and the three LayerNorms are statistics kernels + operand transforms of the convs that
// consume them; the cross-attention K/V projection of the (step-invariant) context goes to pre_ops.
Tensor Net::transformer(const std::string& prefix, const Tensor& x, const Tensor* context, int heads) {
    const int C = x.C, T = x.T;
    Tensor out = talloc(C, T);
    const std::string b = prefix + ".transformer_blocks.0";
    Tensor kv;
    if (context) {                                   // computed once per call: must not share the arena with per-step scratch
        const std::string kk = b + ".attn2.kv#" + std::to_string(Bn) + "x" + std::to_string(context->T);
        auto it = baked.find(kk);
        kv.p = it != baked.end() ? it->second : (baked[kk] = dev_alloc((size_t)Bn * 2 * C * context->T));
        kv.C = 2 * C; kv.T = context->T;
        const bool save = to_pre;
        to_pre = true;
        ConvSpec s;
        s.key = b + ".attn2.kv";
        s.in = {ConvIn{*context}};
        s.w = {WBlock{b + ".attn2.to_k.weight", 0, 0, 0}, WBlock{b + ".attn2.to_v.weight", 0, C, 0}};
        s.Mrows = s.Mout = 2 * C; s.Tout = context->T; s.out = kv;
        conv(s);
        to_pre = save;
    }
    const size_t mk = arena.mark();
    auto lin = [&](const std::string& key, const ConvIn& in, bool bias, const Tensor& resid, const Tensor& dst, bool feeds_ln = false,
                   bool feeds_gn = false, const ConvSpec* proto = nullptr) {
        ConvSpec s;
        if (proto) s = *proto;               // inputs (+ GroupNorm domain) prepared by the caller
        else s.in = {in};
        s.key = key;
        s.want_colstat = feeds_ln;
        s.want_rowstat = feeds_gn;
        s.w = {WBlock{key + ".weight", 0, 0, 0}};
        if (bias) s.bias = {{key + ".bias", 0}};
        s.Mrows = s.Mout = (int)P(key + ".weight").shape[0];
        s.Tout = T; s.resid = resid; s.out = dst;
        return conv(s);
    };
    auto self_attn = [&](const std::string& ap, const ConvIn& in) {
        ConvSpec s;
        s.key = ap + ".qkv";
        s.in = {in};
        s.w = {WBlock{ap + ".to_q.weight", 0, 0, 0}, WBlock{ap + ".to_k.weight", 0, C, 0}, WBlock{ap + ".to_v.weight", 0, 2 * C, 0}};
        s.Mrows = s.Mout = 3 * C; s.Tout = T;
        Tensor qkv = conv(s);
        return attention(ap, qkv, qkv, qkv, C, heads, 0, C, 2 * C);
    };

    ConvSpec pin;
    gn_inputs(pin, prefix + ".norm", {x}, 32, false, 1, 1, 0);
    Tensor h0 = lin(prefix + ".proj_in", ConvIn{}, true, Tensor(), Tensor(), true, false, &pin);
    Tensor a1 = self_attn(b + ".attn1", ln_input(b + ".norm1", h0));
    Tensor h1 = lin(b + ".attn1.to_out.0", ConvIn{a1}, true, h0, Tensor(), true);
    const ConvIn n2 = ln_input(b + ".norm2", h1);
    Tensor a2, h2;
    const bool fold = context && ctx->fold_xattn && context->T <= 32 && C % heads == 0 &&
                      P(b + ".attn2.to_q.weight").numel() == (long long)C * C && P(b + ".attn2.to_out.0.weight").numel() == (long long)C * C;
    if (fold) {
        // Folded cross-attention (kernels.h: XattnFoldArgs, EPI_XSOFTMAX): the key / value side is fixed for the whole call, so
        // to_q and to_out collapse into per-batch-row weight sets G (32 heads x C) and U (C x 32 heads), refreshed by a pre-op:
        //   P = softmax_j((G LN(h1) + Rel) scale) * Cemb      one conv_gemm launch, M = 32 per head
        //   h2 = U P + bias + h1                              one conv_gemm launch, K = 32 per head
        // instead of to_q -> attention kernel -> to_out (3 launches; at C = 512 also half the multiply-adds).
        const int R = 32 * heads, ntok = context->T;
        const std::string gk = b + ".attn2.fold#" + std::to_string(Bn) + "x" + std::to_string(ntok);
        auto it = baked.find(gk);
        float* GU = it != baked.end() ? it->second : (baked[gk] = dev_alloc((size_t)Bn * R * C * 2));
        float* G = GU; float* U = GU + (size_t)Bn * R * C;
        {
            XattnFoldArgs fa{PF(b + ".attn2.to_q.weight"), PF(b + ".attn2.to_out.0.weight"), kv.p, G, U, Bn, C, heads, C / heads, ntok};
            to_pre = true;
            emit([fa](hipStream_t st) { launch_xattn_fold(st, fa); }, OP_SMALL, 0, b + ".attn2 fold K/V into the projections");
            to_pre = false;
        }
        const Param& rel = P(b + ".attn2.relative_position_embedding");
        MUGD_CHECK((int)rel.shape[1] == heads, -2, "attention: head count mismatch at " + b + ".attn2");
        ConvSpec sc;
        sc.key = b + ".attn2.scores";
        sc.in = {n2};
        sc.ext_plain = G;
        sc.Mrows = sc.Mout = R; sc.Tout = T; sc.epi = EPI_XSOFTMAX;
        sc.xs_rel = (const float*)rel.ptr; sc.xs_cemb = PF(b + ".attn2.C_embedding");
        sc.xs_heads = heads; sc.xs_pmax = (int)(rel.shape[0] - 1) / 2; sc.xs_ntok = ntok;
        sc.xs_scale = 1.0f / sqrtf((float)(C / heads));
        Tensor pc = conv(sc);
        ConvSpec so;
        so.key = b + ".attn2.to_out.0";
        so.in = {ConvIn{pc}};
        so.ext_plain = U;
        so.bias = {{so.key + ".bias", 0}};
        so.Mrows = so.Mout = C; so.Tout = T; so.resid = h1; so.want_colstat = true;
        h2 = conv(so);
    } else {
        if (context) {
            Tensor q2 = lin(b + ".attn2.to_q", n2, false, Tensor(), Tensor());
            a2 = attention(b + ".attn2", q2, kv, kv, C, heads, 0, 0, C);
        } else {
            a2 = self_attn(b + ".attn2", n2);
        }
        h2 = lin(b + ".attn2.to_out.0", ConvIn{a2}, true, h1, Tensor(), true);
    }
    Tensor f;
    {
        ConvSpec ff;
        ff.key = b + ".ff.net.0.proj";
        ff.in = {ln_input(b + ".norm3", h2)};
        ff.w = {WBlock{ff.key + ".weight", 0, 0, 0}};
        ff.bias = {{ff.key + ".bias", 0}};
        ff.Mrows = (int)P(ff.key + ".weight").shape[0];
        ff.Mout = ff.Mrows / 2; ff.Tout = T; ff.epi = EPI_GEGLU;
        f = conv(ff);
    }
    if (ctx->fold_proj_out) {
        // out = x + Wp (h2 + W2 f + b2) + bp = x + (Wp W2) f + Wp h2 + (Wp b2 + bp): ONE launch over the K-segments [f | h2] with the
        // product matrix computed once (fp64 accumulation) when the parameters are packed, instead of ff.net.2 and proj_out
        ConvSpec s;
        s.key = prefix + ".proj_out*ff.net.2";
        s.in = {ConvIn{f}, ConvIn{h2}};
        s.w = {WBlock{derive_product(prefix + ".proj_out.weight", b + ".ff.net.2.weight"), 0, 0, 0}, WBlock{prefix + ".proj_out.weight", 1, 0, 0}};
        s.bias = {{derive_bias(prefix + ".proj_out.weight", b + ".ff.net.2.bias", prefix + ".proj_out.bias"), 0}};
        s.Mrows = s.Mout = C; s.Tout = T; s.resid = x; s.out = out; s.want_rowstat = true;
        out = conv(s);
    } else {
        Tensor h3 = lin(b + ".ff.net.2", ConvIn{f}, true, h2, Tensor());
        out = lin(prefix + ".proj_out", ConvIn{h3}, true, x, out, false, true);
    }
    arena.release(mk);
    return out;
}

// mug/model/models.py:94-159 (temb_channels = 0 on this path)
Tensor Net::resnet_block(const std::string& prefix, const Tensor& x, int Cout, int groups, int d0, int d1) {
    const int T = x.T;
    Tensor out = talloc(Cout, T);
    const size_t mk = arena.mark();
    Tensor h1;
    {
        ConvSpec s;
        s.key = prefix + ".conv1";
        gn_inputs(s, prefix + ".norm1", {x}, groups, true, 3, d0, d0);
        span_weights(s, s.key + ".weight", 0, s.in.size());
        s.bias = {{s.key + ".bias", 0}};
        s.Mrows = s.Mout = Cout; s.Tout = T; s.want_rowstat = true;
        h1 = conv(s);
    }
    ConvSpec s;
    gn_inputs(s, prefix + ".norm2", {h1}, groups, true, 3, d1, d1);
    span_weights(s, prefix + ".conv2.weight", 0, s.in.size());
    s.bias = {{prefix + ".conv2.bias", 0}};
    s.Mrows = s.Mout = Cout; s.Tout = T; s.out = out; s.want_rowstat = true;
    if (has(prefix + ".nin_shortcut.weight")) {
        s.key = prefix + ".conv2+nin";
        s.in.push_back(ConvIn{x});
        span_weights(s, prefix + ".nin_shortcut.weight", s.in.size() - 1, 1);
        s.bias.push_back({prefix + ".nin_shortcut.bias", 0});
    } else {
        MUGD_CHECK(x.C == Cout, -2, "resnet_block without shortcut must keep channels: " + prefix);
        s.key = prefix + ".conv2";
        s.resid = x;
    }
    out = conv(s);
    arena.release(mk);
    return out;
}

Tensor Net::downsample(const std::string& prefix, const Tensor& x) {      // models.py:73-91
    return conv_simple(prefix + ".conv", x, 3, 1, 2, 0, 0, (x.T + 1 - 3) / 2 + 1);
}
Tensor Net::upsample(const std::string& prefix, const Tensor& x) {        // models.py:55-70
    return conv_simple(prefix + ".conv", x, 3, 1, 1, 1, 1, 2 * x.T);
}

const float* Net::s4_kernel(const std::string& kp, int H, int L) {
    const std::string key = kp + "#" + std::to_string(L) + (ctx->s4_symmetric ? "s" : "");
    auto it = baked.find(key);
    if (it != baked.end()) return it->second;
    const Param& pl = P(kp + ".L");
    long long Lst = 0;
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    HIP_CHECK(hipMemcpy(&Lst, pl.ptr, sizeof(long long), hipMemcpyDeviceToHost));
    MUGD_CHECK(Lst > 0 && Lst >= L, -6,
               "S4 kernel " + kp + ": stored length L=" + std::to_string(Lst) + " < requested " + std::to_string(L) +
                   " (the host module must run the length-doubling C~ setup first)");
    const Param& pc = P(kp + ".C");
    const int N = (int)pc.shape[pc.shape.size() - 2];
    MUGD_CHECK(pc.numel() == (long long)H * N * 2, -2, "S4: unexpected C shape at " + kp);
    float* k = dev_alloc((size_t)H * L);
    S4GenArgs a{PF(kp + ".C"), PF(kp + ".B"), PF(kp + ".P"), PF(kp + ".inv_w_real"), PF(kp + ".w_imag"), PF(kp + ".log_dt"),
                H, N, (int)Lst, L, nullptr, k, ctx->s4_symmetric ? 1 : 0};
    launch_s4_kernel_gen(ctx->stream, a);
    return baked[key] = k;
}

// mug/diffusion/unet.py:76-91 + mug/model/s4.py:1471-1541
Tensor Net::s4_layer(const std::string& prefix, const Tensor& x) {
    const int H = x.C, L = x.T;
    Tensor out = talloc(H, L);
    const size_t mk = arena.mark();
    const float* aff = nullptr;
    Tensor u = x;
    const bool in_kernel_gn = ctx->fuse_norm && s4_conv_fuses_group_norm(L) && H % 32 == 0;
    if (!ctx->fuse_norm) u = group_norm(prefix + ".norm", {x}, 32, false);
    else if (!in_kernel_gn) aff = gn_stats(prefix + ".norm", {x}, 32);
    const float* k = s4_kernel(prefix + ".s4_model.kernel.kernel", H, L);
    Tensor y = talloc(H, L);
    S4ConvArgs ca{u.p, k, PF(prefix + ".s4_model.D"), y.p, Bn, H, L, aff,
                  in_kernel_gn ? PF(prefix + ".norm.weight") : nullptr, in_kernel_gn ? PF(prefix + ".norm.bias") : nullptr, 32, 1e-6f,
                  (in_kernel_gn && ctx->fuse_stats && x.bmod == 0) ? x.rowstat : nullptr, nullptr};
    if (ca.rowstat) {
        // the producer's GROUP sums instead (ConvArgs::gsink, like a conv consumer: gn_inputs) when it has a free sink: the kernel then loads
        // one finished pair per workgroup instead of summing its group's rows over the wave
        static const bool group_on = !(getenv("MUGD_GN_GROUP") && atoi(getenv("MUGD_GN_GROUP")) == 0);
        if (group_on && H / 32 > 0 && x.prod >= 0 && x.prod < (int)prods.size() && prods[x.prod].rows && prods[x.prod].nsink < 2) {
            double* table = alloc_rowstat((size_t)Bn * 64);
            prods[x.prod].nsink++;
            if (!dry && prods[x.prod].L) {
                ConvArgs& pa = prods[x.prod].L->a;
                const int k = pa.gsink[0].p ? 1 : 0;
                pa.gsink[k].p = table; pa.gsink[k].coff = 0; pa.gsink[k].cg = H / 32;
                ca.gn_table = table; ca.rowstat = nullptr;
                if (getenv("MUGD_GN_GROUP_LOG")) fprintf(stderr, "[mugd] group table: %s (S4 kernel, %d channels per group)\n", prefix.c_str(), H / 32);
            }
        }
        if (ca.rowstat) use_rowstat(x);
    }
    emit([ca](hipStream_t st) { launch_s4_conv(st, ca); }, OP_S4_CONV, 0, prefix + " H=" + std::to_string(H) + " L=" + std::to_string(L));
    ConvSpec gl;
    gl.key = prefix + ".s4_model.output_linear.0";
    gl.in.push_back(ConvIn{y});
    gl.w = {WBlock{gl.key + ".weight", 0, 0, 0}};
    gl.bias = {{gl.key + ".bias", 0}};
    gl.Mrows = 2 * H; gl.Mout = H; gl.Tout = L; gl.epi = EPI_GLU;
    Tensor g = conv(gl);
    out = conv_simple(prefix + ".out_layer", g, 3, 1, 1, 1, 0, L, x, out);
    arena.release(mk);
    return out;
}

// =======================================================================================
// U-Net
// =======================================================================================
std::vector<std::pair<std::string, int>> UNet::resblock_list() const {
    std::vector<std::pair<std::string, int>> r;
    const int mc = cfg.model_channels, nl = (int)cfg.channel_mult.size();
    int idx = 1;
    for (int level = 0; level < nl; ++level) {
        ++idx;                                   // AudioConcatBlock
        for (int i = 0; i < cfg.num_res_blocks; ++i) r.push_back({"input_blocks." + std::to_string(idx++) + ".0", cfg.channel_mult[level] * mc});
        if (level != nl - 1) ++idx;              // Downsample
    }
    r.push_back({"middle_block.0", cfg.channel_mult[nl - 1] * mc});
    r.push_back({"middle_block.2", cfg.channel_mult[nl - 1] * mc});
    idx = 0;
    for (int level = nl - 1; level >= 0; --level) {
        ++idx;
        for (int i = 0; i <= cfg.num_res_blocks; ++i) r.push_back({"output_blocks." + std::to_string(idx++) + ".0", cfg.channel_mult[level] * mc});
    }
    return r;
}

// All 22 `emb_layers` Linear(4mc -> Cout) are stacked into one GEMV so a step needs one launch.
void UNet::prepare_emb() {
    if (baked.count("emb.W")) return;
    auto list = resblock_list();
    int total = 0;
    for (auto& e : list) total += e.second;
    const int K = 4 * cfg.model_channels;
    float* W = dev_alloc((size_t)total * K);
    float* bias = dev_alloc(total);
    int off = 0;
    for (auto& e : list) {
        const Param& w = P(e.first + ".emb_layers.1.weight");
        MUGD_CHECK(w.numel() == (long long)e.second * K, -2, "emb_layers shape mismatch at " + e.first);
        HIP_CHECK(hipMemcpyAsync(W + (size_t)off * K, w.ptr, (size_t)e.second * K * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        HIP_CHECK(hipMemcpyAsync(bias + off, P(e.first + ".emb_layers.1.bias").ptr, (size_t)e.second * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        off += e.second;
    }
    baked["emb.W"] = W;
    baked["emb.b"] = bias;
    rowadd_total = total;
}

// mug/diffusion/unet.py:212-239.  Neither the channel concat [h | audio | skip] nor the two
// GroupNorm+SiLU outputs are materialised: both convs read the raw tensors as K-segments and
// normalise them while staging.
Tensor UNet::resblock(const std::string& prefix, const std::vector<Tensor>& segs, int Cout, int rowadd_off) {
    const int T = segs[0].T;
    Tensor out = talloc(Cout, T);
    const size_t mk = arena.mark();
    ConvSpec c1;
    c1.key = prefix + ".in_layers.2";
    gn_inputs(c1, prefix + ".in_layers.0", segs, 32, true, 3, 1, 1);
    span_weights(c1, c1.key + ".weight", 0, c1.in.size());
    c1.bias = {{c1.key + ".bias", 0}};
    c1.Mrows = c1.Mout = Cout; c1.Tout = T; c1.want_rowstat = true;
    c1.rowadd = emb_rowadd + rowadd_off; c1.rowadd_stride = rowadd_total;
    Tensor h1 = conv(c1);
    ConvSpec s;
    gn_inputs(s, prefix + ".out_layers.0", {h1}, 32, true, 3, 1, 1);
    span_weights(s, prefix + ".out_layers.3.weight", 0, s.in.size());
    s.bias = {{prefix + ".out_layers.3.bias", 0}};
    s.Mrows = s.Mout = Cout; s.Tout = T; s.out = out; s.want_rowstat = true;
    if (has(prefix + ".skip_connection.weight")) {
        s.key = prefix + ".out_layers.3+skip";
        const size_t first = s.in.size();
        for (auto& t : segs) s.in.push_back(ConvIn{t});
        span_weights(s, prefix + ".skip_connection.weight", first, segs.size());
        s.bias.push_back({prefix + ".skip_connection.bias", 0});
    } else {
        MUGD_CHECK(segs.size() == 1 && segs[0].C == Cout, -2, "identity skip needs a single input: " + prefix);
        s.key = prefix + ".out_layers.3";
        s.resid = segs[0];
    }
    out = conv(s);
    arena.release(mk);
    return out;
}

void UNet::build(bool dry_run) {
    dry = dry_run;
    arena.begin(dry_run);
    ops.clear();
    pre_ops.clear();
    emb_ops.clear();
    begin_rowstat();
    Bn = key.B;
    const int mc = cfg.model_channels, nl = (int)cfg.channel_mult.size(), z = key.z;
    const bool attn_any = !cfg.attention_resolutions.empty();
    auto is_attn = [&](int ds) { return std::find(cfg.attention_resolutions.begin(), cfg.attention_resolutions.end(), ds) != cfg.attention_resolutions.end(); };
    (void)attn_any;

    // ---- fixed buffers
    in_x = talloc(cfg.in_channels, z);
    in_ctx = talloc(cfg.context_dim, key.ntok);
    in_audio.clear();
    {
        const int save = Bn;
        if (key.bmod > 0) Bn = key.bmod;            // the audio maps are stored once and shared by both CFG halves
        for (int l = 0; l < nl; ++l) {
            Tensor a = talloc(cfg.audio_channels[l], z >> l);
            a.bmod = key.bmod;
            if (ctx->fuse_norm && ctx->fuse_stats) {          // row sums of the (step-invariant) audio maps: once per call
                const int rows = Bn * a.C;
                // library-owned, kept across recompilations for other (batch, length) keys: keyed like the K/V and S4 buffers
                const std::string rk = "audio.rowstat#" + std::to_string(l) + "x" + std::to_string(rows);
                if (!dry && !baked.count(rk)) baked[rk] = dev_alloc((size_t)rows * 4);
                a.rowstat = dry ? reinterpret_cast<double*>(256) : reinterpret_cast<double*>(baked[rk]);
                const float* xp = a.p; double* rp = a.rowstat; const int T = a.T;
                to_pre = true;
                emit([=](hipStream_t st) { launch_row_sums(st, xp, rp, rows, T); }, OP_SMALL, 0, "audio row sums");
                to_pre = false;
            }
            in_audio.push_back(a);
        }
        Bn = save;
    }
    out_eps = talloc(cfg.out_channels, z);
    t_dev = reinterpret_cast<long long*>(arena.alloc(2 * (size_t)Bn + 2));
    Tensor temb = talloc(mc, 1), e1 = talloc(4 * mc, 1), emb = talloc(4 * mc, 1);
    prepare_emb();
    emb_rowadd = arena.alloc((size_t)Bn * rowadd_total);

    // ---- time embedding (unet.py:522-523) + all resblock emb_layers in one GEMV.  forward() runs these per call (t may
    // differ per batch row); sample() precomputes the rows of every timestep of the schedule instead (see sample()).
    {
        to_emb = true;
        long long* tp = t_dev; float* o = temb.p; int B = Bn, dim = mc;
        emit([=](hipStream_t st) { launch_timestep_embedding(st, tp, nullptr, o, B, dim); });
        LinSmallArgs l1{temb.p, PF("time_embed.0.weight"), PF("time_embed.0.bias"), e1.p, Bn, mc, 4 * mc, 0, 1, mc, 4 * mc};
        emit([=](hipStream_t st) { launch_linear_small(st, l1); });
        LinSmallArgs l2{e1.p, PF("time_embed.2.weight"), PF("time_embed.2.bias"), emb.p, Bn, 4 * mc, 4 * mc, 0, 0, 4 * mc, 4 * mc};
        emit([=](hipStream_t st) { launch_linear_small(st, l2); });
        LinSmallArgs l3{emb.p, baked["emb.W"], baked["emb.b"], emb_rowadd, Bn, 4 * mc, rowadd_total, 1, 0, 4 * mc, rowadd_total};
        emit([=](hipStream_t st) { launch_linear_small(st, l3); });
        to_emb = false;
    }

    // ---- down path (unet.py:341-405, 527-535)
    int ra = 0;                                    // running offset into the stacked emb_layers output
    std::vector<Tensor> hs;
    Tensor h = conv_simple("input_blocks.0.0", in_x, 3, 1, 1, 1, 0, z);
    hs.push_back(h);
    int idx = 1, ds = 1;
    for (int level = 0; level < nl; ++level) {
        ++idx;                                     // AudioConcatBlock: concat is a second K-segment, never materialised
        std::vector<Tensor> segs = {h, in_audio[level]};
        const int Cout = cfg.channel_mult[level] * mc;
        for (int i = 0; i < cfg.num_res_blocks; ++i) {
            const std::string p = "input_blocks." + std::to_string(idx);
            h = resblock(p + ".0", segs, Cout, ra);
            ra += Cout;
            int j = 1;
            if (is_attn(ds)) h = transformer(p + "." + std::to_string(j++), h, &in_ctx, cfg.num_heads);
            if (cfg.s4) h = s4_layer(p + "." + std::to_string(j++), h);
            hs.push_back(h);
            segs = {h};
            ++idx;
        }
        if (level != nl - 1) {
            h = downsample("input_blocks." + std::to_string(idx) + ".0", h);
            hs.push_back(h);
            ++idx;
            ds *= 2;
        }
    }
    // ---- middle (unet.py:412-437)
    {
        const int C = cfg.channel_mult[nl - 1] * mc;
        h = resblock("middle_block.0", {h}, C, ra); ra += C;
        h = transformer("middle_block.1", h, &in_ctx, cfg.num_heads);
        h = resblock("middle_block.2", {h}, C, ra); ra += C;
    }
    // ---- up path (unet.py:440-487, 539-546)
    idx = 0;
    for (int level = nl - 1; level >= 0; --level) {
        ++idx;
        std::vector<Tensor> pend = {h, in_audio[level]};
        const int Cout = cfg.channel_mult[level] * mc;
        for (int i = 0; i <= cfg.num_res_blocks; ++i) {
            const std::string p = "output_blocks." + std::to_string(idx);
            std::vector<Tensor> segs = pend;
            segs.push_back(hs.back());
            hs.pop_back();
            h = resblock(p + ".0", segs, Cout, ra);
            ra += Cout;
            int j = 1;
            if (is_attn(ds)) h = transformer(p + "." + std::to_string(j++), h, &in_ctx, cfg.num_heads);
            if (cfg.s4 && i != cfg.num_res_blocks) h = s4_layer(p + "." + std::to_string(j++), h);
            if (level && i == cfg.num_res_blocks) { h = upsample(p + "." + std::to_string(j++), h); ds /= 2; }
            pend = {h};
            ++idx;
        }
    }
    MUGD_CHECK(ra == rowadd_total, -2, "internal: emb_layers bookkeeping");
    {
        ConvSpec s;
        s.key = "out.2";
        gn_inputs(s, "out.0", {h}, 32, true, 3, 1, 1);
        s.w = {WBlock{"out.2.weight", 0, 0, 0}};
        s.bias = {{"out.2.bias", 0}};
        s.Mrows = s.Mout = cfg.out_channels; s.Tout = z; s.out = out_eps;
        conv(s);
    }
}

void UNet::invalidate() {
    drop_programs();
    Net::invalidate();
    x_state = pred_dev = first_dev = noise_dev = sched_dev = nullptr;
    ttab_dev = nullptr; step_dev = nullptr; ticket_dev = nullptr; emb_table = emb_tmp = nullptr;
    sched_cap = 0; noise_cap = state_cap = 0;
}

void UNet::drop_programs() {
    if (graph) { hipGraphExecDestroy(graph); graph = nullptr; }
    built = false;
    ops.clear();
    pre_ops.clear();
    emb_ops.clear();
}

void UNet::ensure(int B, int z, int ntok, int bmod) {
    Key k{B, z, ntok, bmod};
    if (built && k == key) return;
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    drop_programs();
    MUGD_CHECK(z % (1 << ((int)cfg.channel_mult.size() - 1)) == 0, -2, "latent length must be divisible by 2^(levels-1)");
    key = k;
    build(true);
    arena.reserve(arena.peak());
    build(false);
    finish_stats();
    built = true;
}

static int audio_bmod(int audio_batch, int Bnet, int B) {
    MUGD_CHECK(audio_batch >= 1 && audio_batch <= B && B % audio_batch == 0, -2, "audio_batch must divide the batch size");
    return audio_batch == Bnet ? 0 : audio_batch;
}

void UNet::forward(const float* x, const long long* t, const float* context, int n_tok,
                   const float* const* audio, int audio_batch, float* eps, int B, int z) {
    ensure(B, z, n_tok, audio_bmod(audio_batch, B, B));
    hipStream_t st = ctx->stream;
    const int nl = (int)cfg.channel_mult.size();
    HIP_CHECK(hipMemcpyAsync(in_x.p, x, (size_t)B * in_x.C * z * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIP_CHECK(hipMemcpyAsync(t_dev, t, (size_t)B * sizeof(long long), hipMemcpyDeviceToDevice, st));
    HIP_CHECK(hipMemcpyAsync(in_ctx.p, context, (size_t)B * in_ctx.C * n_tok * sizeof(float), hipMemcpyDeviceToDevice, st));
    for (int l = 0; l < nl; ++l)
        HIP_CHECK(hipMemcpyAsync(in_audio[l].p, audio[l], (size_t)audio_batch * in_audio[l].C * in_audio[l].T * sizeof(float), hipMemcpyDeviceToDevice, st));
    run_pre_ops(st);
    for (auto& o : emb_ops) o.fn(st);
    run_ops(st);
    HIP_CHECK(hipMemcpyAsync(eps, out_eps.p, (size_t)B * out_eps.C * z * sizeof(float), hipMemcpyDeviceToDevice, st));
}

DdimStepArgs UNet::step_args(bool cfg_on, float scale, bool with_noise, bool with_pred, bool with_first, int nstate, int S, int mode) const {
    DdimStepArgs d{};
    d.x = x_state; d.eps = out_eps.p; d.noise = with_noise ? noise_dev : nullptr; d.pred_x0 = with_pred ? pred_dev : nullptr; d.first = with_first ? first_dev : nullptr;
    d.sched = sched_dev; d.step_idx = step_dev; d.ticket = ticket_dev; d.in_x = in_x.p;
    d.emb_table = emb_table; d.emb_rows = emb_rowadd;
    d.n = nstate; d.cfg = cfg_on ? 1 : 0; d.Bnet = key.B; d.emb_total = rowadd_total; d.mode = mode; d.scale = scale;
    d.zero_p = rs_zero_op >= 0 ? rs_base : nullptr; d.zero_n = rs_zero_op >= 0 ? (long long)rs_zero_n : 0;
    return d;
}

// one DDIM step = the U-Net program + ONE more launch (CFG combine, DDIM update, next input, next time-embedding rows, counter)
void UNet::step_body(hipStream_t st, bool cfg_on, float scale, bool with_noise, bool with_pred, bool with_first, int nstate, int S) {
    // the program without its "zero the row-sum accumulators" fill: the per-step kernel of the PREVIOUS step (mode 0 before the first) has
    // zeroed them (DdimStepArgs::zero_p) -- one launch less per step
    for (size_t i = 0; i < ops.size(); ++i)
        if ((int)i != rs_zero_op) ops[i].fn(st);
    launch_ddim_step(st, step_args(cfg_on, scale, with_noise, with_pred, with_first, nstate, S, 1));
}

void UNet::sample(float* x, const float* c, const float* uc, int n_tok, const float* const* audio, int audio_batch,
                  int B, int z, int S, const long long* t_host, const float* sched_host, float scale,
                  const float* noise, float* pred_x0, float* first) {
    const bool cfg_on = (uc != nullptr) && (scale != 1.0f);       // ddim.py:167
    const int Bnet = cfg_on ? 2 * B : B;
    ensure(Bnet, z, n_tok, audio_bmod(audio_batch, Bnet, B));
    hipStream_t st = ctx->stream;
    const int nl = (int)cfg.channel_mult.size();
    const int nstate = B * cfg.in_channels * z;
    const size_t ctx_n = (size_t)B * in_ctx.C * n_tok;

    static_assert(sizeof(long long) == 8, "");
    auto drop_graph = [&] {
        HIP_CHECK(hipStreamSynchronize(st));
        if (graph) { hipGraphExecDestroy(graph); graph = nullptr; }
    };
    if (sched_cap < S) {
        drop_graph();
        sched_cap = std::max(S, 64);
        sched_dev = dev_alloc((size_t)sched_cap * 4);
        ttab_dev = reinterpret_cast<long long*>(dev_alloc((size_t)sched_cap * 2));
        step_dev = reinterpret_cast<int*>(dev_alloc(4));                  // {step, S}
        ticket_dev = reinterpret_cast<int*>(dev_alloc(4, true));
        emb_table = dev_alloc((size_t)sched_cap * rowadd_total);
        emb_tmp = dev_alloc((size_t)sched_cap * 9 * cfg.model_channels);
    }
    if ((size_t)nstate > state_cap) {                    // sampler state sized for this (B, z)
        drop_graph();
        x_state = dev_alloc(nstate);
        pred_dev = dev_alloc(nstate);
        first_dev = dev_alloc(2 * (size_t)nstate);
        state_cap = nstate;
    }
    if (noise) {
        if ((size_t)S * nstate > noise_cap) {
            drop_graph();
            noise_cap = (size_t)S * nstate;
            noise_dev = dev_alloc(noise_cap);
        }
        HIP_CHECK(hipMemcpyAsync(noise_dev, noise, (size_t)S * nstate * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    HIP_CHECK(hipMemcpyAsync(sched_dev, sched_host, (size_t)S * 4 * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemcpyAsync(ttab_dev, t_host, (size_t)S * sizeof(long long), hipMemcpyHostToDevice, st));
    step_init[0] = 0; step_init[1] = S;                   // member: outlives the asynchronous copy
    HIP_CHECK(hipMemcpyAsync(step_dev, step_init, sizeof(step_init), hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemcpyAsync(x_state, x, (size_t)nstate * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (cfg_on) {                                         // batch order [uncond ; cond]  (ddim.py:173)
        HIP_CHECK(hipMemcpyAsync(in_ctx.p, uc, ctx_n * sizeof(float), hipMemcpyDeviceToDevice, st));
        HIP_CHECK(hipMemcpyAsync(in_ctx.p + ctx_n, c, ctx_n * sizeof(float), hipMemcpyDeviceToDevice, st));
    } else {
        HIP_CHECK(hipMemcpyAsync(in_ctx.p, c, ctx_n * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    for (int l = 0; l < nl; ++l)
        HIP_CHECK(hipMemcpyAsync(in_audio[l].p, audio[l], (size_t)audio_batch * in_audio[l].C * in_audio[l].T * sizeof(float), hipMemcpyDeviceToDevice, st));

    run_pre_ops(st);                                      // context-only work (cross-attention K/V): once per call, not per step
    {   // time-embedding rows of all S timesteps (unet.py:522-523 + every ResBlock's emb_layers), once per call
        const int mc = cfg.model_channels;
        float* temb = emb_tmp; float* e1 = emb_tmp + (size_t)S * mc; float* emb = e1 + (size_t)S * 4 * mc;
        launch_timestep_embedding(st, ttab_dev, nullptr, temb, S, mc);
        launch_linear_small(st, LinSmallArgs{temb, PF("time_embed.0.weight"), PF("time_embed.0.bias"), e1, S, mc, 4 * mc, 0, 1, mc, 4 * mc});
        launch_linear_small(st, LinSmallArgs{e1, PF("time_embed.2.weight"), PF("time_embed.2.bias"), emb, S, 4 * mc, 4 * mc, 0, 0, 4 * mc, 4 * mc});
        launch_linear_small(st, LinSmallArgs{emb, baked["emb.W"], baked["emb.b"], emb_table, S, 4 * mc, rowadd_total, 1, 0, 4 * mc, rowadd_total});
    }
    const bool wn = noise != nullptr, wp = pred_x0 != nullptr, wf = first != nullptr;
    launch_ddim_step(st, step_args(cfg_on, scale, wn, wp, wf, nstate, S, 0));      // x_T -> U-Net input, embedding rows of step 0
    if (ctx->use_graph) {
        const int per_graph = ctx->use_graph == 2 ? std::max(S, 1) : 1;        // steps captured into one graph
        if (graph && (graph_steps != per_graph || graph_cfg != cfg_on || graph_noise != wn || graph_pred != wp || graph_first != wf || graph_scale != scale)) {
            HIP_CHECK(hipStreamSynchronize(st));
            hipGraphExecDestroy(graph);
            graph = nullptr;
        }
        if (!graph) {
            hipGraph_t g = nullptr;
            HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < per_graph; ++i) step_body(st, cfg_on, scale, wn, wp, wf, nstate, S);
            HIP_CHECK(hipStreamEndCapture(st, &g));
            HIP_CHECK(hipGraphInstantiate(&graph, g, nullptr, nullptr, 0));
            HIP_CHECK(hipGraphDestroy(g));
            graph_steps = per_graph; graph_cfg = cfg_on; graph_noise = wn; graph_pred = wp; graph_first = wf; graph_scale = scale;
        }
        for (int i = 0; i < S / per_graph; ++i) HIP_CHECK(hipGraphLaunch(graph, st));
    } else {
        for (int i = 0; i < S; ++i) step_body(st, cfg_on, scale, wn, wp, wf, nstate, S);
    }
    HIP_CHECK(hipMemcpyAsync(x, x_state, (size_t)nstate * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (wp) HIP_CHECK(hipMemcpyAsync(pred_x0, pred_dev, (size_t)nstate * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (wf) HIP_CHECK(hipMemcpyAsync(first, first_dev, 2 * (size_t)nstate * sizeof(float), hipMemcpyDeviceToDevice, st));
}

// =======================================================================================
// VAE decoder  (mug/firststage/autoencoder.py:268-354)
// =======================================================================================
void VaeDecoder::build(bool dry_run) {
    dry = dry_run;
    arena.begin(dry_run);
    ops.clear();
    pre_ops.clear();
    emb_ops.clear();
    begin_rowstat();
    Bn = kB;
    const int nres = (int)cfg.channel_mult.size(), g = cfg.num_groups;
    in_z = talloc(cfg.z_channels, kz);
    out_x = talloc(cfg.x_channels, kz << (nres - 1));
    MUGD_CHECK(cfg.scale == 1.0f, -2, "VAE scale != 1 is not supported");    // active config has no `scale` key
    int block_in = cfg.middle_channels * cfg.channel_mult[nres - 1];
    Tensor h = conv_simple("decoder.conv_in", in_z, 3, 1, 1, 1, 0, kz);
    h = resnet_block("decoder.mid.block_1", h, block_in, g, 1, 1);
    h = resnet_block("decoder.mid.block_2", h, block_in, g, 1, 1);
    for (int lvl = nres - 1; lvl >= 0; --lvl) {
        const int block_out = cfg.middle_channels * cfg.channel_mult[lvl];
        for (int ib = 0; ib <= cfg.num_res_blocks; ++ib)
            h = resnet_block("decoder.up." + std::to_string(lvl) + ".block." + std::to_string(ib), h, block_out, g, 1, 1);
        if (lvl != 0) h = upsample("decoder.up." + std::to_string(lvl) + ".upsample", h);
    }
    {
        ConvSpec s;
        s.key = "decoder.conv_out";
        gn_inputs(s, "decoder.norm_out", {h}, g, true, 3, 1, 1);
        s.w = {WBlock{"decoder.conv_out.weight", 0, 0, 0}};
        s.bias = {{"decoder.conv_out.bias", 0}};
        s.Mrows = s.Mout = cfg.x_channels; s.Tout = h.T; s.out = out_x;
        conv(s);
    }
}

void VaeDecoder::decode(const float* z_lat, float* logits, int B, int z) {
    if (!built || B != kB || z != kz) {
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        kB = B; kz = z;
        build(true);
        arena.reserve(arena.peak());
        build(false);
        finish_stats();
        built = true;
    }
    hipStream_t st = ctx->stream;
    HIP_CHECK(hipMemcpyAsync(in_z.p, z_lat, (size_t)B * in_z.C * z * sizeof(float), hipMemcpyDeviceToDevice, st));
    run_ops(st);
    HIP_CHECK(hipMemcpyAsync(logits, out_x.p, (size_t)B * out_x.C * out_x.T * sizeof(float), hipMemcpyDeviceToDevice, st));
}

// =======================================================================================
// VAE encoder  (mug/firststage/autoencoder.py:196-265) -- inpainting / partial regeneration
// =======================================================================================
void VaeEncoder::build(bool dry_run) {
    dry = dry_run;
    arena.begin(dry_run);
    ops.clear();
    pre_ops.clear();
    emb_ops.clear();
    begin_rowstat();
    Bn = kB;
    const int nres = (int)cfg.channel_mult.size(), g = cfg.num_groups;
    MUGD_CHECK(kT % (1 << (nres - 1)) == 0, -2, "VAE encoder: length must be divisible by 2^(levels-1)");
    in_x = talloc(cfg.x_channels, kT);
    out_m = talloc(2 * cfg.z_channels, kT >> (nres - 1));
    Tensor h = conv_simple("encoder.conv_in", in_x, 3, 1, 1, 1, 0, kT);
    for (int lvl = 0; lvl < nres; ++lvl) {
        const int block_out = cfg.middle_channels * cfg.channel_mult[lvl];
        for (int ib = 0; ib < cfg.num_res_blocks; ++ib)
            h = resnet_block("encoder.down." + std::to_string(lvl) + ".block." + std::to_string(ib), h, block_out, g, 1, 1);
        if (lvl != nres - 1) h = downsample("encoder.down." + std::to_string(lvl) + ".downsample", h);
    }
    h = resnet_block("encoder.mid.block_1", h, h.C, g, 1, 1);
    h = resnet_block("encoder.mid.block_2", h, h.C, g, 1, 1);
    ConvSpec s;
    s.key = "encoder.conv_out";
    gn_inputs(s, "encoder.norm_out", {h}, g, true, 3, 1, 1);
    s.w = {WBlock{"encoder.conv_out.weight", 0, 0, 0}};
    s.bias = {{"encoder.conv_out.bias", 0}};
    s.Mrows = s.Mout = 2 * cfg.z_channels; s.Tout = h.T; s.out = out_m;
    conv(s);
}

void VaeEncoder::encode(const float* x, float* moments, int B, int T) {
    if (!built || B != kB || T != kT) {
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        kB = B; kT = T;
        build(true);
        arena.reserve(arena.peak());
        build(false);
        finish_stats();
        built = true;
    }
    hipStream_t st = ctx->stream;
    HIP_CHECK(hipMemcpyAsync(in_x.p, x, (size_t)B * in_x.C * T * sizeof(float), hipMemcpyDeviceToDevice, st));
    run_ops(st);
    HIP_CHECK(hipMemcpyAsync(moments, out_m.p, (size_t)B * out_m.C * out_m.T * sizeof(float), hipMemcpyDeviceToDevice, st));
}

// =======================================================================================
// wave encoder  (mug/cond/wave.py:398-464)
// =======================================================================================
void WaveEncoder::build(bool dry_run) {
    dry = dry_run;
    arena.begin(dry_run);
    ops.clear();
    pre_ops.clear();
    emb_ops.clear();
    begin_rowstat();
    Bn = kB;
    const int nres = (int)cfg.channel_mult.size(), g = cfg.num_groups, mid = cfg.middle_channels;
    in_mel = talloc(cfg.n_freq, kT);
    level_out.clear();
    Tensor h = conv_simple("conv_in", in_mel, 3, 1, 1, 1, 0, kT);
    int ds = 1;
    for (int lvl = 0; lvl < nres; ++lvl) {
        const std::string q = "down." + std::to_string(lvl);
        if (lvl != 0) { h = downsample(q + ".downsample", h); ds *= 2; }
        const int block_out = mid * cfg.channel_mult[lvl];
        const bool at = std::find(cfg.attention_resolutions.begin(), cfg.attention_resolutions.end(), ds) != cfg.attention_resolutions.end();
        for (int ib = 0; ib < cfg.num_res_blocks; ++ib) {
            const bool even = (ib % 2 == 0);
            h = resnet_block(q + ".block." + std::to_string(ib), h, block_out, g, even ? 1 : 4, even ? 2 : 8);
            if (at) h = transformer(q + ".attn." + std::to_string(ib), h, nullptr, cfg.num_heads);
        }
        level_out.push_back(h);
    }
}

void WaveEncoder::encode(const float* mel, float* const* outs, int B, int Ta) {
    if (!built || B != kB || Ta != kT) {
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        kB = B; kT = Ta;
        build(true);
        arena.reserve(arena.peak());
        build(false);
        finish_stats();
        built = true;
    }
    hipStream_t st = ctx->stream;
    HIP_CHECK(hipMemcpyAsync(in_mel.p, mel, (size_t)B * in_mel.C * Ta * sizeof(float), hipMemcpyDeviceToDevice, st));
    run_ops(st);
    for (size_t i = 0; i < level_out.size(); ++i)
        if (outs[i]) HIP_CHECK(hipMemcpyAsync(outs[i], level_out[i].p, (size_t)B * level_out[i].C * level_out[i].T * sizeof(float), hipMemcpyDeviceToDevice, st));
}
