// block_seq.cpp — one supernet transformer block (forward / backward) enqueued natively: the host
// side of the hot path is ONE call per block and direction instead of ~30 framework / ctypes calls
// (the step was launch-bound: ~12.7 ms of host time against ~13 ms of GPU time per step).
//
// Reference semantics: TransformerEncoderLayer.forward, AutoFormer/model/supernet_transformer.py:
// 251-287 (pre-norm block, drop-path on both branches) with AttentionSuper.forward
// (model/module/multihead_super.py:133-160) and the weight-entangled Linear/LayerNorm supers; the
// backward is what autograd derives for it.  Every kernel is the one the per-op C ABI exposes
// (csrc/block_ops.hip, attn_rpe2d.hip, gemm_mfma.hip); this file only sequences them over two HIP
// streams: the weight-gradient GEMMs and the gradient finalisation run on `side_stream` behind
// events, overlapping the HBM-bound passes of the main chain.
//
// Memory: the caller owns everything.  cream_block_fwd_workspace / cream_block_bwd_workspace give
// the byte size of one flat workspace per call and the offsets of the tensors the caller needs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "launch_ev.hpp"

#include <atomic>
#include <mutex>
#include <vector>

#include "cream_amd.h"

namespace {

constexpr int64_t ALIGN = 256;
struct Bump {
    int64_t off = 0;
    int64_t take(int64_t bytes) {
        const int64_t o = off;
        off += (bytes + ALIGN - 1) / ALIGN * ALIGN;
        return o;
    }
};

struct Dims {
    int64_t B, N, E, H, F, Q, M, NP, slabs, P;
    int64_t S2, S1, Sp, Sq;                                       // split factors of the four weight gradients
};
Dims dims_of(const cream_block_desc* d) {
    Dims x;
    x.B = d->B; x.N = d->N; x.E = d->E; x.H = d->H; x.F = d->F;
    x.Q = 64 * x.H; x.M = x.B * x.N; x.NP = cream_attn_rpe2d_padded_len(d->N);
    x.slabs = cream_colsum128_slabs((int)x.M); x.P = cream_ln_partials();
    x.S2 = cream_linear_wgrad_splits_bf16((int)x.M, (int)x.E, (int)x.F);      // (bf16 partial tiles are the default, cream_block_wgrad_bf16)
    x.S1 = cream_linear_wgrad_splits_bf16((int)x.M, (int)x.F, (int)x.E);
    x.Sp = cream_linear_wgrad_splits_bf16((int)x.M, (int)x.E, (int)x.Q);
    x.Sq = cream_linear_wgrad_splits_bf16((int)x.M, 3 * (int)x.Q, (int)x.E);
    return x;
}
bool desc_ok(const cream_block_desc* d) {
    return d && d->B > 0 && d->N > 0 && d->E > 0 && d->H > 0 && d->F > 0 && d->E % 8 == 0 && d->F % 8 == 0 && d->wqkv &&
           d->wqkv_t && d->bqkv && d->wproj && d->wproj_t && d->bproj && d->w1 && d->w1_t && d->b1 && d->w2 && d->w2_t && d->b2 &&
           d->ln1_g && d->ln1_b && d->ln2_g && d->ln2_b && d->tkv && d->tkh && d->tvv && d->tvh;
}

struct FwdLayout {
    int64_t xsum, mean1, rstd1, mean2, rstd2, a, qkv, o, lse, sp, p, x1, c, h, g, f, total;
};
FwdLayout fwd_layout(const Dims& x) {
    Bump b;
    FwdLayout L;
    L.xsum = b.take(x.M * x.E * 4);
    L.mean1 = b.take(x.M * 4); L.rstd1 = b.take(x.M * 4); L.mean2 = b.take(x.M * 4); L.rstd2 = b.take(x.M * 4);
    L.a = b.take(x.M * x.E * 2);
    L.qkv = b.take(x.M * 3 * x.Q * 2);
    L.o = b.take(x.M * x.Q * 2);
    L.lse = b.take(x.B * x.H * x.N * 4);
    L.sp = b.take(x.B * x.H * 64 * x.NP * 2);
    L.p = b.take(x.M * x.E * 2);
    L.x1 = b.take(x.M * x.E * 4);
    L.c = b.take(x.M * x.E * 2);
    L.h = b.take(x.M * x.F * 2);
    L.g = b.take(x.M * x.F * 2);
    L.f = b.take(x.M * x.E * 2);
    L.total = b.off;
    return L;
}

struct BwdLayout {
    int64_t dh, pb1, dc, dx1, dp, pl2, dout, dqkv, dlt, qe, de, delta, dtab, pbq, da, dx, df_prev, pl1, pw2, pw1, pwp, pwq, total;
};
BwdLayout bwd_layout(const Dims& x) {
    Bump b;
    BwdLayout L;
    L.dh = b.take(x.M * x.F * 2);
    L.pb1 = b.take(x.slabs * x.F * 4);
    L.dc = b.take(x.M * x.E * 2);
    L.dx1 = b.take(x.M * x.E * 4);
    L.dp = b.take(x.M * x.E * 2);
    L.pl2 = b.take(x.P * 3 * x.E * 4);
    L.dout = b.take(x.M * x.Q * 2);
    L.dqkv = b.take(x.M * 3 * x.Q * 2);
    L.dlt = b.take(x.B * x.H * 64 * x.NP * 2);
    L.qe = b.take(x.B * x.H * x.NP * 32 * 2);
    L.de = b.take(x.B * x.H * x.NP * 32 * 2);
    L.delta = b.take(x.B * x.H * x.NP * 4);
    L.dtab = b.take(x.B * x.H * 4 * 32 * 64 * 4);
    L.pbq = b.take(x.Sq * 3 * x.Q * 4);
    L.da = b.take(x.M * x.E * 2);
    L.dx = b.take(x.M * x.E * 4);
    L.df_prev = b.take(x.M * x.E * 2);
    L.pl1 = b.take(x.P * 3 * x.E * 4);
    L.pw2 = b.take(x.S2 * x.E * x.F * 4);
    L.pw1 = b.take(x.S1 * x.F * x.E * 4);
    L.pwp = b.take(x.Sp * x.E * x.Q * 4);
    L.pwq = b.take(x.Sq * 3 * x.Q * x.E * 4);
    L.total = b.off;
    return L;
}

// events that order the side stream behind the main stream (re-recorded freely: a wait refers to
// the most recent record at the time it is enqueued).  One pool PER DEVICE: an event belongs to the
// device that was current when it was created, and a process may drive more than one GPU.
constexpr int MAX_DEV = 16, EV_PER_DEV = 8;
std::mutex g_ev_mu;
hipEvent_t g_ev[MAX_DEV][EV_PER_DEV];
int g_ev_n[MAX_DEV] = {}, g_ev_next[MAX_DEV] = {};
}  // namespace
namespace cream { thread_local hipEvent_t tl_stop_event = nullptr; thread_local hipEvent_t tl_start_event = nullptr; }
namespace {

// The events that order the side stream behind dgrad_mul / the LayerNorm backwards / the attention backward ride on those
// kernels' own dispatch packets (launch_ev.hpp) instead of an event record (a marker packet) behind each:
// 9.456 -> 9.407 ms per step (profiles/r04_step_gaps.md).
int fork_on_kernel_mode() { return 1; }

hipEvent_t next_fork_event(int dev);
bool prof_active();

// what the side stream of this thread is already ordered behind: the df_prev output of the last cream_block_bwd call
struct OrderedBehind { const void* df; hipStream_t main, side; };
thread_local OrderedBehind tl_ordered = {nullptr, nullptr, nullptr};

// arm(): hand the next kernel launched through CREAM_LAUNCH on this thread a completion event; join(): make `side` wait
// for it — by a plain record behind the kernel if the launch did not take the event (another code path, or mode 0)
struct KernelFork {
    hipEvent_t ev = nullptr;
    hipStream_t main, side;
    bool armed_ = false;
    KernelFork(hipStream_t m, hipStream_t s) : main(m), side(s) {}
    ~KernelFork() { if (ev && cream::tl_stop_event == ev) cream::tl_stop_event = nullptr; }      // (error return between arm and join)
    bool arm() {
        if (main == side) return true;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0) return false;
        ev = next_fork_event(dev);
        if (!ev) return false;
        armed_ = fork_on_kernel_mode() && !prof_active();                     // (the in-step timing owns the packet's events)
        if (armed_) cream::tl_stop_event = ev;
        return true;
    }
    bool join() {
        if (main == side) return true;
        if (cream::tl_stop_event == ev || !armed_) {                          // not taken: record behind the kernel
            cream::tl_stop_event = nullptr;
            if (hipEventRecord(ev, main) != hipSuccess) return false;
        }
        return hipStreamWaitEvent(side, ev, 0) == hipSuccess;
    }
};

hipEvent_t next_fork_event(int dev) {
    if (dev < 0 || dev >= MAX_DEV) return nullptr;
    std::lock_guard<std::mutex> lock(g_ev_mu);
    if (g_ev_n[dev] < EV_PER_DEV) {
        // Events WITHOUT the system-scope fence: both streams run on this device, the kernels' own agent-scope release /
        // acquire orders their data; the default system-scope release made every record on the main chain cost 4.1 us of
        // step time, 2.7 us without it (CREAM_EXTRA_RECORDS experiment: 64 records per step; same-box A/B x2: 9.99 ->
        // 9.85 ms per step, profiles/r04_step_gaps.md; hipEventReleaseToDevice: no different from plain events).  Ordering by
        // stream memory operations (hipStreamWriteValue32 / WaitValue32 on signal memory) was measured too: 11.29 against 9.35 ms.
        const unsigned flags = hipEventDisableTiming | hipEventDisableSystemFence;
        if (hipEventCreateWithFlags(&g_ev[dev][g_ev_n[dev]], flags) != hipSuccess) return nullptr;
        ++g_ev_n[dev];
    }
    hipEvent_t ev = g_ev[dev][g_ev_next[dev]];
    g_ev_next[dev] = (g_ev_next[dev] + 1) % g_ev_n[dev];
    return ev;
}

bool fork(hipStream_t main, hipStream_t side) {
    if (main == side) return true;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return false;
    hipEvent_t ev = next_fork_event(dev);
    if (!ev) return false;
    return hipEventRecord(ev, main) == hipSuccess && hipStreamWaitEvent(side, ev, 0) == hipSuccess;
}

#define TRY(call) do { const int rc_ = (call); if (rc_ != CREAM_OK) return rc_; } while (0)

// ---- optional in-step kernel timing -------------------------------------------------------------
// HIP events around every launch of cream_block_fwd / cream_block_bwd, each pair recorded on the
// stream the kernel is launched on (main or side) — so the durations are those of the REAL step: both
// streams live, the weight-gradient GEMMs contending with the main chain.  bench.py reads them for the
// roofline entry.  Off by default (one relaxed load per launch).
enum ProfKind { K_LN_FWD = 0, K_GEMM_NT, K_GEMM_NT_GELU, K_GEMM_NT_MUL, K_GEMM_TN, K_ATTN_FWD, K_ATTN_BWD, K_LN_BWD,
                K_GRAD_FINALIZE, K_COUNT };
const char* const kProfNames[K_COUNT] = {"ln_fwd", "gemm_nt", "gemm_nt_gelu", "gemm_nt_mul", "gemm_tn_wgrad", "attn_rpe2d_fwd",
                                         "attn_rpe2d_bwd", "ln_bwd", "grad_finalize"};
// split-K partial tiles of the weight gradients as bf16 (cream_linear_wgrad_parts_bf16): half the partial traffic
std::atomic<int> g_wgrad_bf16{-1};
int wgrad_bf16_mode() {
    int m = g_wgrad_bf16.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = getenv("CREAM_WGRAD_BF16");
        m = e ? (atoi(e) != 0) : 1;              // default ON: -2 % step time in a same-box A/B x3 (profiles/r04_step_ab.md)
        g_wgrad_bf16.store(m, std::memory_order_relaxed);
    }
    return m;
}

// the split-K weight-gradient launch in either partial format (the workspace regions are sized for fp32)
int wgrad_parts(int bf16, float* parts, float* bias_parts, const void* dy, const void* x, int M, int N, int K, int S, void* stream) {
    return bf16 ? cream_linear_wgrad_parts_bf16(parts, bias_parts, dy, x, M, N, K, S, stream)
                : cream_linear_wgrad_parts(parts, bias_parts, dy, x, M, N, K, S, stream);
}
struct ProfRec { int kind; hipEvent_t a, b; double flops, bytes; };
std::atomic<bool> g_prof_on{false};
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_prof_free;
hipEvent_t prof_event() {
    if (!g_prof_free.empty()) { hipEvent_t e = g_prof_free.back(); g_prof_free.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
bool prof_active() { return g_prof_on.load(std::memory_order_relaxed); }
// The pair of events of a record is CARRIED BY THE KERNEL'S OWN DISPATCH PACKET (launch_ev.hpp: CREAM_LAUNCH hands them to
// hipExtLaunchKernelGGL as start / stop events): the elapsed time is the kernel's execution, as rocprofv3 reports it, and no
// marker packets enter the streams (round 3 recorded an event in front of and behind every launch: ~1,400 markers per step,
// 4 us each on the main chain — the timed pass ran 25 % slower than the step it measured).  Multi-kernel operators are
// timed on their last kernel (attention backward: the one-pass kernel, not the 6 us table-image launch in front of it);
// an operator whose launch does not go through CREAM_LAUNCH is not sampled.
struct ProfScope {
    ProfRec r{};
    hipStream_t st;
    bool on;
    ProfScope(int kind, hipStream_t s, double flops, double bytes) : st(s), on(g_prof_on.load(std::memory_order_relaxed)) {
        if (!on) return;
        std::lock_guard<std::mutex> lock(g_prof_mu);
        r.kind = kind; r.flops = flops; r.bytes = bytes;
        r.a = prof_event(); r.b = prof_event();
        on = r.a && r.b;
        if (!on) { give_back(); return; }
        cream::tl_start_event = r.a;
        cream::tl_stop_event = r.b;
    }
    void give_back() {                       // (caller holds g_prof_mu) events of a failed scope return to the pool
        if (r.a) g_prof_free.push_back(r.a);
        if (r.b) g_prof_free.push_back(r.b);
        r.a = r.b = nullptr;
    }
    ~ProfScope() {
        if (!on) return;
        std::lock_guard<std::mutex> lock(g_prof_mu);
        if (cream::tl_stop_event == r.b || cream::tl_start_event == r.a) {   // the launch did not take them: no sample
            cream::tl_stop_event = nullptr;
            cream::tl_start_event = nullptr;
            give_back();
        } else {
            g_prof_recs.push_back(r);
        }
    }
};
#define PTRY(kind, st, flops, bytes, call) do { ProfScope ps_((kind), (hipStream_t)(st), (double)(flops), (double)(bytes)); TRY(call); } while (0)
double attn_flops(const cream_block_desc* d) {      // algorithmic forward flops of the attention core
    const double N = d->N;
    return (double)d->B * d->H * (4.0 * N * N * 64 + 4.0 * N * 64 * 60);
}

template <typename T> T* at(void* base, int64_t off) { return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + off); }
template <typename T> const T* at(const void* base, int64_t off) {
    return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + off);
}

}  // namespace

extern "C" {

int64_t cream_block_fwd_workspace(const cream_block_desc* d, int64_t* off_x, int64_t* off_x1, int64_t* off_f)
{
    if (!desc_ok(d)) return CREAM_ERR_BAD_ARG;
    const FwdLayout L = fwd_layout(dims_of(d));
    if (off_x) *off_x = L.xsum;
    if (off_x1) *off_x1 = L.x1;
    if (off_f) *off_f = L.f;
    return L.total;
}

int cream_block_fwd(const cream_block_desc* d, void* ws, const float* x_in, const void* pend_f, const float* pend_scale,
                    const float* dp1, void* stream)
{
    if (!desc_ok(d) || !ws || !x_in) return CREAM_ERR_BAD_ARG;
    const Dims x = dims_of(d);
    const FwdLayout L = fwd_layout(x);
    const int M = (int)x.M, E = (int)x.E, Q = (int)x.Q, F = (int)x.F, N = (int)x.N;
    const float* xin = x_in;
    // LN1 — with a pending residual branch of the previous block: x = x_in + s_prev * pend_f first
    if (pend_f) {
        PTRY(K_LN_FWD, stream, 0, (double)M * E * 12, cream_add_ln_fwd(at<float>(ws, L.xsum), at<void>(ws, L.a), at<float>(ws, L.mean1), at<float>(ws, L.rstd1), x_in, pend_f,
                             pend_scale, N, d->ln1_g, d->ln1_b, M, E, d->eps1, stream));
        xin = at<float>(ws, L.xsum);
    } else {
        PTRY(K_LN_FWD, stream, 0, (double)M * E * 6, cream_ln_fwd(at<void>(ws, L.a), at<float>(ws, L.mean1), at<float>(ws, L.rstd1), x_in, d->ln1_g, d->ln1_b, M, E, d->eps1,
                         stream));
    }
    // qkv: rows [q | k | v] = the first Q rows of the three de-interleaved parts, bias = plain prefix
    // (qkv_super.py:72-83)
    PTRY(K_GEMM_NT, stream, 2.0 * M * 3 * Q * E, 0, cream_linear_fwd_seg(at<void>(ws, L.qkv), at<void>(ws, L.a), d->wqkv, d->bqkv, M, 3 * Q, E, d->ld_qkv, Q, d->seg_qkv,
                             stream));
    const uint16_t* qkv = at<uint16_t>(ws, L.qkv);
    const int64_t sn = 3 * (int64_t)Q, sb = (int64_t)N * sn;
    PTRY(K_ATTN_FWD, stream, attn_flops(d), 0, cream_attn_rpe2d_fwd_img(at<void>(ws, L.o), at<float>(ws, L.lse), d->inference ? nullptr : at<void>(ws, L.sp), qkv, qkv + Q, qkv + 2 * Q, sb, sn, 64,
                             d->tkv, d->tkh, d->tvv, d->tvh, (int)d->ldt, d->timg, d->B, d->H, N, d->gh, d->gw, d->mr, d->attn_scale,
                             CREAM_BF16, stream));
    PTRY(K_GEMM_NT, stream, 2.0 * M * E * Q, 0, cream_linear_fwd(at<void>(ws, L.p), at<void>(ws, L.o), d->wproj, d->bproj, M, E, Q, d->ld_proj, stream));
    // x1 = x + s1 * p ; c = LN2(x1)
    PTRY(K_LN_FWD, stream, 0, (double)M * E * 12, cream_add_ln_fwd(at<float>(ws, L.x1), at<void>(ws, L.c), at<float>(ws, L.mean2), at<float>(ws, L.rstd2), xin,
                         at<void>(ws, L.p), dp1, N, d->ln2_g, d->ln2_b, M, E, d->eps2, stream));
    // fc1 + gelu in one pass; L.h holds gelu'(h) for the backward, L.g = gelu(h)
    PTRY(K_GEMM_NT_GELU, stream, 2.0 * M * F * E, 0, cream_linear_gelu_fwd_pad(d->inference ? nullptr : at<void>(ws, L.h), at<void>(ws, L.g), at<void>(ws, L.c), d->w1, d->b1, M, F,
                                  d->F_valid > 0 ? d->F_valid : F, E, d->ld_w1, stream));
    PTRY(K_GEMM_NT, stream, 2.0 * M * E * F, 0, cream_linear_fwd(at<void>(ws, L.f), at<void>(ws, L.g), d->w2, d->b2, M, E, F, d->ld_w2, stream));
    return CREAM_OK;
}

int64_t cream_block_bwd_workspace(const cream_block_desc* d, int64_t* off_dx, int64_t* off_df_prev, int64_t* off_pl1)
{
    if (!desc_ok(d)) return CREAM_ERR_BAD_ARG;
    const BwdLayout L = bwd_layout(dims_of(d));
    if (off_dx) *off_dx = L.dx;
    if (off_df_prev) *off_df_prev = L.df_prev;
    if (off_pl1) *off_pl1 = L.pl1;
    return L.total;
}

int cream_block_bwd(const cream_block_desc* d, const cream_block_grads* G, const void* fws, const float* x, void* ws,
                    const float* dx2, const void* df, const float* pb2, int pb2_parts, int64_t pb2_pstride, const float* dp1,
                    const float* prev_scale, int want_prev, void* stream, void* side_stream)
{
    if (!desc_ok(d) || !G || !fws || !x || !ws || !dx2 || !df || !pb2 || pb2_parts <= 0) return CREAM_ERR_BAD_ARG;
    const Dims D = dims_of(d);
    const FwdLayout FL = fwd_layout(D);
    const BwdLayout L = bwd_layout(D);
    const int M = (int)D.M, E = (int)D.E, Q = (int)D.Q, F = (int)D.F, N = (int)D.N, P = (int)D.P, slabs = (int)D.slabs;
    hipStream_t main = (hipStream_t)stream, side = side_stream ? (hipStream_t)side_stream : main;

    // ---- MLP branch ----------------------------------------------------------------------------
    // Every weight gradient is launched on the side stream AS SOON AS its operands exist (launching them in two pairs behind
    // dgrad_mul / the attention backward instead: 11.0 vs 10.83 ms per step, same-box A/B x3); token-sliced partial tiles, added by
    // cream_grad_finalize at the end of the block.
    const int wb16 = wgrad_bf16_mode();           // partial tiles as bf16 (half the partial traffic) or fp32
    const int S2 = (int)D.S2, S1 = (int)D.S1, Sp = (int)D.Sp, Sq = (int)D.Sq;
    {
        // df is complete on main.  If it is the df_prev that the PREVIOUS call on this thread and these streams produced, the
        // side stream is already ordered behind its producer (it waited for that call's last LayerNorm backward before the
        // finalisation): no event — one marker packet less per block on the main chain.  The record is consumed by THIS call
        // whatever it decides (include/cream_amd.h states the contract: df must then be passed on untouched).
        static int reuse = -1;                                             // CREAM_REUSE_ORDER=0: always fork (A/B switch)
        if (reuse < 0) { const char* e = getenv("CREAM_REUSE_ORDER"); reuse = e ? (atoi(e) != 0) : 1; }
        const bool ordered = reuse && df && df == tl_ordered.df && main == tl_ordered.main && side == tl_ordered.side;
        tl_ordered.df = nullptr;
        if (!ordered && !fork(main, side)) return CREAM_ERR_LAUNCH;
        PTRY(K_GEMM_TN, side, 2.0 * M * E * F, 0, wgrad_parts(wb16, at<float>(ws, L.pw2), nullptr, df, at<void>(fws, FL.g), M, E, F, S2, side));
    }
    // dh = (df . W2) * gelu'(h) (saved by the forward) and the fc1 bias partials in the dgrad's epilogue
    KernelFork f_dh(main, side), f_dp(main, side), f_dqkv(main, side), f_dx(main, side);
    if (!f_dh.arm()) return CREAM_ERR_LAUNCH;                             // (the dgrad's own packet carries the event)
    PTRY(K_GEMM_NT_MUL, main, 2.0 * M * E * F, 0, cream_linear_dgrad_mul(at<void>(ws, L.dh), at<float>(ws, L.pb1), df, d->w2_t, at<void>(fws, FL.h), M, E, F, d->ld_w2_t,
                                 main));
    if (!f_dh.join()) return CREAM_ERR_LAUNCH;
    PTRY(K_GEMM_TN, side, 2.0 * M * F * E, 0, wgrad_parts(wb16, at<float>(ws, L.pw1), nullptr, at<void>(ws, L.dh), at<void>(fws, FL.c), M, F, E, S1, side));
    PTRY(K_GEMM_NT, main, 2.0 * M * F * E, 0, cream_linear_dgrad(at<void>(ws, L.dc), at<void>(ws, L.dh), d->w1_t, M, F, E, d->ld_w1_t, main));
    // dx1 = dx2 + dLN2(dc); dp = s1 * dx1 (gradient of the proj output) and its column sums
    if (!f_dp.arm()) return CREAM_ERR_LAUNCH;
    PTRY(K_LN_BWD, main, 0, (double)M * E * 16, cream_ln_bwd(at<float>(ws, L.dx1), at<void>(ws, L.dp), at<float>(ws, L.pl2), at<void>(ws, L.dc), at<float>(fws, FL.x1),
                     at<float>(fws, FL.mean2), at<float>(fws, FL.rstd2), d->ln2_g, dx2, dp1, N, M, E, main));
    // ---- attention branch -----------------------------------------------------------------------
    if (!f_dp.join()) return CREAM_ERR_LAUNCH;                            // dp complete on main
    PTRY(K_GEMM_TN, side, 2.0 * M * E * Q, 0, wgrad_parts(wb16, at<float>(ws, L.pwp), nullptr, at<void>(ws, L.dp), at<void>(fws, FL.o), M, E, Q, Sp, side));
    PTRY(K_GEMM_NT, main, 2.0 * M * E * Q, 0, cream_linear_dgrad(at<void>(ws, L.dout), at<void>(ws, L.dp), d->wproj_t, M, E, Q, d->ld_proj_t, main));
    const uint16_t* qkv = at<uint16_t>(fws, FL.qkv);
    uint16_t* dqkv = at<uint16_t>(ws, L.dqkv);
    const int64_t sn = 3 * (int64_t)Q, sb = (int64_t)N * sn;
    if (!f_dqkv.arm()) return CREAM_ERR_LAUNCH;
    PTRY(K_ATTN_BWD, main, 2.5 * attn_flops(d), 0, cream_attn_rpe2d_bwd_img(dqkv, dqkv + Q, dqkv + 2 * Q, sb, sn, 64, at<float>(ws, L.dtab), at<void>(ws, L.dlt), at<void>(ws, L.qe),
                             at<void>(ws, L.de), at<float>(ws, L.delta), at<void>(ws, L.dout), at<void>(fws, FL.o),
                             at<float>(fws, FL.lse), at<void>(fws, FL.sp), qkv, qkv + Q, qkv + 2 * Q, sb, sn, 64, d->tkv, d->tkh,
                             d->tvv, d->tvh, (int)d->ldt, d->timg, d->B, d->H, N, d->gh, d->gw, d->mr, d->attn_scale, CREAM_BF16, main));
    if (!f_dqkv.join()) return CREAM_ERR_LAUNCH;                          // dqkv complete on main
    // qkv weight gradient (rows [q | k | v]); the bias gradient (column sums of dqkv) rides on it
    PTRY(K_GEMM_TN, side, 2.0 * M * 3 * Q * E, 0, wgrad_parts(wb16, at<float>(ws, L.pwq), at<float>(ws, L.pbq), dqkv, at<void>(fws, FL.a), M, 3 * Q, E, Sq, side));
    PTRY(K_GEMM_NT, main, 2.0 * M * 3 * Q * E, 0, cream_linear_dgrad_seg(at<void>(ws, L.da), dqkv, d->wqkv_t, M, 3 * Q, E, d->ld_qkv_t, Q, d->seg_qkv_t, main));
    if (!f_dx.arm()) return CREAM_ERR_LAUNCH;
    PTRY(K_LN_BWD, main, 0, (double)M * E * 16, cream_ln_bwd(at<float>(ws, L.dx), want_prev ? at<void>(ws, L.df_prev) : nullptr, at<float>(ws, L.pl1), at<void>(ws, L.da), x,
                     at<float>(fws, FL.mean1), at<float>(fws, FL.rstd1), d->ln1_g, at<float>(ws, L.dx1), prev_scale, N, M, E, main));

    // ---- gradient finalisation: every parameter of the block in one launch, on the side stream ----
    if (!f_dx.join()) return CREAM_ERR_LAUNCH;
    cream_grad_job J[18];
    int n = 0;
    auto job = [&](float* dst, int64_t ld, const void* src, int nparts, int64_t pstride, int rows, int cols, int interleave,
                   int bf16) {
        J[n].dst = dst; J[n].src = src; J[n].ld = ld; J[n].pstride = pstride; J[n].nparts = nparts; J[n].rows = rows;
        J[n].cols = cols; J[n].interleave = interleave; J[n].src_bf16 = bf16; J[n].overwrite = 0;
        ++n;
    };
    job(G->w2, G->ld_w2, at<void>(ws, L.pw2), S2, (int64_t)E * F, E, F, 0, wb16);
    job(G->w1, G->ld_w1, at<void>(ws, L.pw1), S1, (int64_t)F * E, F, E, 0, wb16);
    job(G->wproj, G->ld_proj, at<void>(ws, L.pwp), Sp, (int64_t)E * Q, E, Q, 0, wb16);
    job(G->wqkv, G->ld_qkv, at<void>(ws, L.pwq), Sq, 3 * (int64_t)Q * E, 3 * Q, E, Q, wb16);
    job(G->bqkv, 3 * Q, at<void>(ws, L.pbq), Sq, 3 * Q, 1, 3 * Q, 0, 0);
    job(G->b2, E, pb2, pb2_parts, pb2_pstride, 1, E, 0, 0);
    job(G->b1, F, at<void>(ws, L.pb1), slabs, F, 1, F, 0, 0);
    job(G->ln2_g, E, at<float>(ws, L.pl2), P, 3 * (int64_t)E, 1, E, 0, 0);
    job(G->ln2_b, E, at<float>(ws, L.pl2) + E, P, 3 * (int64_t)E, 1, E, 0, 0);
    job(G->bproj, E, at<float>(ws, L.pl2) + 2 * E, P, 3 * (int64_t)E, 1, E, 0, 0);
    const int nb = 2 * d->mr + 2;
    float* tabs[4] = {G->tkv, G->tkh, G->tvv, G->tvh};
    for (int t = 0; t < 4; ++t)
        job(tabs[t], G->ldt, at<float>(ws, L.dtab) + t * 32 * 64, cream_attn_rpe2d_dtab_parts(d->B, d->H), 4 * 32 * 64, nb, 64, 0, 0);
    job(G->ln1_g, E, at<float>(ws, L.pl1), P, 3 * (int64_t)E, 1, E, 0, 0);
    job(G->ln1_b, E, at<float>(ws, L.pl1) + E, P, 3 * (int64_t)E, 1, E, 0, 0);
    PTRY(K_GRAD_FINALIZE, side, 0, 0, cream_grad_finalize(J, n, side));
    if (want_prev && main != side) tl_ordered = OrderedBehind{at<void>(ws, L.df_prev), main, side};
    return CREAM_OK;
}

int cream_block_wgrad_bf16(int on)
{
    const int prev = wgrad_bf16_mode();
    if (on >= 0) g_wgrad_bf16.store(on != 0, std::memory_order_relaxed);
    return prev;
}


int cream_block_prof_enable(int on)
{
    std::lock_guard<std::mutex> lock(g_prof_mu);
    g_prof_on = on != 0;
    return CREAM_OK;
}

int cream_block_prof_kinds(void) { return K_COUNT; }

const char* cream_block_prof_name(int kind) { return kind >= 0 && kind < K_COUNT ? kProfNames[kind] : ""; }

int cream_block_prof_collect(double* total_ms, int64_t* launches, double* flops, double* bytes)
{
    if (!total_ms || !launches || !flops || !bytes) return CREAM_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(g_prof_mu);
    for (int k = 0; k < K_COUNT; ++k) { total_ms[k] = 0; launches[k] = 0; flops[k] = 0; bytes[k] = 0; }
    int rc = CREAM_OK;
    for (const ProfRec& r : g_prof_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            total_ms[r.kind] += ms; launches[r.kind] += 1; flops[r.kind] += r.flops; bytes[r.kind] += r.bytes;
        } else {
            rc = CREAM_ERR_LAUNCH;
        }
        g_prof_free.push_back(r.a);
        g_prof_free.push_back(r.b);
    }
    g_prof_recs.clear();
    return rc;
}

}  // extern "C"
