// attn_rpe2d.hip — fused multi-head attention with AutoFormer's 2-D relative position bias
// on keys and values, forward and backward, for gfx950 (MI355X).
//
// Reference semantics (AutoFormer/model/module/multihead_super.py:135-154),
// per (batch b, head h), head_dim 64, N tokens (token 0 = class token, then a gh x gw grid):
//     Lv = q Tkv^T ; Lh = q Tkh^T                                  (N x nb bucket lookups)
//     A[i,j] = scale * ( q_i.k_j + Lv[i, iv[i,j]] + Lh[i, ih[i,j]] )
//     P = softmax_j(A)
//     Sv[i,u] = sum_{j: iv[i,j]=u} P[i,j] ; Sh likewise
//     O_i = sum_j P[i,j] v_j + Sv[i,:] Tvv + Sh[i,:] Tvh
// Nothing of size N^2 touches HBM.  One workgroup = one (b,h); wave w owns the 32-query
// tile w and keeps the whole row block of scores in registers (N <= 256), so the softmax is
// exact (no online rescaling).  The bias gather / slot sums ride on the MFMAs through the
// one-hot extension described in attn_common.hpp.  Operands shared by the waves of a
// workgroup (K, V, Q, dO, side buffers) are streamed through LDS tile by tile (full lines,
// fetched once per workgroup, two 4-wave staging groups); transposed tiles for the products
// that contract over tokens (V^T, K^T, Q^T, dO^T) are produced by the staging stores.  LDS also
// holds the bucket tables as bf16 operands, the one-hot key operands of the fast geometry and
// the per-wave scratch of the slot <-> bucket shifts.  The AutoFormer geometry (14 x 14 grid,
// max_relative_position 14, bf16) has its own instantiation (FAST): the kernels were
// VALU-issue-bound on index arithmetic, see attn_common.hpp and DESIGN.md 4.2.
//
// dtype = bf16 (throughput mode: bf16 operands, fp32 accumulation/softmax) or fp32 (parity
// mode: v_mfma_f32_32x32x2_f32, exact fp32 products) — one code path, Tr<T> traits.
#include <hip/hip_runtime.h>
#include <hip/hip_bfloat16.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>

#include "attn_common.hpp"
#include "cream_amd.h"
#include "launch_ev.hpp"
#include "cu_budget.hpp"

namespace {
using namespace cream;

constexpr float LOG2E = 1.4426950408889634f;

// Phase timestamps for tools/probes/attn_probe.hip (compiled out of the library).
#ifdef ATTN_PROFILE
__device__ long long* g_attn_prof = nullptr;
#define PROF_DECL long long prof_t[12]; int prof_n = 0;
#define PROF_MARK() do { prof_t[prof_n++] = (long long)__builtin_readcyclecounter(); } while (0)
#define PROF_FLUSH() do { if ((threadIdx.x & 63) == 0 && g_attn_prof) { \
        long long* d_ = g_attn_prof + ((long long)blockIdx.x * 8 + (threadIdx.x >> 6)) * 12; \
        for (int i_ = 0; i_ < 12; ++i_) d_[i_] = i_ < prof_n ? prof_t[i_] : 0; } } while (0)
#else
#define PROF_DECL
#define PROF_MARK() do {} while (0)
#define PROF_FLUSH() do {} while (0)
#endif
#ifdef ATTN_PROFILE_KVLOOP           // sub-phase stamps inside one iteration of the dK/dV tile loop
#define PROF_LOOP(t) do { if ((t) == 3) PROF_MARK(); } while (0)
#else
#define PROF_LOOP(t) do {} while (0)
#endif

struct FwdArgs {
    const void* q; const void* k; const void* v;    // element (b, n, h, :) at b*sb + n*sn + h*sh
    int64_t sb, sn, sh;
    void* out;                                       // (B, N, H, 64) contiguous
    float* lse;                                      // (B, H, N)   log-sum-exp of scaled logits
    void* sp;                                        // (B, H, 64, NP) bucket sums S'^T, dtype T
    const float *tkv, *tkh, *tvv, *tvh;              // (nb, 64) tables, row stride ldt
    int ldt, nb;
    int H, NP;
    RelGeom G;
    float scale;
    int nitems;                                      // B * H (forward: persistent workgroups walk them)
    const void* timg;                                // bf16 operand images of the four tables (cream_attn_rpe2d_table_images) or NULL
    // attention dropout (multihead_super.py:145), DROP instantiations of the tile-streamed kernels only (attn_common.hpp: drop_keep)
    uint32_t drop_thr = 0, drop_seed = 0;            // keep iff hash >= drop_thr; 0: no dropout
    float drop_scale = 1.f;                          // 1 / (1 - p)
};

struct BwdArgs {
    const void* q; const void* k; const void* v;
    int64_t sb, sn, sh;
    void* dq; void* dk; void* dv;                    // same indexing with (dsb, dsn, dsh)
    int64_t dsb, dsn, dsh;
    const void* dout; const void* out;               // (B, N, H, 64) contiguous
    const float* lse;                                // (B, H, N)
    const void* sp;                                  // (B, H, 64, NP) from forward
    void* dlt;                                       // (B, H, 64, NP)  dL'^T            (A -> B)
    void* qe; void* de;                              // (B, H, NP, 32)  slot extensions  (A -> B)
    float* delta;                                    // (B, H, NP)                       (A -> B)
    float* dtab;                                     // (B*H, 4, 32, 64) per-(b,h) table gradients
    const float *tkv, *tkh, *tvv, *tvh;
    int ldt, nb;
    int H, NP;
    RelGeom G;
    float scale;
    int nitems;                                      // B * H (dQ kernel: persistent workgroups walk them)
    int stagger;                                     // one-pass kernel: start offset between the 8 phase groups (x 64 cycles)
    const void* timg;                                // bf16 operand images of the four tables or NULL (then built into dlt)
    uint32_t drop_thr = 0, drop_seed = 0;            // as in FwdArgs: the two-launch backward regenerates the forward's mask
    float drop_scale = 1.f;
};

// ---- pieces shared by forward and backward ---------------------------------------------

template <typename T> __host__ __device__ constexpr int table_pitch() { return sizeof(typename Tr<T>::elem) == 2 ? 72 : 65; }
// bf16: the bucket tables are staged in LDS as operand rows; fp32 (parity mode) reads them from
// global memory (its LDS is full of fp32 tiles)
template <typename T> __host__ __device__ constexpr bool tables_in_lds() { return sizeof(typename Tr<T>::elem) == 2; }

// operand fragments of one row of a row-major (n, 64) matrix (lane = row, steps over d); the
// row pointer is always valid (callers clamp the row index), so the loads are unconditional
// and can be issued a tile ahead of their use
template <typename T>
__device__ __forceinline__ void load_row(typename Tr<T>::frag (&f)[64 / Tr<T>::KI],
                                         const typename Tr<T>::elem* rowp, int g) {
    using TT = Tr<T>;
#pragma unroll
    for (int ks = 0; ks < 64 / TT::KI; ++ks) f[ks] = TT::load(rowp + ks * TT::KI + g * TT::EPL);
}
template <typename T>
__device__ __forceinline__ void load_row32(typename Tr<T>::frag (&f)[32 / Tr<T>::KI],
                                           const typename Tr<T>::elem* rowp, int g) {
    using TT = Tr<T>;
#pragma unroll
    for (int ks = 0; ks < 32 / TT::KI; ++ks) f[ks] = TT::load(rowp + ks * TT::KI + g * TT::EPL);
}
template <typename T, int NF>
__device__ __forceinline__ void zero_frags(typename Tr<T>::frag (&f)[NF]) {
#pragma unroll
    for (int i = 0; i < NF; ++i) f[i] = Tr<T>::zero();
}

// lookups^T (32 buckets x 32 queries) = table(32 x 64) . X^T for the vertical and horizontal
// table, written to the wave's scratch as row[q][u] / row[q][32 + u].
//   tabR : LDS operand rows [64][tp] (rows 0..31 vertical, 32..63 horizontal)  — bf16 mode
//   tv/th: the fp32 tables in global memory                                    — fp32 mode
template <typename T>
__device__ __forceinline__ void table_lookups(float* scr, const typename Tr<T>::frag (&xb)[64 / Tr<T>::KI],
                                              const typename Tr<T>::elem* tabR, const float* tv, const float* th,
                                              int ldt, int nb, int lane) {
    using TT = Tr<T>;
    constexpr int tp = table_pitch<T>();
    const int u = lane & 31, g = lane >> 5;
    f32x16 av = {}, ah = {};
#pragma unroll
    for (int ks = 0; ks < 64 / TT::KI; ++ks) {
        const int k0 = ks * TT::KI + g * TT::EPL;
        if constexpr (tables_in_lds<T>()) {
            av = TT::mma(TT::load(tabR + u * tp + k0), xb[ks], av);
            ah = TT::mma(TT::load(tabR + (u + 32) * tp + k0), xb[ks], ah);
        } else {
            av = TT::mma(TT::load_f32(tv + (int64_t)u * ldt + k0, u < nb), xb[ks], av);
            ah = TT::mma(TT::load_f32(th + (int64_t)u * ldt + k0, u < nb), xb[ks], ah);
        }
    }
    float* row = scr + (lane & 31) * LP;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        row[acc_row(r, g)] = av[r];
        row[32 + acc_row(r, g)] = ah[r];
    }
}

// extension fragments x_i[slot] (B operand, 32 slots) from the wave's scratch
template <typename T>
__device__ __forceinline__ void build_ext(typename Tr<T>::frag (&xe)[32 / Tr<T>::KI], const float* scr,
                                          int lane, int qi, int qr, int qc, const RelGeom& G) {
    using TT = Tr<T>;
    const int g = lane >> 5;
    const float* row = scr + (lane & 31) * LP;
    const float cls = row[0] + row[32];
#pragma unroll
    for (int ks = 0; ks < 32 / TT::KI; ++ks) {
        if constexpr (TT::EPL == 1) {
            xe[ks] = ext_gather(row, cls, ks * 2 + g, qi, qr, qc, G);
        } else {
            f32x8v x;
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = ext_gather(row, cls, ks * 16 + g * 8 + e, qi, qr, qc, G);
            xe[ks] = __builtin_bit_cast(bf16x8, __builtin_convertvector(x, hwbf16x8));
        }
    }
}

// fast geometry (attn_common.hpp): window reads with immediate offsets; bf16 fragments only
__device__ __forceinline__ void build_ext14(bf16x8 (&xe)[2], float* scr, int lane, bool tile0, int qr, int qc) {
    const int g = lane >> 5;
    float* row = scr + (lane & 31) * LP;
    const float cls = row[0] + row[32];
    if (tile0) {
        ext_fix_query0(row, cls, lane);
        wave_lds_fence();
    }
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        float x[8];
        ext_window14(x, row, cls, kh, g, qr, qc);
        xe[kh] = __builtin_bit_cast(bf16x8, __builtin_convertvector((f32x8v{x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]}), hwbf16x8));
    }
}

// slot tile (accumulator, lane = query, rows = slots) -> this lane's 32 bucket values of table
// g (0 vertical / 1 horizontal) in `bk`, and the full bucket rows row[q][0..63] in scratch
__device__ __forceinline__ void slots_to_buckets(float (&bk)[32], float* scr, const f32x16& x, int lane, int qi,
                                                 int qr, int qc, const RelGeom& G) {
    const int g = lane >> 5;
    float* row = scr + (lane & 31) * LP;
#pragma unroll
    for (int r = 0; r < 16; ++r) row[acc_row(r, g)] = x[r];
    wave_lds_fence();
#pragma unroll
    for (int u = 0; u < 32; ++u) bk[u] = bucket_from_slots(row, u, g, qi, qr, qc, G);
    wave_lds_fence();
}

// bucket rows^T (64 buckets x NP queries, dtype T) for the second backward launch
template <typename T>
__device__ __forceinline__ void store_buckets_T(typename Tr<T>::elem* dst /* + qi */, int NP,
                                                const float (&bk)[32], int g) {
#pragma unroll
    for (int u = 0; u < 32; ++u) dst[(int64_t)(g * 32 + u) * NP] = Tr<T>::from_f(bk[u]);
}

// acc^T(64 x 32 queries) += Tab^T(64 x 64 buckets) . rows^T  with Tab^T in LDS ([64][tp], columns
// 0..31 vertical / 32..63 horizontal table) and the bucket rows still in registers: lane (q, g)
// holds the 32 buckets of table g of its query, so in contraction step ks lane group g supplies
// the elements (table g, bucket ks*EPL + e) — the order of the contraction index is free as
// long as both operands agree.  No LDS round trip for the bucket rows.
template <typename T>
__device__ __forceinline__ void add_bucket_product(f32x16 (&o)[2], const typename Tr<T>::elem* tabT,
                                                   const float (&bk)[32], int lane) {
    using TT = Tr<T>;
    constexpr int tp = table_pitch<T>();
    const int g = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 32 / TT::EPL; ++ks) {
        typename TT::frag b;
        if constexpr (TT::EPL == 1) b = bk[ks];
        else {
            f32x8v x;
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = bk[ks * 8 + e];
            b = __builtin_bit_cast(bf16x8, __builtin_convertvector(x, hwbf16x8));
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
            o[dt] = TT::mma(TT::load(tabT + ((lane & 31) + 32 * dt) * tp + g * 32 + ks * TT::EPL), b, o[dt]);
    }
}

// ---- streamed operand tiles -----------------------------------------------------------------
// Every operand that all waves of the workgroup read (K, V, Q, dO, ...) is streamed through LDS
// one 32-token tile at a time: full 128-byte lines, each fetched ONCE per workgroup, register
// staged (loads for tile t+1 are in flight while tile t is consumed; written to the other
// half of a double buffer before the single barrier of the iteration).  Reading such rows
// straight from global memory per wave costs 7x the L2 traffic at a quarter line efficiency
// (16 B of each 128-B line per instruction) — measured 36k of 54k cycles in the dK/dV loop.
template <typename T, int ROWS, int COLS> struct TileRegs {
    static constexpr int V = 16 / sizeof(typename Tr<T>::elem), CPR = COLS / V, CH = ROWS * CPR, NI = (CH + 255) / 256;
    u32x4v v[NI];
};

// rows [row0, row0 + ROWS) of a row-major matrix (row stride `rs` elements), zero beyond `nrows`
template <typename T, int ROWS, int COLS>
__device__ __forceinline__ void tile_load(TileRegs<T, ROWS, COLS>& r, const typename Tr<T>::elem* src, int64_t rs,
                                          int row0, int nrows, int col0 = 0) {
    using R = TileRegs<T, ROWS, COLS>;
#pragma unroll
    for (int i = 0; i < R::NI; ++i) {
        const int c = threadIdx.x + i * blockDim.x;
        const int row = c / R::CPR, cc = c - row * R::CPR;
        r.v[i] = (c < R::CH && row0 + row < nrows)
                     ? *reinterpret_cast<const u32x4v*>(src + (int64_t)(row0 + row) * rs + col0 + cc * R::V)
                     : u32x4v{0, 0, 0, 0};
    }
}

// RM: dst_rm[row][col] (pitch prm);  TR: dst_t[col][row] (pitch pt)
template <typename T, int ROWS, int COLS, bool RM, bool TR>
__device__ __forceinline__ void tile_store(const TileRegs<T, ROWS, COLS>& r, typename Tr<T>::elem* dst_rm, int prm,
                                           typename Tr<T>::elem* dst_t, int pt) {
    using R = TileRegs<T, ROWS, COLS>;
    using E = typename Tr<T>::elem;
#pragma unroll
    for (int i = 0; i < R::NI; ++i) {
        const int c = threadIdx.x + i * blockDim.x;
        if (c < R::CH) {
            const int row = c / R::CPR, cc = c - row * R::CPR;
            union { u32x4v v; E e[R::V]; } u;
            u.v = r.v[i];
            if constexpr (RM) {
                if constexpr (sizeof(E) == 2) {
                    E* d = dst_rm + row * prm + cc * R::V;
                    if ((prm * 2) % 16 == 0) *reinterpret_cast<u32x4v*>(d) = u.v;
                    else {                                   // 8-byte aligned rows (pitch 36)
                        *reinterpret_cast<u32x2v*>(d) = u32x2v{u.v[0], u.v[1]};
                        *reinterpret_cast<u32x2v*>(d + 4) = u32x2v{u.v[2], u.v[3]};
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < R::V; ++e) dst_rm[row * prm + cc * R::V + e] = u.e[e];
                }
            }
            if constexpr (TR) {
#pragma unroll
                for (int e = 0; e < R::V; ++e) dst_t[(cc * R::V + e) * pt + row] = u.e[e];
            }
        }
    }
}

// The same with a GROUP of 256 threads (4 waves) owning the tile: thread `tid` (0..255) of the group
// moves chunks tid, tid + 256, ... — every thread of the group has work (no predicate, no
// exec-mask branch); callers pick the group wave-uniformly.
template <typename T, int ROWS, int COLS>
__device__ __forceinline__ void tile_load_g(TileRegs<T, ROWS, COLS>& r, const typename Tr<T>::elem* src, int64_t rs,
                                            int row0, int nrows, int col0, int tid) {
    using R = TileRegs<T, ROWS, COLS>;
    static_assert(R::CH % 256 == 0, "tile must split evenly over a 256-thread group");
#pragma unroll
    for (int i = 0; i < R::NI; ++i) {
        const int c = tid + i * 256;
        const int row = c / R::CPR, cc = c - row * R::CPR;
        const int rr = min(row0 + row, nrows - 1);
        const u32x4v v = *reinterpret_cast<const u32x4v*>(src + (int64_t)rr * rs + col0 + cc * R::V);
        r.v[i] = row0 + row < nrows ? v : u32x4v{0, 0, 0, 0};
    }
}

template <typename T, int ROWS, int COLS, bool RM, bool TR>
__device__ __forceinline__ void tile_store_g(const TileRegs<T, ROWS, COLS>& r, typename Tr<T>::elem* dst_rm, int prm,
                                             typename Tr<T>::elem* dst_t, int pt, int tid) {
    using R = TileRegs<T, ROWS, COLS>;
    using E = typename Tr<T>::elem;
#pragma unroll
    for (int i = 0; i < R::NI; ++i) {
        const int c = tid + i * 256;
        const int row = c / R::CPR, cc = c - row * R::CPR;
        union { u32x4v v; E e[R::V]; } u;
        u.v = r.v[i];
        if constexpr (RM) {
            if constexpr (sizeof(E) == 2) {
                E* d = dst_rm + row * prm + cc * R::V;
                if ((prm * 2) % 16 == 0) *reinterpret_cast<u32x4v*>(d) = u.v;
                else {                                   // 8-byte aligned rows (pitch 36)
                    *reinterpret_cast<u32x2v*>(d) = u32x2v{u.v[0], u.v[1]};
                    *reinterpret_cast<u32x2v*>(d + 4) = u32x2v{u.v[2], u.v[3]};
                }
            } else {
#pragma unroll
                for (int e = 0; e < R::V; ++e) dst_rm[row * prm + cc * R::V + e] = u.e[e];
            }
        }
        if constexpr (TR) {
#pragma unroll
            for (int e = 0; e < R::V; ++e) dst_t[(cc * R::V + e) * pt + row] = u.e[e];
        }
    }
}

// pitches of staged tiles: row-major [32][RMP(cols)], transposed / bucket-row tiles [rows][TPT]
template <typename T> __host__ __device__ constexpr int rm_pitch(int cols) { return sizeof(typename Tr<T>::elem) == 2 ? cols + 8 : cols + 1; }
template <typename T> __host__ __device__ constexpr int t_pitch() { return sizeof(typename Tr<T>::elem) == 2 ? 36 : 33; }

// workgroup-cooperative table staging.  Two phases so that ALL global loads of a workgroup's
// prologue are in flight together (a load -> convert -> store loop pays one L2/HBM round trip per
// iteration: 6-9 dependent round trips were a quarter of the forward kernel):
//   tab_load  : this thread's float4 pieces (4 d-values of one bucket) of a (v, h) table pair
//   tab_store_T: tabT[d][u] = (u < 32 ? tv[u][d] : th[u-32][d]), zero for u >= nb   (transposed)
//   tab_store_R: tabR[u][d] operand rows, rows 0..31 from tv, 32..63 from th
constexpr int TAB_IT = 4;                        // 64 buckets x 16 pieces = 1024 <= 4 x 256 threads
struct TabRegs { f32x4v v[TAB_IT]; };

__device__ __forceinline__ void tab_load(TabRegs& r, const float* tv, const float* th, int ldt, int nb) {
#pragma unroll
    for (int it = 0; it < TAB_IT; ++it) {
        const int i = threadIdx.x + it * blockDim.x;
        const int d4 = (i & 15) * 4, u = (i >> 4) & 63, uu = u & 31;
        const float* t = u < 32 ? tv : th;
        r.v[it] = (i < 1024 && uu < nb) ? *reinterpret_cast<const f32x4v*>(t + (int64_t)uu * ldt + d4) : f32x4v{0, 0, 0, 0};
    }
}
template <typename T>
__device__ __forceinline__ void tab_store_T(const TabRegs& r, typename Tr<T>::elem* tabT) {
    constexpr int tp = table_pitch<T>();
#pragma unroll
    for (int it = 0; it < TAB_IT; ++it) {
        const int i = threadIdx.x + it * blockDim.x;
        if (i < 1024) {
            const int d4 = (i & 15) * 4, u = i >> 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) tabT[(d4 + e) * tp + u] = Tr<T>::from_f(r.v[it][e]);
        }
    }
}
template <typename T>
__device__ __forceinline__ void tab_store_R(const TabRegs& r, typename Tr<T>::elem* tabR) {
    constexpr int tp = table_pitch<T>();
#pragma unroll
    for (int it = 0; it < TAB_IT; ++it) {
        const int i = threadIdx.x + it * blockDim.x;
        if (i < 1024) {
            const int d4 = (i & 15) * 4, u = i >> 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) tabR[u * tp + d4 + e] = Tr<T>::from_f(r.v[it][e]);
        }
    }
}

// ---- one-hot key operands as ready-made bf16 MFMA fragments (fast path) ------------------------
// ohr[j][c]  ([NP][OHP])     : rows = keys, contraction = slots  (S^T += E . X^T)
// oht[c][j]  ([32][NP + 4])  : rows = slots, contraction = keys  (slot sums += E^T . P^T), read
//                              with load_perm like a transposed V tile
// Built once per workgroup from the geometry; the bit-twiddled operands (Tr::onehot_row /
// onehot_perm, ~45 VALU instructions per key tile and wave) disappear from the tile loops.
constexpr int OHP = 40;
__host__ __device__ constexpr int oht_pitch(int NP) { return NP + 4; }
__host__ __device__ constexpr size_t onehot_bytes(int NP) { return (size_t)NP * OHP * 2 + (size_t)32 * oht_pitch(NP) * 2; }

__device__ __forceinline__ void fill_onehot(short* ohr, short* oht, const uint32_t* masks, int NP) {
    // masks[] must be visible (barrier) before this call
    for (int i = threadIdx.x; i < NP * 4; i += blockDim.x) {          // 8 slots of one key
        const int j = i >> 2, cc = i & 3;
        const uint32_t m = masks[j] >> (8 * cc);
        u32x4v w;
#pragma unroll
        for (int p = 0; p < 4; ++p) w[p] = ((m >> (2 * p)) & 1u) * 0x3F80u + ((m >> (2 * p + 1)) & 1u) * 0x3F800000u;
        *reinterpret_cast<u32x4v*>(ohr + j * OHP + cc * 8) = w;
    }
    const int tpo = oht_pitch(NP);
    for (int i = threadIdx.x; i < 32 * (NP >> 3); i += blockDim.x) {  // 8 keys of one slot
        const int c = i / (NP >> 3), j0 = (i - c * (NP >> 3)) * 8;
        const u32x4v lo = *reinterpret_cast<const u32x4v*>(masks + j0);
        const u32x4v hi = *reinterpret_cast<const u32x4v*>(masks + j0 + 4);
        u32x2v w0, w1;
        w0[0] = ((lo[0] >> c) & 1u) * 0x3F80u + ((lo[1] >> c) & 1u) * 0x3F800000u;
        w0[1] = ((lo[2] >> c) & 1u) * 0x3F80u + ((lo[3] >> c) & 1u) * 0x3F800000u;
        w1[0] = ((hi[0] >> c) & 1u) * 0x3F80u + ((hi[1] >> c) & 1u) * 0x3F800000u;
        w1[1] = ((hi[2] >> c) & 1u) * 0x3F80u + ((hi[3] >> c) & 1u) * 0x3F800000u;
        *reinterpret_cast<u32x2v*>(oht + c * tpo + j0) = w0;
        *reinterpret_cast<u32x2v*>(oht + c * tpo + j0 + 4) = w1;
    }
}

template <typename T>
__device__ __forceinline__ void store_ext_rows(typename Tr<T>::elem* dst /* row of 32 */,
                                               const typename Tr<T>::frag (&xe)[32 / Tr<T>::KI], int g) {
    using TT = Tr<T>;
#pragma unroll
    for (int ks = 0; ks < 32 / TT::KI; ++ks) {
        if constexpr (TT::EPL == 1) dst[ks * 2 + g] = xe[ks];
        else *reinterpret_cast<bf16x8*>(dst + ks * TT::KI + g * TT::EPL) = xe[ks];
    }
}

// accumulator tile (lane = row n of the output, 16 x 2 d-values) -> (.., n, h, :) row pointer
template <typename T>
__device__ __forceinline__ void store_rows_64(typename Tr<T>::elem* op, const f32x16 (&o)[2], int g) {
    using TT = Tr<T>;
    using E = typename TT::elem;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int d = dt * 32 + 8 * r4 + 4 * g;
            if constexpr (sizeof(E) == 2) {
                *reinterpret_cast<u32x2v*>(op + d) = u32x2v{f2bf_pair(o[dt][4 * r4], o[dt][4 * r4 + 1]),
                                                            f2bf_pair(o[dt][4 * r4 + 2], o[dt][4 * r4 + 3])};
            } else {
                *reinterpret_cast<f32x4v*>(op + d) =
                    f32x4v{o[dt][4 * r4], o[dt][4 * r4 + 1], o[dt][4 * r4 + 2], o[dt][4 * r4 + 3]};
            }
        }
}

template <typename T> __host__ __device__ constexpr int tabr_rows(int n_tables) { return tables_in_lds<T>() ? 32 * n_tables : 0; }

// bytes of one set of staged tiles
template <typename T> __host__ __device__ constexpr size_t rm_tile_bytes(int cols) { return (size_t)32 * rm_pitch<T>(cols) * sizeof(typename Tr<T>::elem); }
template <typename T> __host__ __device__ constexpr size_t t_tile_bytes() { return (size_t)64 * t_pitch<T>() * sizeof(typename Tr<T>::elem); }

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
// LDS: [2 x max(K tile, V^T tile)] | value tables^T [64][tp] | key table rows [64][tp] (bf16) | masks [NP] | scratch
template <typename T> size_t fwd_lds_bytes(int NP, int waves, bool fast) {
    using E = typename Tr<T>::elem;
    const size_t tile = rm_tile_bytes<T>(64) > t_tile_bytes<T>() ? rm_tile_bytes<T>(64) : t_tile_bytes<T>();
    return 2 * tile + (size_t)(64 + tabr_rows<T>(2)) * table_pitch<T>() * sizeof(E) + (size_t)NP * 4 +
           (size_t)waves * 32 * LP * 4 + (fast ? onehot_bytes(NP) : 0);
}

// FAST: the AutoFormer geometry (14 x 14 grid, max_relative_position 14, bf16): compile-time
// shapes, window-read shifts and one-hot operands staged in LDS (see attn_common.hpp)
template <typename T, int NT, bool FAST, bool DROP = false>
__global__ __launch_bounds__((NT < 4 ? 4 : NT) * 64) void attn_rpe2d_fwd_kernel(const FwdArgs a) {
    using TT = Tr<T>;
    using E = typename TT::elem;
    using F = typename TT::frag;
    constexpr int KI = TT::KI, EPL = TT::EPL, S64 = 64 / KI, S32 = 32 / KI;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    static_assert(!FAST || (sizeof(E) == 2 && NT == 7), "fast path: bf16, N = 197");
    const RelGeom G = FAST ? RelGeom{197, G14, G14, G14} : a.G;
    const int N = G.n, NP = FAST ? 224 : a.NP;
    const int nt = NP >> 5;
    constexpr int tp = table_pitch<T>(), kp = rm_pitch<T>(64), vp = t_pitch<T>();
    constexpr size_t tile_b = rm_tile_bytes<T>(64) > t_tile_bytes<T>() ? rm_tile_bytes<T>(64) : t_tile_bytes<T>();
    E* buf0 = reinterpret_cast<E*>(smem);
    E* buf1 = reinterpret_cast<E*>(smem + tile_b);
    E* tvt = reinterpret_cast<E*>(smem + 2 * tile_b);                     // value tables^T [64][tp]
    E* tkr = tvt + 64 * tp;                                               // key table rows [64][tp] (bf16)
    uint32_t* masks = reinterpret_cast<uint32_t*>(tkr + tabr_rows<T>(2) * tp);   // [NP]
    float* scratch = reinterpret_cast<float*>(masks + NP);                // [waves][32][LP]
    short* ohr = reinterpret_cast<short*>(scratch + (blockDim.x >> 6) * 32 * LP);   // fast path: one-hot operands
    short* oht = ohr + NP * OHP;
    const int otp = oht_pitch(NP);

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, g = lane >> 5, c32 = lane & 31;
    // waves beyond the query tiles only help staging; the fast instantiation is launched with exactly
    // nt waves, and saying so lets the accumulators start from the MFMA's inline zero operand
    const bool active = FAST ? true : wave < nt;
    const int qi = wave * 32 + c32;
    const bool qok = qi < N;
    const int qr = qi > 0 ? (qi - 1) / G.gw : 0, qc = qi > 0 ? (qi - 1) - qr * G.gw : 0;

    // ---- once per workgroup: tables, key slot masks, one-hot operands (the same for every (b, h)) -----------
    // The workgroup is PERSISTENT: it walks (b, h) items blockIdx.x, + gridDim.x, ... so that this setup — a
    // quarter of a one-item workgroup's time, all CUs bursting on the tables at once — is paid once per CU.
    {
        TabRegs rv, rk;
        tab_load(rv, a.tvv, a.tvh, a.ldt, a.nb);
        if constexpr (tables_in_lds<T>()) tab_load(rk, a.tkv, a.tkh, a.ldt, a.nb);
        tab_store_T<T>(rv, tvt);
        if constexpr (tables_in_lds<T>()) tab_store_R<T>(rk, tkr);
    }
    for (int j = threadIdx.x; j < NP; j += blockDim.x) masks[j] = key_mask(j, G);
    if constexpr (FAST) {
        __syncthreads();
        fill_onehot(ohr, oht, masks, NP);
    }

    for (int item = blockIdx.x; item < a.nitems; item += gridDim.x) {
    const int b = item / a.H, h = item - b * a.H;
    const int64_t base = (int64_t)b * a.sb + (int64_t)h * a.sh;
    const E* qp = reinterpret_cast<const E*>(a.q) + base;
    const E* kpg = reinterpret_cast<const E*>(a.k) + base;
    const E* vpg = reinterpret_cast<const E*>(a.v) + base;

    PROF_DECL
    PROF_MARK();
    F qb[S64];
    load_row<T>(qb, qp + (int64_t)min(qi, N - 1) * a.sn, g);
    // K tiles 0..nt-1 and then V tiles 0..nt-1 form ONE stream of 2*nt staged tiles: tile u is
    // loaded into register set u % PF a full PF tile-iterations before it is written to LDS buffer
    // u & 1 (one iteration of ~1.3k cycles is shorter than an HBM round trip under load: with
    // a single tile in flight every iteration ended in a wait for memory)
    // ONLINE (row blocks of 5+ key tiles): the tiles alternate K_0 V_0 K_1 V_1 ... and the softmax runs online, tile by
    // tile, against a running maximum — the whole score row block (NT x 16 registers beside 48 of accumulators, the
    // operand fragments and the tiles in flight) does not fit the 256 registers of a wave at two waves per SIMD: the
    // exact-softmax form of this kernel kept 690-1050 bytes per lane in scratch at NT = 7 / 8.
    constexpr bool ONLINE = NT > 4;
    // (ONLINE: K tiles travel in ring[0], V tiles in ring[1] — static register indices under a loop that is NOT unrolled)
    constexpr int PF = ONLINE ? 2 : (sizeof(E) == 2 ? 3 : 1);      // exact form in fp32: tiles are twice the registers, one in flight
    TileRegs<T, 32, 64> ring[PF];
    auto issue = [&](int u) {
        if constexpr (ONLINE) {
            if (u < 2 * nt) {
                if (u & 1) tile_load<T, 32, 64>(ring[PF - 1], vpg, a.sn, (u >> 1) * 32, N);
                else tile_load<T, 32, 64>(ring[0], kpg, a.sn, (u >> 1) * 32, N);
            }
        } else {
        if (u < nt) tile_load<T, 32, 64>(ring[u % PF], kpg, a.sn, u * 32, N);
        else if (u < 2 * nt) tile_load<T, 32, 64>(ring[u % PF], vpg, a.sn, (u - nt) * 32, N);
        }
    };
    auto commit = [&](int u) {
        E* dst = (u & 1) ? buf1 : buf0;
        const bool is_k = ONLINE ? !(u & 1) : u < nt;
        if (u >= 2 * nt) return;
        if constexpr (ONLINE) {
            if (is_k) tile_store<T, 32, 64, true, false>(ring[0], buf0, kp, nullptr, 0);
            else if constexpr (sizeof(E) == 2) tile_store<T, 32, 64, true, false>(ring[PF - 1], buf1, kp, nullptr, 0);
            else tile_store<T, 32, 64, false, true>(ring[PF - 1], nullptr, 0, buf1, vp);
            return;
        }
        if (is_k) tile_store<T, 32, 64, true, false>(ring[u % PF], dst, kp, nullptr, 0);
        else {
            // V tiles: bf16 keeps them row-major (the P.V product reads them through ds_read_b64_tr_b16)
            if constexpr (sizeof(E) == 2) tile_store<T, 32, 64, true, false>(ring[u % PF], dst, kp, nullptr, 0);
            else tile_store<T, 32, 64, false, true>(ring[u % PF], nullptr, 0, dst, vp);
        }
    };
#pragma unroll
    for (int u = 0; u < PF; ++u) issue(u);

    // ---- item prologue: first K tile (the barrier also separates this item's staging and scratch use from the
    // previous item's last tile and epilogue, and publishes the once-per-workgroup setup) ---------------------
    __syncthreads();
    commit(0);
    issue(PF);
    __syncthreads();
    PROF_MARK();

    // ---- this wave's query tile: bucket lookups -> slot extension ---------------------------
    float* scr = scratch + wave * 32 * LP;
    if (!qok) zero_frags<T, S64>(qb);
    F qe[S32];
    if (active) {
        table_lookups<T>(scr, qb, tkr, a.tkv, a.tkh, a.ldt, a.nb, lane);
        wave_lds_fence();
        if constexpr (FAST) build_ext14(qe, scr, lane, wave == 0, qr, qc);
        else build_ext<T>(qe, scr, lane, qi, qr, qc, G);
    }
    PROF_MARK();

    f32x16 o[2] = {f32x16{}, f32x16{}};
    f32x16 ox = {};
    float inv_l;
    if constexpr (ONLINE) {
    // ---- one pass over (K_t, V_t): scores of 32 keys, online softmax, [O | slot sums]^T += [V | one-hot]^T . P^T --------
    const float sc = a.scale * LOG2E;
    float m_run = -INFINITY, l = 0.f;                  // running maximum of the raw scores / this lane's half of the running sum
#pragma unroll 1
    for (int t = 0; t < nt; ++t)
        {
            f32x16 st = {};
            if (active) {
#pragma unroll
                for (int ks = 0; ks < S64; ++ks)
                    st = TT::mma(TT::load(buf0 + c32 * kp + ks * KI + g * EPL), qb[ks], st);
                const uint32_t km = masks[t * 32 + c32];
#pragma unroll
                for (int ks = 0; ks < S32; ++ks) st = TT::mma(TT::onehot_row(km, ks, g), qe[ks], st);
            }
            commit(2 * t + 1);                         // V_t
            __syncthreads();
            issue(2 * t + 1 + PF);
#pragma unroll
            for (int r = 0; r < 16; ++r)               // keys >= N (they exist only in the last tile)
                if (t * 32 + acc_row(r, g) >= N) st[r] = -INFINITY;
            float m4[4] = {st[0], st[1], st[2], st[3]};
#pragma unroll
            for (int r = 4; r < 16; ++r) m4[r & 3] = fmaxf(m4[r & 3], st[r]);
            float mt = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
            mt = fmaxf(mt, __shfl_xor(mt, 32));
            const float m_new = fmaxf(m_run, mt);      // (every tile holds a key < N: finite from the first tile on)
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sc);      // first tile: exp2(-inf) = 0 on zeros
            const float msc = m_new * sc;
            m_run = m_new;
            float l4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], sc, -msc));
                st[r] = p;
                l4[r & 3] += p;
            }
            l = __builtin_fmaf(l, alpha, (l4[0] + l4[1]) + (l4[2] + l4[3]));
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; ox[r] *= alpha; }
            if constexpr (DROP) {                      // the normaliser is the undropped sum
                const uint32_t dkey = drop_key(a.drop_seed, item);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    st[r] = drop_keep(dkey, qi, t * 32 + acc_row(r, g), a.drop_thr) ? st[r] * a.drop_scale : 0.f;
            }
            if (active) {
#pragma unroll
                for (int st2 = 0; st2 < S32; ++st2) {
                    const F pb = TT::from_acc(st, st2);
                    o[0] = TT::mma(perm_operand<T>(buf1, kp, buf1, vp, 0, st2, lane), pb, o[0]);
                    o[1] = TT::mma(perm_operand<T>(buf1, kp, buf1, vp, 1, st2, lane), pb, o[1]);
                    ox = TT::mma(TT::onehot_perm(masks + t * 32, st2, g, c32), pb, ox);
                }
            }
            if (t + 1 < nt) {
                commit(2 * t + 2);                     // K_{t+1}
                __syncthreads();
                issue(2 * t + 2 + PF);
            }
        }
    l += __shfl_xor(l, 32);
    inv_l = 1.f / l;
    if (active && qok && g == 0)
        a.lse[((int64_t)b * a.H + h) * N + qi] = (m_run * sc + log2f(l)) * (1.f / LOG2E);
    PROF_MARK();
    } else {
    // ---- S^T tiles: scores of all keys against this wave's 32 queries (K streamed) ------
    f32x16 s[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        s[t] = f32x16{};
        if (t < nt) {
            const E* kb = (t & 1) ? buf1 : buf0;
            if (active) {
#pragma unroll
                for (int ks = 0; ks < S64; ++ks)
                    s[t] = TT::mma(TT::load(kb + c32 * kp + ks * KI + g * EPL), qb[ks], s[t]);
                if constexpr (FAST) {
#pragma unroll
                    for (int ks = 0; ks < S32; ++ks)
                        s[t] = TT::mma(TT::load(ohr + (t * 32 + c32) * OHP + ks * KI + g * EPL), qe[ks], s[t]);
                } else {
                    const uint32_t km = masks[t * 32 + c32];
#pragma unroll
                    for (int ks = 0; ks < S32; ++ks) s[t] = TT::mma(TT::onehot_row(km, ks, g), qe[ks], s[t]);
                }
            }
            commit(t + 1);                             // the last iteration commits V^T tile 0
            __syncthreads();
            issue(t + 1 + PF);
        }
    }
    PROF_MARK();

    // ---- softmax over keys (in-lane + one exchange with the partner lane) ---------------
    // keys >= N exist only in the last tile
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (t == nt - 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (t * 32 + acc_row(r, g) >= N) s[t][r] = -INFINITY;
        }
    // four independent max / sum chains (a single chain of 112 dependent ops is latency-bound)
    float m4[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (t < nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) m4[r & 3] = fmaxf(m4[r & 3], s[t][r]);
        }
    float m = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
    m = fmaxf(m, __shfl_xor(m, 32));
    const float sc = a.scale * LOG2E;
    const float msc = m * sc;
    float l4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (t < nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][r], sc, -msc));
                s[t][r] = p;
                l4[r & 3] += p;
            }
        }
    float l = (l4[0] + l4[1]) + (l4[2] + l4[3]);
    l += __shfl_xor(l, 32);
    inv_l = 1.f / l;
    if (active && qok && g == 0)
        a.lse[((int64_t)b * a.H + h) * N + qi] = (msc + log2f(l)) * (1.f / LOG2E);
    if constexpr (DROP) {                          // the normaliser above is the undropped sum; V product and slot sums take the dropped map
        const uint32_t dkey = drop_key(a.drop_seed, item);
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (t < nt) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    s[t][r] = drop_keep(dkey, qi, t * 32 + acc_row(r, g), a.drop_thr) ? s[t][r] * a.drop_scale : 0.f;
            }
    }
    PROF_MARK();

    // ---- [O | slot sums]^T = [V | one-hot]^T . P^T   (V^T streamed) -------------------------
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (t < nt) {
            const E* vb = ((nt + t) & 1) ? buf1 : buf0;
            if (active) {
#pragma unroll
                for (int st2 = 0; st2 < S32; ++st2) {
                    const F pb = TT::from_acc(s[t], st2);
                    o[0] = TT::mma(perm_operand<T>(vb, kp, vb, vp, 0, st2, lane), pb, o[0]);
                    o[1] = TT::mma(perm_operand<T>(vb, kp, vb, vp, 1, st2, lane), pb, o[1]);
                    if constexpr (FAST) ox = TT::mma(TT::load_perm(oht + c32 * otp + t * 32, st2, g), pb, ox);
                    else ox = TT::mma(TT::onehot_perm(masks + t * 32, st2, g, c32), pb, ox);
                }
            }
            if (t + 1 < nt) {
                commit(nt + t + 1);
                __syncthreads();
                issue(nt + t + 1 + PF);
            }
        }
    }   // exact softmax over the whole row block (NT <= 4)
    if (active) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] *= inv_l; o[1][r] *= inv_l; ox[r] *= inv_l; }
    PROF_MARK();

    // ---- value-side relative position term: slot sums -> bucket sums -> . tables ---------
    float bk[32];
    if constexpr (FAST) slots_to_buckets14(bk, scr, ox, lane, wave == 0, min(qr, G14 - 1), qc);
    else slots_to_buckets(bk, scr, ox, lane, qi, qr, qc, G);
    // S'^T (64 buckets x NP queries) for backward (dTvv / dTvh)
    if (a.sp) store_buckets_T<T>(reinterpret_cast<E*>(a.sp) + ((int64_t)b * a.H + h) * 64 * NP + qi, NP, bk, g);   // (NULL: no backward will follow)
    add_bucket_product<T>(o, tvt, bk, lane);
    PROF_MARK();

    // ---- store O (b, n, h, :) ----------------------------------------------------------------
    if (qok) store_rows_64<T>(reinterpret_cast<E*>(a.out) + (((int64_t)b * N + qi) * a.H + h) * 64, o, g);
    PROF_MARK();
    PROF_FLUSH();
    }   // active
    }   // items
}

// ---------------------------------------------------------------------------------------
// forward, AutoFormer geometry (N = 197, 14 x 14 grid, bf16) with K and V WHOLE in LDS
// ---------------------------------------------------------------------------------------
// The tile-streamed kernel above passes 15 barriers per (b, h) item with 6 MFMAs per wave between two of them: half of a
// wave's cycles were parked at barriers / s_waitcnt.  One item's K and V are only 25 KB each, so here
// they are committed to LDS in ONE piece: 4 barriers per item, and between them every wave runs its whole row block —
// 42 score MFMAs, the exact softmax in registers, 42 P.V / slot-sum MFMAs — without meeting anybody.
// LDS (147.7 KB, one workgroup per CU):  [K 224 x 72] [pad] [V 224 x 72] | tables | masks | one-hot operands.
// The per-wave fp32 shift scratch (7 x 9 KB) is never live together with both matrices: the lookup / slot-extension
// scratch of the prologue lies over [pad | V] (V is committed after the scores), the slot -> bucket scratch of the
// epilogue over [K | pad] (K is dead after the scores).  The next matrix is always in flight in registers: V during
// the scores, the next item's K during P.V.
constexpr int W14_NP = 224, W14_KP = 72, W14_THREADS = 448;
constexpr int W14_MAT = W14_NP * W14_KP * 2;                                    // one matrix in LDS (bytes)
constexpr int W14_SCR = 7 * 32 * LP * 4;                                        // shift scratch of the 7 waves
constexpr int W14_PAD = W14_SCR - W14_MAT;
static_assert(W14_PAD > 0 && W14_PAD % 16 == 0, "scratch = matrix + pad");
constexpr size_t fwd14_lds_bytes() {
    return (size_t)2 * W14_MAT + W14_PAD + (size_t)128 * table_pitch<hip_bfloat16>() * 2 + (size_t)W14_NP * 4 + onehot_bytes(W14_NP);
}

struct MatRegs { u32x4v v[4]; };                                                // 224 x 64 bf16 = 1792 chunks of 16 B / 448 threads
__device__ __forceinline__ void mat_load(MatRegs& r, const short* src, int64_t rs, int nrows) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = threadIdx.x + i * W14_THREADS, row = c >> 3, cc = c & 7;
        const u32x4v v = *reinterpret_cast<const u32x4v*>(src + (int64_t)min(row, nrows - 1) * rs + cc * 8);
        r.v[i] = row < nrows ? v : u32x4v{0, 0, 0, 0};
    }
}
__device__ __forceinline__ void mat_store(const MatRegs& r, short* dst) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = threadIdx.x + i * W14_THREADS, row = c >> 3, cc = c & 7;
        *reinterpret_cast<u32x4v*>(dst + row * W14_KP + cc * 8) = r.v[i];
    }
}

__global__ __launch_bounds__(W14_THREADS) void attn_rpe2d_fwd14_kernel(const FwdArgs a) {
    using T = hip_bfloat16;
    using TT = Tr<T>;
    using F = TT::frag;
    constexpr int NT = 7, N = 197, NP = W14_NP, kp = W14_KP, tp = table_pitch<T>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    short* kbuf = reinterpret_cast<short*>(smem);
    short* vbuf = reinterpret_cast<short*>(smem + W14_MAT + W14_PAD);
    float* scrA = reinterpret_cast<float*>(smem + W14_MAT);                     // [pad | V]: lookups -> slot extension
    float* scrB = reinterpret_cast<float*>(smem);                               // [K | pad]: slot sums -> bucket sums
    short* tvt = reinterpret_cast<short*>(smem + 2 * W14_MAT + W14_PAD);        // value tables^T [64][tp]
    short* tkr = tvt + 64 * tp;                                                 // key table rows [64][tp]
    uint32_t* masks = reinterpret_cast<uint32_t*>(tkr + 64 * tp);
    short* ohr = reinterpret_cast<short*>(masks + NP);
    short* oht = ohr + NP * OHP;
    constexpr int otp = oht_pitch(NP);
    const RelGeom G{N, G14, G14, G14};

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, g = lane >> 5, c32 = lane & 31;
    const int qi = wave * 32 + c32;
    const bool qok = qi < N;
    const int qr = qi > 0 ? (qi - 1) / G14 : 0, qc = qi > 0 ? (qi - 1) - qr * G14 : 0;

    // ---- once per (persistent) workgroup: tables, key slot masks, one-hot operands ----------------------------
    {
        TabRegs rv, rk;
        tab_load(rv, a.tvv, a.tvh, a.ldt, a.nb);
        tab_load(rk, a.tkv, a.tkh, a.ldt, a.nb);
        tab_store_T<T>(rv, tvt);
        tab_store_R<T>(rk, tkr);
    }
    for (int j = threadIdx.x; j < NP; j += blockDim.x) masks[j] = key_mask(j, G);
    __syncthreads();
    fill_onehot(ohr, oht, masks, NP);

    MatRegs mr;
    int item = blockIdx.x;
    if (item < a.nitems) {
        const int b = item / a.H, h = item - b * a.H;
        mat_load(mr, reinterpret_cast<const short*>(a.k) + (int64_t)b * a.sb + (int64_t)h * a.sh, a.sn, N);
    }
    for (; item < a.nitems; item += gridDim.x) {
        const int b = item / a.H, h = item - b * a.H;
        const int64_t base = (int64_t)b * a.sb + (int64_t)h * a.sh;
        const short* qp = reinterpret_cast<const short*>(a.q) + base;
        const short* vpg = reinterpret_cast<const short*>(a.v) + base;
        PROF_DECL
        PROF_MARK();
        F qb[4];
        load_row<T>(qb, qp + (int64_t)min(qi, N - 1) * a.sn, g);
        __syncthreads();                               // previous item: P.V reads of V and the [K | pad] scratch are over
        mat_store(mr, kbuf);
        mat_load(mr, vpg, a.sn, N);                    // V travels during the scores
        __syncthreads();
        PROF_MARK();

        // ---- this wave's query tile: bucket lookups -> slot extension (scratch over [pad | V]) -----------------
        float* scr = scrA + wave * 32 * LP;
        if (!qok) zero_frags<T, 4>(qb);
        F qe[2];
        table_lookups<T>(scr, qb, tkr, a.tkv, a.tkh, a.ldt, a.nb, lane);
        wave_lds_fence();
        build_ext14(qe, scr, lane, wave == 0, qr, qc);
        PROF_MARK();

        // ---- S^T: all keys against this wave's 32 queries, no barrier in between ------------------------------
        f32x16 s[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            s[t] = f32x16{};
            const short* kb = kbuf + t * 32 * kp;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s[t] = TT::mma(TT::load(kb + c32 * kp + ks * 16 + g * 8), qb[ks], s[t]);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) s[t] = TT::mma(TT::load(ohr + (t * 32 + c32) * OHP + ks * 16 + g * 8), qe[ks], s[t]);
        }
        PROF_MARK();
        // ---- softmax over keys (in-lane + one exchange with the partner lane); keys >= N only in the last tile ---
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if ((NT - 1) * 32 + acc_row(r, g) >= N) s[NT - 1][r] = -INFINITY;
        float m4[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) m4[r & 3] = fmaxf(m4[r & 3], s[t][r]);
        float m = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
        m = fmaxf(m, __shfl_xor(m, 32));
        const float sc = a.scale * LOG2E, msc = m * sc;
        float l4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][r], sc, -msc));
                s[t][r] = p;
                l4[r & 3] += p;
            }
        float l = (l4[0] + l4[1]) + (l4[2] + l4[3]);
        l += __shfl_xor(l, 32);
        const float inv_l = 1.f / l;
        if (qok && g == 0) a.lse[((int64_t)b * a.H + h) * N + qi] = (msc + log2f(l)) * (1.f / LOG2E);

        PROF_MARK();
        __syncthreads();                               // everybody is done with K and with the [pad | V] scratch
        mat_store(mr, vbuf);
        {
            const int nxt = item + gridDim.x;          // the next item's K travels during P.V
            if (nxt < a.nitems) {
                const int nb_ = nxt / a.H, nh = nxt - nb_ * a.H;
                mat_load(mr, reinterpret_cast<const short*>(a.k) + (int64_t)nb_ * a.sb + (int64_t)nh * a.sh, a.sn, N);
            }
        }
        __syncthreads();

        PROF_MARK();
        // ---- [O | slot sums]^T = [V | one-hot]^T . P^T ---------------------------------------------------------
        f32x16 o[2] = {f32x16{}, f32x16{}};
        f32x16 ox = {};
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const short* vb = vbuf + t * 32 * kp;
#pragma unroll
            for (int st2 = 0; st2 < 2; ++st2) {
                const F pb = TT::from_acc(s[t], st2);
                o[0] = TT::mma(load_perm_tr(vb, kp, 0, st2, lane), pb, o[0]);
                o[1] = TT::mma(load_perm_tr(vb, kp, 1, st2, lane), pb, o[1]);
                ox = TT::mma(TT::load_perm(oht + c32 * otp + t * 32, st2, g), pb, ox);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] *= inv_l; o[1][r] *= inv_l; ox[r] *= inv_l; }

        PROF_MARK();
        // ---- value-side relative position term: slot sums -> bucket sums -> . tables (scratch over [K | pad]) ---
        float bk[32];
        slots_to_buckets14(bk, scrB + wave * 32 * LP, ox, lane, wave == 0, min(qr, G14 - 1), qc);
        if (a.sp) store_buckets_T<T>(reinterpret_cast<short*>(a.sp) + ((int64_t)b * a.H + h) * 64 * NP + qi, NP, bk, g);
        add_bucket_product<T>(o, tvt, bk, lane);
        PROF_MARK();
        if (qok) store_rows_64<T>(reinterpret_cast<short*>(a.out) + (((int64_t)b * N + qi) * a.H + h) * 64, o, g);
        PROF_MARK();
        PROF_FLUSH();
    }
}

// one persistent workgroup per CU (the forward's LDS footprint allows exactly one)
int fwd_persistent_grid() { return cream::cu_count(); }

bool fast_geometry(const RelGeom& G) { return G.n == 197 && G.gh == G14 && G.gw == G14 && G.mr == G14; }

template <typename T, int NT, bool FAST = false, bool DROP = false>
int launch_fwd_nt(const FwdArgs& a, int B, hipStream_t st) {
    const int waves = a.NP / 32 < 4 ? 4 : a.NP / 32;
    const size_t lds = fwd_lds_bytes<T>(a.NP, waves, FAST);
    auto kern = attn_rpe2d_fwd_kernel<T, NT, FAST, DROP>;
    if (!cream::raise_dynamic_lds(kern, (int)(160 * 1024))) return CREAM_ERR_LAUNCH;
    FwdArgs aa = a;
    aa.nitems = B * a.H;
    const int grid = aa.nitems < fwd_persistent_grid() ? aa.nitems : fwd_persistent_grid();
    CREAM_LAUNCH(kern, dim3(grid), dim3(waves * 64), lds, st, aa);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

#include "attn_rpe2d_bwd1.hpp"
#include "attn_rpe2d_bwd2.hpp"
#include "attn_rpe2d_fwd2.hpp"

// 1: the ping-pong online-softmax forward (attn_rpe2d_fwd2.hpp) whenever the caller hands over the table images
// (AutoFormer geometry, bf16); 0: attn_rpe2d_fwd14.  CREAM_ATTN_FWD2 in the environment sets the initial value.
std::atomic<int> g_fwd2{-1};
int fwd2_mode() {
    int m = g_fwd2.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = getenv("CREAM_ATTN_FWD2");
        m = e ? (atoi(e) != 0) : 1;
        g_fwd2.store(m, std::memory_order_relaxed);
    }
    return m;
}

int fwd_persistent_grid();
int launch_fwd2(const FwdArgs& a, int B, hipStream_t st) {
    if (!cream::raise_dynamic_lds(v3::attn_rpe2d_fwd2_kernel, (int)(160 * 1024))) return CREAM_ERR_LAUNCH;
    FwdArgs aa = a;
    aa.nitems = B * a.H;
    const int grid = aa.nitems < fwd_persistent_grid() ? aa.nitems : fwd_persistent_grid();
    CREAM_LAUNCH(v3::attn_rpe2d_fwd2_kernel, dim3(grid), dim3(v3::F2_THREADS), v3::F2_LDS_B, st, aa, reinterpret_cast<const short*>(a.timg));
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

int launch_fwd14(const FwdArgs& a, int B, hipStream_t st) {
    if (!cream::raise_dynamic_lds(attn_rpe2d_fwd14_kernel, (int)(160 * 1024))) return CREAM_ERR_LAUNCH;
    FwdArgs aa = a;
    aa.nitems = B * a.H;
    const int grid = aa.nitems < fwd_persistent_grid() ? aa.nitems : fwd_persistent_grid();
    CREAM_LAUNCH(attn_rpe2d_fwd14_kernel, dim3(grid), dim3(W14_THREADS), fwd14_lds_bytes(), st, aa);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

template <typename T>
int launch_fwd(const FwdArgs& a, int B, hipStream_t st) {
    const int nt = a.NP / 32;
    if (a.drop_thr) {        // attention dropout: the tile-streamed kernel for every geometry (the score row block is in registers when the mask applies)
        if (nt <= 2) return launch_fwd_nt<T, 2, false, true>(a, B, st);
        if (nt <= 4) return launch_fwd_nt<T, 4, false, true>(a, B, st);
        if (nt <= 7) return launch_fwd_nt<T, 7, false, true>(a, B, st);
        return launch_fwd_nt<T, 8, false, true>(a, B, st);
    }
    if (nt <= 2) return launch_fwd_nt<T, 2>(a, B, st);
    if (nt <= 4) return launch_fwd_nt<T, 4>(a, B, st);
    if constexpr (sizeof(typename Tr<T>::elem) == 2) {
        // (the tile-streamed kernel's own FAST instantiation, launch_fwd_nt<T, 7, true>: 46.5 us against 40.6 us at B = 128, H = 6;
        //  10.56 vs 10.48 ms per step in a same-box A/B x3)
        if (fast_geometry(a.G)) return a.timg && fwd2_mode() ? launch_fwd2(a, B, st) : launch_fwd14(a, B, st);
    }
    if (nt <= 7) return launch_fwd_nt<T, 7>(a, B, st);
    return launch_fwd_nt<T, 8>(a, B, st);
}

// ---------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------
// With Qx = [Q | X] (X = per-query slot extension of the key-side bucket lookups), Kx = [K | E],
// Vx = [V | E] (E = one-hot slots of the keys), S = scale Qx Kx^T, P = softmax(S),
// [O_pv | SA] = P Vx, O = O_pv + shift(SA) [Tvv; Tvh]:
//     dOx = [dO | gather(dO [Tvv;Tvh]^T)]       dP = dOx Vx^T        delta_i = dO_i . O_i
//     dS = scale P o (dP - delta)               dQx = dS Kx          dK = dS^T Q     dV = P^T dO
//     dQ = dQx[:, :64] + shift(dQx[:, 64:]) [Tkv; Tkh]
//     d[Tkv;Tkh] = shift(dQx[:, 64:])^T Q       d[Tvv;Tvh] = shift(SA)^T dO
// Two launches, both one workgroup per (b,h), no atomics, fixed summation order:
//   A  wave = query tile (lanes own queries): dQ, and the side buffers the second launch needs
//      (slot extensions of Q and dO, delta, the shifted bucket gradients dL'^T)
//   B  wave = key tile (lanes own keys): dK, dV (contraction over queries, so Q^T and dO^T
//      live in LDS), then the four table gradients of this (b,h) as eight 32x32 MFMA jobs.
// LDS of the dQ kernel: 2 x [K tile | V tile | K^T tile] | key tables^T | table rows (bf16) | masks | scratch
template <typename T> size_t bwd_q_lds_bytes(int NP, int waves, bool fast) {
    using E = typename Tr<T>::elem;
    return 2 * (2 * rm_tile_bytes<T>(64) + t_tile_bytes<T>()) + (size_t)(64 + tabr_rows<T>(4)) * table_pitch<T>() * sizeof(E) +
           (size_t)NP * 4 + (size_t)waves * 32 * LP * 4 + (fast ? onehot_bytes(NP) : 0);
}

template <typename T, bool FAST, bool DROP = false>
__global__ __launch_bounds__(512) void attn_rpe2d_bwd_q_kernel(const BwdArgs a) {
    using TT = Tr<T>;
    using E = typename TT::elem;
    using F = typename TT::frag;
    constexpr int KI = TT::KI, EPL = TT::EPL, S64 = 64 / KI, S32 = 32 / KI;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    static_assert(!FAST || sizeof(E) == 2, "fast path: bf16");
    const RelGeom G = FAST ? RelGeom{197, G14, G14, G14} : a.G;
    const int N = G.n, NP = FAST ? 224 : a.NP;
    const int nt = NP >> 5;
    constexpr int tp = table_pitch<T>(), rp = rm_pitch<T>(64), ktp = t_pitch<T>();
    constexpr size_t set_b = 2 * rm_tile_bytes<T>(64) + t_tile_bytes<T>();
    // stage set i: K rows [32][rp] | V rows [32][rp] | K^T [64][ktp]
    auto kbuf = [&](int i) { return reinterpret_cast<E*>(smem + i * set_b); };
    auto vbuf = [&](int i) { return reinterpret_cast<E*>(smem + i * set_b + rm_tile_bytes<T>(64)); };
    auto ktbuf = [&](int i) { return reinterpret_cast<E*>(smem + i * set_b + 2 * rm_tile_bytes<T>(64)); };
    E* tkt = reinterpret_cast<E*>(smem + 2 * set_b);                      // key tables^T [64][tp]
    E* tkr = tkt + 64 * tp;                                               // key table rows, then value table rows (bf16)
    E* tvr = tkr + tabr_rows<T>(2) * tp;
    uint32_t* masks = reinterpret_cast<uint32_t*>(tkr + tabr_rows<T>(4) * tp);
    float* scratch = reinterpret_cast<float*>(masks + NP);
    short* ohr = reinterpret_cast<short*>(scratch + nt * 32 * LP);   // fast path: one-hot operands (scratch: one slab per query tile)
    short* oht = ohr + NP * OHP;
    const int otp = oht_pitch(NP);

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, g = lane >> 5, c32 = lane & 31;
    const bool active = wave < nt;
    const int qi = wave * 32 + c32;
    const bool qok = qi < N;
    const int qcl = min(qi, N - 1);
    const int qr = qi > 0 ? (qi - 1) / G.gw : 0, qc = qi > 0 ? (qi - 1) - qr * G.gw : 0;
    const int grp = wave >> 2, gt = threadIdx.x & 255;
    // ---- once per (persistent) workgroup: tables, key slot masks, one-hot operands ----------------------------
    {
        TabRegs rk, rv;
        tab_load(rk, a.tkv, a.tkh, a.ldt, a.nb);
        if constexpr (tables_in_lds<T>()) tab_load(rv, a.tvv, a.tvh, a.ldt, a.nb);
        tab_store_T<T>(rk, tkt);
        if constexpr (tables_in_lds<T>()) {
            tab_store_R<T>(rk, tkr);
            tab_store_R<T>(rv, tvr);
        }
    }
    for (int j = threadIdx.x; j < NP; j += blockDim.x) masks[j] = key_mask(j, G);
    if constexpr (FAST) {
        __syncthreads();
        fill_onehot(ohr, oht, masks, NP);
    }
    for (int item = blockIdx.x; item < a.nitems; item += gridDim.x) {
    __syncthreads();                                  // previous item done with the staged tiles / setup published
    const int b = item / a.H, h = item - b * a.H;
    const int64_t bh = (int64_t)b * a.H + h;
    const int64_t base = (int64_t)b * a.sb + (int64_t)h * a.sh;
    const E* qp = reinterpret_cast<const E*>(a.q) + base;
    const E* kpg = reinterpret_cast<const E*>(a.k) + base;
    const E* vpg = reinterpret_cast<const E*>(a.v) + base;
    const int64_t orow = (int64_t)a.H * 64;                               // token stride of out / dout
    const E* dop = reinterpret_cast<const E*>(a.dout) + ((int64_t)b * N * a.H + h) * 64;
    const E* outp = reinterpret_cast<const E*>(a.out) + ((int64_t)b * N * a.H + h) * 64;

    PROF_DECL
    PROF_MARK();
    F qb[S64], dob[S64];
    float delta = 0.f;
    TileRegs<T, 32, 64> sk;
    {
        F ob[S64];
        load_row<T>(qb, qp + (int64_t)qcl * a.sn, g);
        load_row<T>(dob, dop + (int64_t)qcl * orow, g);
        load_row<T>(ob, outp + (int64_t)qcl * orow, g);
        // the workgroup always has 8 waves: waves 0-3 stage the K tiles (rows + transpose), waves
        // 4-7 the V tiles — one chunk per thread, wave-uniform roles (waves >= nt only stage)
        if (grp == 0) tile_load_g<T, 32, 64>(sk, kpg, a.sn, 0, N, 0, gt);
        else tile_load_g<T, 32, 64>(sk, vpg, a.sn, 0, N, 0, gt);


        // delta_i = dO_i . O_i  (this lane holds half of the 64 d-values; the partner the rest)
        if (!qok) { zero_frags<T, S64>(qb); zero_frags<T, S64>(dob); }
#pragma unroll
        for (int ks = 0; ks < S64; ++ks) {
            if constexpr (EPL == 1) delta += dob[ks] * ob[ks];
            else {
#pragma unroll
                for (int e = 0; e < EPL; ++e) delta += TT::to_f(dob[ks][e]) * TT::to_f(ob[ks][e]);
            }
        }
        delta += __shfl_xor(delta, 32);
    }
    if (active && g == 0) a.delta[bh * NP + qi] = delta;
    if (grp == 0) tile_store_g<T, 32, 64, true, sizeof(E) != 2>(sk, kbuf(0), rp, ktbuf(0), ktp, gt);      // bf16: K^T is read with tr16
    else tile_store_g<T, 32, 64, true, false>(sk, vbuf(0), rp, nullptr, 0, gt);
    __syncthreads();                                  // tables, masks, first tiles in place
    PROF_MARK();

    float* scr = scratch + wave * 32 * LP;
    F qe[S32], de[S32];
    if (active) {
        table_lookups<T>(scr, qb, tkr, a.tkv, a.tkh, a.ldt, a.nb, lane);
        wave_lds_fence();
        if constexpr (FAST) build_ext14(qe, scr, lane, wave == 0, qr, qc);
        else build_ext<T>(qe, scr, lane, qi, qr, qc, G);
        wave_lds_fence();
        table_lookups<T>(scr, dob, tvr, a.tvv, a.tvh, a.ldt, a.nb, lane);
        wave_lds_fence();
        if constexpr (FAST) build_ext14(de, scr, lane, wave == 0, qr, qc);
        else build_ext<T>(de, scr, lane, qi, qr, qc, G);
        wave_lds_fence();
        store_ext_rows<T>(reinterpret_cast<E*>(a.qe) + (bh * NP + qi) * 32, qe, g);
        store_ext_rows<T>(reinterpret_cast<E*>(a.de) + (bh * NP + qi) * 32, de, g);
    }
    PROF_MARK();

    const float sc = a.scale * LOG2E;
    const float m2 = (active && qok) ? a.lse[bh * N + qi] * LOG2E : 0.f;

    f32x16 dq[2] = {f32x16{}, f32x16{}};
    f32x16 dx = {};
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) {
            if (grp == 0) tile_load_g<T, 32, 64>(sk, kpg, a.sn, (t + 1) * 32, N, 0, gt);
            else tile_load_g<T, 32, 64>(sk, vpg, a.sn, (t + 1) * 32, N, 0, gt);
        }
        if (active) {
            const E* kb = kbuf(cur) + c32 * rp;
            const E* vb = vbuf(cur) + c32 * rp;
            f32x16 sacc = {}, pacc = {};
#pragma unroll
            for (int ks = 0; ks < S64; ++ks) {
                sacc = TT::mma(TT::load(kb + ks * KI + g * EPL), qb[ks], sacc);
                pacc = TT::mma(TT::load(vb + ks * KI + g * EPL), dob[ks], pacc);
            }
            uint32_t km = 0;
            if constexpr (!FAST) km = masks[t * 32 + c32];
#pragma unroll
            for (int ks = 0; ks < S32; ++ks) {
                F oh;
                if constexpr (FAST) oh = TT::load(ohr + (t * 32 + c32) * OHP + ks * KI + g * EPL);
                else oh = TT::onehot_row(km, ks, g);
                sacc = TT::mma(oh, qe[ks], sacc);
                pacc = TT::mma(oh, de[ks], pacc);
            }
            // dS^T = scale * P o (dP - delta), keys beyond N contribute nothing
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool ok = t * 32 + acc_row(r, g) < N;
                const float p = ok ? __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[r], sc, -m2)) : 0.f;
                if constexpr (DROP)                   // dP = keep / (1 - p) o (dO Vx^T): the mask sits on dP, P stays undropped
                    pacc[r] = drop_keep(drop_key(a.drop_seed, item), qi, t * 32 + acc_row(r, g), a.drop_thr) ? pacc[r] * a.drop_scale : 0.f;
                sacc[r] = p * (pacc[r] - delta) * a.scale;
            }
            const E* ktb = ktbuf(cur);
#pragma unroll
            for (int st = 0; st < S32; ++st) {
                const F db = TT::from_acc(sacc, st);
                dq[0] = TT::mma(perm_operand<T>(kbuf(cur), rp, ktb, ktp, 0, st, lane), db, dq[0]);
                dq[1] = TT::mma(perm_operand<T>(kbuf(cur), rp, ktb, ktp, 1, st, lane), db, dq[1]);
                if constexpr (FAST) dx = TT::mma(TT::load_perm(oht + c32 * otp + t * 32, st, g), db, dx);
                else dx = TT::mma(TT::onehot_perm(masks + t * 32, st, g, c32), db, dx);
            }
        }
        if (t + 1 < nt) {
            if (grp == 0) tile_store_g<T, 32, 64, true, sizeof(E) != 2>(sk, kbuf(cur ^ 1), rp, ktbuf(cur ^ 1), ktp, gt);
            else tile_store_g<T, 32, 64, true, false>(sk, vbuf(cur ^ 1), rp, nullptr, 0, gt);
            __syncthreads();
        }
    }
    PROF_MARK();
    if (active) {

    float bk[32];
    if constexpr (FAST) slots_to_buckets14(bk, scr, dx, lane, wave == 0, min(qr, G14 - 1), qc);
    else slots_to_buckets(bk, scr, dx, lane, qi, qr, qc, G);
    // dL'^T (64 buckets x NP queries) for the table gradients of launch B
    store_buckets_T<T>(reinterpret_cast<E*>(a.dlt) + bh * 64 * NP + qi, NP, bk, g);
    add_bucket_product<T>(dq, tkt, bk, lane);
    if (qok)
        store_rows_64<T>(reinterpret_cast<E*>(a.dq) + (int64_t)b * a.dsb + (int64_t)qi * a.dsn + (int64_t)h * a.dsh,
                         dq, g);
    PROF_MARK();
    PROF_FLUSH();
    }   // active
    }   // items
}

// LDS of the dK/dV kernel: 2 x [Q | dO rows, Q^T | dO^T, qe | de rows, dL'^T | S'^T tiles] | lse2 | delta
template <typename T> __host__ __device__ constexpr size_t kv_set_bytes() {
    return 2 * rm_tile_bytes<T>(64) + 2 * t_tile_bytes<T>() + 2 * rm_tile_bytes<T>(32) + 2 * t_tile_bytes<T>();
}
template <typename T> size_t bwd_kv_lds_bytes(int NP) { return 2 * kv_set_bytes<T>() + (size_t)NP * 8; }

template <typename T, bool DROP = false>
__global__ __launch_bounds__(512) void attn_rpe2d_bwd_kv_kernel(const BwdArgs a) {
    using TT = Tr<T>;
    using E = typename TT::elem;
    using F = typename TT::frag;
    constexpr int KI = TT::KI, EPL = TT::EPL, S64 = 64 / KI, S32 = 32 / KI;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const RelGeom G = a.G;
    const int N = G.n, NP = a.NP;
    const int nt = NP >> 5;
    constexpr int rp = rm_pitch<T>(64), ep = rm_pitch<T>(32), tpt = t_pitch<T>();
    constexpr size_t RB = rm_tile_bytes<T>(64), TB = t_tile_bytes<T>(), EB = rm_tile_bytes<T>(32), SET = kv_set_bytes<T>();
    // stage set i: Q rows | dO rows | Q^T | dO^T | qe rows | de rows | dL'^T tile [64 u][tpt] | S'^T tile
    auto qbuf = [&](int i) { return reinterpret_cast<E*>(smem + i * SET); };
    auto dbuf = [&](int i) { return reinterpret_cast<E*>(smem + i * SET + RB); };
    auto qtbuf = [&](int i) { return reinterpret_cast<E*>(smem + i * SET + 2 * RB); };
    auto dtbuf = [&](int i) { return reinterpret_cast<E*>(smem + i * SET + 2 * RB + TB); };
    auto qebuf = [&](int i) { return reinterpret_cast<E*>(smem + i * SET + 2 * RB + 2 * TB); };
    auto debuf = [&](int i) { return reinterpret_cast<E*>(smem + i * SET + 2 * RB + 2 * TB + EB); };
    auto dlbuf = [&](int i) { return reinterpret_cast<E*>(smem + i * SET + 2 * RB + 2 * TB + 2 * EB); };
    auto spbuf = [&](int i) { return reinterpret_cast<E*>(smem + i * SET + 2 * RB + 3 * TB + 2 * EB); };
    float* lse2 = reinterpret_cast<float*>(smem + 2 * SET);               // [NP]
    float* dlt_s = lse2 + NP;                                             // delta [NP]

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int lane = threadIdx.x & 63, g = lane >> 5, c32 = lane & 31;
    // persistent workgroup: (b, h) items blockIdx.x, + gridDim.x, ...  The item-specific operands of the NEXT item (own
    // K / V rows, first staged tile set, lse / delta) are requested into registers before the current item's epilogue
    // (dK / dV and table-gradient stores), so an item starts with its prologue loads already landed: the prologue was
    // 21 % of an item (phase stamps of tools/probes/attn_probe).
    struct ItemPtrs {
        const E *qp, *kpg, *vpg, *dop, *qep, *dep, *dltp, *spp;
        int64_t bh;
        int b, h;
    };
    const int64_t orow = (int64_t)a.H * 64;
    auto item_ptrs = [&](int item) {
        ItemPtrs P;
        P.b = item / a.H;
        P.h = item - P.b * a.H;
        P.bh = (int64_t)P.b * a.H + P.h;
        const int64_t base = (int64_t)P.b * a.sb + (int64_t)P.h * a.sh;
        P.qp = reinterpret_cast<const E*>(a.q) + base;
        P.kpg = reinterpret_cast<const E*>(a.k) + base;
        P.vpg = reinterpret_cast<const E*>(a.v) + base;
        P.dop = reinterpret_cast<const E*>(a.dout) + ((int64_t)P.b * N * a.H + P.h) * 64;
        P.qep = reinterpret_cast<const E*>(a.qe) + P.bh * NP * 32;
        P.dep = reinterpret_cast<const E*>(a.de) + P.bh * NP * 32;
        P.dltp = reinterpret_cast<const E*>(a.dlt) + P.bh * 64 * NP;
        P.spp = reinterpret_cast<const E*>(a.sp) + P.bh * 64 * NP;
        return P;
    };
    const bool active = wave < nt;
    const int kj = wave * 32 + c32;
    const bool kok = active && kj < N;

    // staging is split over two groups of four waves (the workgroup always has 8 waves; waves
    // >= nt only stage): group 0 moves the Q tile (+ its transpose), the qe | de rows and the S'^T
    // tile, group 1 the dO tile (+ transpose) and the dL'^T tile — wave-uniform roles, every
    // thread of a group has exactly one chunk per tile
    const int grp = wave >> 2, gt = threadIdx.x & 255;
    TileRegs<T, 32, 64> sq;                                  // group 0: Q rows      group 1: dO rows
    TileRegs<T, 64, 32> sdl;                                 // group 0: S'^T tile   group 1: dL'^T tile
    constexpr int EH = TileRegs<T, 32, 32>::CH;              // chunks of one 32 x 32 extension tile
    TileRegs<T, 32, 32> sqe;                                 // group 0 only: qe (first half of the group) | de
    auto load_set = [&](const ItemPtrs& P, int t) {
        if (grp == 0) {
            tile_load_g<T, 32, 64>(sq, P.qp, a.sn, t * 32, N, 0, gt);
            tile_load_g<T, 64, 32>(sdl, P.spp, NP, 0, 64, t * 32, gt);
            if constexpr (EH == 128) {                       // bf16: 128 chunks each -> half a group per tile
                const int c = gt & 127, row = c >> 2, cc = c & 3;
                const E* src = (gt < 128 ? P.qep : P.dep) + (int64_t)(t * 32 + row) * 32 + cc * 8;
                sqe.v[0] = *reinterpret_cast<const u32x4v*>(src);
            } else {                                         // fp32: 256 chunks each -> two per thread
                tile_load_g<T, 32, 32>(sqe, P.qep, 32, t * 32, NP, 0, gt);
            }
        } else {
            tile_load_g<T, 32, 64>(sq, P.dop, orow, t * 32, N, 0, gt);
            tile_load_g<T, 64, 32>(sdl, P.dltp, NP, 0, 64, t * 32, gt);
        }
    };
    TileRegs<T, 32, 32> sde;                                 // fp32 only: de rows (group 0)
    auto load_set_extra = [&](const ItemPtrs& P, int t) {
        if constexpr (EH != 128) { if (grp == 0) tile_load_g<T, 32, 32>(sde, P.dep, 32, t * 32, NP, 0, gt); }
    };
    auto store_set = [&](int i) {
        if (grp == 0) {
            tile_store_g<T, 32, 64, true, sizeof(E) != 2>(sq, qbuf(i), rp, qtbuf(i), tpt, gt);     // bf16: Q^T / dO^T are read with tr16
            tile_store_g<T, 64, 32, true, false>(sdl, spbuf(i), tpt, nullptr, 0, gt);
            if constexpr (EH == 128) {
                const int c = gt & 127, row = c >> 2, cc = c & 3;
                E* d = (gt < 128 ? qebuf(i) : debuf(i)) + row * ep + cc * 8;
                *reinterpret_cast<u32x4v*>(d) = sqe.v[0];
            } else {
                tile_store_g<T, 32, 32, true, false>(sqe, qebuf(i), ep, nullptr, 0, gt);
                tile_store_g<T, 32, 32, true, false>(sde, debuf(i), ep, nullptr, 0, gt);
            }
        } else {
            tile_store_g<T, 32, 64, true, sizeof(E) != 2>(sq, dbuf(i), rp, dtbuf(i), tpt, gt);
            tile_store_g<T, 64, 32, true, false>(sdl, dlbuf(i), tpt, nullptr, 0, gt);
        }
    };
    // the one-hot slot operand of this lane's key: geometry only
    F oh[S32];
    {
        const uint32_t km = key_mask(kj, G);
#pragma unroll
        for (int ks = 0; ks < S32; ++ks) oh[ks] = TT::onehot_row(km, ks, g);
    }
    const float sc = a.scale * LOG2E;

    // requests of an item's prologue operands (NP <= 256 < blockDim: one lse / delta value per thread)
    F kb[S64], vb[S64];
    float lse_r = INFINITY, dlt_r = 0.f;
    auto request_item = [&](const ItemPtrs& P) {
        load_row<T>(kb, P.kpg + (int64_t)min(kj, N - 1) * a.sn, g);
        load_row<T>(vb, P.vpg + (int64_t)min(kj, N - 1) * a.sn, g);
        load_set(P, 0);
        load_set_extra(P, 0);
        const int i = threadIdx.x;
        lse_r = i < N ? a.lse[P.bh * N + i] * LOG2E : INFINITY;           // P = 0 for padding queries
        dlt_r = i < N ? a.delta[P.bh * NP + i] * a.scale : 0.f;
    };
    // table-gradient jobs of this wave: job = tab*2 + dt (tab 0/1 key tables v/h: X = Q, R = dL';
    // tab 2/3 value tables: X = dO, R = S'), dT^T(64 d x 32 u) = X^T(d x q) . R(q x u).  The tables are shared by all
    // (b, h): the accumulators live across the workgroup's items and ONE partial per workgroup leaves the kernel
    // (256 x 32 KB instead of B*H x 32 KB written here and read back by the gradient finalisation).
    const int job0 = wave, job1 = wave + nwaves;      // nwaves >= 4 -> at most two jobs per wave
    f32x16 tacc[2] = {f32x16{}, f32x16{}};

    int item = blockIdx.x;
    if (item >= a.nitems) return;
    ItemPtrs P = item_ptrs(item);
    request_item(P);

    for (;;) {
    __syncthreads();                                  // previous item done with the staged tiles and lse / delta
    const int b = P.b, h = P.h;
    [[maybe_unused]] const int64_t bh = P.bh;
    PROF_DECL
    PROF_MARK();
    if (threadIdx.x < NP) { lse2[threadIdx.x] = lse_r; dlt_s[threadIdx.x] = dlt_r; }
    store_set(0);
    __syncthreads();
    PROF_MARK();


    f32x16 dk[2] = {f32x16{}, f32x16{}}, dv[2] = {f32x16{}, f32x16{}};
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        PROF_LOOP(t);
        if (t + 1 < nt) { load_set(P, t + 1); load_set_extra(P, t + 1); }
        if (active) {
            const E* qrow = qbuf(cur) + c32 * rp;
            const E* drow = dbuf(cur) + c32 * rp;
            const E* qerow = qebuf(cur) + c32 * ep;
            const E* derow = debuf(cur) + c32 * ep;
            f32x16 sacc = {}, pacc = {};
#pragma unroll
            for (int ks = 0; ks < S64; ++ks) {
                sacc = TT::mma(TT::load(qrow + ks * KI + g * EPL), kb[ks], sacc);
                pacc = TT::mma(TT::load(drow + ks * KI + g * EPL), vb[ks], pacc);
            }
#pragma unroll
            for (int ks = 0; ks < S32; ++ks) {
                sacc = TT::mma(TT::load(qerow + ks * KI + g * EPL), oh[ks], sacc);
                pacc = TT::mma(TT::load(derow + ks * KI + g * EPL), oh[ks], pacc);
            }
            PROF_LOOP(t);
            // lane = key, registers = queries t*32 + acc_row(r, g) — four runs of four consecutive
            // queries, so lse / delta come as 16-byte LDS reads; padding queries have lse2 = +inf
            // (P = 0).  Padding KEYS (lanes kj >= N) need no masking: a key is a COLUMN of every
            // product below, its values never mix into other lanes, and its rows are not stored.
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const f32x4v l4 = *reinterpret_cast<const f32x4v*>(lse2 + t * 32 + 8 * r4 + 4 * g);
                const f32x4v d4 = *reinterpret_cast<const f32x4v*>(dlt_s + t * 32 + 8 * r4 + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * r4 + e;
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[r], sc, -l4[e]));
                    if constexpr (DROP) {                                         // dV takes the dropped map, dS the mask on dP
                        const bool keep = drop_keep(drop_key(a.drop_seed, item), t * 32 + 8 * r4 + 4 * g + e, kj, a.drop_thr);
                        sacc[r] = keep ? p * a.drop_scale : 0.f;
                        pacc[r] = p * __builtin_fmaf(keep ? pacc[r] * a.drop_scale : 0.f, a.scale, -d4[e]);
                    } else {
                    sacc[r] = p;
                    pacc[r] = p * __builtin_fmaf(pacc[r], a.scale, -d4[e]);      // dlt_s holds scale * delta
                    }
                }
            }
            PROF_LOOP(t);
#pragma unroll
            for (int st = 0; st < S32; ++st) {
                const F pb = TT::from_acc(sacc, st);
                const F db = TT::from_acc(pacc, st);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dv[dt] = TT::mma(perm_operand<T>(dbuf(cur), rp, dtbuf(cur), tpt, dt, st, lane), pb, dv[dt]);
                    dk[dt] = TT::mma(perm_operand<T>(qbuf(cur), rp, qtbuf(cur), tpt, dt, st, lane), db, dk[dt]);
                }
            }
        }
        PROF_LOOP(t);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int job = jj == 0 ? job0 : job1;
            if (job < 8) {
                const int tab = job >> 1, dt = job & 1;
                const E* xR = tab < 2 ? qbuf(cur) : dbuf(cur);
                const E* xT = tab < 2 ? qtbuf(cur) : dtbuf(cur);
                const E* rT = (tab < 2 ? dlbuf(cur) : spbuf(cur)) + ((tab & 1) * 32 + c32) * tpt;
#pragma unroll
                for (int st = 0; st < S32; ++st)
                    tacc[jj] = TT::mma(perm_operand<T>(xR, rp, xT, tpt, dt, st, lane), TT::load_perm(rT, st, g), tacc[jj]);
            }
        }
        PROF_LOOP(t);
        if (t + 1 < nt) {
            store_set(cur ^ 1);
            PROF_LOOP(t);
            __syncthreads();
        }
        PROF_LOOP(t);
    }
    PROF_MARK();
    // the next item's prologue operands travel while this item's results are stored
    const int next = item + gridDim.x;
    const bool more = next < a.nitems;
    if (more) {
        P = item_ptrs(next);
        request_item(P);
    }
    if (kok) {
        const int64_t off = (int64_t)b * a.dsb + (int64_t)kj * a.dsn + (int64_t)h * a.dsh;
        store_rows_64<T>(reinterpret_cast<E*>(a.dk) + off, dk, g);
        store_rows_64<T>(reinterpret_cast<E*>(a.dv) + off, dv, g);
    }
    PROF_MARK();
    PROF_MARK();
    PROF_FLUSH();
    if (!more) break;
    item = next;
    }   // items
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int job = jj == 0 ? job0 : job1;
        if (job < 8) {
            // lane = bucket u (column), registers = d rows
            const int tab = job >> 1, dt = job & 1;
            float* dst = a.dtab + (((int64_t)blockIdx.x * 4 + tab) * 32 + c32) * 64 + dt * 32;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                *reinterpret_cast<f32x4v*>(dst + 8 * r4 + 4 * g) =
                    f32x4v{tacc[jj][4 * r4], tacc[jj][4 * r4 + 1], tacc[jj][4 * r4 + 2], tacc[jj][4 * r4 + 3]};
        }
    }
}

template <typename T, bool FAST, bool DROP = false>
int launch_bwd_impl(const BwdArgs& a, int B, hipStream_t st) {
    auto kq = attn_rpe2d_bwd_q_kernel<T, FAST, DROP>;
    auto kkv = attn_rpe2d_bwd_kv_kernel<T, DROP>;
    if (!cream::raise_dynamic_lds(kq, 160 * 1024) || !cream::raise_dynamic_lds(kkv, 160 * 1024)) return CREAM_ERR_LAUNCH;
    BwdArgs aa = a;
    aa.nitems = B * a.H;
    const int pgrid = aa.nitems < fwd_persistent_grid() ? aa.nitems : fwd_persistent_grid();
    hipLaunchKernelGGL(kq, dim3(pgrid), dim3(512), bwd_q_lds_bytes<T>(a.NP, a.NP / 32, FAST), st, aa);
    if (hipGetLastError() != hipSuccess) return CREAM_ERR_LAUNCH;
#ifdef PROBE_SKIP_KV                     // tools/probes/attn_probe.hip: keep the dQ kernel's phase stamps
    (void)kkv;
    return CREAM_OK;
#endif
    CREAM_LAUNCH(kkv, dim3(pgrid), dim3(512), bwd_kv_lds_bytes<T>(a.NP), st, aa);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

// The backward of the AutoFormer geometry in bf16: 0 = the two-launch backward (what every other geometry and fp32 run);
// 1 = the one-pass kernel with both roles on every wave (attn_rpe2d_bwd1.hpp, 7 waves); 2 = the one-pass kernel with the roles
// on separate waves and the table gradients in registers (attn_rpe2d_bwd2.hpp, 12 waves) — the DEFAULT: 91.6 against 102.6 us
// alone at B = 128, H = 6, same-call step A/B x3 8.475 -> 8.321 ms (profiles/r06_attn_bwd2.md).
// CREAM_ATTN_BWD1 in the environment sets the initial value; cream_attn_rpe2d_bwd_mode() switches it (A/B runs).
std::atomic<int> g_bwd_onepass{-1};
int bwd_onepass_mode() {
    int m = g_bwd_onepass.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = getenv("CREAM_ATTN_BWD1");
        m = e ? atoi(e) : 2;
        g_bwd_onepass.store(m, std::memory_order_relaxed);
    }
    return m;
}

// the side buffer `dlt` of the two-launch path (B*H*64*NP elements, far more than the 32 KB needed) carries the bf16
// operand images of the four tables
int launch_bwd1(const BwdArgs& a, int B, hipStream_t st) {
    if (!cream::raise_dynamic_lds(v2::attn_rpe2d_bwd1_kernel, (int)(160 * 1024))) return CREAM_ERR_LAUNCH;
    BwdArgs aa = a;
    aa.nitems = B * a.H;
    aa.stagger = 0;                                  // (de-phasing the workgroups' first items was measured: no gain, profiles/r04_attn_bwd1.md)
    if (bwd_onepass_mode() == 2) { static const int stg = getenv("CREAM_ATTN_BWD_STAGGER") ? atoi(getenv("CREAM_ATTN_BWD_STAGGER")) : 0; aa.stagger = stg; }   // (< 0: items in (b, h) order)
    const short* img = reinterpret_cast<const short*>(a.timg);
    if (!img) {                                      // no images from the caller: built into the side buffer, one more launch
        short* own = reinterpret_cast<short*>(a.dlt);
        hipLaunchKernelGGL(v2::table_images_kernel, dim3(8), dim3(256), 0, st, own, a.tkv, a.tkh, a.tvv, a.tvh, a.ldt, a.nb);
        if (hipGetLastError() != hipSuccess) return CREAM_ERR_LAUNCH;
        img = own;
    }
    const int grid = aa.nitems < fwd_persistent_grid() ? aa.nitems : fwd_persistent_grid();
    if (bwd_onepass_mode() == 2) {
        if (!cream::raise_dynamic_lds(v4::attn_rpe2d_bwd2_kernel, (int)(160 * 1024))) return CREAM_ERR_LAUNCH;
        CREAM_LAUNCH(v4::attn_rpe2d_bwd2_kernel, dim3(grid), dim3(v4::THREADS), v4::LDS_B, st, aa, img);
    } else {
        CREAM_LAUNCH(v2::attn_rpe2d_bwd1_kernel, dim3(grid), dim3(v2::THREADS), v2::LDS_B, st, aa, img);
    }
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

template <typename T>
int launch_bwd(const BwdArgs& a, int B, hipStream_t st) {
    if (a.drop_thr) return launch_bwd_impl<T, false, true>(a, B, st);       // attention dropout: the two-launch backward regenerates the mask
    if constexpr (sizeof(typename Tr<T>::elem) == 2) {
        if (fast_geometry(a.G)) {
            if (bwd_onepass_mode() && (size_t)B * a.H * 64 * a.NP >= (size_t)v2::IMG_ELEMS) return launch_bwd1(a, B, st);
            return launch_bwd_impl<T, true>(a, B, st);
        }
    }
    return launch_bwd_impl<T, false>(a, B, st);
}

bool geom_ok(int N, int gh, int gw, int mr, int nb) {
    return N >= 1 && N <= 256 && gh >= 0 && gw >= 1 && gh * gw + 1 == N && gh < CB && gw <= 32 - CB && mr >= 0 &&
           nb == 2 * mr + 2 && nb <= 32;
}

}  // namespace

extern "C" {

int cream_attn_rpe2d_padded_len(int N) { return N <= 0 ? 0 : ((N + 31) / 32) * 32; }

int cream_attn_rpe2d_bwd_mode(int onepass)
{
    const int prev = bwd_onepass_mode();
    if (onepass >= 0) g_bwd_onepass.store(onepass > 2 ? 1 : onepass, std::memory_order_relaxed);
    return prev;
}

int cream_attn_rpe2d_dtab_parts(int B, int H)
{
    if (B <= 0 || H <= 0) return 0;
    const int64_t items = (int64_t)B * H;
    return (int)(items < fwd_persistent_grid() ? items : fwd_persistent_grid());
}

int cream_attn_rpe2d_fwd_mode(int fwd2)
{
    const int prev = fwd2_mode();
    if (fwd2 >= 0) g_fwd2.store(fwd2 != 0, std::memory_order_relaxed);
    return prev;
}

int64_t cream_attn_rpe2d_table_image_bytes(void) { return (int64_t)v2::IMG_ELEMS * 2; }

int cream_attn_rpe2d_table_images(void* img, const float* tkv, const float* tkh, const float* tvv, const float* tvh, int ldt,
                                  int mr, void* stream)
{
    if (!img || !tkv || !tkh || !tvv || !tvh || ldt < 64 || mr < 0 || 2 * mr + 2 > 32 || ((uintptr_t)img) % 16) return CREAM_ERR_BAD_ARG;
    hipLaunchKernelGGL(v2::table_images_kernel, dim3(8), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<short*>(img), tkv, tkh,
                       tvv, tvh, ldt, 2 * mr + 2);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

int cream_attn_rpe2d_fwd(void* out, float* lse, void* sp, const void* q, const void* k, const void* v,
                         int64_t sb, int64_t sn, int64_t sh, const float* tkv, const float* tkh,
                         const float* tvv, const float* tvh, int ldt, int B, int H, int N, int gh, int gw,
                         int mr, float scale, int dtype, void* stream)
{
    return cream_attn_rpe2d_fwd_img(out, lse, sp, q, k, v, sb, sn, sh, tkv, tkh, tvv, tvh, ldt, nullptr, B, H, N, gh, gw, mr, scale,
                                    dtype, stream);
}

int cream_attn_rpe2d_fwd_img(void* out, float* lse, void* sp, const void* q, const void* k, const void* v,
                             int64_t sb, int64_t sn, int64_t sh, const float* tkv, const float* tkh,
                             const float* tvv, const float* tvh, int ldt, const void* timg, int B, int H, int N, int gh, int gw,
                             int mr, float scale, int dtype, void* stream)
{
    return cream_attn_rpe2d_fwd_drop(out, lse, sp, q, k, v, sb, sn, sh, tkv, tkh, tvv, tvh, ldt, timg, B, H, N, gh, gw, mr, scale,
                                     0.f, 0u, dtype, stream);
}

int cream_attn_rpe2d_fwd_drop(void* out, float* lse, void* sp, const void* q, const void* k, const void* v,
                              int64_t sb, int64_t sn, int64_t sh, const float* tkv, const float* tkh,
                              const float* tvv, const float* tvh, int ldt, const void* timg, int B, int H, int N, int gh, int gw,
                              int mr, float scale, float dropout_p, uint32_t dropout_seed, int dtype, void* stream)
{
    if (B < 0 || H < 0) return CREAM_ERR_BAD_ARG;
    if (!(dropout_p >= 0.f) || dropout_p >= 1.f) return CREAM_ERR_BAD_ARG;
    if (B == 0 || H == 0) return CREAM_OK;
    if (!out || !lse || !q || !k || !v || !tkv || !tkh || !tvv || !tvh) return CREAM_ERR_BAD_ARG;     // (sp == NULL: forward only)
    if (!geom_ok(N, gh, gw, mr, 2 * mr + 2)) return CREAM_ERR_TOO_LARGE;
    if (ldt < 64) return CREAM_ERR_BAD_ARG;
    const int esz = dtype == CREAM_F32 ? 4 : 2;
    // 16-byte vector loads of operand rows
    if ((sb * esz) % 16 || (sn * esz) % 16 || (sh * esz) % 16) return CREAM_ERR_BAD_ARG;
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) % 16) return CREAM_ERR_BAD_ARG;
    if (ldt % 4 || ((uintptr_t)tkv | (uintptr_t)tkh) % 16 || ((uintptr_t)timg) % 16) return CREAM_ERR_BAD_ARG;
    FwdArgs a;
    a.timg = timg;
    a.q = q; a.k = k; a.v = v; a.sb = sb; a.sn = sn; a.sh = sh;
    a.out = out; a.lse = lse; a.sp = sp;
    a.tkv = tkv; a.tkh = tkh; a.tvv = tvv; a.tvh = tvh; a.ldt = ldt; a.nb = 2 * mr + 2;
    a.H = H; a.NP = cream_attn_rpe2d_padded_len(N);
    a.G = RelGeom{N, gh, gw, mr};
    a.scale = scale;
    a.drop_thr = drop_threshold(dropout_p);
    a.drop_seed = dropout_seed;
    a.drop_scale = 1.f / (1.f - dropout_p);
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case CREAM_BF16: return launch_fwd<hip_bfloat16>(a, B, st);
        case CREAM_F32: return launch_fwd<float>(a, B, st);
        default: return CREAM_ERR_BAD_DTYPE;
    }
}

int cream_attn_rpe2d_bwd(void* dq, void* dk, void* dv, int64_t dsb, int64_t dsn, int64_t dsh, float* dtab,
                         void* dlt, void* qe, void* de, float* delta,
                         const void* dout, const void* out, const float* lse, const void* sp,
                         const void* q, const void* k, const void* v, int64_t sb, int64_t sn, int64_t sh,
                         const float* tkv, const float* tkh, const float* tvv, const float* tvh, int ldt,
                         int B, int H, int N, int gh, int gw, int mr, float scale, int dtype, void* stream)
{
    return cream_attn_rpe2d_bwd_img(dq, dk, dv, dsb, dsn, dsh, dtab, dlt, qe, de, delta, dout, out, lse, sp, q, k, v, sb, sn, sh, tkv, tkh,
                                    tvv, tvh, ldt, nullptr, B, H, N, gh, gw, mr, scale, dtype, stream);
}

int cream_attn_rpe2d_bwd_img(void* dq, void* dk, void* dv, int64_t dsb, int64_t dsn, int64_t dsh, float* dtab,
                             void* dlt, void* qe, void* de, float* delta,
                             const void* dout, const void* out, const float* lse, const void* sp,
                             const void* q, const void* k, const void* v, int64_t sb, int64_t sn, int64_t sh,
                             const float* tkv, const float* tkh, const float* tvv, const float* tvh, int ldt, const void* timg,
                             int B, int H, int N, int gh, int gw, int mr, float scale, int dtype, void* stream)
{
    return cream_attn_rpe2d_bwd_drop(dq, dk, dv, dsb, dsn, dsh, dtab, dlt, qe, de, delta, dout, out, lse, sp, q, k, v, sb, sn, sh, tkv,
                                     tkh, tvv, tvh, ldt, timg, B, H, N, gh, gw, mr, scale, 0.f, 0u, dtype, stream);
}

int cream_attn_rpe2d_bwd_drop(void* dq, void* dk, void* dv, int64_t dsb, int64_t dsn, int64_t dsh, float* dtab,
                              void* dlt, void* qe, void* de, float* delta,
                              const void* dout, const void* out, const float* lse, const void* sp,
                              const void* q, const void* k, const void* v, int64_t sb, int64_t sn, int64_t sh,
                              const float* tkv, const float* tkh, const float* tvv, const float* tvh, int ldt, const void* timg,
                              int B, int H, int N, int gh, int gw, int mr, float scale, float dropout_p, uint32_t dropout_seed,
                              int dtype, void* stream)
{
    if (B < 0 || H < 0) return CREAM_ERR_BAD_ARG;
    if (!(dropout_p >= 0.f) || dropout_p >= 1.f) return CREAM_ERR_BAD_ARG;
    if (B == 0 || H == 0) return CREAM_OK;
    if (!dq || !dk || !dv || !dtab || !dlt || !qe || !de || !delta || !dout || !out || !lse || !sp || !q || !k ||
        !v || !tkv || !tkh || !tvv || !tvh)
        return CREAM_ERR_BAD_ARG;
    if (!geom_ok(N, gh, gw, mr, 2 * mr + 2)) return CREAM_ERR_TOO_LARGE;
    if (ldt < 64 || ldt % 4) return CREAM_ERR_BAD_ARG;
    const int esz = dtype == CREAM_F32 ? 4 : 2;
    if ((sb * esz) % 16 || (sn * esz) % 16 || (sh * esz) % 16 || (dsb * esz) % 16 || (dsn * esz) % 16 ||
        (dsh * esz) % 16)
        return CREAM_ERR_BAD_ARG;
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv |
         (uintptr_t)dout | (uintptr_t)out | (uintptr_t)sp | (uintptr_t)dlt | (uintptr_t)qe | (uintptr_t)de |
         (uintptr_t)dtab | (uintptr_t)tkv | (uintptr_t)tkh | (uintptr_t)tvv | (uintptr_t)tvh | (uintptr_t)timg) % 16)
        return CREAM_ERR_BAD_ARG;
    BwdArgs a{};
    a.timg = timg;
    a.q = q; a.k = k; a.v = v; a.sb = sb; a.sn = sn; a.sh = sh;
    a.dq = dq; a.dk = dk; a.dv = dv; a.dsb = dsb; a.dsn = dsn; a.dsh = dsh;
    a.dout = dout; a.out = out; a.lse = lse; a.sp = sp;
    a.dlt = dlt; a.qe = qe; a.de = de; a.delta = delta; a.dtab = dtab;
    a.tkv = tkv; a.tkh = tkh; a.tvv = tvv; a.tvh = tvh; a.ldt = ldt; a.nb = 2 * mr + 2;
    a.H = H; a.NP = cream_attn_rpe2d_padded_len(N);
    a.G = RelGeom{N, gh, gw, mr};
    a.scale = scale;
    a.drop_thr = drop_threshold(dropout_p);
    a.drop_seed = dropout_seed;
    a.drop_scale = 1.f / (1.f - dropout_p);
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case CREAM_BF16: return launch_bwd<hip_bfloat16>(a, B, st);
        case CREAM_F32: return launch_bwd<float>(a, B, st);
        default: return CREAM_ERR_BAD_DTYPE;
    }
}

}  // extern "C"
