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
// one-hot extension described in attn_common.hpp.  Row-major operands (Q, K, dO, ...) are
// read straight from global memory (they are L2-resident: 25 KB per head); LDS holds only
// the transposed tiles the matrix cores need with the contraction index contiguous (V^T,
// K^T, Q^T, dO^T), the transposed value tables and the per-wave shift scratch.
//
// dtype = bf16 (throughput mode: bf16 operands, fp32 accumulation/softmax) or fp32 (parity
// mode: v_mfma_f32_32x32x2_f32, exact fp32 products) — one code path, Tr<T> traits.
#include <hip/hip_runtime.h>
#include <hip/hip_bfloat16.h>
#include <stdint.h>

#include "attn_common.hpp"
#include "cream_amd.h"

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
#pragma unroll
    for (int u = 0; u < 32; ++u) row[g * 32 + u] = bk[u];
    wave_lds_fence();
}

// bucket rows^T (64 buckets x NP queries, dtype T) for the second backward launch
template <typename T>
__device__ __forceinline__ void store_buckets_T(typename Tr<T>::elem* dst /* + qi */, int NP,
                                                const float (&bk)[32], int g) {
#pragma unroll
    for (int u = 0; u < 32; ++u) dst[(int64_t)(g * 32 + u) * NP] = Tr<T>::from_f(bk[u]);
}

// acc^T(64 x 32 queries) += Tab^T(64 x 64 buckets) . rows^T  with Tab^T in LDS ([64][tp]) and the
// bucket rows in the wave's fp32 scratch
template <typename T>
__device__ __forceinline__ void add_bucket_product(f32x16 (&o)[2], const typename Tr<T>::elem* tabT,
                                                   const float* scr, int lane) {
    using TT = Tr<T>;
    constexpr int tp = table_pitch<T>();
    const int g = lane >> 5;
    const float* row = scr + (lane & 31) * LP;
#pragma unroll
    for (int ks = 0; ks < 64 / TT::KI; ++ks) {
        const int k0 = ks * TT::KI + g * TT::EPL;
        const typename TT::frag b = TT::load_f32(row + k0, true);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
            o[dt] = TT::mma(TT::load(tabT + ((lane & 31) + 32 * dt) * tp + k0), b, o[dt]);
    }
}

// workgroup-cooperative: transposed tile  dst[d][n] = src(n, d)  for n < NP (zero beyond N).
// 16-byte chunks in row-major order: consecutive lanes read consecutive chunks (full lines).
template <typename T>
__device__ __forceinline__ void fill_transposed(typename Tr<T>::elem* dst, int pitch,
                                                const typename Tr<T>::elem* src, int64_t sn, int N, int NP) {
    using E = typename Tr<T>::elem;
    constexpr int V = 16 / sizeof(E), CPR = 64 / V;                     // chunks per row
    const int total = NP * CPR;
    constexpr int U = 4;
    for (int c0 = threadIdx.x; c0 < total; c0 += blockDim.x * U) {
        u32x4v buf[U];
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int c = c0 + i * blockDim.x;
            const int n = c / CPR, cc = c - n * CPR;
            buf[i] = (c < total && n < N) ? *reinterpret_cast<const u32x4v*>(src + (int64_t)n * sn + cc * V)
                                          : u32x4v{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int c = c0 + i * blockDim.x;
            if (c < total) {
                const int n = c / CPR, cc = c - n * CPR;
                union { u32x4v v; E e[V]; } u;
                u.v = buf[i];
#pragma unroll
                for (int e = 0; e < V; ++e) dst[(cc * V + e) * pitch + n] = u.e[e];
            }
        }
    }
}

// workgroup-cooperative: tabT[d][u] = (u < 32 ? tv[u][d] : th[u-32][d]), zero for u >= nb
template <typename T>
__device__ __forceinline__ void fill_tables_T(typename Tr<T>::elem* tabT, const float* tv, const float* th,
                                              int ldt, int nb) {
    constexpr int tp = table_pitch<T>();
    for (int i = threadIdx.x; i < 64 * 16; i += blockDim.x) {          // float4 = 4 d-values of one bucket
        const int d4 = (i & 15) * 4, u = i >> 4, uu = u & 31;
        const float* t = u < 32 ? tv : th;
        const f32x4v x = uu < nb ? *reinterpret_cast<const f32x4v*>(t + (int64_t)uu * ldt + d4) : f32x4v{0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 4; ++e) tabT[(d4 + e) * tp + u] = Tr<T>::from_f(x[e]);
    }
}

// workgroup-cooperative (bf16 mode): tabR[u][d] operand rows, rows 0..31 from tv, 32..63 from th
template <typename T>
__device__ __forceinline__ void fill_tables_R(typename Tr<T>::elem* tabR, const float* tv, const float* th,
                                              int ldt, int nb) {
    constexpr int tp = table_pitch<T>();
    for (int i = threadIdx.x; i < 64 * 16; i += blockDim.x) {
        const int d4 = (i & 15) * 4, u = i >> 4, uu = u & 31;
        const float* t = u < 32 ? tv : th;
        const f32x4v x = uu < nb ? *reinterpret_cast<const f32x4v*>(t + (int64_t)uu * ldt + d4) : f32x4v{0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 4; ++e) tabR[u * tp + d4 + e] = Tr<T>::from_f(x[e]);
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

// LDS carve shared by the forward and the dQ kernel:
//   tT [64][NP+PADT] transposed tile | tabT [64][tp] | tabR [128][tp] (bf16 only) | masks [NP] | scratch
template <typename T> __host__ __device__ constexpr int tabr_rows(int n_tables) { return tables_in_lds<T>() ? 32 * n_tables : 0; }
template <typename T> size_t q_side_lds_bytes(int NP, int n_tables) {
    using E = typename Tr<T>::elem;
    return (size_t)64 * (NP + Tr<T>::PADT) * sizeof(E) + (size_t)(64 + tabr_rows<T>(n_tables)) * table_pitch<T>() * sizeof(E) +
           (size_t)NP * 4 + (size_t)(NP / 32) * 32 * LP * 4;
}

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
template <typename T, int NT>
__global__ __launch_bounds__(NT * 64) void attn_rpe2d_fwd_kernel(const FwdArgs a) {
    using TT = Tr<T>;
    using E = typename TT::elem;
    using F = typename TT::frag;
    constexpr int KI = TT::KI, EPL = TT::EPL, S64 = 64 / KI, S32 = 32 / KI;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const RelGeom G = a.G;
    const int N = G.n, NP = a.NP;
    const int nt = NP >> 5;
    const int vp = NP + TT::PADT;
    constexpr int tp = table_pitch<T>();
    E* vt = reinterpret_cast<E*>(smem);                                   // V^T [64][vp]
    E* tvt = vt + 64 * vp;                                                // value tables^T [64][tp]
    E* tkr = tvt + 64 * tp;                                               // key table rows [64][tp] (bf16)
    uint32_t* masks = reinterpret_cast<uint32_t*>(tkr + tabr_rows<T>(2) * tp);   // [NP]
    float* scratch = reinterpret_cast<float*>(masks + NP);                // [nt][32][LP]

    const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, g = lane >> 5, c32 = lane & 31;
    const int64_t base = (int64_t)b * a.sb + (int64_t)h * a.sh;
    const E* qp = reinterpret_cast<const E*>(a.q) + base;
    const E* kp = reinterpret_cast<const E*>(a.k) + base;
    const E* vpg = reinterpret_cast<const E*>(a.v) + base;

    PROF_DECL
    PROF_MARK();
    // this wave's query rows and the first key tile: issued before the LDS fills
    const int qi = wave * 32 + c32;
    const bool qok = qi < N;
    const int qr = qi > 0 ? (qi - 1) / G.gw : 0, qc = qi > 0 ? (qi - 1) - qr * G.gw : 0;
    F qb[S64], ka[S64];
    load_row<T>(qb, qp + (int64_t)min(qi, N - 1) * a.sn, g);
    load_row<T>(ka, kp + (int64_t)min(c32, N - 1) * a.sn, g);

    // ---- workgroup prologue: V^T, tables, key slot masks --------------------------------
    fill_transposed<T>(vt, vp, vpg, a.sn, N, NP);
    fill_tables_T<T>(tvt, a.tvv, a.tvh, a.ldt, a.nb);
    if constexpr (tables_in_lds<T>()) fill_tables_R<T>(tkr, a.tkv, a.tkh, a.ldt, a.nb);
    for (int j = threadIdx.x; j < NP; j += blockDim.x) masks[j] = key_mask(j, G);
    __syncthreads();
    PROF_MARK();

    // ---- this wave's query tile: bucket lookups -> slot extension ---------------------------
    float* scr = scratch + wave * 32 * LP;
    if (!qok) zero_frags<T, S64>(qb);
    table_lookups<T>(scr, qb, tkr, a.tkv, a.tkh, a.ldt, a.nb, lane);
    wave_lds_fence();
    PROF_MARK();
    F qe[S32];
    build_ext<T>(qe, scr, lane, qi, qr, qc, G);
    PROF_MARK();

    // ---- S^T tiles: scores of all keys against this wave's 32 queries -------------------
    f32x16 s[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        s[t] = f32x16{};
        if (t < nt) {
            F kn[S64];
            if (t + 1 < nt) load_row<T>(kn, kp + (int64_t)min((t + 1) * 32 + c32, N - 1) * a.sn, g);
#pragma unroll
            for (int ks = 0; ks < S64; ++ks) s[t] = TT::mma(ka[ks], qb[ks], s[t]);
            const uint32_t km = masks[t * 32 + c32];
#pragma unroll
            for (int ks = 0; ks < S32; ++ks) s[t] = TT::mma(TT::onehot_row(km, ks, g), qe[ks], s[t]);
            if (t + 1 < nt) {
#pragma unroll
                for (int ks = 0; ks < S64; ++ks) ka[ks] = kn[ks];
            }
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
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (t < nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) m = fmaxf(m, s[t][r]);
        }
    m = fmaxf(m, __shfl_xor(m, 32));
    const float sc = a.scale * LOG2E;
    const float msc = m * sc;
    float l = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (t < nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(s[t][r] * sc - msc);
                s[t][r] = p;
                l += p;
            }
        }
    l += __shfl_xor(l, 32);
    const float inv_l = 1.f / l;
    if (qok && g == 0)
        a.lse[((int64_t)b * a.H + h) * N + qi] = (msc + log2f(l)) * (1.f / LOG2E);
    PROF_MARK();

    // ---- [O | slot sums]^T = [V | one-hot]^T . P^T ----------------------------------------
    f32x16 o[2] = {f32x16{}, f32x16{}};
    f32x16 ox = {};
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (t < nt) {
#pragma unroll
            for (int st = 0; st < S32; ++st) {
                const F pb = TT::from_acc(s[t], st);
                o[0] = TT::mma(TT::load_perm(vt + c32 * vp + t * 32, st, g), pb, o[0]);
                o[1] = TT::mma(TT::load_perm(vt + (c32 + 32) * vp + t * 32, st, g), pb, o[1]);
                ox = TT::mma(TT::onehot_perm(masks + t * 32, st, g, c32), pb, ox);
            }
        }
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] *= inv_l; o[1][r] *= inv_l; ox[r] *= inv_l; }
    PROF_MARK();

    // ---- value-side relative position term: slot sums -> bucket sums -> . tables ---------
    float bk[32];
    slots_to_buckets(bk, scr, ox, lane, qi, qr, qc, G);
    PROF_MARK();
    // S'^T (64 buckets x NP queries) for backward (dTvv / dTvh)
    store_buckets_T<T>(reinterpret_cast<E*>(a.sp) + ((int64_t)b * a.H + h) * 64 * NP + qi, NP, bk, g);
    PROF_MARK();
    add_bucket_product<T>(o, tvt, scr, lane);
    PROF_MARK();

    // ---- store O (b, n, h, :) ----------------------------------------------------------------
    if (qok) store_rows_64<T>(reinterpret_cast<E*>(a.out) + (((int64_t)b * N + qi) * a.H + h) * 64, o, g);
    PROF_MARK();
    PROF_FLUSH();
}

template <typename T> size_t fwd_lds_bytes(int NP) { return q_side_lds_bytes<T>(NP, 2); }

template <typename T, int NT>
int launch_fwd_nt(const FwdArgs& a, int B, hipStream_t st) {
    const size_t lds = fwd_lds_bytes<T>(a.NP);
    auto kern = attn_rpe2d_fwd_kernel<T, NT>;
    static bool attr_done = false;           // per instantiation
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            return CREAM_ERR_LAUNCH;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(B * a.H), dim3((a.NP / 32) * 64), lds, st, a);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

template <typename T>
int launch_fwd(const FwdArgs& a, int B, hipStream_t st) {
    const int nt = a.NP / 32;
    if (nt <= 2) return launch_fwd_nt<T, 2>(a, B, st);
    if (nt <= 4) return launch_fwd_nt<T, 4>(a, B, st);
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
template <typename T>
__global__ __launch_bounds__(512) void attn_rpe2d_bwd_q_kernel(const BwdArgs a) {
    using TT = Tr<T>;
    using E = typename TT::elem;
    using F = typename TT::frag;
    constexpr int KI = TT::KI, EPL = TT::EPL, S64 = 64 / KI, S32 = 32 / KI;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const RelGeom G = a.G;
    const int N = G.n, NP = a.NP;
    const int nt = NP >> 5;
    const int kpch = NP + TT::PADT;
    constexpr int tp = table_pitch<T>();
    E* kt = reinterpret_cast<E*>(smem);                                   // K^T [64][kpch]
    E* tkt = kt + 64 * kpch;                                              // key tables^T [64][tp]
    E* tkr = tkt + 64 * tp;                                               // key table rows, then value table rows (bf16)
    E* tvr = tkr + tabr_rows<T>(2) * tp;
    uint32_t* masks = reinterpret_cast<uint32_t*>(tkr + tabr_rows<T>(4) * tp);
    float* scratch = reinterpret_cast<float*>(masks + NP);

    const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
    const int64_t bh = (int64_t)b * a.H + h;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, g = lane >> 5, c32 = lane & 31;
    const int64_t base = (int64_t)b * a.sb + (int64_t)h * a.sh;
    const E* qp = reinterpret_cast<const E*>(a.q) + base;
    const E* kp = reinterpret_cast<const E*>(a.k) + base;
    const E* vpg = reinterpret_cast<const E*>(a.v) + base;
    const int64_t orow = (int64_t)a.H * 64;                               // token stride of out / dout
    const E* dop = reinterpret_cast<const E*>(a.dout) + ((int64_t)b * N * a.H + h) * 64;
    const E* outp = reinterpret_cast<const E*>(a.out) + ((int64_t)b * N * a.H + h) * 64;

    const int qi = wave * 32 + c32;
    const bool qok = qi < N;
    const int qcl = min(qi, N - 1);
    const int qr = qi > 0 ? (qi - 1) / G.gw : 0, qc = qi > 0 ? (qi - 1) - qr * G.gw : 0;
    F qb[S64], dob[S64], ka[S64], va[S64];
    float delta = 0.f;
    {
        F ob[S64];
        load_row<T>(qb, qp + (int64_t)qcl * a.sn, g);
        load_row<T>(dob, dop + (int64_t)qcl * orow, g);
        load_row<T>(ob, outp + (int64_t)qcl * orow, g);
        load_row<T>(ka, kp + (int64_t)min(c32, N - 1) * a.sn, g);
        load_row<T>(va, vpg + (int64_t)min(c32, N - 1) * a.sn, g);

        fill_transposed<T>(kt, kpch, kp, a.sn, N, NP);
        fill_tables_T<T>(tkt, a.tkv, a.tkh, a.ldt, a.nb);
        if constexpr (tables_in_lds<T>()) {
            fill_tables_R<T>(tkr, a.tkv, a.tkh, a.ldt, a.nb);
            fill_tables_R<T>(tvr, a.tvv, a.tvh, a.ldt, a.nb);
        }
        for (int j = threadIdx.x; j < NP; j += blockDim.x) masks[j] = key_mask(j, G);

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
    if (g == 0) a.delta[bh * NP + qi] = delta;
    __syncthreads();                                  // K^T, tables, masks in place

    float* scr = scratch + wave * 32 * LP;
    F qe[S32], de[S32];
    table_lookups<T>(scr, qb, tkr, a.tkv, a.tkh, a.ldt, a.nb, lane);
    wave_lds_fence();
    build_ext<T>(qe, scr, lane, qi, qr, qc, G);
    wave_lds_fence();
    table_lookups<T>(scr, dob, tvr, a.tvv, a.tvh, a.ldt, a.nb, lane);
    wave_lds_fence();
    build_ext<T>(de, scr, lane, qi, qr, qc, G);
    wave_lds_fence();
    store_ext_rows<T>(reinterpret_cast<E*>(a.qe) + (bh * NP + qi) * 32, qe, g);
    store_ext_rows<T>(reinterpret_cast<E*>(a.de) + (bh * NP + qi) * 32, de, g);

    const float sc = a.scale * LOG2E;
    const float m2 = qok ? a.lse[bh * N + qi] * LOG2E : 0.f;

    f32x16 dq[2] = {f32x16{}, f32x16{}};
    f32x16 dx = {};
    for (int t = 0; t < nt; ++t) {
        F kn[S64], vn[S64];
        if (t + 1 < nt) {
            const int64_t ro = (int64_t)min((t + 1) * 32 + c32, N - 1) * a.sn;
            load_row<T>(kn, kp + ro, g);
            load_row<T>(vn, vpg + ro, g);
        }
        f32x16 sacc = {}, pacc = {};
#pragma unroll
        for (int ks = 0; ks < S64; ++ks) {
            sacc = TT::mma(ka[ks], qb[ks], sacc);
            pacc = TT::mma(va[ks], dob[ks], pacc);
        }
        const uint32_t km = masks[t * 32 + c32];
#pragma unroll
        for (int ks = 0; ks < S32; ++ks) {
            const F oh = TT::onehot_row(km, ks, g);
            sacc = TT::mma(oh, qe[ks], sacc);
            pacc = TT::mma(oh, de[ks], pacc);
        }
        // dS^T = scale * P o (dP - delta), keys beyond N contribute nothing
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool ok = t * 32 + acc_row(r, g) < N;
            const float p = ok ? __builtin_amdgcn_exp2f(sacc[r] * sc - m2) : 0.f;
            sacc[r] = p * (pacc[r] - delta) * a.scale;
        }
#pragma unroll
        for (int st = 0; st < S32; ++st) {
            const F db = TT::from_acc(sacc, st);
            dq[0] = TT::mma(TT::load_perm(kt + c32 * kpch + t * 32, st, g), db, dq[0]);
            dq[1] = TT::mma(TT::load_perm(kt + (c32 + 32) * kpch + t * 32, st, g), db, dq[1]);
            dx = TT::mma(TT::onehot_perm(masks + t * 32, st, g, c32), db, dx);
        }
        if (t + 1 < nt) {
#pragma unroll
            for (int ks = 0; ks < S64; ++ks) { ka[ks] = kn[ks]; va[ks] = vn[ks]; }
        }
    }

    float bk[32];
    slots_to_buckets(bk, scr, dx, lane, qi, qr, qc, G);
    // dL'^T (64 buckets x NP queries) for the table gradients of launch B
    store_buckets_T<T>(reinterpret_cast<E*>(a.dlt) + bh * 64 * NP + qi, NP, bk, g);
    add_bucket_product<T>(dq, tkt, scr, lane);
    if (qok)
        store_rows_64<T>(reinterpret_cast<E*>(a.dq) + (int64_t)b * a.dsb + (int64_t)qi * a.dsn + (int64_t)h * a.dsh,
                         dq, g);
}

template <typename T>
__global__ __launch_bounds__(512) void attn_rpe2d_bwd_kv_kernel(const BwdArgs a) {
    using TT = Tr<T>;
    using E = typename TT::elem;
    using F = typename TT::frag;
    constexpr int KI = TT::KI, EPL = TT::EPL, S64 = 64 / KI, S32 = 32 / KI;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const RelGeom G = a.G;
    const int N = G.n, NP = a.NP;
    const int nt = NP >> 5;
    const int pch = NP + TT::PADT;
    E* qt = reinterpret_cast<E*>(smem);                                   // Q^T  [64][pch]
    E* dot = qt + 64 * pch;                                               // dO^T [64][pch]
    float* lse2 = reinterpret_cast<float*>(dot + 64 * pch);               // [NP]
    float* dlt_s = lse2 + NP;                                             // delta [NP]

    const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
    const int64_t bh = (int64_t)b * a.H + h;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, g = lane >> 5, c32 = lane & 31;
    const int64_t base = (int64_t)b * a.sb + (int64_t)h * a.sh;
    const E* qp = reinterpret_cast<const E*>(a.q) + base;
    const E* kp = reinterpret_cast<const E*>(a.k) + base;
    const E* vpg = reinterpret_cast<const E*>(a.v) + base;
    const int64_t orow = (int64_t)a.H * 64;
    const E* dop = reinterpret_cast<const E*>(a.dout) + ((int64_t)b * N * a.H + h) * 64;
    const E* qep = reinterpret_cast<const E*>(a.qe) + bh * NP * 32;
    const E* dep = reinterpret_cast<const E*>(a.de) + bh * NP * 32;

    const int kj = wave * 32 + c32;
    const bool kok = kj < N;
    F kb[S64], vb[S64], oh[S32];
    F qa[S64], da[S64], qx[S32], dxe[S32];
    load_row<T>(kb, kp + (int64_t)min(kj, N - 1) * a.sn, g);
    load_row<T>(vb, vpg + (int64_t)min(kj, N - 1) * a.sn, g);
    {   // first query tile (rows c32)
        const int q0 = min(c32, N - 1);
        load_row<T>(qa, qp + (int64_t)q0 * a.sn, g);
        load_row<T>(da, dop + (int64_t)q0 * orow, g);
        load_row32<T>(qx, qep + (int64_t)c32 * 32, g);
        load_row32<T>(dxe, dep + (int64_t)c32 * 32, g);
    }

    fill_transposed<T>(qt, pch, qp, a.sn, N, NP);
    fill_transposed<T>(dot, pch, dop, orow, N, NP);
    for (int i = threadIdx.x; i < NP; i += blockDim.x) {
        lse2[i] = i < N ? a.lse[bh * N + i] * LOG2E : INFINITY;           // P = 0 for padding queries
        dlt_s[i] = i < N ? a.delta[bh * NP + i] : 0.f;
    }
    {
        const uint32_t km = key_mask(kj, G);
#pragma unroll
        for (int ks = 0; ks < S32; ++ks) oh[ks] = TT::onehot_row(km, ks, g);
    }
    const float sc = a.scale * LOG2E;
    __syncthreads();

    f32x16 dk[2] = {f32x16{}, f32x16{}}, dv[2] = {f32x16{}, f32x16{}};
    for (int t = 0; t < nt; ++t) {
        F qn[S64], dn[S64], qxn[S32], dxn[S32];
        if (t + 1 < nt) {
            const int qn_i = (t + 1) * 32 + c32, qn_c = min(qn_i, N - 1);
            load_row<T>(qn, qp + (int64_t)qn_c * a.sn, g);
            load_row<T>(dn, dop + (int64_t)qn_c * orow, g);
            load_row32<T>(qxn, qep + (int64_t)qn_i * 32, g);
            load_row32<T>(dxn, dep + (int64_t)qn_i * 32, g);
        }
        f32x16 sacc = {}, pacc = {};
#pragma unroll
        for (int ks = 0; ks < S64; ++ks) {
            sacc = TT::mma(qa[ks], kb[ks], sacc);
            pacc = TT::mma(da[ks], vb[ks], pacc);
        }
#pragma unroll
        for (int ks = 0; ks < S32; ++ks) {
            sacc = TT::mma(qx[ks], oh[ks], sacc);
            pacc = TT::mma(dxe[ks], oh[ks], pacc);
        }
        // lane = key, registers = queries t*32 + acc_row(r, g); rows of padding queries hold
        // clamped (finite) data and are switched off by lse2 = +inf
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = t * 32 + acc_row(r, g);
            const float p = kok ? __builtin_amdgcn_exp2f(sacc[r] * sc - lse2[qq]) : 0.f;
            sacc[r] = p;
            pacc[r] = p * (pacc[r] - dlt_s[qq]) * a.scale;
        }
#pragma unroll
        for (int st = 0; st < S32; ++st) {
            const F pb = TT::from_acc(sacc, st);
            const F db = TT::from_acc(pacc, st);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                dv[dt] = TT::mma(TT::load_perm(dot + (c32 + 32 * dt) * pch + t * 32, st, g), pb, dv[dt]);
                dk[dt] = TT::mma(TT::load_perm(qt + (c32 + 32 * dt) * pch + t * 32, st, g), db, dk[dt]);
            }
        }
        if (t + 1 < nt) {
#pragma unroll
            for (int ks = 0; ks < S64; ++ks) { qa[ks] = qn[ks]; da[ks] = dn[ks]; }
#pragma unroll
            for (int ks = 0; ks < S32; ++ks) { qx[ks] = qxn[ks]; dxe[ks] = dxn[ks]; }
        }
    }
    if (kok) {
        const int64_t off = (int64_t)b * a.dsb + (int64_t)kj * a.dsn + (int64_t)h * a.dsh;
        store_rows_64<T>(reinterpret_cast<E*>(a.dk) + off, dk, g);
        store_rows_64<T>(reinterpret_cast<E*>(a.dv) + off, dv, g);
    }

    // ---- table gradients of this (b,h): dT^T(64 d x 32 u) = X^T(d x q) . R(q x u) ------------
    //   job = tab*2 + dt;  tab 0/1 = key tables v/h (X = Q, R = dL'), 2/3 = value tables (X = dO, R = S')
    const E* dltp = reinterpret_cast<const E*>(a.dlt) + bh * 64 * NP;
    const E* spp = reinterpret_cast<const E*>(a.sp) + bh * 64 * NP;
    for (int job = wave; job < 8; job += nt) {
        const int tab = job >> 1, dt = job & 1;
        const E* xT = (tab < 2 ? qt : dot) + (c32 + 32 * dt) * pch;
        const E* rT = (tab < 2 ? dltp : spp) + (int64_t)((tab & 1) * 32 + c32) * NP;
        f32x16 acc = {};
        for (int t = 0; t < nt; ++t) {
#pragma unroll
            for (int st = 0; st < S32; ++st)
                acc = TT::mma(TT::load_perm(xT + t * 32, st, g), TT::load_perm(rT + t * 32, st, g), acc);
        }
        // lane = bucket u (column), registers = d rows
        float* dst = a.dtab + ((bh * 4 + tab) * 32 + c32) * 64 + dt * 32;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
            *reinterpret_cast<f32x4v*>(dst + 8 * r4 + 4 * g) =
                f32x4v{acc[4 * r4], acc[4 * r4 + 1], acc[4 * r4 + 2], acc[4 * r4 + 3]};
    }
}

template <typename T> size_t bwd_q_lds_bytes(int NP) { return q_side_lds_bytes<T>(NP, 4); }
template <typename T> size_t bwd_kv_lds_bytes(int NP) {
    return (size_t)2 * 64 * (NP + Tr<T>::PADT) * sizeof(typename Tr<T>::elem) + (size_t)NP * 8;
}

template <typename T>
int launch_bwd(const BwdArgs& a, int B, hipStream_t st) {
    auto kq = attn_rpe2d_bwd_q_kernel<T>;
    auto kkv = attn_rpe2d_bwd_kv_kernel<T>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kq), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(kkv), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            return CREAM_ERR_LAUNCH;
        attr_done = true;
    }
    const dim3 grid(B * a.H), block((a.NP / 32) * 64);
    hipLaunchKernelGGL(kq, grid, block, bwd_q_lds_bytes<T>(a.NP), st, a);
    if (hipGetLastError() != hipSuccess) return CREAM_ERR_LAUNCH;
    hipLaunchKernelGGL(kkv, grid, block, bwd_kv_lds_bytes<T>(a.NP), st, a);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

bool geom_ok(int N, int gh, int gw, int mr, int nb) {
    return N >= 1 && N <= 256 && gh >= 0 && gw >= 1 && gh * gw + 1 == N && gh + gw + 1 <= 32 && mr >= 0 &&
           nb == 2 * mr + 2 && nb <= 32;
}

}  // namespace

extern "C" {

int cream_attn_rpe2d_padded_len(int N) { return N <= 0 ? 0 : ((N + 31) / 32) * 32; }

int cream_attn_rpe2d_fwd(void* out, float* lse, void* sp, const void* q, const void* k, const void* v,
                         int64_t sb, int64_t sn, int64_t sh, const float* tkv, const float* tkh,
                         const float* tvv, const float* tvh, int ldt, int B, int H, int N, int gh, int gw,
                         int mr, float scale, int dtype, void* stream)
{
    if (B < 0 || H < 0) return CREAM_ERR_BAD_ARG;
    if (B == 0 || H == 0) return CREAM_OK;
    if (!out || !lse || !sp || !q || !k || !v || !tkv || !tkh || !tvv || !tvh) return CREAM_ERR_BAD_ARG;
    if (!geom_ok(N, gh, gw, mr, 2 * mr + 2)) return CREAM_ERR_TOO_LARGE;
    if (ldt < 64) return CREAM_ERR_BAD_ARG;
    const int esz = dtype == CREAM_F32 ? 4 : 2;
    // 16-byte vector loads of operand rows
    if ((sb * esz) % 16 || (sn * esz) % 16 || (sh * esz) % 16) return CREAM_ERR_BAD_ARG;
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) % 16) return CREAM_ERR_BAD_ARG;
    if (ldt % 4 || ((uintptr_t)tkv | (uintptr_t)tkh) % 16) return CREAM_ERR_BAD_ARG;
    FwdArgs a;
    a.q = q; a.k = k; a.v = v; a.sb = sb; a.sn = sn; a.sh = sh;
    a.out = out; a.lse = lse; a.sp = sp;
    a.tkv = tkv; a.tkh = tkh; a.tvv = tvv; a.tvh = tvh; a.ldt = ldt; a.nb = 2 * mr + 2;
    a.H = H; a.NP = cream_attn_rpe2d_padded_len(N);
    a.G = RelGeom{N, gh, gw, mr};
    a.scale = scale;
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
    if (B < 0 || H < 0) return CREAM_ERR_BAD_ARG;
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
         (uintptr_t)dtab | (uintptr_t)tkv | (uintptr_t)tkh | (uintptr_t)tvv | (uintptr_t)tvh) % 16)
        return CREAM_ERR_BAD_ARG;
    BwdArgs a;
    a.q = q; a.k = k; a.v = v; a.sb = sb; a.sn = sn; a.sh = sh;
    a.dq = dq; a.dk = dk; a.dv = dv; a.dsb = dsb; a.dsn = dsn; a.dsh = dsh;
    a.dout = dout; a.out = out; a.lse = lse; a.sp = sp;
    a.dlt = dlt; a.qe = qe; a.de = de; a.delta = delta; a.dtab = dtab;
    a.tkv = tkv; a.tkh = tkh; a.tvv = tvv; a.tvh = tvh; a.ldt = ldt; a.nb = 2 * mr + 2;
    a.H = H; a.NP = cream_attn_rpe2d_padded_len(N);
    a.G = RelGeom{N, gh, gw, mr};
    a.scale = scale;
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case CREAM_BF16: return launch_bwd<hip_bfloat16>(a, B, st);
        case CREAM_F32: return launch_bwd<float>(a, B, st);
        default: return CREAM_ERR_BAD_DTYPE;
    }
}

}  // extern "C"
