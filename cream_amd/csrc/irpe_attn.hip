// irpe_attn.hip — fused attention with image relative position encoding (iRPE, contextual mode) on
// queries, keys and values, forward and backward, for gfx950 (MI355X).
//
// Reference semantics (iRPE/DeiT-with-iRPE/rpe_vision_transformer.py:68-97 `RPEAttention.forward`
// with irpe.py:585-687), per (batch b, head h), head_dim 64, L tokens, nb <= 64 buckets, s = scale:
//     LK = (s q) Wk                 (L x nb)   rpe_k lookups   (irpe.py:639-642)
//     LQ = (s k) Wq                 (L x nb)   rpe_q lookups
//     A[i,j] = (s q_i).k_j + LK[i, idk[i,j]] + LQ[j, idq[j,i]]            (:74-83)
//     P = softmax_j(A)
//     SV[i,u] = sum_{j: idv[i,j]=u} P[i,j]                                 (irpe.py:649-687)
//     O_i = sum_j P[i,j] v_j + SV[i,:] Wv                                  (:88-92)
// The reference materialises four to eight (B,H,L,L) tensors per layer (1 GB each in fp32 at
// B=64, H=12, L=577) around the rpe_index gather; here nothing of size L^2 touches HBM — the only
// side buffers are (B,H,L,64) bf16 rows (lookups, bucket sums and their gradients).
//
// Shape of the kernels.  One workgroup = 4 waves = 128 tokens of one (b,h); a wave owns 32 of them and
// computes 32 x 32 score tiles against the streamed other side with the swapped product of
// attn_common.hpp, so a lane holds 16 partners of ONE own token: the bias gather is 16 LDS reads from
// lookup rows lk[token][bucket] (bf16 like the reference's autocast matmul result; pitch 33 words: the
// bucket of a pair is data, so the reads are a random bank pattern whatever the pitch — an odd one
// keeps rows from aliasing), the bucket ids come as bytes: the 16 ids of a lane and tile are ONE 16-byte
// load from a byte matrix into registers (round 6; rounds 2-5 staged (128 x 32)-byte id tiles in LDS:
// 9 KB per table, and with them the kv / qkv kernels did not fit twice on a CU).  The bucket tables are
// converted once per (table, device) to padded uint8 matrices in BOTH orientations (query-major for the
// kernels whose lanes own queries, key-major for the dK/dV kernel), every 32-byte group in lane order
// (bucket_bytes_kernel), 370 KB at L=577: L2 resident.  Scatter-adds (value-side bucket sums, bucket gradients) are
// LDS float atomics into rows owned by the wave.
//
// Forward softmax is two-pass (max first, then exp / sums / P.V): the bucket sums cannot be rescaled
// cheaply when a running maximum moves, and the second pass is exactly what backward needs anyway
// (P from the saved log-sum-exp).  K is streamed twice through LDS (L2 hits), V once.
//
// Backward, with G = dO Wv^T (the value-side lookups of dO) and delta_i = dO_i . O_i:
//     dP[i,j] = dO_i . v_j + G[i, idv[i,j]]        dS = P o (dP - delta)
//     dLK[i,u] = sum_{j: idk[i,j]=u} dS[i,j]       dLQ[j,u] = sum_{i: idq[j,i]=u} dS[i,j]
//     dq = s (dS k + dLK Wk^T)     dk = dS^T (s q) + s dLQ Wq^T     dv = P^T dO
//     dWk = (s q)^T dLK            dWq = (s k)^T dLQ                dWv = SV^T dO     (summed over b, (h), tokens)
// Launch A (lanes own queries): delta, dq, dLK rows; hands LK and G rows to launch B.  Launch B (lanes
// own keys): dk, dv, dLQ rows.  The three table gradients are per-(b,h) 64 x 64 products over tokens
// (`irpe_table_grad_kernel`), reduced over b (and h for shared tables) by the caller.  No global atomics.
#include <hip/hip_runtime.h>
#include <hip/hip_bfloat16.h>
#include <stdint.h>

#include <type_traits>

#include "attn_common.hpp"
#include "cream_amd.h"

namespace {
using namespace cream;
using TT = Tr<hip_bfloat16>;
using F = TT::frag;

constexpr float LOG2E = 1.4426950408889634f;
constexpr int KP = 72;      // pitch (bf16) of row-major [32][64] tiles and of the [64][64] tables
constexpr int VTP = 36;     // pitch (bf16) of transposed [64][32] tiles read with load_perm
// Pitches of the per-token rows in LDS: fp32 scatter-add rows (bucket sums / bucket gradients) and bf16 lookup rows, both an odd
// number of words.  SM: tables of at most 51 buckets (every configuration of the zoo: product 50, cross 50, euclidean / quant fewer)
// take rows of 51 floats / 27 words instead of 65 / 33 — with the id tiles gone (ids_load) that is what lets the kernels with rpe on
// q, k and v stay twice on a CU (160 KB): forward 86 -> 74 KB, dK/dV launch 90 -> 77 KB.  Columns past the pitch do not exist: every
// 64-wide access of a row is guarded by its column (compile-time for most, the lane group g for the rest).
template <bool SM> struct Pitch {
    static constexpr int LK = SM ? 51 : 65;     // fp32 scatter-add rows
    static constexpr int LB = SM ? 54 : 66;     // bf16 lookup rows: 27 / 33 words
#ifdef IRPE_LQ_NARROW
    static constexpr int LQ = LB;
#else
    static constexpr int LQ = 66;               // rpe_q lookup rows of the streamed key tile (forward, dQ launch): the narrow pitch costs the
#endif                                          // side gathers 8 % (forward q 224 vs 207 us, q + k + v 620 vs 567) and 1.5 KB fits
};
constexpr int SM_MAX_NB = 51;
// ONE V tile buffer instead of two in the forward and the dQ launch (4.6 KB less LDS, paid with a second workgroup barrier per tile)
// exactly where it buys a third workgroup on the CU — the register budgets of these kernels allow three waves per SIMD, not four.
// Same-call A/B at config 4 (tools/probe_irpe_variants.sh, IRPE_V_DOUBLE = never): forward k + v 431 -> 338 us, backward with rpe
// on k 881 -> 750 (dQ launch 58.4 -> 53.8 KB); everywhere else the extra barrier costs 3-16 % and buys nothing.
constexpr int wgs_per_cu(int lds_bytes) { return 160 * 1024 / lds_bytes; }
template <template <bool, bool, bool, bool, int> class T, bool HQ, bool HK, bool HV, bool SM> constexpr int pick_vbufs() {
#ifdef IRPE_V_DOUBLE
    return 2;
#else
    return (wgs_per_cu(T<HQ, HK, HV, SM, 1>::total) > wgs_per_cu(T<HQ, HK, HV, SM, 2>::total) && wgs_per_cu(T<HQ, HK, HV, SM, 1>::total) <= 3) ? 1 : 2;
#endif
}
constexpr int QW = 4;       // waves (32-token tiles) per workgroup

struct Args {
    const short *q, *k, *v;
    int64_t sb, sn, sh;
    short* out;                       // (B, L, H, 64)
    float* lse;                       // (B, H, L)
    short* sv;                        // (B, H, NP, 64) bucket sums of P (value side), bf16
    const float *wq, *wk, *wv;        // (H', 64, nb), (H', 64, nb), (H', nb, 64) fp32
    int64_t wq_hs, wk_hs, wv_hs;      // head strides (0: shared)
    const float *bq, *bk;             // bias mode (irpe.py:622-624): (H', nb) tables, used when wq / wk is null
    int64_t bq_hs, bk_hs;
    const uint8_t *idq, *idk, *idv;         // (NP, NP) query-major:  [i][j] = bucket_q[j][i], bucket_k[i][j], bucket_v[i][j]
    const uint8_t *idq_t, *idk_t, *idv_t;   // (NP, NP) key-major:    [j][i] of the same
    int B, H, L, NP, nb;
    float scale;
    int causal;                       // key j takes part for query i only if j <= i (text towers); 0: full attention
    // attention dropout (rpe_vision_transformer.py:86 `attn = self.attn_drop(attn)`): P[i,j] -> keep[i,j] P[i,j] / (1 - p)
    // AFTER the softmax normalisation, for the value product and the value-side bucket sums alike.  keep is a pure
    // function of (seed, b, h, i, j) — see drop_keep — so that the backward launches regenerate it.
    uint32_t drop_thr;                // keep iff hash >= drop_thr = round(p 2^32); 0: no dropout
    uint32_t drop_seed;
    float drop_scale;                 // 1 / (1 - p)
    // backward
    const short* dout;                // (B, L, H, 64)
    short *dq, *dk, *dv;
    int64_t dsb, dsn, dsh;
    float* delta;                     // (B, H, NP)
    short *lkg, *gg;                  // (B, H, NP, 64) rpe_k lookups / value-side lookups of dO   (A -> B)
    short *dlk, *dlq;                 // (B, H, NP, 64) bucket gradients (-> table gradients)
};

// (keep mask: mix32 / drop_key / drop_keep in attn_common.hpp)

// consecutive logical workgroups (the blocks of one (b,h), which share the streamed side) on one XCD
__device__ __forceinline__ int xcd_order(int bid, int n) {
    if (n & 7) return bid;
    return (bid & 7) * (n >> 3) + (bid >> 3);
}

__device__ __forceinline__ F scaled(const F x, float s) {
    f32x8v y;
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = bf2f(x[e]) * s;
    return __builtin_bit_cast(F, __builtin_convertvector(y, hwbf16x8));
}
__device__ __forceinline__ u32x4v scaled_raw(const u32x4v x, float s) {
    return __builtin_bit_cast(u32x4v, scaled(__builtin_bit_cast(F, x), s));
}

// dst[c][r] = src[r][c] (src: nrow x ncol fp32, zero outside) as bf16 operand rows of pitch KP
__device__ __forceinline__ void stage_table_T(short* dst, const float* src, int nrow, int ncol) {
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i & 63, r = i >> 6;
        dst[c * KP + r] = f2bf((r < nrow && c < ncol) ? src[(int64_t)r * ncol + c] : 0.f);
    }
}
// dst[r][c] = src[r][c]
__device__ __forceinline__ void stage_table_R(short* dst, const float* src, int nrow, int ncol) {
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i & 63, r = i >> 6;
        dst[r * KP + c] = f2bf((r < nrow && c < ncol) ? src[(int64_t)r * ncol + c] : 0.f);
    }
}

// ---- staged tiles ------------------------------------------------------------------------------------
// row-major [32][64] tile: thread -> (row tid >> 3, 16-byte chunk tid & 7)
__device__ __forceinline__ u32x4v rows_load(const short* base, int64_t rs, int row0, int nrows, bool zero_pad) {
    const int row = threadIdx.x >> 3, cc = threadIdx.x & 7;
    const int j = min(row0 + row, nrows - 1);
    const u32x4v x = *reinterpret_cast<const u32x4v*>(base + (int64_t)j * rs + cc * 8);
    return (!zero_pad || row0 + row < nrows) ? x : u32x4v{0, 0, 0, 0};
}
__device__ __forceinline__ void rows_store(short* dst, const u32x4v& x) {
    const int row = threadIdx.x >> 3, cc = threadIdx.x & 7;
    *reinterpret_cast<u32x4v*>(dst + row * KP + cc * 8) = x;
}
__device__ __forceinline__ void rows_store_T(short* dst, const u32x4v& x, int pitch = VTP) {
    const int row = threadIdx.x >> 3, cc = threadIdx.x & 7;
    union { u32x4v v; short e[8]; } u;
    u.v = x;
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[(cc * 8 + e) * pitch + row] = u.e[e];
}
// lookup rows (bf16 [32][64] contiguous in global) -> [32][LBP]
template <int LBP> __device__ __forceinline__ void lrows_store(short* dst, const u32x4v& x) {
    const int row = threadIdx.x >> 3, cc = threadIdx.x & 7;
    uint32_t* d = reinterpret_cast<uint32_t*>(dst + row * LBP + cc * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (LBP >= 64 || cc * 8 + 2 * i + 1 < LBP) d[i] = x[i];
}
// The 16 bucket ids of this lane's (own row, partners acc_row(r, g), r = 0..15) of streamed tile t: ONE 16-byte load from the
// byte matrix straight into registers.  bucket_bytes_kernel stores every 32-byte group of a row in that order (lane group g = 0
// first), so the two lanes of an own row read the two halves of one 32-byte piece and a wave reads 32 such pieces — the access
// pattern the staged id tiles had (round 2-5: a (128 x 32)-byte tile per table, double-buffered in LDS, 27 KB with rpe on q, k
// and v), without the LDS tile, its store and its four reads per lane.
__device__ __forceinline__ u32x4v ids_load(const uint8_t* tab, int NP, int row, int t, int g) {
    return *reinterpret_cast<const u32x4v*>(tab + (int64_t)min(row, NP - 1) * NP + t * 32 + g * 16);
}
// the bytes hold 2 * bucket id: the byte offset of the bucket in a bf16 lookup row
__device__ __forceinline__ int off2_of(const u32x4v& w, int r) { return (w[r >> 2] >> (8 * (r & 3))) & 0xffu; }
// acc += (bf16 at byte offset off2 of row): one v_dot2c_f32_bf16 with (x, 0) . (1, 1) instead of shift + add
typedef __bf16 hwbf16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float add_bf16_at(float acc, const short* row, int off2) {
    const uint32_t x = *reinterpret_cast<const unsigned short*>(reinterpret_cast<const char*>(row) + off2);
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(hwbf16x2v, x), __builtin_bit_cast(hwbf16x2v, 0x3F803F80u), acc, false);   // (1, 1): a literal — the inline constant 1.0 is not (1, 0) for packed bf16
}

// row[id_r] += val_r for the 16 (bucket id, value) pairs of this lane — without LDS float atomics
// (ds_add_f32 retires a few lanes per clock: the kernels were 6x slower with it, profiles/r02_irpe_attention.md).
// The two lanes that share a row (g = 0 / 1) take turns — the LDS operations of a wave execute in order, so
// the second half adds onto the first half's sums — and each half goes 8 pairs at a time: 8 reads,
// duplicates inside the group resolved in registers (the latest earlier match carries the running sum,
// and the last write to an address is the complete one), 8 writes.
#ifndef IRPE_SCATTER_GROUP
#define IRPE_SCATTER_GROUP 4      // pairs per read-modify-write group.  Same-call A/B at config 4 (tools/probe_irpe_variants.sh, two workgroups per
#endif                            // CU): 16 / 8 / 4 / 2 -> forward v 413 / 313 / 274 / 274 us, backward k 1,042 / 930 / 870 / 870: with a second wave on
                                  // the SIMD the extra LDS round trips are hidden and the compare / select pairs (G (G - 1) / 2 per group) are not
__device__ __forceinline__ void scatter_add16(float* row, const u32x4v& w, const f32x16& val, int g) {
    constexpr int G = IRPE_SCATTER_GROUP;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (g == half) {
#pragma unroll
            for (int grp = 0; grp < 16 / G; ++grp) {
                int id[G];
                float sum[G];
#pragma unroll
                for (int v = 0; v < G; ++v) {
                    id[v] = off2_of(w, grp * G + v);
                    sum[v] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(row) + 2 * id[v]);
                }
#pragma unroll
                for (int v = 0; v < G; ++v) {
                    float base = sum[v];
#pragma unroll
                    for (int u = 0; u < v; ++u) base = (id[u] == id[v]) ? sum[u] : base;
                    sum[v] = base + val[grp * G + v];
                }
#pragma unroll
                for (int v = 0; v < G; ++v) *reinterpret_cast<float*>(reinterpret_cast<char*>(row) + 2 * id[v]) = sum[v];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
}

template <int K> using Slot = std::integral_constant<int, K>;
constexpr int PF = 1;       // streamed tiles in flight in registers.  Measured at L = 577: 3 in flight cost 48 more VGPRs
                            // (one workgroup less per CU) and made every kernel slower (fwd 327 -> 424 us) — the loop
                            // is VALU / LDS bound, not latency bound
// body(u, Slot<u % PF>) for u = 0 .. n-1
template <typename Body> __device__ __forceinline__ void stream(int n, Body&& body) {
    for (int u0 = 0; u0 < n; u0 += PF) {
        body(u0, Slot<0>{});
        if constexpr (PF > 1) { if (u0 + 1 < n) body(u0 + 1, Slot<1 % PF>{}); }
        if constexpr (PF > 2) { if (u0 + 2 < n) body(u0 + 2, Slot<2 % PF>{}); }
    }
}

// lookups^T (64 buckets x 32 own rows) = tab(64 buckets x 64 d) . X^T  ->  scr[row][bucket] (bf16) * mul
template <int LBP> __device__ __forceinline__ void lookups_to_lds(short* scr, const short* tab, const F (&xb)[4], float mul, int lane) {
    const int c32 = lane & 31, g = lane >> 5;
    f32x16 a0 = {}, a1 = {};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        a0 = TT::mma(TT::load(tab + c32 * KP + ks * 16 + g * 8), xb[ks], a0);
        a1 = TT::mma(TT::load(tab + (c32 + 32) * KP + ks * 16 + g * 8), xb[ks], a1);
    }
    short* row = scr + c32 * LBP;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        row[acc_row(r, g)] = f2bf(a0[r] * mul);
        if (LBP >= 64 || 32 + acc_row(r, g) < LBP) row[32 + acc_row(r, g)] = f2bf(a1[r] * mul);
    }
}
// bias mode: the lookups do not depend on the token — every row of a [32][LBP] block is the head's bias table
template <int LBP> __device__ __forceinline__ void bias_rows_to_lds(short* scr, const float* bias, int nb, int lane) {
    const int c32 = lane & 31, g = lane >> 5;
    short* row = scr + c32 * LBP + g * 32;
#pragma unroll 8
    for (int e = 0; e < 32; ++e)
        if (LBP >= 64 || g * 32 + e < LBP) row[e] = f2bf(g * 32 + e < nb ? bias[g * 32 + e] : 0.f);
}

// this lane's half of a bf16 lookup row (buckets ks*16 + g*8 .. +7, ks = 0..3) -> global row of 64
__device__ __forceinline__ void lrow_to_global(short* dst, const short* row, int g) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint32_t* s = reinterpret_cast<const uint32_t*>(row + ks * 16 + g * 8);
        *reinterpret_cast<u32x4v*>(dst + ks * 16 + g * 8) = u32x4v{s[0], s[1], s[2], s[3]};
    }
}
// fragment of 8 consecutive values (buckets ks*16 + g*8 ..) of an fp32 scatter-add row, times mul
template <int LKP> __device__ __forceinline__ F srow_frag(const float* row, int ks, int g, float mul) {
    f32x8v x;
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (LKP >= 64 || ks * 16 + g * 8 + e < LKP) ? row[ks * 16 + g * 8 + e] * mul : 0.f;
    return __builtin_bit_cast(F, __builtin_convertvector(x, hwbf16x8));
}
__device__ __forceinline__ void store_row64(short* op, const f32x16 (&o)[2], int g, float mul) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int d = dt * 32 + 8 * r4 + 4 * g;
            *reinterpret_cast<u32x2v*>(op + d) = u32x2v{f2bf_pair(o[dt][4 * r4] * mul, o[dt][4 * r4 + 1] * mul),
                                                        f2bf_pair(o[dt][4 * r4 + 2] * mul, o[dt][4 * r4 + 3] * mul)};
        }
}
__device__ __forceinline__ void load_frags(F (&f)[4], const short* row, int g) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) f[ks] = TT::load(row + ks * 16 + g * 8);
}

// A operand of a product that contracts over the 32 streamed tokens of a ROW-major [32][KP] tile (rows = tokens):
// MFMA row = column dt*32 + (lane & 31) of the tile, contraction indices in the permuted order of from_acc
// (acc_row(s2*8 + e, g)): two ds_read_b64_tr_b16 — the transpose happens on the way to the matrix cores, so no
// transposed copy of the tile is staged (8 conflicting 2-byte LDS stores per thread and tile, and 9 KB per buffer).
__device__ __forceinline__ F load_perm_tr(const short* rows, int dt, int s2, int lane) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    const int gi = lane & 15, q = lane >> 4;
    const short* p0 = rows + (16 * s2 + 4 * (q >> 1) + (gi >> 2)) * KP + dt * 32 + 16 * (q & 1) + (gi & 3) * 4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        reinterpret_cast<__attribute__((address_space(3))) s16x4*>(reinterpret_cast<uintptr_t>(p0)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        reinterpret_cast<__attribute__((address_space(3))) s16x4*>(reinterpret_cast<uintptr_t>(p0 + 8 * KP)));
    return F{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
template <bool HQ, bool HK, bool HV, bool SM, int VB> struct LdsFv {
    static constexpr int LBP = Pitch<SM>::LB, LKP = Pitch<SM>::LK, VBUFS = VB;
    static constexpr int kbuf = 0;                                   // 2 x [32][KP] bf16
    static constexpr int vbuf = kbuf + 2 * 32 * KP * 2;              // 2 x [32][KP] bf16 (read transposed: load_perm_tr)
    static constexpr int wkT = 0;                                    // [64 buckets][KP] bf16: prologue only, over the tile area
    static constexpr int wvT = 0;                                    // [64 d][KP] bf16 (columns = buckets): epilogue only, over the tile area
    static constexpr int wqT = vbuf + VBUFS * 32 * KP * 2;           // [64 buckets][KP] bf16: every key tile (lq_tile)
    static constexpr int lk = wqT + (HQ ? 64 * KP * 2 : 0);          // QW x [32][LBP] bf16
    static constexpr int sv = lk + (HK ? QW * 32 * LBP * 2 : 0);     // QW x [32][LKP] fp32
    static constexpr int lq = sv + (HV ? QW * 32 * LKP * 4 : 0);     // 2 x [32 keys][LBP] bf16
    static constexpr int total = lq + (HQ ? 2 * 32 * Pitch<SM>::LQ * 2 : 0);
};
template <bool HQ, bool HK, bool HV, bool SM> using LdsF = LdsFv<HQ, HK, HV, SM, pick_vbufs<LdsFv, HQ, HK, HV, SM>()>;

// S^T tile (rows = streamed tokens, column = own token) with the relative position terms:
//   own_row[id_own]             lookups indexed by the lane's own token        (ids in own_ids)
//   side[partner][id_side]      lookups indexed by the streamed token          (ids in side_ids)
template <bool OWN, bool SIDE, int LBP>
__device__ __forceinline__ f32x16 score_tile(const short* rows, const F (&own)[4], const u32x4v& own_ids,
                                             const u32x4v& side_ids, const short* own_row, const short* side, int lane) {
    const int c32 = lane & 31, g = lane >> 5;
    f32x16 s = {};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) s = TT::mma(TT::load(rows + c32 * KP + ks * 16 + g * 8), own[ks], s);
    if constexpr (OWN) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = add_bf16_at(s[r], own_row, off2_of(own_ids, r));
    }
    if constexpr (SIDE) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = add_bf16_at(s[r], side + acc_row(r, g) * LBP, off2_of(side_ids, r));
    }
    return s;
}

// rpe_q lookups of a staged key tile, shared by the four waves: waves 0 and 1 each compute one half of
// the buckets.  Ends with a workgroup barrier.
template <int LBP>
__device__ __forceinline__ void lq_tile(short* dst, const short* wqT, const short* kb, float scale, int wave, int lane, bool ctx) {
    if (!ctx) return;                                  // bias mode: both tile buffers were filled once, nothing changes per tile
    if (wave < 2) {
        const int c32 = lane & 31, g = lane >> 5;
        f32x16 acc = {};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            acc = TT::mma(TT::load(wqT + (c32 + 32 * wave) * KP + ks * 16 + g * 8), TT::load(kb + c32 * KP + ks * 16 + g * 8),
                          acc);
        short* row = dst + c32 * LBP + 32 * wave;                     // lane = key, rows = buckets
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (LBP >= 64 || 32 * wave + acc_row(r, g) < LBP) row[acc_row(r, g)] = f2bf(acc[r] * scale);   // (k * scale) Wq (:82)
    }
    __syncthreads();
}

template <bool HQ, bool HK, bool HV, bool DROP = false, bool SM = false>      // DROP: attention dropout (its own instantiations: the
// without any rpe term (TinyCLIP's towers): 108 VGPRs — held to 102 (96 used, 16 bytes of scratch) a fifth workgroup fits: 153 -> 144.5 us
// at config 4's shape (same-call A/B, tools/probe_irpe_variants.sh)
__attribute__((amdgpu_waves_per_eu((!HQ && !HK && !HV && !DROP) ? 5 : 2)))
__global__ __launch_bounds__(256) void irpe_attn_fwd_kernel(const Args a) {       // hash temporaries cost 6-40 VGPRs); SM: Pitch
    using L = LdsF<HQ, HK, HV, SM>;
    constexpr int LBP = Pitch<SM>::LB, LKP = Pitch<SM>::LK, LQP = Pitch<SM>::LQ, VBUFS = L::VBUFS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int NT = a.NP >> 5, QB = (NT + QW - 1) / QW;
    const int lb = xcd_order(blockIdx.x, gridDim.x);
    const int bh = lb / QB, qblk = lb - bh * QB;
    const int b = bh / a.H, h = bh - b * a.H;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, g = lane >> 5, c32 = lane & 31;
    const int q0 = qblk * 32 * QW;
    const bool active = qblk * QW + wave < NT;
    const int qrow = wave * 32 + c32;                 // row of this lane's query inside the workgroup's block
    const int qi = q0 + qrow;
    const bool qok = active && qi < a.L;

    const int64_t base = (int64_t)b * a.sb + (int64_t)h * a.sh;
    const short* qp = a.q + base;
    const short* kp = a.k + base;
    const short* vp = a.v + base;

    short* kbuf = reinterpret_cast<short*>(smem + L::kbuf);
    short* vbuf = reinterpret_cast<short*>(smem + L::vbuf);
    short* wkT = reinterpret_cast<short*>(smem + L::wkT);
    short* wqT = reinterpret_cast<short*>(smem + L::wqT);
    short* wvT = reinterpret_cast<short*>(smem + L::wvT);
    short* lkw = reinterpret_cast<short*>(smem + L::lk) + wave * 32 * LBP;
    float* svw = reinterpret_cast<float*>(smem + L::sv) + wave * 32 * LKP;
    short* lqs = reinterpret_cast<short*>(smem + L::lq);

    // ---- prologue: this lane's query row (scaled), tables, first key tile -----------------------
    F qs[4];
    load_frags(qs, qp + (int64_t)min(qi, a.L - 1) * a.sn, g);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qs[ks] = scaled(qs[ks], a.scale);          // q * scale in q's dtype (:73)
    u32x4v sk, sv4 = {}, sik = {}, siv = {}, siq = {};       // s*: the NEXT tile (in flight), c*: the current tile's ids
    u32x4v cik = {}, civ = {}, ciq = {};
    sk = rows_load(kp, a.sn, 0, a.L, false);
    sv4 = rows_load(vp, a.sn, 0, a.L, true);
    if constexpr (HK) cik = ids_load(a.idk, a.NP, qi, 0, g);
    if constexpr (HQ) ciq = ids_load(a.idq, a.NP, qi, 0, g);
    if constexpr (HV) civ = ids_load(a.idv, a.NP, qi, 0, g);
    if constexpr (HK) { if (a.wk) stage_table_T(wkT, a.wk + (int64_t)h * a.wk_hs, 64, a.nb); }
    if constexpr (HQ) {
        if (a.wq) stage_table_T(wqT, a.wq + (int64_t)h * a.wq_hs, 64, a.nb);
        else if (wave < 2) bias_rows_to_lds<LQP>(lqs + wave * 32 * LQP, a.bq + (int64_t)h * a.bq_hs, a.nb, lane);   // both tile buffers
    }
    if constexpr (HV) { for (int i = lane; i < 32 * LKP; i += 64) svw[i] = 0.f; }
    if constexpr (HK) {
        __syncthreads();                               // the rpe_k table lies over the tile area: use it before the first tiles land
        if (active) {
            if (a.wk) lookups_to_lds<LBP>(lkw, wkT, qs, 1.f, lane);
            else bias_rows_to_lds<LBP>(lkw, a.bk + (int64_t)h * a.bk_hs, a.nb, lane);
        }
        __syncthreads();
    }
    rows_store(kbuf, sk);
    rows_store(vbuf, sv4);
    __syncthreads();
    if constexpr (HQ) lq_tile<LQP>(lqs, wqT, kbuf, a.scale, wave, lane, a.wq != nullptr);

    f32x16 o[2] = {f32x16{}, f32x16{}};
    float l4[4] = {0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY;
    {
        // ---- ONE pass, online softmax with a LAZY reference maximum: m moves only when a tile's maximum exceeds it by more
        // than RESCALE_T (probabilities then stay below e^8 — exact in fp32 sums, fine as bf16 MFMA operands), so that the
        // rescaling of the running state — o, l and, with rpe on values, the 64 bucket sums of the row in LDS — is a rare,
        // wave-uniform branch instead of per-tile work.  (Round 2 ran TWO passes whenever rpe_v was present — all scores
        // and both bias gathers twice — because the bucket sums could not be rescaled cheaply every tile.)
        constexpr float RESCALE_T = 8.f;
        float lrun = 0.f;
        for (int t = 0; t < NT; ++t) {
            const int cur = t & 1, nxt = cur ^ 1;
            const bool more = t + 1 < NT;
            if (more) {
                sk = rows_load(kp, a.sn, (t + 1) * 32, a.L, false);
                sv4 = rows_load(vp, a.sn, (t + 1) * 32, a.L, true);
                if constexpr (HK) sik = ids_load(a.idk, a.NP, qi, t + 1, g);
                if constexpr (HQ) siq = ids_load(a.idq, a.NP, qi, t + 1, g);
                if constexpr (HV) siv = ids_load(a.idv, a.NP, qi, t + 1, g);
            }
            f32x16 s = {};
            if (active) {
                s = score_tile<HK, HQ, LQP>(kbuf + cur * 32 * KP, qs, cik, ciq, lkw + c32 * LBP, lqs + cur * 32 * LQP, lane);
                if (t == NT - 1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (t * 32 + acc_row(r, g) >= a.L) s[r] = -INFINITY;
                }
                if (a.causal && t * 32 + 31 > q0 + wave * 32) {      // (wave-uniform: tiles at or beyond the diagonal; key 0 is always visible)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (t * 32 + acc_row(r, g) > qi) s[r] = -INFINITY;
                }
                float t4[4] = {s[0], s[1], s[2], s[3]};
#pragma unroll
                for (int r = 4; r < 16; ++r) t4[r & 3] = fmaxf(t4[r & 3], s[r]);
                float tm = fmaxf(fmaxf(t4[0], t4[1]), fmaxf(t4[2], t4[3]));
                tm = fmaxf(tm, __shfl_xor(tm, 32));                  // finite: every tile has a valid key
                const bool grow = tm > m + RESCALE_T;                // (always on the first tile: m = -inf)
                if (__any(grow)) {
                    const float mn = grow ? tm : m;
                    const float alpha = __builtin_amdgcn_exp2f((m - mn) * LOG2E);      // 1 for the rows that keep their reference
                    m = mn;
                    lrun *= alpha;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
                    if constexpr (HV) {
                        float* row = svw + c32 * LKP + 32 * g;       // this lane's half of the row's 64 bucket sums
#pragma unroll
                        for (int i = 0; i < 32; ++i)
                            if (LKP >= 64 || 32 * g + i < LKP) row[i] *= alpha;   // (odd row pitch: scalar accesses, conflict-free)
                        wave_lds_fence();
                    }
                }
                const float mLn = m * LOG2E;
                float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], LOG2E, -mLn));
                    s[r] = p;
                    ps[r & 3] += p;
                }
                lrun += (ps[0] + ps[1]) + (ps[2] + ps[3]);
                if constexpr (DROP) {                                // the normaliser above is the undropped sum
                    const uint32_t dkey = drop_key(a.drop_seed, bh);
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        s[r] = drop_keep(dkey, qi, t * 32 + acc_row(r, g), a.drop_thr) ? s[r] * a.drop_scale : 0.f;
                }
            }
            if constexpr (VBUFS == 1) __syncthreads();     // the one V buffer was refilled behind the previous tile's barrier
            if (active) {
                const short* vb = vbuf + (VBUFS == 2 ? cur : 0) * 32 * KP;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const F pb = TT::from_acc(s, s2);
                    o[0] = TT::mma(load_perm_tr(vb, 0, s2, lane), pb, o[0]);
                    o[1] = TT::mma(load_perm_tr(vb, 1, s2, lane), pb, o[1]);
                }
                if constexpr (HV) scatter_add16(svw + c32 * LKP, civ, s, g);
            }
            if (more) {
                rows_store(kbuf + nxt * 32 * KP, sk);
                if constexpr (VBUFS == 2) rows_store(vbuf + nxt * 32 * KP, sv4);
                if constexpr (HK) cik = sik;
                if constexpr (HQ) ciq = siq;
                if constexpr (HV) civ = siv;
                __syncthreads();
                if constexpr (VBUFS == 1) rows_store(vbuf, sv4);                // every wave is past this tile's value product
                if constexpr (HQ) lq_tile<LQP>(lqs + nxt * 32 * LQP, wqT, kbuf + nxt * 32 * KP, a.scale, wave, lane, a.wq != nullptr);
            }
        }
        l4[0] = lrun;                                  // (this lane's half; the halves are added below)
    }
    if constexpr (HV) {
        __syncthreads();                               // every wave is done with the tile area: the value table goes over it
        stage_table_T(wvT, a.wv + (int64_t)h * a.wv_hs, a.nb, 64);            // Wv (nb x 64): dst[d][u]
        __syncthreads();
    }
    if (!active) return;
    float l = (l4[0] + l4[1]) + (l4[2] + l4[3]);
    l += __shfl_xor(l, 32);
    const float inv_l = 1.f / l;
    if (qok && g == 0) a.lse[(int64_t)bh * a.L + qi] = m + __logf(l);

    // ---- value-side term: (normalised bucket sums) . Wv;  the bucket sums are kept for backward ------
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] *= inv_l; o[1][r] *= inv_l; }
    if constexpr (HV) {
        const float* row = svw + c32 * LKP;
        short* svg = a.sv + ((int64_t)bh * a.NP + qi) * 64;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const F sb = srow_frag<LKP>(row, ks, g, inv_l);
            *reinterpret_cast<F*>(svg + ks * 16 + g * 8) = sb;
            o[0] = TT::mma(TT::load(wvT + c32 * KP + ks * 16 + g * 8), sb, o[0]);
            o[1] = TT::mma(TT::load(wvT + (c32 + 32) * KP + ks * 16 + g * 8), sb, o[1]);
        }
    }
    if (qok) store_row64(a.out + (((int64_t)b * a.L + qi) * a.H + h) * 64, o, g, 1.f);
}

// ---------------------------------------------------------------------------------------------------
// backward A: lanes own queries — delta, dq, dLK; LK / G rows for launch B
// ---------------------------------------------------------------------------------------------------
template <bool HQ, bool HK, bool HV, bool SM, int VB> struct LdsAv {
    static constexpr int LBP = Pitch<SM>::LB, LKP = Pitch<SM>::LK, VBUFS = VB;
    static constexpr int kbuf = 0;                                   // 2 x [32][KP]   K rows
    static constexpr int vbuf = kbuf + 2 * 32 * KP * 2;              // 2 x [32][KP]   V rows
    static constexpr int tile_end = vbuf + VBUFS * 32 * KP * 2;
    static constexpr int tables = (HV ? 2 : (HK ? 1 : 0)) * 64 * KP * 2;             // Wk^T (slot 0) / Wv (slot 1) are staged over the tile area outside the loop:
    static constexpr int stage_end = tile_end > tables ? tile_end : tables;          // room for both even with ONE V buffer
    static constexpr int lk = stage_end;                             // QW x [32][LBP] bf16
    static constexpr int gl = lk + (HK ? QW * 32 * LBP * 2 : 0);     // QW x [32][LBP] bf16
    static constexpr int dlk = gl + (HV ? QW * 32 * LBP * 2 : 0);    // QW x [32][LKP] fp32
    static constexpr int lq = dlk + (HK ? QW * 32 * LKP * 4 : 0);    // 2 x [32][LBP] bf16
    static constexpr int total = lq + (HQ ? 2 * 32 * Pitch<SM>::LQ * 2 : 0);
};
template <bool HQ, bool HK, bool HV, bool SM> using LdsA = LdsAv<HQ, HK, HV, SM, pick_vbufs<LdsAv, HQ, HK, HV, SM>()>;
#ifndef IRPE_V_DOUBLE
static_assert(LdsF<false, true, true, true>::VBUFS == 1 && LdsA<false, true, false, true>::VBUFS == 1, "the two kernels the third workgroup is for");
static_assert(LdsF<true, true, true, true>::VBUFS == 2 && LdsA<true, true, true, true>::VBUFS == 2 && LdsF<false, false, true, true>::VBUFS == 2 &&
              LdsF<false, true, false, true>::VBUFS == 2 && LdsA<false, true, true, true>::VBUFS == 2, "no barrier where it buys no workgroup");
static_assert(LdsF<true, true, true, true>::total <= 80 * 1024 && LdsA<true, true, true, true>::total <= 80 * 1024, "q + k + v: twice per CU");
#endif

template <bool HQ, bool HK, bool HV, bool DROP = false, bool SM = false>      // DROP: attention dropout (its own instantiations: the
__global__ __launch_bounds__(256) void irpe_attn_bwd_q_kernel(const Args a) {       // hash temporaries cost 6-40 VGPRs); SM: Pitch
    using L = LdsA<HQ, HK, HV, SM>;
    constexpr int LBP = Pitch<SM>::LB, LKP = Pitch<SM>::LK, LQP = Pitch<SM>::LQ, VBUFS = L::VBUFS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int NT = a.NP >> 5, QB = (NT + QW - 1) / QW;
    const int lb = xcd_order(blockIdx.x, gridDim.x);
    const int bh = lb / QB, qblk = lb - bh * QB;
    const int b = bh / a.H, h = bh - b * a.H;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, g = lane >> 5, c32 = lane & 31;
    const int q0 = qblk * 32 * QW;
    const bool active = qblk * QW + wave < NT;
    const int qrow = wave * 32 + c32;
    const int qi = q0 + qrow;
    const bool qok = active && qi < a.L;
    const int qcl = min(qi, a.L - 1);

    const int64_t base = (int64_t)b * a.sb + (int64_t)h * a.sh;
    const short* kp = a.k + base;
    const short* vp = a.v + base;
    const int64_t orow = (int64_t)a.H * 64;
    const short* dop = a.dout + ((int64_t)b * a.L * a.H + h) * 64;
    const short* outp = a.out + ((int64_t)b * a.L * a.H + h) * 64;

    short* kbuf = reinterpret_cast<short*>(smem + L::kbuf);
    short* vbuf = reinterpret_cast<short*>(smem + L::vbuf);
    short* tab0 = reinterpret_cast<short*>(smem);                    // tables staged over the tile area
    short* tab1 = tab0 + 64 * KP;
    short* lkw = reinterpret_cast<short*>(smem + L::lk) + wave * 32 * LBP;
    short* glw = reinterpret_cast<short*>(smem + L::gl) + wave * 32 * LBP;
    float* dlkw = reinterpret_cast<float*>(smem + L::dlk) + wave * 32 * LKP;
    short* lqs = reinterpret_cast<short*>(smem + L::lq);

    // ---- prologue ---------------------------------------------------------------------------------
    F qs[4], dob[4];
    float delta = 0.f;
    {
        F ob[4];
        load_frags(qs, a.q + base + (int64_t)qcl * a.sn, g);
        load_frags(dob, dop + (int64_t)qcl * orow, g);
        load_frags(ob, outp + (int64_t)qcl * orow, g);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qs[ks] = scaled(qs[ks], a.scale);
#pragma unroll
            for (int e = 0; e < 8; ++e) delta += bf2f(dob[ks][e]) * bf2f(ob[ks][e]);
        }
        delta += __shfl_xor(delta, 32);
        if (!qok) {
            delta = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) dob[ks] = TT::zero();
        }
    }
    if (active && g == 0) a.delta[(int64_t)bh * a.NP + qi] = delta;
    const float lseL = qok ? a.lse[(int64_t)bh * a.L + qi] * LOG2E : INFINITY;

    if constexpr (HK) { if (a.wk) stage_table_T(tab0, a.wk + (int64_t)h * a.wk_hs, 64, a.nb); }      // [bucket][d]
    if constexpr (HV) stage_table_R(tab1, a.wv + (int64_t)h * a.wv_hs, a.nb, 64);       // [bucket][d]
    // rpe_q lookups of the streamed keys: contextual — rows of (k * scale) Wq written by lq_rows_kernel (launched in front of
    // this kernel) into the dlq buffer, which launch B overwrites with the bucket gradients only at its end; staged per key tile
    // like K and V (rounds 2-5 recomputed them per tile from a resident Wq^T: 9 KB of LDS, two MFMA chains and a second
    // workgroup barrier per tile).  Bias mode: the head's table in both tile buffers, once.
    const short* lqg = HQ ? a.dlq + (int64_t)bh * a.NP * 64 : nullptr;
    const bool lq_rows = HQ && a.wq != nullptr;
    if constexpr (HQ) {
        if (!a.wq && wave < 2) bias_rows_to_lds<LQP>(lqs + wave * 32 * LQP, a.bq + (int64_t)h * a.bq_hs, a.nb, lane);
    }
    if constexpr (HK) { for (int i = lane; i < 32 * LKP; i += 64) dlkw[i] = 0.f; }
    __syncthreads();
    if (active) {
        if constexpr (HK) {
            if (a.wk) lookups_to_lds<LBP>(lkw, tab0, qs, 1.f, lane);
            else bias_rows_to_lds<LBP>(lkw, a.bk + (int64_t)h * a.bk_hs, a.nb, lane);
            wave_lds_fence();
            lrow_to_global(a.lkg + ((int64_t)bh * a.NP + qi) * 64, lkw + c32 * LBP, g);
        }
        if constexpr (HV) {
            lookups_to_lds<LBP>(glw, tab1, dob, 1.f, lane);
            wave_lds_fence();
            lrow_to_global(a.gg + ((int64_t)bh * a.NP + qi) * 64, glw + c32 * LBP, g);
        }
    }
    __syncthreads();                                   // tables consumed: the tile area is free
    static_assert(PF == 1, "the ids of the tile being processed live in one register set (cik / ciq / civ)");
    struct Stage { u32x4v k, v, lq, ik, iv, iq; } st[PF];
    u32x4v cik = {}, ciq = {}, civ = {};               // the current tile's ids of this lane (ids_load)
    auto issue = [&](Stage& r, int t) {
        r.k = rows_load(kp, a.sn, t * 32, a.L, false);
        r.v = rows_load(vp, a.sn, t * 32, a.L, true);
        if constexpr (HQ) { if (lq_rows) r.lq = rows_load(lqg, 64, t * 32, a.NP, false); }
        if constexpr (HK) r.ik = ids_load(a.idk, a.NP, qi, t, g);
        if constexpr (HQ) r.iq = ids_load(a.idq, a.NP, qi, t, g);
        if constexpr (HV) r.iv = ids_load(a.idv, a.NP, qi, t, g);
    };
    auto commit = [&](const Stage& r, int t) {
        const int buf = t & 1;
        rows_store(kbuf + buf * 32 * KP, r.k);
        if constexpr (VBUFS == 2) rows_store(vbuf + buf * 32 * KP, r.v);
        if constexpr (HQ) { if (lq_rows) lrows_store<LQP>(lqs + buf * 32 * LQP, r.lq); }
        if constexpr (HK) cik = r.ik;
        if constexpr (HQ) ciq = r.iq;
        if constexpr (HV) civ = r.iv;
    };
#pragma unroll
    for (int t = 0; t < PF; ++t)
        if (t < NT) issue(st[t], t);
    commit(st[0], 0);
    if constexpr (VBUFS == 1) rows_store(vbuf, st[0].v);
    __syncthreads();

    // ---- key tiles ----------------------------------------------------------------------------------
    f32x16 dq[2] = {f32x16{}, f32x16{}};
    stream(NT, [&](int t, auto slot) {
        constexpr int K = decltype(slot)::value;
        const int cur = t & 1;
        if (t + PF < NT) issue(st[K], t + PF);
        f32x16 s = {};
        if (active) s = score_tile<HK, HQ, LQP>(kbuf + cur * 32 * KP, qs, cik, ciq, lkw + c32 * LBP, lqs + cur * 32 * LQP, lane);
        if constexpr (VBUFS == 1) __syncthreads();     // the one V buffer was refilled behind the previous tile's barrier
        if (active) {
            f32x16 dp = {};
            const short* vb = vbuf + (VBUFS == 2 ? cur : 0) * 32 * KP;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) dp = TT::mma(TT::load(vb + c32 * KP + ks * 16 + g * 8), dob[ks], dp);
            if constexpr (HV) {
                const short* row = glw + c32 * LBP;
#pragma unroll
                for (int r = 0; r < 16; ++r) dp[r] = add_bf16_at(dp[r], row, off2_of(civ, r));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = t * 32 + acc_row(r, g);
                const bool ok = (t < NT - 1 || key < a.L) && !(a.causal && key > qi);
                const float p = ok ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], LOG2E, -lseL)) : 0.f;
                float dpr = dp[r];                                   // gradient of the DROPPED map -> of the softmax output
                if constexpr (DROP) dpr = drop_keep(drop_key(a.drop_seed, bh), qi, key, a.drop_thr) ? dpr * a.drop_scale : 0.f;
                s[r] = p * (dpr - delta);
            }
            if constexpr (HK) scatter_add16(dlkw + c32 * LKP, cik, s, g);
            const short* kb = kbuf + cur * 32 * KP;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const F db = TT::from_acc(s, s2);
                dq[0] = TT::mma(load_perm_tr(kb, 0, s2, lane), db, dq[0]);
                dq[1] = TT::mma(load_perm_tr(kb, 1, s2, lane), db, dq[1]);
            }
        }
        if (t + 1 < NT) {
            commit(st[(K + 1) % PF], t + 1);
            __syncthreads();
            if constexpr (VBUFS == 1) rows_store(vbuf, st[(K + 1) % PF].v);    // every wave is past this tile's dP product
        }
    });

    // ---- dq += dLK Wk^T; bucket gradient rows out -------------------------------------------------
    if constexpr (HK) {
        const bool ctx = a.wk != nullptr;              // bias mode: no dependence on q, only the bucket gradient rows leave
        if (ctx) {
            __syncthreads();                           // every wave is done with the tile area
            stage_table_R(tab0, a.wk + (int64_t)h * a.wk_hs, 64, a.nb);  // [d][bucket]
            __syncthreads();
        }
        if (active) {
            const float* row = dlkw + c32 * LKP;
            short* dst = a.dlk + ((int64_t)bh * a.NP + qi) * 64;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const F db = srow_frag<LKP>(row, ks, g, 1.f);
                *reinterpret_cast<F*>(dst + ks * 16 + g * 8) = db;
                if (ctx) {
                    dq[0] = TT::mma(TT::load(tab0 + c32 * KP + ks * 16 + g * 8), db, dq[0]);
                    dq[1] = TT::mma(TT::load(tab0 + (c32 + 32) * KP + ks * 16 + g * 8), db, dq[1]);
                }
            }
        }
    }
    if (qok) store_row64(a.dq + (int64_t)b * a.dsb + (int64_t)qi * a.dsn + (int64_t)h * a.dsh, dq, g, a.scale);
}

// rpe_q lookup rows of every key, (k * scale) Wq as bf16 (the values lq_tile puts in LDS in the forward: product first, scale
// after, :82) -> dst (B, H, NP, 64).  Launched in front of backward A with dst = the dlq buffer.
__global__ __launch_bounds__(256) void irpe_lq_rows_kernel(const Args a, short* dst) {
    __shared__ __attribute__((aligned(16))) short tab[64 * KP];
    __shared__ __attribute__((aligned(16))) short scr[QW * 32 * 66];
    const int NT = a.NP >> 5, KB = (NT + QW - 1) / QW;
    const int bh = blockIdx.x / KB, kblk = blockIdx.x - bh * KB;
    const int b = bh / a.H, h = bh - b * a.H;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, g = lane >> 5, c32 = lane & 31;
    const int kj = kblk * 32 * QW + wave * 32 + c32;
    stage_table_T(tab, a.wq + (int64_t)h * a.wq_hs, 64, a.nb);      // [bucket][d]
    __syncthreads();
    if (kblk * QW + wave >= NT) return;
    F kf[4];
    load_frags(kf, a.k + (int64_t)b * a.sb + (int64_t)h * a.sh + (int64_t)min(kj, a.L - 1) * a.sn, g);
    short* rows = scr + wave * 32 * 66;
    lookups_to_lds<66>(rows, tab, kf, a.scale, lane);
    wave_lds_fence();
    lrow_to_global(dst + ((int64_t)bh * a.NP + kj) * 64, rows + c32 * 66, g);
}

// ---------------------------------------------------------------------------------------------------
// backward B: lanes own keys — dk, dv, dLQ
// ---------------------------------------------------------------------------------------------------
template <bool HQ, bool HK, bool HV, bool SM> struct LdsB {
    static constexpr int LBP = Pitch<SM>::LB, LKP = Pitch<SM>::LK;
    static constexpr int qbuf = 0;                                   // 2 x [32][KP]   (s q) rows
    static constexpr int dobuf = qbuf + 2 * 32 * KP * 2;             // 2 x [32][KP]   dO rows
    static constexpr int stage_end = dobuf + 2 * 32 * KP * 2;
    static constexpr int lkt = stage_end;                            // 2 x [32 queries][LBP]  rpe_k lookups of the query tile
    static constexpr int gt = lkt + (HK ? 2 * 32 * LBP * 2 : 0);     // 2 x [32 queries][LBP]  value-side lookups of dO
    static constexpr int lqk = gt + (HV ? 2 * 32 * LBP * 2 : 0);     // QW x [32 keys][LBP]    rpe_q lookups (own keys)
    static constexpr int dlq = lqk + (HQ ? QW * 32 * LBP * 2 : 0);   // QW x [32 keys][LKP] fp32
    static constexpr int stats = dlq + (HQ ? QW * 32 * LKP * 4 : 0); // lse*log2e [NP], delta [NP]
    static constexpr int fixed = stats;
};

template <bool HQ, bool HK, bool HV, bool DROP = false, bool SM = false>      // DROP: attention dropout (its own instantiations: the
// without any rpe term the launch is register-bound at two waves per SIMD (184 VGPRs, 18 KB of LDS): held to 170 registers (164 used, no
// scratch) a third workgroup fits — 300 -> 265 us at config 4's shape; with rpe on k the same bound costs 52 bytes of scratch and 2 %
__attribute__((amdgpu_waves_per_eu((!HQ && !HK && !HV && !DROP) ? 3 : 2)))
__global__ __launch_bounds__(256) void irpe_attn_bwd_kv_kernel(const Args a) {       // hash temporaries cost 6-40 VGPRs); SM: Pitch
    using L = LdsB<HQ, HK, HV, SM>;
    constexpr int LBP = Pitch<SM>::LB, LKP = Pitch<SM>::LK;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int NT = a.NP >> 5, KB = (NT + QW - 1) / QW;
    const int lb = xcd_order(blockIdx.x, gridDim.x);
    const int bh = lb / KB, kblk = lb - bh * KB;
    const int b = bh / a.H, h = bh - b * a.H;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, g = lane >> 5, c32 = lane & 31;
    const int k0 = kblk * 32 * QW;
    const bool active = kblk * QW + wave < NT;
    const int krow = wave * 32 + c32;
    const int kj = k0 + krow;
    const bool kok = active && kj < a.L;
    const int kcl = min(kj, a.L - 1);

    const int64_t base = (int64_t)b * a.sb + (int64_t)h * a.sh;
    const short* qp = a.q + base;
    const int64_t orow = (int64_t)a.H * 64;
    const short* dop = a.dout + ((int64_t)b * a.L * a.H + h) * 64;
    const short* lkg = a.lkg + (int64_t)bh * a.NP * 64;
    const short* gg = a.gg + (int64_t)bh * a.NP * 64;

    short* qbuf = reinterpret_cast<short*>(smem + L::qbuf);
    short* dobuf = reinterpret_cast<short*>(smem + L::dobuf);
    short* tab0 = reinterpret_cast<short*>(smem);
    short* lkt = reinterpret_cast<short*>(smem + L::lkt);
    short* gt = reinterpret_cast<short*>(smem + L::gt);
    short* lqw = reinterpret_cast<short*>(smem + L::lqk) + wave * 32 * LBP;
    float* dlqw = reinterpret_cast<float*>(smem + L::dlq) + wave * 32 * LKP;
    float* lse_s = reinterpret_cast<float*>(smem + L::stats);
    float* delta_s = lse_s + a.NP;

    // ---- prologue ---------------------------------------------------------------------------------
    F kf[4], vf[4];
    load_frags(kf, a.k + base + (int64_t)kcl * a.sn, g);
    load_frags(vf, a.v + base + (int64_t)kcl * a.sn, g);
    for (int i = threadIdx.x; i < a.NP; i += 256) {
        lse_s[i] = i < a.L ? a.lse[(int64_t)bh * a.L + i] * LOG2E : INFINITY;
        delta_s[i] = a.delta[(int64_t)bh * a.NP + i];
    }
    if constexpr (HQ) {
        if (a.wq) stage_table_T(tab0, a.wq + (int64_t)h * a.wq_hs, 64, a.nb);      // [bucket][d]
        for (int i = lane; i < 32 * LKP; i += 64) dlqw[i] = 0.f;
        __syncthreads();
        if (active) {
            if (a.wq) {
                F ksf[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) ksf[ks] = scaled(kf[ks], a.scale);
                lookups_to_lds<LBP>(lqw, tab0, ksf, 1.f, lane);               // (k * scale) Wq (:82)
            } else {
                bias_rows_to_lds<LBP>(lqw, a.bq + (int64_t)h * a.bq_hs, a.nb, lane);
            }
            wave_lds_fence();
        }
    }
    __syncthreads();
    struct Stage { u32x4v q, dout, lk, gl, ik, iv, iq; } st[PF];
    u32x4v cik = {}, ciq = {}, civ = {};               // the current tile's (key-major) ids of this lane
    auto issue = [&](Stage& r, int t) {
        r.q = scaled_raw(rows_load(qp, a.sn, t * 32, a.L, false), a.scale);
        r.dout = rows_load(dop, orow, t * 32, a.L, true);
        if constexpr (HK) r.lk = rows_load(lkg, 64, t * 32, a.NP, false);
        if constexpr (HV) r.gl = rows_load(gg, 64, t * 32, a.NP, false);
        if constexpr (HK) r.ik = ids_load(a.idk_t, a.NP, kj, t, g);
        if constexpr (HQ) r.iq = ids_load(a.idq_t, a.NP, kj, t, g);
        if constexpr (HV) r.iv = ids_load(a.idv_t, a.NP, kj, t, g);
    };
    auto commit = [&](const Stage& r, int t) {
        const int buf = t & 1;
        rows_store(qbuf + buf * 32 * KP, r.q);
        rows_store(dobuf + buf * 32 * KP, r.dout);
        if constexpr (HK) lrows_store<LBP>(lkt + buf * 32 * LBP, r.lk);
        if constexpr (HV) lrows_store<LBP>(gt + buf * 32 * LBP, r.gl);
        if constexpr (HK) cik = r.ik;
        if constexpr (HQ) ciq = r.iq;
        if constexpr (HV) civ = r.iv;
    };
#pragma unroll
    for (int t = 0; t < PF; ++t)
        if (t < NT) issue(st[t], t);
    commit(st[0], 0);
    __syncthreads();

    // ---- query tiles --------------------------------------------------------------------------------
    f32x16 dk[2] = {f32x16{}, f32x16{}}, dv[2] = {f32x16{}, f32x16{}};
    stream(NT, [&](int t, auto slot) {
        constexpr int K = decltype(slot)::value;
        const int cur = t & 1;
        if (t + PF < NT) issue(st[K], t + PF);
        if (active) {
            // rows = queries of the tile, column = own key: own lookups = rpe_q, side lookups = rpe_k
            f32x16 s = score_tile<HQ, HK, LBP>(qbuf + cur * 32 * KP, kf, ciq, cik, lqw + c32 * LBP, lkt + cur * 32 * LBP, lane);
            f32x16 dp = {};
            const short* db = dobuf + cur * 32 * KP;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) dp = TT::mma(TT::load(db + c32 * KP + ks * 16 + g * 8), vf[ks], dp);
            if constexpr (HV) {
                const short* side = gt + cur * 32 * LBP;
#pragma unroll
                for (int r = 0; r < 16; ++r) dp[r] = add_bf16_at(dp[r], side + acc_row(r, g) * LBP, off2_of(civ, r));
            }
            f32x16 ds;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const f32x4v ls = *reinterpret_cast<const f32x4v*>(lse_s + t * 32 + 8 * rr + 4 * g);
                const f32x4v dl = *reinterpret_cast<const f32x4v*>(delta_s + t * 32 + 8 * rr + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * rr + e;
                    float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], LOG2E, -ls[e]));
                    if (a.causal && t * 32 + acc_row(r, g) < kj) p = 0.f;          // query before this lane's key
                    float dpr = dp[r], pd = p;
                    if constexpr (DROP) {                                          // dv takes the dropped map, dS the mask on dP
                        const bool keep = drop_keep(drop_key(a.drop_seed, bh), t * 32 + acc_row(r, g), kj, a.drop_thr);
                        dpr = keep ? dpr * a.drop_scale : 0.f;
                        pd = keep ? p * a.drop_scale : 0.f;
                    }
                    s[r] = pd;
                    ds[r] = p * (dpr - dl[e]);
                }
            }
            if constexpr (HQ) scatter_add16(dlqw + c32 * LKP, ciq, ds, g);
            const short* qb = qbuf + cur * 32 * KP;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const F pb = TT::from_acc(s, s2);
                const F sb = TT::from_acc(ds, s2);
                dv[0] = TT::mma(load_perm_tr(db, 0, s2, lane), pb, dv[0]);
                dv[1] = TT::mma(load_perm_tr(db, 1, s2, lane), pb, dv[1]);
                dk[0] = TT::mma(load_perm_tr(qb, 0, s2, lane), sb, dk[0]);
                dk[1] = TT::mma(load_perm_tr(qb, 1, s2, lane), sb, dk[1]);
            }
        }
        if (t + 1 < NT) {
            commit(st[(K + 1) % PF], t + 1);
            __syncthreads();
        }
    });

    // ---- dk += s dLQ Wq^T; bucket gradient rows out ------------------------------------------------
    if constexpr (HQ) {
        const bool ctx = a.wq != nullptr;
        if (ctx) {
            __syncthreads();
            stage_table_R(tab0, a.wq + (int64_t)h * a.wq_hs, 64, a.nb);  // [d][bucket]
            __syncthreads();
        }
        if (active) {
            const float* row = dlqw + c32 * LKP;
            short* dst = a.dlq + ((int64_t)bh * a.NP + kj) * 64;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                *reinterpret_cast<F*>(dst + ks * 16 + g * 8) = srow_frag<LKP>(row, ks, g, 1.f);
                if (ctx) {
                    const F db = srow_frag<LKP>(row, ks, g, a.scale);
                    dk[0] = TT::mma(TT::load(tab0 + c32 * KP + ks * 16 + g * 8), db, dk[0]);
                    dk[1] = TT::mma(TT::load(tab0 + (c32 + 32) * KP + ks * 16 + g * 8), db, dk[1]);
                }
            }
        }
    }
    if (kok) {
        const int64_t off = (int64_t)b * a.dsb + (int64_t)kj * a.dsn + (int64_t)h * a.dsh;
        store_row64(a.dk + off, dk, g, 1.f);
        store_row64(a.dv + off, dv, g, 1.f);
    }
}

// ---------------------------------------------------------------------------------------------------
// table gradients: out[bh][a][c] = mul * sum_i X[b,i,h][a] * Y[b,i,h][c]     (64 x 64 per (b,h))
// ---------------------------------------------------------------------------------------------------
struct TgArgs {
    const short *x, *y;
    int64_t xsb, xsn, xsh, ysb, ysn, ysh;
    float* out;
    int H, L;
    float mul;
};
constexpr int TGP = 40;      // pitch of the transposed tiles here: rows are read 16 bytes at a time

__global__ __launch_bounds__(256) void irpe_table_grad_kernel(const TgArgs a) {
    __shared__ __attribute__((aligned(16))) short xt[2][64 * TGP];
    __shared__ __attribute__((aligned(16))) short yt[2][64 * TGP];
    const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 5, c32 = lane & 31;
    const short* xp = a.x + (int64_t)b * a.xsb + (int64_t)h * a.xsh;
    const short* yp = a.y + (int64_t)b * a.ysb + (int64_t)h * a.ysh;
    const int NT = (a.L + 31) >> 5;
    const int ta = wave & 1, tc = wave >> 1;
    u32x4v sx = rows_load(xp, a.xsn, 0, a.L, true), sy = rows_load(yp, a.ysn, 0, a.L, true);
    rows_store_T(xt[0], sx, TGP);
    rows_store_T(yt[0], sy, TGP);
    __syncthreads();
    f32x16 acc = {};
    for (int t = 0; t < NT; ++t) {
        const int cur = t & 1;
        if (t + 1 < NT) {
            sx = rows_load(xp, a.xsn, (t + 1) * 32, a.L, true);
            sy = rows_load(yp, a.ysn, (t + 1) * 32, a.L, true);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            acc = TT::mma(TT::load(xt[cur] + (ta * 32 + c32) * TGP + ks * 16 + g * 8),
                          TT::load(yt[cur] + (tc * 32 + c32) * TGP + ks * 16 + g * 8), acc);
        if (t + 1 < NT) {
            rows_store_T(xt[cur ^ 1], sx, TGP);
            rows_store_T(yt[cur ^ 1], sy, TGP);
            __syncthreads();
        }
    }
    float* o = a.out + (int64_t)bh * 64 * 64;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[(ta * 32 + acc_row(r, g)) * 64 + tc * 32 + c32] = acc[r] * a.mul;
}

// int32 (Lq x Lk) bucket ids -> zero-padded uint8 (NP x NP) holding 2 * id, optionally transposed; every 32-byte group of a row
// (one streamed tile) in the order the lanes of the kernels take it (ids_load): byte 16 g + r = partner acc_row(r, g) of the tile
__global__ void bucket_bytes_kernel(uint8_t* dst, const int32_t* ids, int Lq, int Lk, int NP, int transpose) {
    const int i = blockIdx.y, pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= NP) return;
    const int j = (pos & ~31) + acc_row(pos & 15, (pos >> 4) & 1);
    int v = 0;
    if (!transpose) { if (i < Lq && j < Lk) v = ids[(int64_t)i * Lk + j]; }
    else            { if (j < Lq && i < Lk) v = ids[(int64_t)j * Lk + i]; }
    dst[(int64_t)i * NP + pos] = (uint8_t)(2 * v);    // byte offset of the bucket in a bf16 lookup row
}

template <typename K>
int launch(K kern, const Args& a, size_t lds, hipStream_t st) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
        hipSuccess)
        return CREAM_ERR_LAUNCH;
    const int NT = a.NP >> 5, QB = (NT + QW - 1) / QW;
    hipLaunchKernelGGL(kern, dim3(a.B * a.H * QB), dim3(256), lds, st, a);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

template <bool HQ, bool HK, bool HV, bool DROP, bool SM> int launch_fwd2(const Args& a, hipStream_t st) {
    return launch(irpe_attn_fwd_kernel<HQ, HK, HV, DROP, SM>, a, LdsF<HQ, HK, HV, SM>::total, st);
}
template <bool HQ, bool HK, bool HV, bool DROP, bool SM> int launch_bwd2(const Args& a, hipStream_t st) {
    if (HQ && a.wq) {                                  // rpe_q lookup rows of the keys for launch A, in the buffer launch B fills last
        const int NT = a.NP >> 5, KB = (NT + QW - 1) / QW;
        hipLaunchKernelGGL(irpe_lq_rows_kernel, dim3(a.B * a.H * KB), dim3(256), 0, st, a, a.dlq);
        if (hipGetLastError() != hipSuccess) return CREAM_ERR_LAUNCH;
    }
    const int rc = launch(irpe_attn_bwd_q_kernel<HQ, HK, HV, DROP, SM>, a, LdsA<HQ, HK, HV, SM>::total, st);
    if (rc) return rc;
    return launch(irpe_attn_bwd_kv_kernel<HQ, HK, HV, DROP, SM>, a, LdsB<HQ, HK, HV, SM>::fixed + (size_t)a.NP * 8, st);
}
// the narrow row pitches only where a table is present (without one there are no per-token rows)
template <bool HQ, bool HK, bool HV> int launch_fwd(const Args& a, hipStream_t st) {
    constexpr bool ANY = HQ || HK || HV;
    if (ANY && a.nb <= SM_MAX_NB) return a.drop_thr ? launch_fwd2<HQ, HK, HV, true, ANY>(a, st) : launch_fwd2<HQ, HK, HV, false, ANY>(a, st);
    return a.drop_thr ? launch_fwd2<HQ, HK, HV, true, false>(a, st) : launch_fwd2<HQ, HK, HV, false, false>(a, st);
}
template <bool HQ, bool HK, bool HV> int launch_bwd(const Args& a, hipStream_t st) {
    constexpr bool ANY = HQ || HK || HV;
    if (ANY && a.nb <= SM_MAX_NB) return a.drop_thr ? launch_bwd2<HQ, HK, HV, true, ANY>(a, st) : launch_bwd2<HQ, HK, HV, false, ANY>(a, st);
    return a.drop_thr ? launch_bwd2<HQ, HK, HV, true, false>(a, st) : launch_bwd2<HQ, HK, HV, false, false>(a, st);
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

int check(const cream_irpe_attn_desc* d, bool bwd) {
    if (!d || !d->q || !d->k || !d->v || !d->out || !d->lse) return CREAM_ERR_BAD_ARG;
    if (d->B <= 0 || d->H <= 0 || d->L <= 0 || d->nb <= 0 || d->nb > 64) return CREAM_ERR_BAD_ARG;
    if (d->NP != (d->L + 31) / 32 * 32) return CREAM_ERR_BAD_ARG;
    if (d->NP > 2048) return CREAM_ERR_TOO_LARGE;
    if (!(d->dropout_p >= 0.f) || d->dropout_p >= 1.f) return CREAM_ERR_BAD_ARG;          // (NP <= 2048: (i, j) fit 16 bits each)
    if (d->sn % 8 || d->sh % 8 || d->sb % 8 || !aligned16(d->q) || !aligned16(d->k) || !aligned16(d->v) || !aligned16(d->out))
        return CREAM_ERR_BAD_ARG;
    const bool hq = d->wq != nullptr || d->bq != nullptr, hk = d->wk != nullptr || d->bk != nullptr, hv = d->wv != nullptr;
    if ((d->wq && d->bq) || (d->wk && d->bk)) return CREAM_ERR_BAD_ARG;
    if ((hq && !d->idq) || (hk && !d->idk) || (hv && (!d->idv || !d->sv))) return CREAM_ERR_BAD_ARG;
    if ((hq && !aligned16(d->idq)) || (hk && !aligned16(d->idk)) || (hv && !aligned16(d->idv))) return CREAM_ERR_BAD_ARG;
    if (bwd) {
        if (!d->dout || !d->dq || !d->dk || !d->dv || !d->delta) return CREAM_ERR_BAD_ARG;
        if (d->dsn % 4 || d->dsh % 4 || d->dsb % 4 || !aligned16(d->dout)) return CREAM_ERR_BAD_ARG;
        if ((hq && (!d->idq_t || !d->dlq)) || (hk && (!d->idk_t || !d->lkg || !d->dlk)) || (hv && (!d->idv_t || !d->gg)))
            return CREAM_ERR_BAD_ARG;
    }
    return CREAM_OK;
}

Args to_args(const cream_irpe_attn_desc* d) {
    Args a{};
    a.q = (const short*)d->q; a.k = (const short*)d->k; a.v = (const short*)d->v;
    a.sb = d->sb; a.sn = d->sn; a.sh = d->sh;
    a.out = (short*)d->out; a.lse = d->lse; a.sv = (short*)d->sv;
    a.wq = d->wq; a.wk = d->wk; a.wv = d->wv;
    a.wq_hs = d->wq_hs; a.wk_hs = d->wk_hs; a.wv_hs = d->wv_hs;
    a.bq = d->bq; a.bk = d->bk; a.bq_hs = d->bq_hs; a.bk_hs = d->bk_hs;
    a.idq = d->idq; a.idk = d->idk; a.idv = d->idv;
    a.idq_t = d->idq_t; a.idk_t = d->idk_t; a.idv_t = d->idv_t;
    a.B = d->B; a.H = d->H; a.L = d->L; a.NP = d->NP; a.nb = d->nb; a.scale = d->scale; a.causal = d->causal;
    a.drop_thr = drop_threshold(d->dropout_p);
    a.drop_seed = d->dropout_seed;
    a.drop_scale = 1.f / (1.f - d->dropout_p);
    a.dout = (const short*)d->dout;
    a.dq = (short*)d->dq; a.dk = (short*)d->dk; a.dv = (short*)d->dv;
    a.dsb = d->dsb; a.dsn = d->dsn; a.dsh = d->dsh;
    a.delta = d->delta; a.lkg = (short*)d->lkg; a.gg = (short*)d->gg; a.dlk = (short*)d->dlk; a.dlq = (short*)d->dlq;
    return a;
}

template <template <bool, bool, bool> class Fn>
int dispatch(const Args& a, hipStream_t st) {
    const int key = ((a.wq || a.bq) ? 4 : 0) | ((a.wk || a.bk) ? 2 : 0) | (a.wv ? 1 : 0);
    switch (key) {
        case 0: return Fn<false, false, false>::run(a, st);
        case 1: return Fn<false, false, true>::run(a, st);
        case 2: return Fn<false, true, false>::run(a, st);
        case 3: return Fn<false, true, true>::run(a, st);
        case 4: return Fn<true, false, false>::run(a, st);
        case 5: return Fn<true, false, true>::run(a, st);
        case 6: return Fn<true, true, false>::run(a, st);
        default: return Fn<true, true, true>::run(a, st);
    }
}
template <bool HQ, bool HK, bool HV> struct FwdFn { static int run(const Args& a, hipStream_t st) { return launch_fwd<HQ, HK, HV>(a, st); } };
template <bool HQ, bool HK, bool HV> struct BwdFn { static int run(const Args& a, hipStream_t st) { return launch_bwd<HQ, HK, HV>(a, st); } };

}  // namespace

extern "C" {

int cream_irpe_padded_len(int L) { return (L + 31) / 32 * 32; }

int cream_irpe_bucket_bytes(uint8_t* dst, const int32_t* ids, int Lq, int Lk, int NP, int transpose, void* stream)
{
    if (!dst || !ids || Lq <= 0 || Lk <= 0 || NP % 32 || NP < Lq || NP < Lk) return CREAM_ERR_BAD_ARG;
    hipLaunchKernelGGL(bucket_bytes_kernel, dim3((NP + 255) / 256, NP), dim3(256), 0, (hipStream_t)stream, dst, ids, Lq,
                       Lk, NP, transpose);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

int cream_irpe_attn_fwd(const cream_irpe_attn_desc* d, void* stream)
{
    const int rc = check(d, false);
    if (rc) return rc;
    return dispatch<FwdFn>(to_args(d), (hipStream_t)stream);
}

int cream_irpe_attn_bwd(const cream_irpe_attn_desc* d, void* stream)
{
    const int rc = check(d, true);
    if (rc) return rc;
    return dispatch<BwdFn>(to_args(d), (hipStream_t)stream);
}

int cream_irpe_table_grad(float* out, const void* x, int64_t xsb, int64_t xsn, int64_t xsh, const void* y, int64_t ysb,
                          int64_t ysn, int64_t ysh, int B, int H, int L, float mul, void* stream)
{
    if (!out || !x || !y || B <= 0 || H <= 0 || L <= 0) return CREAM_ERR_BAD_ARG;
    if (xsn % 8 || xsh % 8 || xsb % 8 || ysn % 8 || ysh % 8 || ysb % 8 || !aligned16(x) || !aligned16(y)) return CREAM_ERR_BAD_ARG;
    TgArgs a{(const short*)x, (const short*)y, xsb, xsn, xsh, ysb, ysn, ysh, out, H, L, mul};
    hipLaunchKernelGGL(irpe_table_grad_kernel, dim3(B * H), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

}  // extern "C"
