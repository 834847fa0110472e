// attn_common.hpp — shared device helpers of the fused attention kernels (gfx950).
//
// Tiling convention used everywhere ("swapped" products): a score tile is computed as
//   S^T(32 keys x 32 queries) = K_tile(32 x 64) . Q_tile^T(64 x 32)
// with v_mfma_f32_32x32x16_bf16 (bf16 data) or v_mfma_f32_32x32x2_f32 (fp32 data, exact
// fp32 products — the parity mode).  In the 32x32 accumulator layout lane l owns COLUMN
// c = l & 31 and 16 rows  row(r, g) = (r & 3) + 8 * (r >> 2) + 4 * g,  g = l >> 5, so a
// lane holds 16 keys of ONE query: softmax statistics are in-lane, and the probabilities
// can be fed straight back as the B operand of the next MFMA (P^T or G^T) as long as the
// A operand enumerates the contraction index in the same permuted order — which is what
// `load_perm` / `onehot_perm` below do (no LDS round trip, no cross-lane shuffles for P).
//
// Relative position bias without gathers.  AutoFormer's 2-D bias of (query i, key j)
// (multihead_super.py:40-62) depends only on (i, grid row of j) and (i, grid column of j):
//   bias[i,j] = a_i[slotA(j)] + b_i[slotB(j)]
// so it is a dot product of a per-query "extension" vector x_i (one entry per slot) with a
// one-hot vector of the key.  Appending x_i to q_i and the one-hot to k_j turns the bias
// gather into 32 more contraction columns of the QK^T MFMA, and appending the one-hot to
// v_j makes the P.V MFMA produce the per-slot probability sums needed by the value-side
// bias — the rpe_index gather/scatter become matrix-core work with LDS only used for the
// per-query shift  slot <-> bucket  (once per 32-query tile, not per key tile).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bfloat16.h>
#include <stdint.h>

namespace cream {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int acc_row(int r, int g) { return (r & 3) + 8 * (r >> 2) + 4 * g; }

typedef __bf16 hwbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 hwbf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x8v __attribute__((ext_vector_type(8)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ short f2bf(float f) {            // round-to-nearest-even (v_cvt_pk_bf16_f32)
    return __builtin_bit_cast(short, (__bf16)f);
}
__device__ __forceinline__ uint32_t f2bf_pair(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2v{lo, hi}), hwbf16x2));
}
__device__ __forceinline__ float bf2f(short h) { return __uint_as_float(((uint32_t)(uint16_t)h) << 16); }

__device__ __forceinline__ int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

// The slots of a key are kept as a 32-bit mask (bit c set <=> the key sits in slot c).

// ---- per-dtype traits ---------------------------------------------------------------
template <typename T> struct Tr;

template <> struct Tr<hip_bfloat16> {
    using elem = short;                       // raw bf16 bits
    using frag = bf16x8;                      // MFMA operand of one lane
    static constexpr int KI = 16;             // contraction length of one MFMA
    static constexpr int EPL = 8;             // operand elements per lane per MFMA
    static constexpr int PADR = 8;            // row padding (elements) of [n][64] LDS tiles
    static constexpr int PADT = 12;           // row padding of transposed [64][n] LDS tiles
    static constexpr short ONE = 0x3F80;
    static __device__ __forceinline__ elem from_f(float f) { return f2bf(f); }
    static __device__ __forceinline__ float to_f(elem e) { return bf2f(e); }
    static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ frag zero() { return frag{0, 0, 0, 0, 0, 0, 0, 0}; }
    // EPL contiguous elements starting at p (16-byte aligned, LDS or global)
    static __device__ __forceinline__ frag load(const elem* p) {
        union { u32x4v v; frag f; } u;
        u.v = *reinterpret_cast<const u32x4v*>(p);
        return u.f;
    }
    // EPL elements from fp32 memory (tables / fp32 scratch), converted
    static __device__ __forceinline__ frag load_f32(const float* p, bool ok) {
        f32x8v x;
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = ok ? p[e] : 0.f;
        return __builtin_bit_cast(frag, __builtin_convertvector(x, hwbf16x8));
    }
    // operand built from accumulator registers [s*EPL, s*EPL+EPL) (the permuted-k trick)
    static __device__ __forceinline__ frag from_acc(const f32x16& a, int s) {
        f32x8v x;
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = a[s * 8 + e];
        return __builtin_bit_cast(frag, __builtin_convertvector(x, hwbf16x8));
    }
    // matching A operand: row pointer `rowp` of a TRANSPOSED tile ([d][n]), contraction
    // indices acc_row(s*8 + e, g): two runs of 4 consecutive columns (8-byte aligned)
    static __device__ __forceinline__ frag load_perm(const elem* rowp, int s, int g) {
        union { u32x2v v[2]; frag f; } u;
        u.v[0] = *reinterpret_cast<const u32x2v*>(rowp + 16 * s + 4 * g);
        u.v[1] = *reinterpret_cast<const u32x2v*>(rowp + 16 * s + 8 + 4 * g);
        return u.f;
    }
    // one-hot operand of a key with slot mask `mask`: contraction slots c = ks*16 + 8g + e
    static __device__ __forceinline__ frag onehot_row(uint32_t mask, int ks, int g) {
        const uint32_t m = mask >> (ks * 16 + 8 * g);
        union { uint32_t w[4]; frag f; } u;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            u.w[p] = ((m >> (2 * p)) & 1u) * 0x3F80u + ((m >> (2 * p + 1)) & 1u) * 0x3F800000u;
        return u.f;
    }
    // one-hot A operand, rows = slots (this lane: slot c), contraction = keys in the permuted
    // order of from_acc; `km` points at the slot masks of the 32-key tile (16-byte aligned)
    static __device__ __forceinline__ frag onehot_perm(const uint32_t* km, int s, int g, int c) {
        const u32x4v lo = *reinterpret_cast<const u32x4v*>(km + 16 * s + 4 * g);
        const u32x4v hi = *reinterpret_cast<const u32x4v*>(km + 16 * s + 8 + 4 * g);
        union { uint32_t w[4]; frag f; } u;
        u.w[0] = ((lo[0] >> c) & 1u) * 0x3F80u + ((lo[1] >> c) & 1u) * 0x3F800000u;
        u.w[1] = ((lo[2] >> c) & 1u) * 0x3F80u + ((lo[3] >> c) & 1u) * 0x3F800000u;
        u.w[2] = ((hi[0] >> c) & 1u) * 0x3F80u + ((hi[1] >> c) & 1u) * 0x3F800000u;
        u.w[3] = ((hi[2] >> c) & 1u) * 0x3F80u + ((hi[3] >> c) & 1u) * 0x3F800000u;
        return u.f;
    }
};

template <> struct Tr<float> {
    using elem = float;
    using frag = float;
    static constexpr int KI = 2;
    static constexpr int EPL = 1;
    static constexpr int PADR = 1;
    static constexpr int PADT = 1;
    static __device__ __forceinline__ elem from_f(float f) { return f; }
    static __device__ __forceinline__ float to_f(elem e) { return e; }
    static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ frag zero() { return 0.f; }
    static __device__ __forceinline__ frag load(const elem* p) { return *p; }
    static __device__ __forceinline__ frag load_f32(const float* p, bool ok) { return ok ? *p : 0.f; }
    static __device__ __forceinline__ frag from_acc(const f32x16& a, int s) { return a[s]; }
    static __device__ __forceinline__ frag load_perm(const elem* rowp, int s, int g) {
        return rowp[acc_row(s, g)];
    }
    static __device__ __forceinline__ frag onehot_row(uint32_t mask, int ks, int g) {
        return (float)((mask >> (ks * 2 + g)) & 1u);
    }
    static __device__ __forceinline__ frag onehot_perm(const uint32_t* km, int s, int g, int c) {
        return (float)((km[acc_row(s, g)] >> c) & 1u);
    }
};

// Geometry of AutoFormer's 2-D relative position index (multihead_super.py:40-62):
// token 0 = class token, tokens 1.. = gh x gw grid (row-major); bucket of (query i, key j)
//   vertical   : clamp(kr - qr, +-mr) + mr + 1     horizontal: clamp(kc - qc, +-mr) + mr + 1
//   bucket 0 when i == 0 or j == 0.
// Slots of a key: [0, gh) = grid row, gh = "class token key", [CB, CB+gw) = grid column with
// CB = 16 (so the first 16-slot contraction step is the vertical table, the second one the
// horizontal table); the fused kernels need gh <= 15, gw <= 16 and 2*mr + 2 <= 32.
struct RelGeom {
    int n;          // tokens (gh*gw + 1)
    int gh, gw;     // grid
    int mr;         // max_relative_position
};

constexpr int CB = 16;      // first grid-column slot

// slot mask of key j (0 for padding keys j >= n)
__device__ __forceinline__ uint32_t key_mask(int j, const RelGeom& G) {
    if (j >= G.n) return 0u;
    if (j == 0) return 1u << G.gh;
    const int r = (j - 1) / G.gw, c = (j - 1) - r * G.gw;
    return (1u << r) | (1u << (CB + c));
}

constexpr int LP = 71;      // pitch (floats) of the per-wave shift scratch rows: [32][64] bucket
                            // lookups, or the zero-padded slot windows of slots_to_buckets14

// x_i[c]: extension of query i for slot c, from its bucket lookups row[0..31] (vertical
// table) and row[32..63] (horizontal table).  Used for the key-side bias (row = q.T_k^T)
// and, in backward, for d(slot sums) (row = dO.T_v^T).  Branch-free: one LDS read per slot.
__device__ __forceinline__ float ext_gather(const float* row, float cls, int c, int qi, int qr, int qc,
                                            const RelGeom& G) {
    const int iv = clampi(c - qr, -G.mr, G.mr) + G.mr + 1;
    const int ih = 32 + clampi(c - CB - qc, -G.mr, G.mr) + G.mr + 1;
    float x = row[c < CB ? iv : ih];
    x = c == G.gh ? cls : x;
    x = ((c > G.gh && c < CB) || c >= CB + G.gw) ? 0.f : x;
    const float x0 = c <= G.gh ? cls : 0.f;          // class-token query: every key has bucket 0
    return qi == 0 ? x0 : x;
}

// Adjoint of ext_gather, as a gather: bucket u (0..31) of table `tab` (0 vertical, 1 horizontal)
// of query i from its 32 slot values slot[0..31] (LDS row).
//   u = 0            : class-token key slot (or, for the class-token query, all slots <= gh)
//   u = d + mr + 1   : the slot at relative distance d, plus everything clamped onto it
__device__ __forceinline__ float bucket_from_slots(const float* slot, int u, int tab, int qi, int qr, int qc,
                                                   const RelGeom& G) {
    const int lim = tab == 0 ? G.gh : G.gw, base = tab == 0 ? 0 : CB, pos0 = tab == 0 ? qr : qc;
    const int d = u - G.mr - 1;
    const int pos = pos0 + d;
    const bool inside = u >= 1 && u <= 2 * G.mr + 1 && pos >= 0 && pos < lim;
    float x = slot[inside ? base + pos : G.gh];       // u == 0 reads the class-token key slot
    x = (inside || u == 0) ? x : 0.f;
    if (lim - 1 > G.mr) {                             // clamped distances exist (wave-uniform)
        if (d == -G.mr) for (int p = 0; p < pos && p < lim; ++p) x += slot[base + p];
        if (d == G.mr) for (int p = (pos < 0 ? 0 : pos + 1); p < lim; ++p) x += slot[base + p];
    }
    if (qi == 0) {                                    // class-token query: sum of all key slots
        x = 0.f;
        if (u == 0) for (int c = 0; c <= G.gh; ++c) x += slot[c];
    }
    return x;
}

// A operand of a product that contracts over the 32 tokens of a ROW-major [32][pitch] bf16 tile (rows = tokens):
// MFMA row = column dt*32 + (lane & 31) of the tile, contraction indices in the permuted order of from_acc /
// load_perm (acc_row(s*8 + e, g)).  Two ds_read_b64_tr_b16: the transpose happens on the way to the matrix cores,
// so no transposed copy of the tile has to be staged (that copy cost 8 bank-conflicting 2-byte LDS stores per
// thread and tile).  Lane i of a 16-lane group supplies the address of row i>>2, columns 4(i&3).. of a 4 x 16
// block and receives column i, rows 0..3; with a pitch of 72 elements the four rows fall on disjoint banks.
__device__ __forceinline__ bf16x8 load_perm_tr(const short* rows, int pitch, int dt, int s, int lane) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    const int gi = lane & 15, q = lane >> 4;
    const short* p0 = rows + (16 * s + 4 * (q >> 1) + (gi >> 2)) * pitch + dt * 32 + 16 * (q & 1) + (gi & 3) * 4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        reinterpret_cast<__attribute__((address_space(3))) s16x4*>(reinterpret_cast<uintptr_t>(p0)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        reinterpret_cast<__attribute__((address_space(3))) s16x4*>(reinterpret_cast<uintptr_t>(p0 + 8 * pitch)));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// the same operand for either data type: bf16 reads the row-major tile through the transposing LDS read, fp32
// (no 32-bit transposing read) reads the transposed copy
template <typename T>
__device__ __forceinline__ typename Tr<T>::frag perm_operand(const typename Tr<T>::elem* rm, int rm_pitch,
                                                             const typename Tr<T>::elem* tr, int tr_pitch, int dt, int s,
                                                             int lane) {
    if constexpr (sizeof(typename Tr<T>::elem) == 2) return load_perm_tr(rm, rm_pitch, dt, s, lane);
    else return Tr<T>::load_perm(tr + ((lane & 31) + 32 * dt) * tr_pitch, s, lane >> 5);
}

// make LDS traffic of this wave visible to its own later reads (in-order LDS, compiler fence)
__device__ __forceinline__ void wave_lds_fence() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// ---- the AutoFormer geometry (14 x 14 grid, max_relative_position 14, N = 197) ---------------
// No relative distance is clamped (13 <= 14), so the slot <-> bucket shift of a query is a pure
// WINDOW: x_i[row slot c] = Lv[i][c - qr + 15] — 14 consecutive lookups starting at 15 - qr — and
// the adjoint is a window of the zero-padded slot vector.  With one per-lane base address all
// index arithmetic folds into the immediate offsets of ds_read_b32: the generic gathers above
// cost ~8 VALU instructions per element, and these kernels are VALU-issue-bound.
constexpr int G14 = 14;

// 16 slot values (one 16-slot half: kh = 0 vertical incl. the class-token slot, kh = 1
// horizontal) of this lane's query: lane group g supplies slots kh*16 + 8g + e, e = 0..7.
// `row` holds the bucket lookups (row[u] vertical, row[32 + u] horizontal table).
__device__ __forceinline__ void ext_window14(float (&x)[8], const float* row, float cls, int kh, int g, int qr, int qc) {
    const float* p = row + (kh == 0 ? 15 - qr : 32 + 15 - qc) + 8 * g;
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = p[e];
    // slots 14 (class token key), 15 and 30, 31 (unused) sit in lane group 1, e = 6, 7
    x[6] = g ? (kh == 0 ? cls : 0.f) : x[6];
    x[7] = g ? 0.f : x[7];
}

// class-token query (query 0: lane pair 0 of the first query tile): every key has bucket 0, so
// x_0[c] = cls for the row slots and the class-token slot, 0 for the column slots.  Rewrites
// that query's lookup row so that ext_window14 returns exactly this.  Call between the cls read
// and the window reads, wave-uniformly for the first query tile only.
__device__ __forceinline__ void ext_fix_query0(float* row, float cls, int lane) {
    if ((lane & 31) == 0) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            row[15 + u] = cls;
            row[32 + 15 + u] = 0.f;
        }
    }
}

// Adjoint: slot tile (accumulator layout: this lane holds slots acc_row(r, g) of its query) ->
// the 32 bucket values of table g (0 vertical / 1 horizontal) of this lane's query.
// Row layout (floats): [0,14) zeros | [14,28) row slots | [28,42) zeros | [42,56) column slots |
// [56,70) zeros | [70] class-token slot.   bucket u = d + 15 of a query at grid position pos0 is
// the slot at pos0 + d: index 14 + pos0 + d = pos0 + u - 1 (+28 for the horizontal table).
__device__ __forceinline__ void slots_to_buckets14(float (&bk)[32], float* scr, const f32x16& x, int lane, bool tile0,
                                                   int qr, int qc) {
    const int g = lane >> 5;
    float* row = scr + (lane & 31) * LP;
    // zero pads: lane group 0 writes [0,14) and [28,42), lane group 1 [56,70) and (again) [28,42)
    {
        float* z = row + (g ? 56 : 0);
#pragma unroll
        for (int i = 0; i < 14; ++i) { z[i] = 0.f; row[28 + i] = 0.f; }
    }
    wave_lds_fence();                  // the two lane groups of a query write overlapping zeros first
    // slots c = c0 + 4g, c0 = (r & 3) + 8 * (r >> 2): r < 8 -> vertical half, r >= 8 -> horizontal
    {
        float* wv = row + 14 + 4 * g;                  // slot c -> index 14 + c      (c = 0..13)
        float* wh = row + 42 - 16 + 4 * g;             // slot c -> index 42 + c - 16 (c = 16..29)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int c0 = (r & 3) + 8 * (r >> 2);
            // slot 14 (c0 = 10, g = 1) is the class-token key slot; slot 15 (c0 = 11, g = 1) is
            // never set in a key mask, its value is exactly 0 and lands in the zero pad
            if (c0 == 10) *(g ? row + 70 : wv + c0) = x[r];
            else wv[c0] = x[r];
        }
#pragma unroll
        for (int r = 8; r < 16; ++r) {
            const int c0 = (r & 3) + 8 * (r >> 2);     // 16..19, 24..27 (+4g): slots 30, 31 are exactly 0
            wh[c0] = x[r];
        }
    }
    wave_lds_fence();
    {
        const float* p = row + (g ? 28 + qc : qr);
        bk[0] = row[70];
#pragma unroll
        for (int u = 1; u < 30; ++u) bk[u] = p[u - 1];
        bk[30] = 0.f;
        bk[31] = 0.f;
    }
    if (tile0) {                        // class-token query: bucket 0 collects every key (row slots + cls)
        float sum = row[70];
#pragma unroll
        for (int c = 0; c < G14; ++c) sum += row[14 + c];
        const bool q0 = (lane & 31) == 0;
#pragma unroll
        for (int u = 0; u < 32; ++u) bk[u] = q0 ? (u == 0 ? sum : 0.f) : bk[u];
    }
    wave_lds_fence();                  // the row is reused by the caller
}

// ---- attention dropout: counter-based keep mask -----------------------------------------------------------------------
// P[i,j] -> keep[i,j] P[i,j] / (1 - p) AFTER the softmax normalisation (multihead_super.py:145, rpe_vision_transformer.py:86
// `attn = self.attn_drop(attn)`).  keep is a pure function of (seed, b * H + h, i, j) — a 32-bit avalanche mix (two
// multiply-xorshift rounds) of a per-(b, h) key and the pair (query i, key j) — so that the backward launches regenerate it
// instead of reading a mask from memory.  Shared by the AutoFormer kernels (attn_rpe2d.hip) and the iRPE kernels
// (irpe_attn.hip); cream_amd/irpe_fused.py `dropout_keep_mask` restates it in numpy for the tests.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t drop_key(uint32_t seed, int bh) { return mix32(seed ^ ((uint32_t)(bh + 1) * 0x9E3779B9u)); }
__device__ __forceinline__ bool drop_keep(uint32_t key, int i, int j, uint32_t thr) {
    return mix32(key ^ (((uint32_t)i << 16) | (uint32_t)j)) >= thr;
}
// keep iff hash >= thr = round(p 2^32) clamped to [1, 2^32 - 1]; 0 means "no dropout"
inline uint32_t drop_threshold(float p) {
    const double thr = (double)p * 4294967296.0;
    return p > 0.f ? (uint32_t)(thr < 1.0 ? 1.0 : (thr > 4294967295.0 ? 4294967295.0 : thr)) : 0u;
}

}  // namespace cream
