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

__device__ __forceinline__ short f2bf(float f) {            // round-to-nearest-even
    uint32_t x = __float_as_uint(f);
    x += 0x7fffu + ((x >> 16) & 1u);
    return (short)(x >> 16);
}
__device__ __forceinline__ float bf2f(short h) { return __uint_as_float(((uint32_t)(uint16_t)h) << 16); }

__device__ __forceinline__ int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

// slot pair of a key, packed (slotA | slotB << 8); 0xFF = none
__device__ __forceinline__ bool slot_hit(uint32_t packed, int c) {
    return (int)(packed & 0xFFu) == c || (int)((packed >> 8) & 0xFFu) == c;
}

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
        frag f;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = ok ? f2bf(p[e]) : (short)0;
        return f;
    }
    // operand built from accumulator registers [s*EPL, s*EPL+EPL) (the permuted-k trick)
    static __device__ __forceinline__ frag from_acc(const f32x16& a, int s) {
        frag f;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = f2bf(a[s * 8 + e]);
        return f;
    }
    // matching A operand: row pointer `rowp` of a TRANSPOSED tile ([d][n]), contraction
    // indices acc_row(s*8 + e, g): two runs of 4 consecutive columns (8-byte aligned)
    static __device__ __forceinline__ frag load_perm(const elem* rowp, int s, int g) {
        union { u32x2v v[2]; frag f; } u;
        u.v[0] = *reinterpret_cast<const u32x2v*>(rowp + 16 * s + 4 * g);
        u.v[1] = *reinterpret_cast<const u32x2v*>(rowp + 16 * s + 8 + 4 * g);
        return u.f;
    }
    // one-hot A operand, rows = keys: contraction slots c = ks*16 + 8g + e
    static __device__ __forceinline__ frag onehot_row(uint32_t packed, int ks, int g) {
        frag f;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = slot_hit(packed, ks * 16 + 8 * g + e) ? ONE : (short)0;
        return f;
    }
    // one-hot A operand, rows = slots (this lane: slot c), contraction = keys in the permuted
    // order of from_acc; `sl` points at the packed slots of the 32-key tile
    static __device__ __forceinline__ frag onehot_perm(const uint16_t* sl, int s, int g, int c) {
        union { u32x2v v; uint16_t h[4]; } lo, hi;
        lo.v = *reinterpret_cast<const u32x2v*>(sl + 16 * s + 4 * g);
        hi.v = *reinterpret_cast<const u32x2v*>(sl + 16 * s + 8 + 4 * g);
        frag f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f[e] = slot_hit(lo.h[e], c) ? ONE : (short)0;
            f[4 + e] = slot_hit(hi.h[e], c) ? ONE : (short)0;
        }
        return f;
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
    static __device__ __forceinline__ frag onehot_row(uint32_t packed, int ks, int g) {
        return slot_hit(packed, ks * 2 + g) ? 1.f : 0.f;
    }
    static __device__ __forceinline__ frag onehot_perm(const uint16_t* sl, int s, int g, int c) {
        return slot_hit(sl[acc_row(s, g)], c) ? 1.f : 0.f;
    }
};

// Geometry of AutoFormer's 2-D relative position index (multihead_super.py:40-62):
// token 0 = class token, tokens 1.. = gh x gw grid (row-major); bucket of (query i, key j)
//   vertical   : clamp(kr - qr, +-mr) + mr + 1     horizontal: clamp(kc - qc, +-mr) + mr + 1
//   bucket 0 when i == 0 or j == 0.
// Slots of a key: [0, gh) = grid row, gh = "class token key", [gh+1, gh+1+gw) = grid column;
// the fused kernels need gh + gw + 1 <= 32 and 2*mr + 2 <= 32.
struct RelGeom {
    int n;          // tokens (gh*gw + 1)
    int gh, gw;     // grid
    int mr;         // max_relative_position
};

// packed slots of key j (0xFFFF for padding keys j >= n)
__device__ __forceinline__ uint16_t key_slots(int j, const RelGeom& G) {
    if (j >= G.n) return 0xFFFFu;
    if (j == 0) return (uint16_t)(G.gh | 0xFF00u);
    const int r = (j - 1) / G.gw, c = (j - 1) - r * G.gw;
    return (uint16_t)(r | ((G.gh + 1 + c) << 8));
}

constexpr int LP = 65;      // pitch (floats) of the per-wave [32][64] shift scratch rows

// x_i[c]: extension of query i for slot c, from its bucket lookups row[0..31] (vertical
// table) and row[32..63] (horizontal table).  Used for the key-side bias (row = q.T_k^T)
// and, in backward, for d(slot sums) (row = dO.T_v^T).
__device__ __forceinline__ float ext_gather(const float* row, int c, int qi, int qr, int qc,
                                            const RelGeom& G) {
    const float cls = row[0] + row[32];
    if (qi == 0) return c <= G.gh ? cls : 0.f;
    if (c < G.gh) return row[clampi(c - qr, -G.mr, G.mr) + G.mr + 1];
    if (c == G.gh) return cls;
    if (c <= G.gh + G.gw) return row[32 + clampi(c - G.gh - 1 - qc, -G.mr, G.mr) + G.mr + 1];
    return 0.f;
}

// adjoint of ext_gather: add the slot value x (slot c of query i) into the bucket rows
__device__ __forceinline__ void ext_scatter(float* row, int c, float x, int qi, int qr, int qc,
                                            const RelGeom& G) {
    if (qi == 0) {
        if (c <= G.gh) { row[0] += x; row[32] += x; }
        return;
    }
    if (c < G.gh) row[clampi(c - qr, -G.mr, G.mr) + G.mr + 1] += x;
    else if (c == G.gh) { row[0] += x; row[32] += x; }
    else if (c <= G.gh + G.gw) row[32 + clampi(c - G.gh - 1 - qc, -G.mr, G.mr) + G.mr + 1] += x;
}

// make LDS traffic of this wave visible to its own later reads (in-order LDS, compiler fence)
__device__ __forceinline__ void wave_lds_fence() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

}  // namespace cream
