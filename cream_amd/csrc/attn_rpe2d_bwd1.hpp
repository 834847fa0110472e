// attn_rpe2d_bwd1.hpp — ONE-PASS backward of the fused attention for the AutoFormer geometry (N = 197, 14 x 14 grid,
// max_relative_position 14, bf16).  Included by attn_rpe2d.hip inside its anonymous namespace (uses BwdArgs and the
// helpers of attn_common.hpp); reference semantics: AutoFormer/model/module/multihead_super.py:133-160, SURVEY App. B.
//
// Why one pass.  The two-launch backward (attn_rpe2d_bwd_q / _bwd_kv) computes the score tile and dP twice (38 MFMAs per
// (query tile, key tile) pair against 26 here), re-reads q, k, v, dO in both launches and hands slot extensions / bucket
// gradients / delta from the first launch to the second through HBM (354 MB per launch pair against ~155 MB algorithmic).
// Here one workgroup (7 waves) owns one (b, h) with K, V, Q and dO WHOLE in LDS (4 x 28 KB, XOR-swizzled, no padding),
// and every wave plays two roles:
//   * OWNER OF QUERY TILE w ("producer"): in step s it takes key tile j = (w + s) mod 7, computes S^T and dP^T
//     (12 MFMAs, lanes own queries), P and dS in registers, accumulates dQx^T += Kx_j^T dS^T (6 MFMAs) and publishes
//     P and dS as two 32 x 32 bf16 tiles in LDS ([query][key] rows);
//   * OWNER OF KEY TILE w ("consumer"): it picks up the pair of tiles published for its key tile in the previous step
//     and accumulates dV_w^T += dO^T P and dK_w^T += Q^T dS (8 MFMAs; both operands through ds_read_b64_tr_b16, the
//     products contract over queries).
// The rotation (w + s) mod 7 gives every key tile exactly one tile pair per step, so the exchange buffer is 7 slots, and
// no gradient is ever reduced across waves: dQ lives in its query owner's registers, dK / dV in their key owner's.
// Everything a launch used to hand to the next one stays on chip: delta, lse and the slot extensions are per-lane
// registers of the query owner; the bucket gradients dL' go through the exchange slots to the table-gradient jobs.
// The table gradients are accumulated per workgroup in global memory (read-modify-write by the owning lanes, fixed
// order: bit-reproducible), so no accumulator registers are pinned across items.
//
// LDS (161,280 B, one workgroup per CU):
//   K | V | Q | dO      4 x [224][64] bf16, 16-byte chunks XOR-swizzled with swz128(row)
//   OH                  [224][32] bf16 one-hot slot rows of the keys (chunk ^ ((row >> 2) & 3))
//   7 exchange slots    4608 B each: P tile | dS tile ([32 q][32 keys] bf16, 8-byte columns ^ ((q >> 1) & 7));
//                       the slot of wave w doubles as its shift scratch ([32][72] bf16) in the item prologue /
//                       epilogue and carries its dL' tile ([32 q][64 buckets] bf16) to the table-gradient jobs.
// The bucket tables are read from bf16 operand images in global memory (table_images_kernel below; 32 KB, L2 /
// L1 resident) — LDS is full.
#pragma once

// phase stamps for tools/probes/attn_bwd1_probe.hip (compiled out of the library): 20 slots per wave
#ifdef ATTN_PROFILE
#define V2_PROF_DECL long long prof_t[20]; int prof_n = 0;
#define V2_PROF_FLUSH() do { if ((threadIdx.x & 63) == 0 && g_attn_prof) { \
        long long* d_ = g_attn_prof + ((long long)blockIdx.x * 8 + (threadIdx.x >> 6)) * 20; \
        for (int i_ = 0; i_ < 20; ++i_) d_[i_] = i_ < prof_n ? prof_t[i_] : 0; } } while (0)
#else
#define V2_PROF_DECL
#define V2_PROF_FLUSH() do {} while (0)
#endif
#ifdef ATTN_PROFILE_STEPS                    // sub-phase stamps inside step 3 of the one-pass backward
#define PROF_STEP(s) do { if ((s) == 3) PROF_MARK(); } while (0)
#else
#define PROF_STEP(s) do {} while (0)
#endif

#ifndef BWD1_PREFETCH
#define BWD1_PREFETCH 0
#endif
// timing experiments of tools/probes/attn_bwd1_probe.hip (results are WRONG with either switch): leave out the global
// stores of dq / dk / dv, or the loads of the next item's matrices (the LDS keeps the first item's)
#ifndef BWD1_EXP_NOSTORE
#define BWD1_EXP_NOSTORE 0
#endif
#ifndef BWD1_EXP_NOLOAD
#define BWD1_EXP_NOLOAD 0
#endif

namespace v2 {

constexpr int NT = 7, N14 = 197, NP14 = 224, THREADS = 448;
constexpr int MAT_B = NP14 * 128;                    // one [224][64] bf16 matrix
constexpr int OFF_K = 0, OFF_V = MAT_B, OFF_Q = 2 * MAT_B, OFF_D = 3 * MAT_B;
constexpr int OFF_OH = 4 * MAT_B, OH_B = NP14 * 64;
constexpr int OFF_X = OFF_OH + OH_B;
constexpr int SLOT_B = 4608, XT_B = 2048;            // slot = P tile | dS tile | pad  (scratch rows: 72 bf16 = 144 B)
constexpr int OFF_SINK = OFF_X + NT * SLOT_B;        // 1 KB landing zone of the L2 prefetch (never read)
constexpr int LDS_B = OFF_SINK + 1024;
constexpr int SCRP = 72;                             // shift scratch pitch (bf16 elements)
static_assert(LDS_B <= 160 * 1024, "LDS budget");
static_assert(32 * SCRP * 2 <= SLOT_B, "shift scratch fits the wave's exchange slot");

// image layout (bf16 elements): key rows [64 u'][64 d] | key^T [64 d][64 u'] | value rows | value^T,
// u' = bucket of the vertical table (0..31) or 32 + bucket of the horizontal table; rows >= nb are zero
constexpr int IMG_KR = 0, IMG_KT = 4096, IMG_VR = 8192, IMG_VT = 12288, IMG_ELEMS = 16384;

__global__ __launch_bounds__(256) void table_images_kernel(short* img, const float* tkv, const float* tkh, const float* tvv,
                                                           const float* tvh, int ldt, int nb) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * 4096; i += gridDim.x * blockDim.x) {
        const int pair = i >> 12, e = i & 4095, u2 = e >> 6, d = e & 63, u = u2 & 31;
        const float* t = pair == 0 ? (u2 < 32 ? tkv : tkh) : (u2 < 32 ? tvv : tvh);
        const short x = u < nb ? f2bf(t[(int64_t)u * ldt + d]) : (short)0;
        img[pair * 8192 + u2 * 64 + d] = x;                  // rows
        img[pair * 8192 + 4096 + d * 64 + u2] = x;           // transposed
    }
}

__device__ __forceinline__ int swz128(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }

typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x4 lds_tr16(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        reinterpret_cast<__attribute__((address_space(3))) s16x4*>(reinterpret_cast<uintptr_t>(p)));
}
__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* p0, const unsigned char* p1) {
    const s16x4 lo = lds_tr16(p0), hi = lds_tr16(p1);
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ bf16x8 lds_b128(const unsigned char* p) {
    union { u32x4v v; bf16x8 f; } u;
    u.v = *reinterpret_cast<const u32x4v*>(p);
    return u.f;
}
__device__ __forceinline__ bf16x8 mk8(const short (&x)[8]) { return bf16x8{x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]}; }
__device__ __forceinline__ f32x16 mma16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

// per-lane LDS offsets (bytes) that do not depend on the tile: computed once per kernel
struct LaneOffs {
    int row[4];        // A / B operand rows of a [32][64] tile: lane = row c32, k-step ks -> 16 bytes
    int ohrow[2];      // the same for a [32][32] one-hot tile
    int tr[2][2];      // ds_read_b64_tr_b16 of a [32][64] tile: [dt][h] (+ 2048 per k-step of 16 tokens)
    int ohtr[2];       // the same for a [32][32] one-hot tile     (+ 1024 per k-step)
    int xr[2];         // the same for an exchange tile            (+ 1024 per k-step)
    int xw[2][2];      // exchange tile write: [st][h], 8 bytes
};
__device__ __forceinline__ LaneOffs lane_offs(int lane) {
    LaneOffs o;
    const int c32 = lane & 31, g = lane >> 5, gi = lane & 15, q4 = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) o.row[ks] = c32 * 128 + (((2 * ks + g) ^ swz128(c32)) << 4);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) o.ohrow[ks] = c32 * 64 + (((2 * ks + g) ^ ((c32 >> 2) & 3)) << 4);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int rh = 4 * g + (gi >> 2) + 8 * h;                  // token row inside a 16-token k-step
        const int inner = 8 * (gi & 1), c0 = 2 * (q4 & 1) + ((gi & 3) >> 1);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) o.tr[dt][h] = rh * 128 + (((4 * dt + c0) ^ swz128(rh)) << 4) + inner;
        o.ohtr[h] = rh * 64 + ((c0 ^ ((rh >> 2) & 3)) << 4) + inner;
        o.xr[h] = rh * 64 + (((4 * (q4 & 1) + (gi & 3)) ^ ((rh >> 1) & 7)) << 3);
#pragma unroll
        for (int st = 0; st < 2; ++st) o.xw[st][h] = c32 * 64 + (((4 * st + 2 * h + g) ^ ((c32 >> 1) & 7)) << 3);
    }
    return o;
}

__device__ __forceinline__ void launder_offs(LaneOffs& o) {
    int* p = reinterpret_cast<int*>(&o);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(LaneOffs) / sizeof(int)); ++i) asm volatile("" : "+v"(p[i]));
}

// workgroup barrier for LDS traffic only: ds operations are retired (lgkmcnt), global loads and the prefetch DMA stay in
// flight across it (__syncthreads() would also drain vmcnt)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// one 1 KB global -> LDS DMA read (16 bytes per lane, LDS destination = wave-uniform `lds_dst` + 16 lane) as INLINE ASM:
// hipcc orders every later ds_read behind a builtin LDS-DMA with s_waitcnt vmcnt(0) (it is a pending LDS write), which
// would drain the prefetch at every step; an asm statement is outside its bookkeeping (cdna_hip_programming.md 5.7).
// Nothing ever reads the destination, so no wait belongs to it; M0 is restored.
__device__ __forceinline__ void prefetch_dma(const short* src, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_dst));
}

// ---- the item's matrices travel global -> LDS by DMA (no register staging) --------------------------------------------
// A register-staged prefetch of the next item does not survive here: with ~230 live registers the compiler waits for every
// prefetched vector at once and parks it in scratch (round-4 builds: the "prefetch" was a load, a wait, a scratch store and
// a scratch reload).  LDS-DMA needs no registers: one wave instruction moves 1 KB = 8 rows x 128 B into a lane-linear image;
// the chunk swizzle of the tile (chunk ^ swz128(row)) is applied to the SOURCE address, rows beyond N come from a zero line.
__device__ __attribute__((aligned(16))) const uint32_t g_zero_line[4] = {0u, 0u, 0u, 0u};
__device__ __forceinline__ void dma_1k(const short* src, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
}
// rows [0, 224) of one matrix (row stride rs elements) into the LDS region at byte address lds_base: 28 pieces, 4 per wave
__device__ __forceinline__ void mat_dma(const short* src, int64_t rs, uint32_t lds_base, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = wave + NT * i, row = piece * 8 + (lane >> 3), cc = lane & 7;
        const short* s = row < N14 ? src + (int64_t)row * rs + ((cc ^ swz128(row)) << 3) : reinterpret_cast<const short*>(g_zero_line);
        dma_1k(s, lds_base + piece * 1024);
    }
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// one-hot slot rows of the keys (geometry only): OH[j][c] = 1 iff key j sits in slot c
__device__ __forceinline__ void fill_onehot_swz(unsigned char* oh) {
    const RelGeom G{N14, G14, G14, G14};
    for (int i = threadIdx.x; i < NP14 * 4; i += THREADS) {
        const int j = i >> 2, cc = i & 3;
        // keys >= N carry slot 15 (unused by the geometry): the query extension holds -2^15 there, so their
        // probabilities underflow to exactly 0 without a select per score (slot 15 of every gradient stays exactly 0)
        const uint32_t m = (key_mask(j, G) | (j >= N14 ? 1u << 15 : 0u)) >> (8 * cc);
        u32x4v w;
#pragma unroll
        for (int p = 0; p < 4; ++p) w[p] = ((m >> (2 * p)) & 1u) * 0x3F80u + ((m >> (2 * p + 1)) & 1u) * 0x3F800000u;
        *reinterpret_cast<u32x4v*>(oh + j * 64 + ((cc ^ ((j >> 2) & 3)) << 4)) = w;
    }
}

// ---- slot <-> bucket shifts through the wave's bf16 scratch ([32 queries][SCRP]) ---------------------------------
// lookups^T (buckets x queries) of the vertical / horizontal table against this lane's row fragments xb, then the
// window shift of attn_common.hpp (ext_window14) -> slot extension fragments xe[2] (B operand, lane = query).
// `rows` = bf16 image [64 u'][64 d] in global memory.  Rounding the lookups to bf16 before the shift gives the same
// bits as rounding the shifted values (the shift is a permutation); the class-token value is summed in fp32 first.
// one lookup set -> scratch -> window shift (attn_common.hpp: ext_window14).  slot15 = raw bf16 placed in the unused slot 15
// (0, or -2^15 for the key-side extension: padding-key mask)
__device__ __forceinline__ void ext_from_lookups14(bf16x8 (&xe)[2], const f32x16& av, const f32x16& ah, unsigned char* scr, int lane,
                                                   bool tile0, int qr, int qc, short slot15) {
    const int c32 = lane & 31, g = lane >> 5;
    unsigned char* row = scr + c32 * (SCRP * 2);
    // buckets acc_row(4i + e, g) = 8i + 4g + e: four consecutive bf16 per store
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<u32x2v*>(row + (8 * i + 4 * g) * 2) =
            u32x2v{f2bf_pair(av[4 * i], av[4 * i + 1]), f2bf_pair(av[4 * i + 2], av[4 * i + 3])};
        *reinterpret_cast<u32x2v*>(row + (32 + 8 * i + 4 * g) * 2) =
            u32x2v{f2bf_pair(ah[4 * i], ah[4 * i + 1]), f2bf_pair(ah[4 * i + 2], ah[4 * i + 3])};
    }
    if (g == 0) *reinterpret_cast<float*>(row + 128) = av[0] + ah[0];          // bucket 0 of both tables (fp32)
    wave_lds_fence();
    const short cls = f2bf(*reinterpret_cast<const float*>(row + 128));
    if (tile0) {                                                               // class-token query (attn_common.hpp: ext_fix_query0)
        if (c32 == 0 && g == 0) {
            short* r = reinterpret_cast<short*>(row);
#pragma unroll
            for (int u = 0; u < 16; ++u) { r[15 + u] = cls; r[32 + 15 + u] = 0; }
        }
        wave_lds_fence();
    }
    const short* r = reinterpret_cast<const short*>(row);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        const short* p = r + (kh == 0 ? 15 - qr : 32 + 15 - qc) + 8 * g;
        short x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = p[e];
        x[6] = g ? (kh == 0 ? cls : (short)0) : x[6];
        x[7] = g ? (kh == 0 ? slot15 : (short)0) : x[7];
        xe[kh] = mk8(x);
    }
    wave_lds_fence();
}

// slot extensions of q (key tables) and of dO (value tables): lookups^T (buckets x queries) of the vertical / horizontal
// table against this lane's row fragments, then the window shifts through the wave's scratch.  `img` = the bf16 images
// in global memory; all 16 row fragments are requested before the first MFMA.  Rounding the lookups to bf16 before the
// shift gives the same bits as rounding the shifted values (the shift is a permutation); the class-token value is
// summed in fp32 first.
struct TabFrags { bf16x8 tk[2][4], tv[2][4]; };       // row fragments of the key / value table images (vertical, horizontal)
__device__ __forceinline__ void load_tab_frags(TabFrags& f, const short* img, int lane) {
    const int c32 = lane & 31, g = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f.tk[t][ks] = *reinterpret_cast<const bf16x8*>(img + IMG_KR + (32 * t + c32) * 64 + ks * 16 + g * 8);
            f.tv[t][ks] = *reinterpret_cast<const bf16x8*>(img + IMG_VR + (32 * t + c32) * 64 + ks * 16 + g * 8);
        }
}
__device__ __forceinline__ void lookups_ext14_pair(bf16x8 (&qe)[2], bf16x8 (&de)[2], const bf16x8 (&qb)[4], const bf16x8 (&dob)[4],
                                                   const TabFrags& tf, unsigned char* scr, int lane, bool tile0, int qr, int qc) {
    const bf16x8 (&tk)[2][4] = tf.tk;
    const bf16x8 (&tv)[2][4] = tf.tv;
    f32x16 kv = {}, kh = {}, vv = {}, vh = {};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        kv = mma16(tk[0][ks], qb[ks], kv);
        kh = mma16(tk[1][ks], qb[ks], kh);
        vv = mma16(tv[0][ks], dob[ks], vv);
        vh = mma16(tv[1][ks], dob[ks], vh);
    }
    ext_from_lookups14(qe, kv, kh, scr, lane, tile0, qr, qc, (short)0xC700);
    ext_from_lookups14(de, vv, vh, scr, lane, tile0, qr, qc, (short)0);
}

// adjoint: slot tile (accumulator: this lane holds slots acc_row(r, g) of its query) -> the 32 bucket values of table g
// (0 vertical / 1 horizontal) of this lane's query as four bf16 operand fragments bk[ks] = buckets 8 ks .. 8 ks + 7
// (attn_common.hpp: slots_to_buckets14; the class-token query's sum is taken in fp32 from the registers)
__device__ __forceinline__ void slots_to_buckets14_bf16(bf16x8 (&bk)[4], unsigned char* scr, const f32x16& x, int lane,
                                                        bool tile0, int qr, int qc) {
    const int c32 = lane & 31, g = lane >> 5;
    unsigned char* rowb = scr + c32 * (SCRP * 2);
    short* row = reinterpret_cast<short*>(rowb);
    {
        // zero pads: lane group 0 writes [0,14) and [28,42), lane group 1 [56,70) and (again) [28,42)
        uint32_t* z = reinterpret_cast<uint32_t*>(rowb + (g ? 112 : 0));
        uint32_t* z2 = reinterpret_cast<uint32_t*>(rowb + 56);
#pragma unroll
        for (int i = 0; i < 7; ++i) { z[i] = 0u; z2[i] = 0u; }
    }
    wave_lds_fence();
    {
        // vertical half: slots c = c0 + 4g, c0 = (r & 3) + 8 (r >> 2), r < 8  -> index 14 + c; slot 14 (c0 = 10, g = 1)
        // is the class-token key slot (index 70), slot 15 is exactly 0
#pragma unroll
        for (int rp = 0; rp < 4; ++rp) {
            const int r0 = 2 * rp, c0 = (r0 & 3) + 8 * (r0 >> 2);
            float lo = x[r0], hi = x[r0 + 1];
            if (c0 == 10) { lo = g ? 0.f : lo; hi = g ? 0.f : hi; }
            *reinterpret_cast<uint32_t*>(rowb + (14 + c0 + 4 * g) * 2) = f2bf_pair(lo, hi);
        }
        if (g) row[70] = f2bf(x[6]);
        // horizontal half: slots 16 + c', r >= 8 -> index 42 + c - 16 (slots 30, 31 are exactly 0 and land in the pad)
#pragma unroll
        for (int rp = 4; rp < 8; ++rp) {
            const int r0 = 2 * rp, c0 = (r0 & 3) + 8 * (r0 >> 2);      // 16,18,24,26
            *reinterpret_cast<uint32_t*>(rowb + (42 - 16 + c0 + 4 * g) * 2) = f2bf_pair(x[r0], x[r0 + 1]);
        }
    }
    wave_lds_fence();
    short b[32];
    {
        const short* p = row + (g ? 28 + qc : qr);
        b[0] = row[70];
#pragma unroll
        for (int u = 1; u < 30; ++u) b[u] = p[u - 1];
        b[30] = 0;
        b[31] = 0;
    }
    if (tile0) {                                    // class-token query: bucket 0 collects every key (row slots + cls slot)
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) sum += x[r];    // slots 0..15 of the vertical half (slot 15 is exactly 0)
        sum += __shfl_xor(sum, 32);
        const bool q0 = c32 == 0;
        const short s16 = f2bf(sum);
#pragma unroll
        for (int u = 0; u < 32; ++u) b[u] = q0 ? (u == 0 ? s16 : (short)0) : b[u];
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        bk[ks] = bf16x8{b[8 * ks], b[8 * ks + 1], b[8 * ks + 2], b[8 * ks + 3], b[8 * ks + 4], b[8 * ks + 5], b[8 * ks + 6], b[8 * ks + 7]};
    wave_lds_fence();                               // the slot is reused by the caller
}

// accumulator tile (lane = token row c32 of this wave's 32-token tile, 64 d-values) -> global rows, COALESCED: the tile goes
// through a wave-private 4 KB LDS stage ([32][128 B], chunks ^ swz128) and leaves as whole 128-byte rows (8 lanes x 16 B
// per row, 8 rows per wave instruction).  Storing straight from the accumulator layout (a lane owns a row, 32 rows at a
// 2.3 KB stride per instruction, 32 bytes each) made the write-back of dq / dk / dv the slowest phase of an item: vmcnt
// retires in order, so the first load consumer behind those stores waited ~20k cycles for them (ablation in
// tools/probes/attn_bwd1_probe: an item 81k -> 52k cycles without the stores).
// The two lane groups of a row hold alternating runs of 4 d-values; one v_permlane32_swap per dword pairs them up into
// 16-byte pieces (cdna_hip_programming.md T21).  `tile_row0` = first token of the tile; rows >= N14 are not written.
__device__ __forceinline__ void store_tile_staged(unsigned char* stage, short* gbase, int64_t rs, int tile_row0,
                                                  const f32x16 (&o)[2], int lane) {
    const int c32 = lane & 31, g = lane >> 5;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int k = 0; k < 4; k += 2) {
            uint32_t ax = f2bf_pair(o[dt][4 * k], o[dt][4 * k + 1]), ay = f2bf_pair(o[dt][4 * k + 2], o[dt][4 * k + 3]);
            uint32_t bx = f2bf_pair(o[dt][4 * k + 4], o[dt][4 * k + 5]), by = f2bf_pair(o[dt][4 * k + 6], o[dt][4 * k + 7]);
            const auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
            const auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
            // lanes 0-31: [own run k | partner's run k] = d 8k .. 8k+7 (chunk 4 dt + k);  lanes 32-63: chunk 4 dt + k + 1
            *reinterpret_cast<u32x4v*>(stage + c32 * 128 + (((4 * dt + k + g) ^ swz128(c32)) << 4)) = u32x4v{rx[0], ry[0], rx[1], ry[1]};
        }
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = i * 8 + (lane >> 3), cc = lane & 7;
        const u32x4v v = *reinterpret_cast<const u32x4v*>(stage + row * 128 + ((cc ^ swz128(row)) << 4));
        if (tile_row0 + row < N14) *reinterpret_cast<u32x4v*>(gbase + (int64_t)row * rs + cc * 8) = v;
    }
    wave_lds_fence();                               // the stage may be rewritten by the caller
}

__global__ __launch_bounds__(THREADS) void attn_rpe2d_bwd1_kernel(const BwdArgs a, const short* img) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const float sc = a.scale * LOG2E;
    const int64_t orow = (int64_t)a.H * 64;

    struct Item {
        const short *qp, *kpg, *vpg, *dop, *outp;
        int64_t bh;
        int b, h;
    };
    auto item_of = [&](int item) {
        Item I;
        I.b = item / a.H;
        I.h = item - I.b * a.H;
        I.bh = (int64_t)I.b * a.H + I.h;
        const int64_t base = (int64_t)I.b * a.sb + (int64_t)I.h * a.sh;
        I.qp = reinterpret_cast<const short*>(a.q) + base;
        I.kpg = reinterpret_cast<const short*>(a.k) + base;
        I.vpg = reinterpret_cast<const short*>(a.v) + base;
        I.dop = reinterpret_cast<const short*>(a.dout) + ((int64_t)I.b * N14 * a.H + I.h) * 64;
        I.outp = reinterpret_cast<const short*>(a.out) + ((int64_t)I.b * N14 * a.H + I.h) * 64;
        return I;
    };

    fill_onehot_swz(smem + OFF_OH);

    // STAGGER.  All workgroups of a launch do the same work on equal items, so they ask HBM for their 137 KB and write their
    // 75 KB at the same moments — a convoy: every memory phase runs at 1 / 256 of the chip's bandwidth while HBM idles
    // under the step loops.  Eight phase groups (workgroups b, b + 64, ... share an XCD: the groups cut ACROSS the XCDs, so
    // each XCD's fabric link sees an eighth of its workgroups at a time) start `stagger` x 64 cycles apart.
    if (a.stagger > 0) {
        const int grp = (blockIdx.x >> 3) & 7;
        for (int i = 0; i < grp * a.stagger; ++i) __builtin_amdgcn_s_sleep(1);
    }

    // ---- software pipeline over the workgroup's items: the NEXT item's operands are requested during the epilogue of
    // the current one (K, V as soon as the step loop is over — their LDS regions are dead then —, Q, dO, this lane's
    // row of O and its softmax statistic before the table-gradient jobs), so an item starts with its loads landed.
    // All workgroups of a launch run in lockstep: without this every CU asked HBM for its 112 KB at the same moment and
    // the load phase was 20k of an item's 90k cycles (phase stamps, tools/probes/attn_bwd1_probe).
    int item = blockIdx.x;
    if (item >= a.nitems) return;
    Item I = item_of(item);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>(smem));
    {
        const int w0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l0 = threadIdx.x & 63;
        mat_dma(I.kpg, a.sn, lds0 + OFF_K, w0, l0);
        mat_dma(I.vpg, a.sn, lds0 + OFF_V, w0, l0);
        mat_dma(I.qp, a.sn, lds0 + OFF_Q, w0, l0);
        mat_dma(I.dop, orow, lds0 + OFF_D, w0, l0);
    }

    for (;;) {
        V2_PROF_DECL
        PROF_MARK();
        // Everything derived from the thread index is recomputed per item from an OPAQUE copy: the compiler would
        // otherwise hoist every LDS / global address of the loop body out of the item loop as an invariant — hundreds of
        // address registers, spilled around every phase (the first build: 160 spilled VGPRs, reloads in the epilogue).
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int lane = tid & 63, g = lane >> 5, c32 = lane & 31;
        const LaneOffs lo = lane_offs(lane);
        const int qi = wave * 32 + c32;              // this lane's query (producer role) and key (consumer role)
        const bool tok_ok = qi < N14;
        const int qcl = min(qi, N14 - 1);
        const int qr = qi > 0 ? (qi - 1) / G14 : 0, qc = qi > 0 ? (qi - 1) - qr * G14 : 0;
        unsigned char* myslot = smem + OFF_X + wave * SLOT_B;
        const int b = I.b, h = I.h;
        const int64_t bh = I.bh;
        // this lane's row of O (for delta) and its softmax statistic
        bf16x8 ob[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ob[ks] = *reinterpret_cast<const bf16x8*>(I.outp + (int64_t)qcl * orow + ks * 16 + g * 8);
        const float lse_r = a.lse[bh * N14 + qcl];
        TabFrags tf;                                 // (requested here: their round trip runs under the wait for the matrices)
        load_tab_frags(tf, img, lane);
        dma_wait_all();                              // this wave's pieces of K, V, Q, dO have landed ...
        __syncthreads();                             // ... and everybody's
        const float m2 = tok_ok ? lse_r * LOG2E : INFINITY;                    // padding queries: P = 0
        PROF_MARK();

        // ---- query-owner prologue: delta, slot extensions of q and dO ----------------------------------------
        bf16x8 qe[2], de[2];
        float dsc;                                   // scale * delta_i
        {
            bf16x8 qb[4], dob[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                qb[ks] = lds_b128(smem + OFF_Q + wave * 4096 + lo.row[ks]);
                dob[ks] = lds_b128(smem + OFF_D + wave * 4096 + lo.row[ks]);
            }
            float delta = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) delta += bf2f(dob[ks][e]) * bf2f(ob[ks][e]);
            delta += __shfl_xor(delta, 32);
            dsc = tok_ok ? delta * a.scale : 0.f;
            lookups_ext14_pair(qe, de, qb, dob, tf, myslot, lane, wave == 0, qr, qc);
        }

        PROF_MARK();
        f32x16 dq[2] = {f32x16{}, f32x16{}}, dx = {};
        f32x16 dk[2] = {f32x16{}, f32x16{}}, dv[2] = {f32x16{}, f32x16{}};

        // consumer: the tile pair published for key tile `wave` by the owner of query tile qt
        auto consume = [&](int qt) {
            const unsigned char* xp = myslot;                               // P tile, then dS tile
            const unsigned char* dt_ = smem + OFF_D + qt * 4096;
            const unsigned char* qt_ = smem + OFF_Q + qt * 4096;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const bf16x8 pb = tr_pair(xp + st * 1024 + lo.xr[0], xp + st * 1024 + lo.xr[1]);
                const bf16x8 db = tr_pair(xp + XT_B + st * 1024 + lo.xr[0], xp + XT_B + st * 1024 + lo.xr[1]);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dv[dt] = mma16(tr_pair(dt_ + st * 2048 + lo.tr[dt][0], dt_ + st * 2048 + lo.tr[dt][1]), pb, dv[dt]);
                    dk[dt] = mma16(tr_pair(qt_ + st * 2048 + lo.tr[dt][0], qt_ + st * 2048 + lo.tr[dt][1]), db, dk[dt]);
                }
            }
        };

        // L2 PREFETCH of the next item.  Its K, V, Q, dO and O rows cannot land anywhere while this item runs (LDS and
        // registers are full), but all workgroups run in lockstep, so without help every CU asks HBM for its 137 KB in the
        // same few microseconds after the step loop (the loads were 30k of a middle item's 82k cycles).  During the step
        // loop — when HBM is otherwise idle — every wave therefore issues three 1 KB global -> LDS DMA reads per step
        // into a sink that is never read: the lines are in L2 / the memory-side cache when the real loads come.
        const int next = item + (int)gridDim.x;
        const bool more = next < a.nitems;
        Item In = I;
        if (more) In = item_of(next);
        const int pf_row = lane >> 3, pf_col = (lane & 7) * 8;
        const uint32_t sink = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>(smem + OFF_SINK));

#pragma unroll 1
        for (int s = 0; s < NT; ++s) {
            const int j = wave + s < NT ? wave + s : wave + s - NT;         // this step's key tile (wave-uniform)
            PROF_STEP(s);
            if (BWD1_PREFETCH && more) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int p = (s * NT + wave) * 3 + i;                  // 0 .. 146: 25 pieces of 8 rows per matrix (5 matrices)
                    const int m = p / 25, piece = p - m * 25;
                    if (m < 5) {
                        const short* base = m == 0 ? In.kpg : m == 1 ? In.vpg : m == 2 ? In.qp : m == 3 ? In.dop : In.outp;
                        const int64_t rs = m < 3 ? a.sn : orow;
                        const short* src = base + (int64_t)min(piece * 8 + pf_row, N14 - 1) * rs + pf_col;
                        prefetch_dma(src, sink);
                    }
                }
            }
            // tiles published in step s - 1 for key tile `wave` come from the owner of query tile (wave - (s - 1)) mod 7
            if (s > 0) consume(wave - (s - 1) >= 0 ? wave - (s - 1) : wave - (s - 1) + NT);
            PROF_STEP(s);
            __builtin_amdgcn_sched_barrier(0);
            // ---- producer: S^T and dP^T of (key tile j, query tile wave) ---------------------------------------
            const unsigned char* kt = smem + OFF_K + j * 4096;
            const unsigned char* vt = smem + OFF_V + j * 4096;
            const unsigned char* oh = smem + OFF_OH + j * 2048;
            const unsigned char* qrow = smem + OFF_Q + wave * 4096;
            const unsigned char* drow = smem + OFF_D + wave * 4096;
            f32x16 sacc = {}, pacc = {};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                sacc = mma16(lds_b128(kt + lo.row[ks]), lds_b128(qrow + lo.row[ks]), sacc);
                pacc = mma16(lds_b128(vt + lo.row[ks]), lds_b128(drow + lo.row[ks]), pacc);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 o1 = lds_b128(oh + lo.ohrow[ks]);
                sacc = mma16(o1, qe[ks], sacc);
                pacc = mma16(o1, de[ks], pacc);
            }
            PROF_STEP(s);
            __builtin_amdgcn_sched_barrier(0);
            // P = exp2(S sc - m2), dS = P (dP scale - delta scale); keys >= N: P = 0 through slot 15 (fill_onehot_swz)
            uint32_t pw[8], dw[8];                   // bf16 pairs: operand fragments AND the published tiles
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[r], sc, -m2));
                const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[r + 1], sc, -m2));
                pw[r >> 1] = f2bf_pair(p0, p1);
                dw[r >> 1] = f2bf_pair(p0 * __builtin_fmaf(pacc[r], a.scale, -dsc), p1 * __builtin_fmaf(pacc[r + 1], a.scale, -dsc));
            }
            bf16x8 pb[2], db[2];
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                pb[st] = __builtin_bit_cast(bf16x8, (u32x4v{pw[4 * st], pw[4 * st + 1], pw[4 * st + 2], pw[4 * st + 3]}));
                db[st] = __builtin_bit_cast(bf16x8, (u32x4v{dw[4 * st], dw[4 * st + 1], dw[4 * st + 2], dw[4 * st + 3]}));
            }
            PROF_STEP(s);
            __builtin_amdgcn_sched_barrier(0);
            // dQx^T += Kx_j^T dS^T
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                dq[0] = mma16(tr_pair(kt + st * 2048 + lo.tr[0][0], kt + st * 2048 + lo.tr[0][1]), db[st], dq[0]);
                dq[1] = mma16(tr_pair(kt + st * 2048 + lo.tr[1][0], kt + st * 2048 + lo.tr[1][1]), db[st], dq[1]);
                dx = mma16(tr_pair(oh + st * 1024 + lo.ohtr[0], oh + st * 1024 + lo.ohtr[1]), db[st], dx);
            }
            PROF_STEP(s);
            lds_barrier();                           // every consumer has read the tiles of step s - 1
            PROF_STEP(s);
            // publish P and dS for the owner of key tile j: [32 q][32 keys], lane = query row, four runs of 4 keys
            {
                unsigned char* xs = smem + OFF_X + j * SLOT_B;
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        *reinterpret_cast<u32x2v*>(xs + lo.xw[st][hh]) = u32x2v{pw[4 * st + 2 * hh], pw[4 * st + 2 * hh + 1]};
                        *reinterpret_cast<u32x2v*>(xs + XT_B + lo.xw[st][hh]) = u32x2v{dw[4 * st + 2 * hh], dw[4 * st + 2 * hh + 1]};
                    }
            }
            PROF_STEP(s);
            lds_barrier();
            PROF_STEP(s);
        }
        PROF_MARK();

        // ---- epilogue ------------------------------------------------------------------------------------------
        // K and V are dead (every wave is past its last producer phase): the next item's K, V start travelling NOW, by
        // DMA straight into their LDS regions; Q and dO follow after the table-gradient jobs (their last readers)
        if (more && !BWD1_EXP_NOLOAD) {
            mat_dma(In.kpg, a.sn, lds0 + OFF_K, wave, lane);
            mat_dma(In.vpg, a.sn, lds0 + OFF_V, wave, lane);
        }
        consume(wave + 1 < NT ? wave + 1 : 0);       // step 6's tiles: query tile (wave - 6) mod 7
        PROF_MARK();
        // key-owner results: the rows of dK, dV of key tile `wave` leave through this wave's exchange slot (its P / dS tiles
        // are consumed) as whole 128-byte rows
        const int64_t goff = (int64_t)b * a.dsb + (int64_t)(wave * 32) * a.dsn + (int64_t)h * a.dsh;
        if (!BWD1_EXP_NOSTORE) {
            store_tile_staged(myslot, reinterpret_cast<short*>(a.dk) + goff, a.dsn, wave * 32, dk, lane);
            store_tile_staged(myslot, reinterpret_cast<short*>(a.dv) + goff, a.dsn, wave * 32, dv, lane);
        }
        PROF_MARK();
        // job = tab * 2 + dt.  Wave w takes job w; the eighth job goes to wave 3 — the only wave that has its SIMD to itself
        // (waves w and w + 4 share one), so every SIMD runs two jobs
        const int job0 = wave, job1 = wave == 3 ? 7 : 8;
        {
            bf16x8 bk[4];
            slots_to_buckets14_bf16(bk, myslot, dx, lane, wave == 0, min(qr, G14 - 1), qc);
            // dq^T += [Tkv; Tkh]^T dL'^T : lane group g supplies the buckets of table g (order of the contraction index is free)
            const short* kt_img = img + IMG_KT;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    dq[dt] = mma16(*reinterpret_cast<const bf16x8*>(kt_img + (c32 + 32 * dt) * 64 + g * 32 + ks * 8), bk[ks], dq[dt]);
            // dq rows out through the slot, THEN the dL' tile takes it (its fragments are still in registers)
            if (!BWD1_EXP_NOSTORE) store_tile_staged(myslot, reinterpret_cast<short*>(a.dq) + goff, a.dsn, wave * 32, dq, lane);
            // dL' tile [32 q][64 u'] for the table-gradient jobs (chunks 4g + ks of row q)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                union { bf16x8 f; u32x4v v; } u;
                u.f = bk[ks];
                *reinterpret_cast<u32x4v*>(myslot + c32 * 128 + (((4 * g + ks) ^ swz128(c32)) << 4)) = u.v;
            }
        }
        PROF_MARK();
        __syncthreads();                             // all dL' tiles are in place
        PROF_MARK();

        // ---- table gradients: job = tab * 2 + dt;  dT^T (64 d x 32 u) = X^T (d x q) . R (q x u)
        //      tab 0 / 1: X = Q, R = dL' (vertical / horizontal);  tab 2 / 3: X = dO, R = S' (bucket sums of the forward)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int job = jj == 0 ? job0 : job1;
            if (job < 8) {
                const int tab = job >> 1, dt = job & 1;
                // lane = bucket u (column), registers = d rows; partial of THIS workgroup, accumulated over its items
                float* dst = a.dtab + (((int64_t)blockIdx.x * 4 + tab) * 32 + c32) * 64 + dt * 32 + 4 * g;
                const bool first = item == (int)blockIdx.x;
                f32x4v old[4];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) old[r4] = first ? f32x4v{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4v*>(dst + 8 * r4);
                f32x16 acc = {};
                const unsigned char* xbase = smem + (tab < 2 ? OFF_Q : OFF_D);
                if (tab < 2) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int st = 0; st < 2; ++st) {
                            const unsigned char* xt = xbase + t * 4096 + st * 2048;
                            const unsigned char* dl = smem + OFF_X + t * SLOT_B + st * 2048;
                            acc = mma16(tr_pair(xt + lo.tr[dt][0], xt + lo.tr[dt][1]), tr_pair(dl + lo.tr[tab & 1][0], dl + lo.tr[tab & 1][1]), acc);
                        }
                } else {
                    const short* spr = reinterpret_cast<const short*>(a.sp) + (bh * 64 + (tab & 1) * 32 + c32) * NP14;
                    bf16x8 rb[NT][2];
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int st = 0; st < 2; ++st) rb[t][st] = Tr<hip_bfloat16>::load_perm(spr + t * 32, st, g);
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int st = 0; st < 2; ++st) {
                            const unsigned char* xt = xbase + t * 4096 + st * 2048;
                            acc = mma16(tr_pair(xt + lo.tr[dt][0], xt + lo.tr[dt][1]), rb[t][st], acc);
                        }
                }
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
                    *reinterpret_cast<f32x4v*>(dst + 8 * r4) = f32x4v{old[r4][0] + acc[4 * r4], old[r4][1] + acc[4 * r4 + 1],
                                                                      old[r4][2] + acc[4 * r4 + 2], old[r4][3] + acc[4 * r4 + 3]};
            }
        }
        if (more && !BWD1_EXP_NOLOAD) {
            __syncthreads();                         // every table-gradient job is done with Q, dO (and the slots)
            mat_dma(In.qp, a.sn, lds0 + OFF_Q, wave, lane);
            mat_dma(In.dop, orow, lds0 + OFF_D, wave, lane);
        }
        PROF_MARK();
#ifdef ATTN_PROFILE_ITEM1
        if (item == (int)(blockIdx.x + gridDim.x)) V2_PROF_FLUSH();          // the SECOND item: one with a predecessor and a successor
#else
        V2_PROF_FLUSH();
#endif
        if (!more) break;
        item = next;
        I = In;
    }   // items
}

}  // namespace v2
