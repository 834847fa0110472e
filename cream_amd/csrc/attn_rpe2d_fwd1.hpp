// attn_rpe2d_fwd1.hpp — forward of the fused attention for the AutoFormer geometry (N = 197, 14 x 14 grid,
// max_relative_position 14, bf16) with K and V travelling global -> LDS by DMA.  Included by attn_rpe2d.hip after
// attn_rpe2d_bwd1.hpp (uses FwdArgs and the v2 helpers); reference semantics:
// AutoFormer/model/module/multihead_super.py:133-160, SURVEY App. B.
//
// What changes against attn_rpe2d_fwd14_kernel (same algebra, same operand roundings):
//   * K and V of an item are requested by LDS-DMA (global_load_lds_dwordx4, 1 KB per wave instruction, no registers:
//     the register-staged matrices of fwd14 were committed with 4 ds_write_b128 per thread behind a wait for HBM, twice
//     per item) into unpadded, XOR-swizzled [224][64] images (chunk ^ swz128(row)); the NEXT item's K is requested as
//     soon as the scores are done, its V as soon as P.V is done — both land under the rest of the item;
//   * the shift scratch is bf16 and private to the wave (4.6 KB), so it overlays neither matrix: TWO barriers per item
//     (K dead + V landed; V dead + next K landed) instead of four, none of them behind a fresh memory round trip;
//   * keys >= N are masked through the unused slot 15 (one-hot bit 15 x a -2^15 entry of the query extension) instead
//     of a select per score; the bucket-table operand images (key rows, value^T) sit in LDS, built once per workgroup;
//   * the rows of O leave as whole 128-byte lines through the wave's scratch (store_tile_staged).
// LDS (120,320 B, one workgroup per CU): K | V | one-hot rows | 7 wave scratches | key-table rows | value tables^T.
#pragma once

namespace v2 {

constexpr int F_OFF_K = 0, F_OFF_V = MAT_B, F_OFF_OH = 2 * MAT_B, F_OFF_X = F_OFF_OH + OH_B;
constexpr int F_OFF_TKR = F_OFF_X + NT * SLOT_B, F_OFF_TVT = F_OFF_TKR + 8192;
constexpr int FWD1_LDS_B = F_OFF_TVT + 8192;
static_assert(FWD1_LDS_B <= 160 * 1024, "LDS budget");

// bf16 operand images of the bucket tables, XOR-swizzled like the matrices:
//   TKR [64 u'][64 d]  rows of the key tables (u' = bucket of the vertical table, or 32 + bucket of the horizontal one)
//   TVT [64 d][64 u']  value tables transposed;  rows u >= nb are zero
__device__ __forceinline__ void fill_table_images(unsigned char* tkr, unsigned char* tvt, const FwdArgs& a) {
    for (int i = threadIdx.x; i < 4096; i += THREADS) {
        const int u2 = i >> 6, d = i & 63, u = u2 & 31;
        const bool in = u < a.nb;
        const float xk = in ? (u2 < 32 ? a.tkv : a.tkh)[(int64_t)u * a.ldt + d] : 0.f;
        const float xv = in ? (u2 < 32 ? a.tvv : a.tvh)[(int64_t)u * a.ldt + d] : 0.f;
        *reinterpret_cast<short*>(tkr + u2 * 128 + (((d >> 3) ^ swz128(u2)) << 4) + (d & 7) * 2) = f2bf(xk);
        *reinterpret_cast<short*>(tvt + d * 128 + (((u2 >> 3) ^ swz128(d)) << 4) + (u2 & 7) * 2) = f2bf(xv);
    }
}

__global__ __launch_bounds__(THREADS) void attn_rpe2d_fwd1_kernel(const FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const float sc = a.scale * LOG2E;
    const int64_t orow = (int64_t)a.H * 64;

    int item = blockIdx.x;
    if (item >= a.nitems) return;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>(smem));
    bf16x8 qn[4];                                    // this lane's query row of the NEXT item (requested an item ahead)
    {
        const int w0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l0 = threadIdx.x & 63;
        const int b0 = item / a.H, h0 = item - b0 * a.H;
        const int64_t base = (int64_t)b0 * a.sb + (int64_t)h0 * a.sh;
        mat_dma(reinterpret_cast<const short*>(a.k) + base, a.sn, lds0 + F_OFF_K, w0, l0);
        mat_dma(reinterpret_cast<const short*>(a.v) + base, a.sn, lds0 + F_OFF_V, w0, l0);
        const int q0 = min(w0 * 32 + (l0 & 31), N14 - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            qn[ks] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const short*>(a.q) + base + (int64_t)q0 * a.sn + ks * 16 + (l0 >> 5) * 8);
    }
    fill_onehot_swz(smem + F_OFF_OH);
    fill_table_images(smem + F_OFF_TKR, smem + F_OFF_TVT, a);
    dma_wait_all();
    __syncthreads();

    for (;;) {
        V2_PROF_DECL
        PROF_MARK();
        // (everything derived from the thread index is recomputed per item from an opaque copy: see attn_rpe2d_bwd1.hpp)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int lane = tid & 63, g = lane >> 5, c32 = lane & 31;
        const LaneOffs lo = lane_offs(lane);
        const int qi = wave * 32 + c32;
        const bool qok = qi < N14;
        const int qr = qi > 0 ? (qi - 1) / G14 : 0, qc = qi > 0 ? (qi - 1) - qr * G14 : 0;
        unsigned char* myslot = smem + F_OFF_X + wave * SLOT_B;
        const int b = item / a.H, h = item - b * a.H;
        const int64_t bh = (int64_t)b * a.H + h;
        const int next = item + (int)gridDim.x;
        const bool more = next < a.nitems;
        const int nb_ = more ? next / a.H : b, nh_ = more ? next - nb_ * a.H : h;
        const int64_t nbase = (int64_t)nb_ * a.sb + (int64_t)nh_ * a.sh;

        // ---- this wave's query tile: bucket lookups (key tables) -> slot extension ------------------------------
        bf16x8 qb[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qb[ks] = qok ? qn[ks] : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        bf16x8 qe[2];
        {
            f32x16 kv = {}, kh = {};
            const unsigned char* tkr = smem + F_OFF_TKR;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                kv = mma16(lds_b128(tkr + lo.row[ks]), qb[ks], kv);
                kh = mma16(lds_b128(tkr + 4096 + lo.row[ks]), qb[ks], kh);
            }
            ext_from_lookups14(qe, kv, kh, myslot, lane, wave == 0, qr, qc, (short)0xC700);
        }
        PROF_MARK();

        // ---- S^T: all keys against this wave's 32 queries ------------------------------------------------------
        f32x16 s[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            s[t] = f32x16{};
            const unsigned char* kt = smem + F_OFF_K + t * 4096;
            const unsigned char* oh = smem + F_OFF_OH + t * 2048;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s[t] = mma16(lds_b128(kt + lo.row[ks]), qb[ks], s[t]);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) s[t] = mma16(lds_b128(oh + lo.ohrow[ks]), qe[ks], s[t]);
        }
        PROF_MARK();
        dma_wait_all();                              // this wave's pieces of V have landed (requested an item ago) ...
        __syncthreads();                             // ... everybody's too, and nobody reads K any more
        if (more) mat_dma(reinterpret_cast<const short*>(a.k) + nbase, a.sn, lds0 + F_OFF_K, wave, lane);
        PROF_MARK();

        // ---- softmax over keys (in-lane + one exchange with the partner lane); keys >= N sit at -2^15 -----------
        float m4[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) m4[r & 3] = fmaxf(m4[r & 3], s[t][r]);
        float m = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
        m = fmaxf(m, __shfl_xor(m, 32));
        const float msc = m * sc;
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
        if (qok && g == 0) a.lse[bh * N14 + qi] = (msc + log2f(l)) * (1.f / LOG2E);
        PROF_MARK();

        // ---- [O | slot sums]^T = [V | one-hot]^T . P^T ---------------------------------------------------------
        f32x16 o[2] = {f32x16{}, f32x16{}};
        f32x16 ox = {};
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const unsigned char* vt = smem + F_OFF_V + t * 4096;
            const unsigned char* oh = smem + F_OFF_OH + t * 2048;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const bf16x8 pb = Tr<hip_bfloat16>::from_acc(s[t], st);
                o[0] = mma16(tr_pair(vt + st * 2048 + lo.tr[0][0], vt + st * 2048 + lo.tr[0][1]), pb, o[0]);
                o[1] = mma16(tr_pair(vt + st * 2048 + lo.tr[1][0], vt + st * 2048 + lo.tr[1][1]), pb, o[1]);
                ox = mma16(tr_pair(oh + st * 1024 + lo.ohtr[0], oh + st * 1024 + lo.ohtr[1]), pb, ox);
            }
        }
        PROF_MARK();
        dma_wait_all();                              // this wave's pieces of the next K have landed ...
        __syncthreads();                             // ... everybody's too, and nobody reads V any more
        if (more) {
            mat_dma(reinterpret_cast<const short*>(a.v) + nbase, a.sn, lds0 + F_OFF_V, wave, lane);
            const int qn_ = min(qi, N14 - 1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                qn[ks] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const short*>(a.q) + nbase + (int64_t)qn_ * a.sn + ks * 16 + g * 8);
        }
        PROF_MARK();
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] *= inv_l; o[1][r] *= inv_l; ox[r] *= inv_l; }

        // ---- value-side relative position term: slot sums -> bucket sums -> . value tables -----------------------
        {
            bf16x8 bk[4];
            slots_to_buckets14_bf16(bk, myslot, ox, lane, wave == 0, min(qr, G14 - 1), qc);
            if (a.sp && qi < NP14) {                 // S'^T (64 buckets x NP queries) for the table gradients of the backward
                short* dst = reinterpret_cast<short*>(a.sp) + (bh * 64 + g * 32) * NP14 + qi;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int e = 0; e < 8; ++e) dst[(int64_t)(8 * ks + e) * NP14] = bk[ks][e];
            }
            const unsigned char* tvt = smem + F_OFF_TVT;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    o[dt] = mma16(lds_b128(tvt + dt * 4096 + c32 * 128 + (((4 * g + ks) ^ swz128(c32)) << 4)), bk[ks], o[dt]);
        }
        PROF_MARK();
        store_tile_staged(myslot, reinterpret_cast<short*>(a.out) + ((int64_t)b * N14 + wave * 32) * orow + (int64_t)h * 64, orow,
                          wave * 32, o, lane);
        PROF_MARK();
#ifdef ATTN_PROFILE_ITEM1
        if (item == (int)(blockIdx.x + gridDim.x)) V2_PROF_FLUSH();          // the SECOND item: one with a predecessor and a successor
#else
        V2_PROF_FLUSH();
#endif
        if (!more) break;
        item = next;
    }
}

}  // namespace v2
