// attn_rpe2d_fwd2.hpp — forward of the fused attention for the AutoFormer geometry (N = 197, 14 x 14 grid,
// max_relative_position 14, bf16) with TILE-GRANULAR online softmax and two half-workgroups in PING-PONG.
// Included by attn_rpe2d.hip inside its anonymous namespace, after attn_rpe2d_bwd1.hpp (shares its primitives: DMA,
// swizzles, transposing LDS reads, the bf16 slot <-> bucket shifts, the staged row store).
// Reference semantics: AutoFormer/model/module/multihead_super.py:133-154.
//
// Why.  attn_rpe2d_fwd14 keeps the whole row block of scores of a query tile in registers (7 tiles x 16 = 112 VGPRs; 256
// in all), so a CU holds ONE 7-wave workgroup and all seven waves are in the same phase at the same time: the two MFMA
// phases are 29 % of an item, the matrix cores idle under the softmax / shift / commit phases and vice versa (13 % MFMA
// busy).  Here the score row block never exists: per 32-key tile a wave computes S^T (6 MFMAs: 4 over d,
// 2 over the one-hot slot extension = the relative position bias), the softmax numerators against a LAZY running
// maximum (it moves only when a tile exceeds it by more than 2^8: the rescale of O is a rare wave-uniform branch),
// and [O | slot sums]^T += [V | one-hot]^T P^T (6 MFMAs) — 48 accumulator registers, <= 128 VGPRs in all, so a CU holds
// 14 waves: ONE workgroup of two 7-wave HALVES, each working on its own (b, h) item, half a period apart:
//
//   phase p, half (p & 1):      epilogue of item p - 2 (normalise, slot sums -> bucket sums -> S'^T out, + bucket sums . value
//                               tables, rows of O out, lse) and prologue of item p (Q rows, bucket lookups q . T_k^T, window
//                               shift -> slot extension): VALU / LDS / global traffic
//   phase p, the other half:    asks for K, V of item p by DMA into the FIRST half's matrices (dead since the last barrier),
//                               then runs the 7-tile loop of item p - 1: MFMA + exp
//   one s_barrier per phase.
//
// So at any time seven waves multiply while seven do element-wise / memory work (the two pipes of a SIMD overlap across
// waves), every phase has one item's K / V in flight under it (the HBM read stream never pauses), and nothing is
// committed through registers.  The half that multiplies issues the OTHER half's DMA: a wave's vmcnt retires in issue
// order, so loads a wave issues behind its own DMA requests would wait for them; this way the element-wise half only has
// ordinary, compiler-counted loads and the multiplying half only waits for the DMA at the end of its loop.
//
// LDS (161,280 B, one workgroup per CU):
//   half 0: K | V, half 1: K | V     4 x [224][64] bf16, 16-byte chunks XOR-swizzled with swz128(row)
//   OH                               [224][32] bf16 one-hot slot rows of the keys (shared; keys >= N carry slot 15)
//   7 scratch slots                  4608 B each: shift scratch / row staging of wave w of whichever half is in its
//                                    element-wise phase (the halves alternate, one barrier between them)
// The bucket tables are read from the bf16 operand images (table_images_kernel / cream_attn_rpe2d_table_images).
#pragma once

#ifdef ATTN_PROFILE
#define F2_PROF_DECL long long f2t[16]; int f2n = 0;
#define F2_MARK() do { if (f2n < 16) f2t[f2n++] = (long long)__builtin_readcyclecounter(); } while (0)
#define F2_FLUSH() do { if ((threadIdx.x & 63) == 0 && g_attn_prof) { \
        long long* d_ = g_attn_prof + ((long long)blockIdx.x * 16 + (threadIdx.x >> 6)) * 16; \
        for (int i_ = 0; i_ < 16; ++i_) d_[i_] = i_ < f2n ? f2t[i_] : 0; } } while (0)
#else
#define F2_PROF_DECL
#define F2_MARK() do {} while (0)
#define F2_FLUSH() do {} while (0)
#endif

namespace v3 {
using namespace v2;

constexpr int F2_THREADS = 2 * THREADS;                 // 14 waves
constexpr int F2_OFF_OH = 4 * MAT_B;
constexpr int F2_OFF_X = F2_OFF_OH + OH_B;
constexpr int F2_LDS_B = F2_OFF_X + NT * SLOT_B;
static_assert(F2_LDS_B <= 160 * 1024, "LDS budget");
// lazy maximum: the reference of the exponentials moves when a tile's maximum exceeds it by more than this many powers of two
constexpr float F2_LAZY_LOG2 = 8.f;

// x *= alpha IN PLACE (tied operand).  Written in C the rare rescale branch is a diamond on the three accumulators: the register
// allocator then keeps two homes for O and copies 32 registers per tile between them (first build: 32 v_mov_b64 per iteration
// and the Q fragments in scratch).
// alpha comes straight from v_exp_f32: a transcendental result needs a wait state before a VALU instruction reads it, and the
// compiler's hazard recognizer does not look inside asm statements (first build: element 0 of O was NaN for a quarter of the lanes)
// — hence the s_nop in front of the first multiplication.
__device__ __forceinline__ void scale_in_place(f32x16& v, float alpha) {
    asm volatile("s_nop 1" ::: );
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float x = v[r];
        asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x) : "v"(alpha));
        v[r] = x;
    }
}
__device__ __forceinline__ float max3f(float a, float b, float c) {          // (fmaxf on MFMA results costs a canonicalising v_max each)
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// max over the two lanes of a query (lane, lane ^ 32): one v_permlane32_swap instead of a ds_bpermute round trip
__device__ __forceinline__ float pair_max(float x) {
    const uint32_t u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));        // (compiler-visible: it knows the swap's hazards)
}

struct F2Lane {                                         // thread-derived values, recomputed per phase from an opaque thread index
    int lane, g, c32, qi, qr, qc;
    bool tok_ok;
};
__device__ __forceinline__ F2Lane f2_lane(int wave) {
    // OPAQUE copy of the thread index: the compiler would otherwise hoist every thread-derived address out of the item loop and
    // keep it in registers across all three roles (attn_rpe2d_bwd1.hpp)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    F2Lane L;
    L.lane = tid & 63; L.g = L.lane >> 5; L.c32 = L.lane & 31;
    L.qi = wave * 32 + L.c32;
    L.tok_ok = L.qi < N14;
    L.qr = L.qi > 0 ? (L.qi - 1) / G14 : 0;
    L.qc = L.qi > 0 ? (L.qi - 1) - L.qr * G14 : 0;
    return L;
}

// ---- prologue of an item: this lane's Q row, bucket lookups q . [Tkv; Tkh]^T, window shift -> slot extension ----------------------
__device__ __forceinline__ void f2_prologue(bf16x8 (&qb)[4], bf16x8 (&qe)[2], const FwdArgs& a, const short* img, int item, int wave,
                                            unsigned char* myslot) {
    const F2Lane L = f2_lane(wave);
    const int b = item / a.H, h = item - b * a.H;
    const int qcl = min(L.qi, N14 - 1);
    const short* qp = reinterpret_cast<const short*>(a.q) + (int64_t)b * a.sb + (int64_t)h * a.sh + (int64_t)qcl * a.sn;
    bf16x8 tk[2][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qb[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16 + L.g * 8);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int t = 0; t < 2; ++t) tk[t][ks] = *reinterpret_cast<const bf16x8*>(img + IMG_KR + (32 * t + L.c32) * 64 + ks * 16 + L.g * 8);
    if (!L.tok_ok) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qb[ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    f32x16 kv = {}, kh = {};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        kv = mma16(tk[0][ks], qb[ks], kv);
        kh = mma16(tk[1][ks], qb[ks], kh);
    }
    // key-side slot extension; slot 15 = -2^15: padding keys (one-hot slot 15) get probability exactly 0
    ext_from_lookups14(qe, kv, kh, myslot, L.lane, wave == 0, L.qr, L.qc, (short)0xC700);
}

// ---- epilogue of an item: normalise, slot sums -> bucket sums (S'^T out), + bucket sums . value tables, rows of O, lse ------------
__device__ __forceinline__ void f2_epilogue(f32x16 (&o)[2], f32x16& ox, float m_ref, float l, float sc, const FwdArgs& a, const short* img,
                                            int item, int wave, unsigned char* myslot) {
    const F2Lane L = f2_lane(wave);
    const int b = item / a.H, h = item - b * a.H;
    const int64_t bh = (int64_t)b * a.H + h;
    l += __shfl_xor(l, 32);
    const float inv_l = 1.f / l;
    if (L.tok_ok && L.g == 0) a.lse[bh * N14 + L.qi] = (m_ref + log2f(l)) * (1.f / LOG2E);      // (m_ref is in log2 units)
#pragma unroll
    for (int r = 0; r < 16; ++r) ox[r] *= inv_l;
    // slot sums -> bucket sums (bf16: what the backward reads and what the matrix cores take)
    bf16x8 bk[4];
    slots_to_buckets14_bf16(bk, myslot, ox, L.lane, wave == 0, min(L.qr, G14 - 1), L.qc);
    if (a.sp) {
        short* dst = reinterpret_cast<short*>(a.sp) + (bh * 64 + L.g * 32) * NP14 + L.qi;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) dst[(int64_t)(ks * 8 + e) * NP14] = bk[ks][e];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] *= inv_l; o[1][r] *= inv_l; }
    // O^T += [Tvv; Tvh]^T S'^T : lane group g supplies the buckets of table g (transposed value-table image)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
            o[dt] = mma16(*reinterpret_cast<const bf16x8*>(img + IMG_VT + (L.c32 + 32 * dt) * 64 + L.g * 32 + ks * 8), bk[ks], o[dt]);
    store_tile_staged(myslot, reinterpret_cast<short*>(a.out) + (((int64_t)b * N14 + wave * 32) * a.H + h) * 64, (int64_t)a.H * 64,
                      wave * 32, o, L.lane);
}

// ---- K, V of an item -> the LDS matrices at lds_kv (by the waves of the half that multiplies) ---------------------------------------
__device__ __forceinline__ void f2_request_kv(const FwdArgs& a, int item, uint32_t lds_kv, int wave) {
    const F2Lane L = f2_lane(wave);
    const int b = item / a.H, h = item - b * a.H;
    const int64_t base = (int64_t)b * a.sb + (int64_t)h * a.sh;
    mat_dma(reinterpret_cast<const short*>(a.k) + base, a.sn, lds_kv, wave, L.lane);
    mat_dma(reinterpret_cast<const short*>(a.v) + base, a.sn, lds_kv + MAT_B, wave, L.lane);
}

// ---- the 7-tile loop of an item: online softmax against a lazy maximum --------------------------------------------------------------
__device__ __forceinline__ void f2_tiles(f32x16 (&o)[2], f32x16& ox, float& m_ref, float& l, const bf16x8 (&qb)[4], const bf16x8 (&qe)[2],
                                         const unsigned char* kbase, const unsigned char* ohb, float sc, int wave) {
    const F2Lane L = f2_lane(wave);
    // per-lane byte offsets inside a 32-token tile; every other operand address is one XOR / ADD away:
    //   row[ks] = r0 ^ (ks << 5)        ohrow[1] = h0 ^ 32
    //   tr[dt][hh] = ((t0 + 1024 hh) ^ (32 hh)) ^ (64 dt)      ohtr[hh] = (u0 + 512 hh) ^ (32 hh)
    // (lane_offs of attn_rpe2d_bwd1.hpp, which keeps all twelve in registers)
    const int gi = L.lane & 15, q4 = L.lane >> 4;
    const int rh = 4 * L.g + (gi >> 2), inner = 8 * (gi & 1), c0 = 2 * (q4 & 1) + ((gi & 3) >> 1);
    int r0 = L.c32 * 128 + ((L.g ^ swz128(L.c32)) << 4);
    int h0 = L.c32 * 64 + ((L.g ^ ((L.c32 >> 2) & 3)) << 4);
    int t0 = rh * 128 + ((c0 ^ swz128(rh)) << 4) + inner;
    int u0 = rh * 64 + ((c0 ^ ((rh >> 2) & 3)) << 4) + inner;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; ox[r] = 0.f; }
    m_ref = -INFINITY;
    l = 0.f;
#pragma unroll 1
    for (int t = 0; t < NT; ++t) {
        const unsigned char* kt = kbase + t * 4096;
        const unsigned char* vt = kt + MAT_B;
        const unsigned char* oh = ohb + t * 2048;
        asm volatile("" : "+v"(r0), "+v"(h0), "+v"(t0), "+v"(u0));   // (derived addresses stay inside the iteration)
        // ---- S^T (32 keys x 32 queries) = [K | one-hot] . [Q | X]^T -------------------------------------
        f32x16 s = {};
        {
            const bf16x8 k0 = lds_b128(kt + r0), k1 = lds_b128(kt + (r0 ^ 32));
            s = mma16(k0, qb[0], s);
            s = mma16(k1, qb[1], s);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const bf16x8 k2 = lds_b128(kt + (r0 ^ 64)), k3 = lds_b128(kt + (r0 ^ 96));
            s = mma16(k2, qb[2], s);
            s = mma16(k3, qb[3], s);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const bf16x8 e0 = lds_b128(oh + h0), e1 = lds_b128(oh + (h0 ^ 32));
            s = mma16(e0, qe[0], s);
            s = mma16(e1, qe[1], s);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- lazy running maximum (per query = lane pair), in log2 units ---------------------------------
        // The FIRST consumer of the MFMA result is an ordinary multiplication: the compiler puts the wait states an MFMA result
        // needs in front of it — not in front of an asm statement (a v_max3 written in asm as first reader saw partial tiles).
        float tl[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) tl[r] = s[r] * sc;
        float mt = max3f(tl[0], tl[1], tl[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) mt = max3f(mt, tl[r], tl[r + 1]);
        mt = pair_max(fmaxf(mt, tl[15]));
        if (!__all(mt <= m_ref + F2_LAZY_LOG2)) {
            // every P.V product issued so far is in o / ox (the MFMAs of the previous tile are complete in program order): O, the
            // slot sums and l are rescaled together, nothing else is at the old scale
            const float mn = fmaxf(m_ref, mt);
            const float alpha = __builtin_amdgcn_exp2f(m_ref - mn);
            m_ref = mn;
            l *= alpha;
            scale_in_place(o[0], alpha);
            scale_in_place(o[1], alpha);
            scale_in_place(ox, alpha);
        }
        uint32_t pw[8];
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const float p0 = __builtin_amdgcn_exp2f(tl[r] - m_ref);
            const float p1 = __builtin_amdgcn_exp2f(tl[r + 1] - m_ref);
            l += p0 + p1;
            pw[r >> 1] = f2bf_pair(p0, p1);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- [O | slot sums]^T += [V | one-hot]^T P^T ---------------------------------------------------
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const bf16x8 pb = __builtin_bit_cast(bf16x8, (u32x4v{pw[4 * st], pw[4 * st + 1], pw[4 * st + 2], pw[4 * st + 3]}));
            const unsigned char* vs = vt + st * 2048;
            const int t1 = (t0 + 1024) ^ 32;
            const bf16x8 v0 = tr_pair(vs + t0, vs + t1);
            const bf16x8 v1 = tr_pair(vs + (t0 ^ 64), vs + (t1 ^ 64));
            const bf16x8 e = tr_pair(oh + st * 1024 + u0, oh + st * 1024 + ((u0 + 512) ^ 32));
            o[0] = mma16(v0, pb, o[0]);
            o[1] = mma16(v1, pb, o[1]);
            ox = mma16(e, pb, ox);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

__global__ __launch_bounds__(F2_THREADS) void attn_rpe2d_fwd2_kernel(const FwdArgs a, const short* img) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // (wave-uniform floats are computed on the vector ALU: pinned to SGPRs by hand — in the tile loop every VGPR counts)
    const float sc = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(a.scale * LOG2E)));

    fill_onehot_swz(smem + F2_OFF_OH);

    const int G = (int)gridDim.x, B0 = (int)blockIdx.x;
    const int nk = ((int)a.nitems - B0 + G - 1) / G;    // items of this workgroup: B0 + k G, k < nk
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>(smem));
    const int wave14 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = wave14 >= NT ? 1 : 0;
    const int wave = wave14 - half * NT;
    unsigned char* myslot = smem + F2_OFF_X + wave * SLOT_B;
    const unsigned char* kmine = smem + half * 2 * MAT_B;                   // this half's K | V
    const uint32_t kv_other = lds0 + (uint32_t)((1 - half) * 2 * MAT_B);    // the other half's (DMA destination)

    __syncthreads();                                    // the one-hot image is complete

    // Phases 0 .. nk + 1, one barrier each.  Half h works on the items k = h, h + 2, ...:
    //   phase k      element-wise: epilogue of item k - 2, prologue of item k
    //   phase k + 1  matrix: request K, V of item k + 1 for the OTHER half (its matrices are dead since the last barrier), tile loop of k
    // so both halves run the same loop body, half 1 one phase behind half 0 (its phase 0 = the request of item 0).
    int phase = 0;
    if (half) {
        f2_request_kv(a, B0, kv_other, wave);
        dma_wait_all();
        lds_barrier();
        phase = 1;
    }
    f32x16 o[2], ox;
    float m_ref = 0.f, l = 0.f;
    int kprev = -1;
    F2_PROF_DECL
#pragma unroll 1
    for (int k = half; k < nk; k += 2) {
        bf16x8 qb[4], qe[2];
        F2_MARK();
        if (kprev >= 0) f2_epilogue(o, ox, m_ref, l, sc, a, img, B0 + kprev * G, wave, myslot);
        F2_MARK();
        f2_prologue(qb, qe, a, img, B0 + k * G, wave, myslot);
        F2_MARK();
        lds_barrier();
        F2_MARK();
        if (k + 1 < nk) f2_request_kv(a, B0 + (k + 1) * G, kv_other, wave);
        F2_MARK();
        f2_tiles(o, ox, m_ref, l, qb, qe, kmine, smem + F2_OFF_OH, sc, wave);
        F2_MARK();
        dma_wait_all();                                 // this wave's pieces of the other half's K, V have landed
        F2_MARK();
        lds_barrier();
        kprev = k;
        phase += 2;
    }
    F2_MARK();
    if (kprev >= 0) f2_epilogue(o, ox, m_ref, l, sc, a, img, B0 + kprev * G, wave, myslot);
    F2_MARK();
    F2_FLUSH();
#pragma unroll 1
    for (; phase < nk + 2; ++phase) lds_barrier();
}

}  // namespace v3
