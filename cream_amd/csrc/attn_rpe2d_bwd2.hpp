// attn_rpe2d_bwd2.hpp — the one-pass attention backward with the two roles of attn_rpe2d_bwd1.hpp on SEPARATE waves
// (round 6; AutoFormer geometry N = 197, 14 x 14 grid, max_relative_position 14, bf16).  Included by attn_rpe2d.hip behind
// attn_rpe2d_bwd1.hpp (same LDS layout, same helpers: namespace v2); reference semantics:
// AutoFormer/model/module/multihead_super.py:133-160, SURVEY App. B.
//
// Why.  In bwd1 every one of the 7 waves is owner of a query tile AND owner of a key tile: 112 accumulator registers of
// gradients + 32 of scores pin 238 VGPRs, so a CU holds 7 waves (2 + 2 + 2 + 1 on its SIMDs) and every phase of an item
// lasts as long as one wave's serial chain: per step  consume (855 cycles) -> S, dP (885) -> exp / dS / dQx (1,060) ->
// barrier -> publish -> barrier; the eight table-gradient jobs of an item run two per wave behind the row stores
// (9.6-14k of an item's 63-66k cycles), dK / dV / dQ leave one after the other through the same wave.
// Here a workgroup has 12 waves of <= 168 VGPRs (three per SIMD):
//   * 7 PRODUCERS (wave w = owner of query tile w): delta and slot extensions, then per step S^T, dP^T (12 MFMAs), P, dS,
//     publish, dQx^T += Kx_j^T dS^T (6 MFMAs) — 48 accumulator registers of gradients;
//   * 5 CONSUMERS share the 14 key-side jobs (key tile j, dV or dK): job q = 2 j + t goes to consumer q mod 5, which
//     picks up the tile published for key tile j in the previous step and accumulates dV_j^T += dO^T P or dK_j^T += Q^T dS
//     (4 MFMAs per job and step, 32 accumulator registers per job) WHILE the producers compute the next score tiles.
// Same exchange protocol (7 slots, two LDS-only barriers per step), same contraction order per output element: dq, dk, dv
// are BIT-IDENTICAL to bwd1's (probe + GPU test).
// What moves out of the serial chain of an item, and out of HBM:
//   * the consume phase of every step (other waves, other issue slots);
//   * the TABLE gradients never leave the chip between items: the eight 32 x 32 jobs (table, half of d) are dealt to the
//     consumers (table below), each job's MFMA accumulator lives in its consumer's registers across ALL items of the
//     workgroup and is written once at the end of the kernel — bwd1 read-modify-writes its 32 KB partial per item
//     (64 KB x 768 items = 49 MB of the 234 MB a launch moved).  The sums are the same numbers added in one fp32 chain
//     instead of per-item chains plus additions: equal to bwd1's within fp32 rounding, bit-reproducible run to run;
//   * the four VALUE-table jobs (X = dO, R = the forward's bucket sums S') are spread over the step loop — the two
//     MFMAs of query tile s in step s, the S' fragments of tile s + 1 requested a step ahead: 28 KB per item drawn from
//     HBM while nothing else is (the first build ran them at the top of the item: another burst beside K, V, Q, dO);
//   * dK / dV rows (consumers, staged through the V region — dead since step 6) leave beside the producers'
//     slot -> bucket shifts, dq product and dQ rows;
//   * the four KEY-table jobs (X = Q, R = dL') run behind [D] on three consumers.
//     consumer   key-side jobs q = 2 j + t     value-table jobs   key-table jobs
//        0        0  5 10                       4
//        1        1  6 11                       5
//        2        2  7 12                                          0
//        3        3  8 13                                          1 2
//        4        4  9                          6 7                3
// LDS as in bwd1 (K | V | Q | dO | one-hot rows | 7 exchange slots = 161,280 B).
#pragma once

// timing experiments of tools/probes/attn_bwd1_probe.hip (results are WRONG with either switch): leave out the global row
// stores of dq / dk / dv, or the loads of the next item's matrices and O rows (the LDS keeps the first item's)
#ifndef BWD2_EXP_NOSTORE
#define BWD2_EXP_NOSTORE 0
#endif
#ifndef BWD2_EXP_NOLOAD
#define BWD2_EXP_NOLOAD 0
#endif

// phase stamps for tools/probes/attn_bwd1_probe.hip (compiled out of the library): 12 slots per wave, 12 waves per workgroup
#ifdef ATTN_PROFILE
#define V4_PROF_DECL long long pt4[12]; int pn4 = 0;
#define V4_MARK() do { pt4[pn4++] = (long long)__builtin_readcyclecounter(); } while (0)
#define V4_FLUSH() do { if ((threadIdx.x & 63) == 0 && g_attn_prof) { \
        long long* d_ = g_attn_prof + ((long long)blockIdx.x * 12 + (threadIdx.x >> 6)) * 12; \
        for (int i_ = 0; i_ < 12; ++i_) d_[i_] = i_ < pn4 ? pt4[i_] : 0; } } while (0)
#else
#define V4_PROF_DECL
#define V4_MARK() do {} while (0)
#define V4_FLUSH() do {} while (0)
#endif

namespace v4 {

using v2::NT; using v2::N14; using v2::NP14; using v2::MAT_B;
using v2::OFF_K; using v2::OFF_V; using v2::OFF_Q; using v2::OFF_D; using v2::OFF_OH; using v2::OFF_X;
using v2::SLOT_B; using v2::XT_B; using v2::SCRP;
using v2::IMG_KR; using v2::IMG_KT; using v2::IMG_VR;
using v2::LaneOffs; using v2::lane_offs; using v2::swz128; using v2::tr_pair; using v2::lds_b128; using v2::mma16;
using v2::lds_barrier; using v2::dma_1k; using v2::dma_wait_all; using v2::ext_from_lookups14;
using v2::slots_to_buckets14_bf16; using v2::store_tile_staged;

constexpr int NPROD = 7, NCONS = 5, WAVES = NPROD + NCONS, THREADS = WAVES * 64;
constexpr int LDS_B = v2::OFF_SINK;                  // (no prefetch sink)
constexpr int JOBS = 3;                              // key-side jobs per consumer (14 over 5: 3 3 3 3 2)
static_assert(LDS_B <= 160 * 1024, "LDS budget");

// rows [0, 224) of one matrix into the LDS region at byte address lds_base: 28 pieces of 1 KB over 12 waves
__device__ __forceinline__ void mat_dma12(const short* src, int64_t rs, uint32_t lds_base, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int piece = wave + WAVES * i;
        if (piece < 28) {                            // (wave-uniform)
            const int row = piece * 8 + (lane >> 3), cc = lane & 7;
            const short* s = row < N14 ? src + (int64_t)row * rs + ((cc ^ swz128(row)) << 3) : reinterpret_cast<const short*>(v2::g_zero_line);
            dma_1k(s, lds_base + piece * 1024);
        }
    }
}

__device__ __forceinline__ void fill_onehot12(unsigned char* oh) {
    const RelGeom G{N14, G14, G14, G14};
    for (int i = threadIdx.x; i < NP14 * 4; i += THREADS) {
        const int j = i >> 2, cc = i & 3;
        const uint32_t m = (key_mask(j, G) | (j >= N14 ? 1u << 15 : 0u)) >> (8 * cc);     // (slot 15: padding keys, see v2::fill_onehot_swz)
        u32x4v w;
#pragma unroll
        for (int p = 0; p < 4; ++p) w[p] = ((m >> (2 * p)) & 1u) * 0x3F80u + ((m >> (2 * p + 1)) & 1u) * 0x3F800000u;
        *reinterpret_cast<u32x4v*>(oh + j * 64 + ((cc ^ ((j >> 2) & 3)) << 4)) = w;
    }
}

// global address of job's 32 x 32 block in this workgroup's partial: job = tab * 2 + dt -> rows u (lane c32), columns dt * 32 + ...
__device__ __forceinline__ float* table_job_dst(const BwdArgs& a, int job, int lane) {
    const int g = lane >> 5, c32 = lane & 31;
    return a.dtab + (((int64_t)blockIdx.x * 4 + (job >> 1)) * 32 + c32) * 64 + (job & 1) * 32 + 4 * g;
}
__device__ __forceinline__ void table_job_store(const BwdArgs& a, int job, const f32x16& acc, int lane) {
    float* dst = table_job_dst(a, job, lane);
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4)
        *reinterpret_cast<f32x4v*>(dst + 8 * r4) = f32x4v{acc[4 * r4], acc[4 * r4 + 1], acc[4 * r4 + 2], acc[4 * r4 + 3]};
}

struct Item {
    const short *qp, *kpg, *vpg, *dop, *outp;
    int64_t bh;
    int b, h;
};
// (items in (b, h) order.  Measured and dropped, profiles/r06_attn_bwd2.md: an XCD-aware order — XCD x takes the images
// b = x mod 8 with all their heads, so that the adjacent 128-byte pieces of an image's token rows meet in one L2 — 92.1 against
// 92.4 us; starting every second workgroup of an XCD 6k .. 38k cycles late: only the delay shows.)
__device__ __forceinline__ Item item_of(const BwdArgs& a, int item) {
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
}

// ================================================== PRODUCER: owner of query tile `wave` (all items of the workgroup) =========
__device__ __forceinline__ void producer_items(const BwdArgs& a, const short* img, unsigned char* smem, uint32_t lds0, int item, Item I) {
    const float sc = a.scale * LOG2E;
    const int64_t orow = (int64_t)a.H * 64;
    // this lane's row of O (for delta) and its softmax statistic: requested an item AHEAD (behind the step loop of the previous
    // item) — at the top of an item a fresh HBM round trip is the longest thing a wave waits for (9k cycles of the 12k
    // there even with no operand DMA at all: ablation builds, profiles/r06_attn_bwd2.md)
    bf16x8 ob[4];
    float lse_r;
    {
        const int lane0 = threadIdx.x & 63, w0 = threadIdx.x >> 6;
        const int qcl0 = min(w0 * 32 + (lane0 & 31), N14 - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ob[ks] = *reinterpret_cast<const bf16x8*>(I.outp + (int64_t)qcl0 * orow + ks * 16 + (lane0 >> 5) * 8);
        lse_r = a.lse[I.bh * N14 + qcl0];
    }
    for (;;) {
        // (thread-derived values from an OPAQUE copy per item: see v2 — no address of the loop body is hoisted across items)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int lane = tid & 63, g = lane >> 5, c32 = lane & 31;
        const LaneOffs lo = lane_offs(lane);
        const int b = I.b, h = I.h;
        const int64_t bh = I.bh;
        const int next = item + (int)gridDim.x;
        const bool more = next < a.nitems;
        Item In = I;
        if (more) In = item_of(a, next);
        V4_PROF_DECL
        V4_MARK();                                   // 0: item start
        const int qi = wave * 32 + c32;
        const bool tok_ok = qi < N14;
        const int qcl = min(qi, N14 - 1);
        const int qr = qi > 0 ? (qi - 1) / G14 : 0, qc = qi > 0 ? (qi - 1) - qr * G14 : 0;
        unsigned char* myslot = smem + OFF_X + wave * SLOT_B;
        bf16x8 tk[2][4];                             // row fragments of the key-table image (vertical, horizontal)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int t = 0; t < 2; ++t) tk[t][ks] = *reinterpret_cast<const bf16x8*>(img + IMG_KR + (32 * t + c32) * 64 + ks * 16 + g * 8);
        bf16x8 tv[2][4];                             // ... and of the value-table image
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int t = 0; t < 2; ++t) tv[t][ks] = *reinterpret_cast<const bf16x8*>(img + IMG_VR + (32 * t + c32) * 64 + ks * 16 + g * 8);
        dma_wait_all();                              // this wave's pieces of K, V, Q, dO have landed ...
        lds_barrier();                               // ... and everybody's                                    [top]
        V4_MARK();                                   // 1: operands landed
        const float m2 = tok_ok ? lse_r * LOG2E : INFINITY;
        bf16x8 qe[2], de[2];
        float dsc;
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
            f32x16 kv = {}, kh = {};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                kv = mma16(tk[0][ks], qb[ks], kv);
                kh = mma16(tk[1][ks], qb[ks], kh);
            }
            ext_from_lookups14(qe, kv, kh, myslot, lane, wave == 0, qr, qc, (short)0xC700);
            f32x16 vv = {}, vh = {};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                vv = mma16(tv[0][ks], dob[ks], vv);
                vh = mma16(tv[1][ks], dob[ks], vh);
            }
            ext_from_lookups14(de, vv, vh, myslot, lane, wave == 0, qr, qc, (short)0);
        }

        f32x16 dq[2] = {f32x16{}, f32x16{}}, dx = {};
        V4_MARK();                                   // 2: prologue done
#pragma unroll 1
        for (int s = 0; s < NT; ++s) {
            const int j = wave + s < NT ? wave + s : wave + s - NT;             // this step's key tile (wave-uniform)
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
            __builtin_amdgcn_sched_barrier(0);
            // P = exp2(S sc - m2), dS = P (dP scale - delta scale); keys >= N: P = 0 through slot 15
            uint32_t pw[8], dw[8];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[r], sc, -m2));
                const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[r + 1], sc, -m2));
                pw[r >> 1] = f2bf_pair(p0, p1);
                dw[r >> 1] = f2bf_pair(p0 * __builtin_fmaf(pacc[r], a.scale, -dsc), p1 * __builtin_fmaf(pacc[r + 1], a.scale, -dsc));
            }
            bf16x8 db[2];
#pragma unroll
            for (int st = 0; st < 2; ++st)
                db[st] = __builtin_bit_cast(bf16x8, (u32x4v{dw[4 * st], dw[4 * st + 1], dw[4 * st + 2], dw[4 * st + 3]}));
            __builtin_amdgcn_sched_barrier(0);
            lds_barrier();                           // [A] the consumers have read the tiles of step s - 1
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
            // dQx^T += Kx_j^T dS^T (under the write-back of the published tiles)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                dq[0] = mma16(tr_pair(kt + st * 2048 + lo.tr[0][0], kt + st * 2048 + lo.tr[0][1]), db[st], dq[0]);
                dq[1] = mma16(tr_pair(kt + st * 2048 + lo.tr[1][0], kt + st * 2048 + lo.tr[1][1]), db[st], dq[1]);
                dx = mma16(tr_pair(oh + st * 1024 + lo.ohtr[0], oh + st * 1024 + lo.ohtr[1]), db[st], dx);
            }
            lds_barrier();                           // [B] the tiles of step s are in place
        }
        V4_MARK();                                   // 3: steps done
        // K and V are dead.  Every region of the next item is requested as soon as its last reader is through: K now, dO behind
        // [C] (the consumers stage their rows through the V region), V behind [D], Q behind [E]
        if (more && !BWD2_EXP_NOLOAD) mat_dma12(In.kpg, a.sn, lds0 + OFF_K, wave, lane);
        if (more) {                                  // (ob, lse_r are dead since the prologue)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) ob[ks] = *reinterpret_cast<const bf16x8*>(In.outp + (int64_t)qcl * orow + ks * 16 + g * 8);
            lse_r = a.lse[In.bh * N14 + qcl];
        }
        // the transposed key-table image for the dq product: requested in front of [C] (L2 round trip under the barrier)
        bf16x8 ktf[4][2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) ktf[ks][dt] = *reinterpret_cast<const bf16x8*>(img + IMG_KT + (c32 + 32 * dt) * 64 + g * 32 + ks * 8);
        lds_barrier();                               // [C] the consumers are through with the last tiles: the slots are free
        V4_MARK();                                   // 4: behind [C]
        if (more && !BWD2_EXP_NOLOAD) v2::mat_dma(In.dop, orow, lds0 + OFF_D, wave, lane);         // (7 waves x 4 pieces)
        const int64_t goff = (int64_t)b * a.dsb + (int64_t)(wave * 32) * a.dsn + (int64_t)h * a.dsh;
        {
            bf16x8 bk[4];
            slots_to_buckets14_bf16(bk, myslot, dx, lane, wave == 0, min(qr, G14 - 1), qc);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) dq[dt] = mma16(ktf[ks][dt], bk[ks], dq[dt]);
            if (!BWD2_EXP_NOSTORE) store_tile_staged(myslot, reinterpret_cast<short*>(a.dq) + goff, a.dsn, wave * 32, dq, lane);
            else { asm volatile("" :: "v"(dq[0]), "v"(dq[1])); }
            // dL' tile [32 q][64 u'] for the key-table jobs (chunks 4g + ks of row q)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                union { bf16x8 f; u32x4v v; } u;
                u.f = bk[ks];
                *reinterpret_cast<u32x4v*>(myslot + c32 * 128 + (((4 * g + ks) ^ swz128(c32)) << 4)) = u.v;
            }
        }
        V4_MARK();                                   // 5: epilogue done
        lds_barrier();                               // [D] all dL' tiles are in place
        V4_MARK();                                   // 6: behind [D]
        if (more && !BWD2_EXP_NOLOAD) mat_dma12(In.vpg, a.sn, lds0 + OFF_V, wave, lane);
        lds_barrier();                               // [E] the key-table jobs are done with Q and the slots
        V4_MARK();                                   // 7: behind [E]
#ifdef ATTN_PROFILE
        if (item == (int)(blockIdx.x + gridDim.x)) V4_FLUSH();                  // the SECOND item: one with a predecessor and a successor
#endif
        if (!more) break;
        // Q of the next item (its last readers are behind [E])
        if (!BWD2_EXP_NOLOAD) mat_dma12(In.qp, a.sn, lds0 + OFF_Q, wave, lane);
        item = next;
        I = In;
    }
}

// ================================================== CONSUMER c: KJ key-side jobs q = c + 5 jb, NV value-table jobs vj0 ..,
// NK key-table jobs kj0 .. (the table above); the table accumulators live across all items =====================================
template <int KJ, int NV, int NK>
__device__ __forceinline__ void consumer_items(const BwdArgs& a, unsigned char* smem, uint32_t lds0, int item, Item I, int c, int vj0, int kj0) {
    const int64_t orow = (int64_t)a.H * 64;
    f32x16 va[NV > 0 ? NV : 1], ka[NK > 0 ? NK : 1];
    bf16x8 rb[NV > 0 ? NV : 1][2];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int lane0 = threadIdx.x & 63;
        const short* sp0 = reinterpret_cast<const short*>(a.sp) + (I.bh * 64 + (((vj0 + v) >> 1) & 1) * 32 + (lane0 & 31)) * NP14;
#pragma unroll
        for (int st = 0; st < 2; ++st) rb[v][st] = Tr<hip_bfloat16>::load_perm(sp0, st, lane0 >> 5);
    }
#pragma unroll
    for (int v = 0; v < (NV > 0 ? NV : 1); ++v) va[v] = f32x16{};
#pragma unroll
    for (int k = 0; k < (NK > 0 ? NK : 1); ++k) ka[k] = f32x16{};
    for (;;) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int lane = tid & 63, g = lane >> 5, c32 = lane & 31;
        const LaneOffs lo = lane_offs(lane);
        const int b = I.b, h = I.h;
        const int64_t bh = I.bh;
        const int next = item + (int)gridDim.x;
        const bool more = next < a.nitems;
        Item In = I;
        if (more) In = item_of(a, next);
        V4_PROF_DECL
        V4_MARK();                                   // 0: item start
        // value-table jobs: S'^T row fragments of query tile t (bucket c32 of table `tab & 1`), one tile ahead (tile 0: an item ahead)
        const short* spr[NV > 0 ? NV : 1];
#pragma unroll
        for (int v = 0; v < NV; ++v) spr[v] = reinterpret_cast<const short*>(a.sp) + (bh * 64 + (((vj0 + v) >> 1) & 1) * 32 + c32) * NP14;
        dma_wait_all();
        lds_barrier();                               //                                                          [top]
        V4_MARK();                                   // 1
        f32x16 acc[KJ][2];
#pragma unroll
        for (int jb = 0; jb < KJ; ++jb) { acc[jb][0] = f32x16{}; acc[jb][1] = f32x16{}; }
        // tiles published in step sp for key tile j come from the owner of query tile (j - sp) mod 7
        auto consume = [&](int sp) {
#pragma unroll
            for (int jb = 0; jb < KJ; ++jb) {
                const int q = c + NCONS * jb;
                const int j = q >> 1, t = q & 1;
                const int qt = j - sp >= 0 ? j - sp : j - sp + NT;
                const unsigned char* xp = smem + OFF_X + j * SLOT_B + (t ? XT_B : 0);
                const unsigned char* xt_ = smem + (t ? OFF_Q : OFF_D) + qt * 4096;
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const bf16x8 tb = tr_pair(xp + st * 1024 + lo.xr[0], xp + st * 1024 + lo.xr[1]);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
                        acc[jb][dt] = mma16(tr_pair(xt_ + st * 2048 + lo.tr[dt][0], xt_ + st * 2048 + lo.tr[dt][1]), tb, acc[jb][dt]);
                }
            }
        };
        V4_MARK();                                   // 2
#pragma unroll 1
        for (int s = 0; s < NT; ++s) {
            if (s > 0) consume(s - 1);
            // value tables: dT^T (64 d x 32 u) += dO_s^T (d x q) . S'_s (q x u)
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int dtx = ((vj0 + v) & 1) << 6;                           // lo.tr[1][h] == lo.tr[0][h] ^ 64
                const unsigned char* xt = smem + OFF_D + s * 4096;
#pragma unroll
                for (int st = 0; st < 2; ++st)
                    va[v] = mma16(tr_pair(xt + st * 2048 + (lo.tr[0][0] ^ dtx), xt + st * 2048 + (lo.tr[0][1] ^ dtx)), rb[v][st], va[v]);
                if (s + 1 < NT) {
#pragma unroll
                    for (int st = 0; st < 2; ++st) rb[v][st] = Tr<hip_bfloat16>::load_perm(spr[v] + (s + 1) * 32, st, g);
                }
            }
            lds_barrier();                           // [A]
            lds_barrier();                           // [B]
        }
        V4_MARK();                                   // 3: steps done
        if (more && !BWD2_EXP_NOLOAD) mat_dma12(In.kpg, a.sn, lds0 + OFF_K, wave, lane);
        consume(NT - 1);
        lds_barrier();                               // [C] (V is dead since step 6: its region stages the rows)
        V4_MARK();                                   // 4: behind [C]
        unsigned char* stage = smem + OFF_V + c * 4096;
#pragma unroll
        for (int jb = 0; jb < KJ; ++jb) {
            const int q = c + NCONS * jb;
            const int j = q >> 1, t = q & 1;
            const int64_t goff = (int64_t)b * a.dsb + (int64_t)(j * 32) * a.dsn + (int64_t)h * a.dsh;
            if (!BWD2_EXP_NOSTORE) store_tile_staged(stage, reinterpret_cast<short*>(t ? a.dk : a.dv) + goff, a.dsn, j * 32, acc[jb], lane);
            else { asm volatile("" :: "v"(acc[jb][0]), "v"(acc[jb][1])); }
        }
        V4_MARK();                                   // 5: rows out
        lds_barrier();                               // [D]
        V4_MARK();                                   // 6: behind [D]
        if (more && !BWD2_EXP_NOLOAD) mat_dma12(In.vpg, a.sn, lds0 + OFF_V, wave, lane);
        if (more) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const short* spn = reinterpret_cast<const short*>(a.sp) + (In.bh * 64 + (((vj0 + v) >> 1) & 1) * 32 + c32) * NP14;
#pragma unroll
                for (int st = 0; st < 2; ++st) rb[v][st] = Tr<hip_bfloat16>::load_perm(spn, st, g);
            }
        }
        // key tables: dT^T (64 d x 32 u) += Q^T (d x q) . dL' (q x u)
        if constexpr (NK > 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const unsigned char* xt = smem + OFF_Q + t * 4096 + st * 2048;
                    const unsigned char* dl = smem + OFF_X + t * SLOT_B + st * 2048;
#pragma unroll
                    for (int k = 0; k < NK; ++k) {
                        const int job = kj0 + k;
                        const int dtx = (job & 1) << 6, tbx = ((job >> 1) & 1) << 6;
                        ka[k] = mma16(tr_pair(xt + (lo.tr[0][0] ^ dtx), xt + (lo.tr[0][1] ^ dtx)), tr_pair(dl + (lo.tr[0][0] ^ tbx), dl + (lo.tr[0][1] ^ tbx)), ka[k]);
                    }
                }
        }
        lds_barrier();                               // [E]
        V4_MARK();                                   // 7: behind [E]
#ifdef ATTN_PROFILE
        if (item == (int)(blockIdx.x + gridDim.x)) V4_FLUSH();
#endif
        if (!more) break;
        if (!BWD2_EXP_NOLOAD) mat_dma12(In.qp, a.sn, lds0 + OFF_Q, wave, lane);
        item = next;
        I = In;
    }
    // this workgroup's table-gradient partial: written ONCE
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int v = 0; v < NV; ++v) table_job_store(a, vj0 + v, va[v], lane);
#pragma unroll
    for (int k = 0; k < NK; ++k) table_job_store(a, kj0 + k, ka[k], lane);
}

__global__ __launch_bounds__(THREADS) void attn_rpe2d_bwd2_kernel(const BwdArgs a, const short* img) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    fill_onehot12(smem + OFF_OH);
    const int item = blockIdx.x;
    if (item >= a.nitems) return;
    const Item I = item_of(a, item);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>(smem));
    const int w0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l0 = threadIdx.x & 63;
    mat_dma12(I.kpg, a.sn, lds0 + OFF_K, w0, l0);
    mat_dma12(I.vpg, a.sn, lds0 + OFF_V, w0, l0);
    mat_dma12(I.qp, a.sn, lds0 + OFF_Q, w0, l0);
    mat_dma12(I.dop, (int64_t)a.H * 64, lds0 + OFF_D, w0, l0);
    // DE-PHASING.  An item is a compute phase (prologue + step loop: ~28k cycles with HBM idle) and a memory phase (75 KB of rows
    // out, 137 KB of the next item in: ~27k cycles in which every CU asks at once).  Every second workgroup of an XCD starts
    // `stagger` x 64 clocks late, so that one half of the chip computes while the other half moves its bytes.
    if (a.stagger > 0 && ((blockIdx.x >> 3) & 1)) {
        for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(1);
    }
    // every role runs the same barriers per item ([top], 7 x ([A], [B]), [C], [D], [E]) over the same items
    if (w0 < NPROD) producer_items(a, img, smem, lds0, item, I);
    else if (w0 == NPROD + 0) consumer_items<3, 1, 0>(a, smem, lds0, item, I, 0, 4, 0);
    else if (w0 == NPROD + 1) consumer_items<3, 1, 0>(a, smem, lds0, item, I, 1, 5, 0);
    else if (w0 == NPROD + 2) consumer_items<3, 0, 1>(a, smem, lds0, item, I, 2, 0, 0);
    else if (w0 == NPROD + 3) consumer_items<3, 0, 2>(a, smem, lds0, item, I, 3, 0, 1);
    else consumer_items<2, 2, 1>(a, smem, lds0, item, I, 4, 6, 3);
}

}  // namespace v4
