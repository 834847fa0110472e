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
// Same exchange protocol (7 slots, two LDS-only barriers per step), same contraction order per output element, same
// per-workgroup table-gradient partials in the same order: results are BIT-IDENTICAL to bwd1 (probe + GPU test).
// What moves out of the serial chain of an item:
//   * the consume phase of every step (other waves, other issue slots);
//   * the four VALUE-table jobs (X = dO, R = the forward's bucket sums): consumers 0-3 run them at the top of the item,
//     under the producers' prologue (they have nothing to consume before step 1; their loads queue behind their own
//     stores of the previous item, not behind the producers');
//   * dK / dV rows (consumers, staged through the dO region — dead after the last consume) leave beside the producers'
//     slot -> bucket shifts, dq product and dQ rows;
//   * the four KEY-table jobs run one per consumer instead of two per wave.
// LDS as in bwd1 (K | V | Q | dO | one-hot rows | 7 exchange slots = 161,280 B).
#pragma once

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

// one table-gradient job (v2: "job = tab * 2 + dt"): dT^T (64 d x 32 u) = X^T (d x q) . R (q x u), accumulated into this
// workgroup's partial in global memory (read-modify-write by the owning lanes, fixed order)
__device__ __forceinline__ void table_job(const BwdArgs& a, unsigned char* smem, const LaneOffs& lo, int job, int64_t bh, bool first, int lane) {
    const int g = lane >> 5, c32 = lane & 31;
    const int tab = job >> 1, dt = job & 1;
    float* dst = a.dtab + (((int64_t)blockIdx.x * 4 + tab) * 32 + c32) * 64 + dt * 32 + 4 * g;
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

__global__ __launch_bounds__(THREADS) void attn_rpe2d_bwd2_kernel(const BwdArgs a, const short* img) {
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

    fill_onehot12(smem + OFF_OH);

    int item = blockIdx.x;
    if (item >= a.nitems) return;
    Item I = item_of(item);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>(smem));
    {
        const int w0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l0 = threadIdx.x & 63;
        mat_dma12(I.kpg, a.sn, lds0 + OFF_K, w0, l0);
        mat_dma12(I.vpg, a.sn, lds0 + OFF_V, w0, l0);
        mat_dma12(I.qp, a.sn, lds0 + OFF_Q, w0, l0);
        mat_dma12(I.dop, orow, lds0 + OFF_D, w0, l0);
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
        if (more) In = item_of(next);
        const bool first = item == (int)blockIdx.x;

        if (wave < NPROD) {
            // =========================================== PRODUCER: owner of query tile `wave` ===========================
            const int qi = wave * 32 + c32;
            const bool tok_ok = qi < N14;
            const int qcl = min(qi, N14 - 1);
            const int qr = qi > 0 ? (qi - 1) / G14 : 0, qc = qi > 0 ? (qi - 1) - qr * G14 : 0;
            unsigned char* myslot = smem + OFF_X + wave * SLOT_B;
            bf16x8 ob[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) ob[ks] = *reinterpret_cast<const bf16x8*>(I.outp + (int64_t)qcl * orow + ks * 16 + g * 8);
            const float lse_r = a.lse[bh * N14 + qcl];
            bf16x8 tk[2][4];                         // row fragments of the key-table image (vertical, horizontal)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int t = 0; t < 2; ++t) tk[t][ks] = *reinterpret_cast<const bf16x8*>(img + IMG_KR + (32 * t + c32) * 64 + ks * 16 + g * 8);
            dma_wait_all();                          // this wave's pieces of K, V, Q, dO have landed ...
            lds_barrier();                           // ... and everybody's                                    [top]
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
                bf16x8 tv[2][4];                     // (requested here: their round trip runs under the first window shift)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int t = 0; t < 2; ++t) tv[t][ks] = *reinterpret_cast<const bf16x8*>(img + IMG_VR + (32 * t + c32) * 64 + ks * 16 + g * 8);
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
#pragma unroll 1
            for (int s = 0; s < NT; ++s) {
                const int j = wave + s < NT ? wave + s : wave + s - NT;         // this step's key tile (wave-uniform)
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
                lds_barrier();                       // [A] the consumers have read the tiles of step s - 1
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
                lds_barrier();                       // [B] the tiles of step s are in place
            }
            // K and V are dead: the next item's start travelling now (28 pieces over all 12 waves)
            if (more) {
                mat_dma12(In.kpg, a.sn, lds0 + OFF_K, wave, lane);
                mat_dma12(In.vpg, a.sn, lds0 + OFF_V, wave, lane);
            }
            lds_barrier();                           // [C] the consumers are through with the last tiles: the slots are free
            const int64_t goff = (int64_t)b * a.dsb + (int64_t)(wave * 32) * a.dsn + (int64_t)h * a.dsh;
            {
                bf16x8 bk[4];
                slots_to_buckets14_bf16(bk, myslot, dx, lane, wave == 0, min(qr, G14 - 1), qc);
                const short* kt_img = img + IMG_KT;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
                        dq[dt] = mma16(*reinterpret_cast<const bf16x8*>(kt_img + (c32 + 32 * dt) * 64 + g * 32 + ks * 8), bk[ks], dq[dt]);
                store_tile_staged(myslot, reinterpret_cast<short*>(a.dq) + goff, a.dsn, wave * 32, dq, lane);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    union { bf16x8 f; u32x4v v; } u;
                    u.f = bk[ks];
                    *reinterpret_cast<u32x4v*>(myslot + c32 * 128 + (((4 * g + ks) ^ swz128(c32)) << 4)) = u.v;
                }
            }
            lds_barrier();                           // [D] all dL' tiles are in place
            lds_barrier();                           // [E] the key-table jobs are done with Q and the slots
        } else {
            // =========================================== CONSUMER `wave - 7`: key-side jobs q = c, c + 5, c + 10 ========
            const int c = wave - NPROD;
            dma_wait_all();
            lds_barrier();                           //                                                          [top]
            if (c < 4) table_job(a, smem, lo, 4 + c, bh, first, lane);         // value tables: X = dO, R = S' of the forward
            f32x16 acc[JOBS][2];
#pragma unroll
            for (int jb = 0; jb < JOBS; ++jb) { acc[jb][0] = f32x16{}; acc[jb][1] = f32x16{}; }
            // tiles published in step sp for key tile j come from the owner of query tile (j - sp) mod 7
            auto consume = [&](int sp) {
#pragma unroll
                for (int jb = 0; jb < JOBS; ++jb) {
                    const int q = c + NCONS * jb;
                    if (q < 2 * NT) {                // (wave-uniform)
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
                }
            };
#pragma unroll 1
            for (int s = 0; s < NT; ++s) {
                if (s > 0) consume(s - 1);
                lds_barrier();                       // [A]
                lds_barrier();                       // [B]
            }
            if (more) {
                mat_dma12(In.kpg, a.sn, lds0 + OFF_K, wave, lane);
                mat_dma12(In.vpg, a.sn, lds0 + OFF_V, wave, lane);
            }
            consume(NT - 1);
            lds_barrier();                           // [C] dO is dead (value-table jobs ran at the top): its region stages the rows
            unsigned char* stage = smem + OFF_D + c * 4096;
#pragma unroll
            for (int jb = 0; jb < JOBS; ++jb) {
                const int q = c + NCONS * jb;
                if (q < 2 * NT) {
                    const int j = q >> 1, t = q & 1;
                    const int64_t goff = (int64_t)b * a.dsb + (int64_t)(j * 32) * a.dsn + (int64_t)h * a.dsh;
                    store_tile_staged(stage, reinterpret_cast<short*>(t ? a.dk : a.dv) + goff, a.dsn, j * 32, acc[jb], lane);
                }
            }
            lds_barrier();                           // [D]
            if (c < 4) table_job(a, smem, lo, c, bh, first, lane);             // key tables: X = Q, R = dL'
            lds_barrier();                           // [E]
        }
        if (!more) break;
        // Q and dO of the next item (their last readers are behind [E])
        mat_dma12(In.qp, a.sn, lds0 + OFF_Q, wave, lane);
        mat_dma12(In.dop, orow, lds0 + OFF_D, wave, lane);
        item = next;
        I = In;
    }   // items
}

}  // namespace v4
