// gemm_tn8.hpp — the weight-gradient (TN) product on the macro tile and the phase-interleaved loop of gemm_nt8.hpp (gfx950).
//
// Reference semantics (unchanged): what autograd derives for F.linear on the active block W[:out, :in]
// (AutoFormer/model/module/Linear_super.py:71-81, qkv_super.py:72-83):  dW(N x K) = dY(M x N)^T . X(M x K), contraction over
// the M = B * 197 tokens, split S ways over workgroups; every split writes ONE bf16 partial tile set [S][N][K] that
// cream_grad_finalize adds in fp32 in fixed order (no atomics).
//
// Why a second kernel: gemm_tn_kernel's 128 x 128 tile moves 32 KB through the CU's 64 B/clk L2 -> LDS path per 64-token step
// for 16 MFMAs per wave — two workgroups per CU saturate that path at HALF the matrix-core rate, and its two-stage ring waits for
// every step's loads with vmcnt(0) one step after issuing them.  Here:
//   * output tile up to 256 (n) x 256 (k): 64 KB per 64-token K-tile for 32 MFMAs per wave — half the bytes per flop;
//     8 waves as 4 (n) x 2 (k), ONE workgroup per CU, 128 KB of LDS = 2 K-tile buffers x 4 half-tile slots [64 tokens][128 columns]
//     [Y_q0 | Y_q1 | X_q0 | X_q1] (the images and the transpose reads — ds_read_b64_tr_b16, two per fragment — are those of
//     gemm_tn_kernel: both operands are token-major, the MFMA wants token-contiguous fragments);
//   * the 32 x 32 blocks of the tile are dealt round-robin to the waves — wave row wm owns row blocks {wm, wm + 4}, wave column
//     wn column blocks {wn, wn + 2, wn + 4, wn + 6} — so a tile that sticks out of the matrix (N, K are multiples of 64 in this
//     model family, rarely of 256) costs every wave the same smaller number of MFMAs instead of idling whole waves;
//   * a K-tile (64 tokens) is four phases of one quadrant (1 row block x 2 column blocks x 4 sub-steps = 8 MFMAs) each:
//       phase 1  (tm 0; tn 0, 1)  reads Y_q0 (4 fragments), X_q0 (8)      phase 3  (tm 1; tn 2, 3)  reads X_q1 (8)
//       phase 2  (tm 1; tn 0, 1)  reads Y_q1 (4)                           phase 4  (tm 0; tn 2, 3)  reads nothing
//     with the schedule of gemm_nt8.hpp: the two halves of the workgroup (waves 0-3 / 4-7: one wave of each per SIMD) run one
//     barrier apart, every phase requests one half-tile (2 LDS-DMA per wave, scalar base + 32-bit lane offsets) into a slot that
//     died two phases earlier, vmcnt(8) keeps a whole K-tile of requests in flight across every barrier, a half-tile is read
//     five phases after its request;
//   * the partial tile leaves straight from the accumulators: bf16 pairs, v_permlane32_swap to 16-byte chunks, 16-byte stores.
//
// Measured and dropped (profiles/r05_tn8_classes.md): a slice count per tile class (cheap last-row / last-column tiles get fewer
// slices) and a two-phase loop over three half-tiles for tiles of at most 128 rows or columns — 5-25 % shorter in the cold probe,
// +1.2 % / +-0 on the training step (more partial tiles through HBM; the early-finishing workgroups were handing their CUs to the
// main chain anyway).
//
// Limits (the launcher keeps gemm_tn_kernel otherwise): bf16 partials, no bias partials, N % 8 == 0, K % 8 == 0.
#pragma once
#include "gemm_nt8.hpp"

namespace cream {
namespace gemm {

constexpr int TN8_LDS_BYTES = 2 * 65536;

// BIAS: bias_parts[split][n] = column sums of dY over the split's tokens (F.linear's bias gradient) ride along as MFMAs against a
// ones fragment, in the workgroups of the first column tile only: wave column 0 sums row block tm 0 in phase 4, wave column 1
// row block tm 1 in phase 3 (the two phases without Y reads; both waves of a wave row hold both Y fragments anyway).
template <bool BIAS = false>
__global__ __launch_bounds__(512, 2) void gemm_tn8_kernel(const TnParams p)
{
    constexpr uint32_t SLOT = 16384, KTB = 65536;
    constexpr uint32_t S_Y0 = 0, S_Y1 = SLOT, S_X0 = 2 * SLOT, S_X1 = 3 * SLOT;
    extern __shared__ __attribute__((aligned(1024))) char tn8_lds[];
    char* const smem = tn8_lds;

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 5, c32 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, grp = wave >> 2;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>(smem));
    const int ntc = (p.K + 255) / 256, T = ntc * ((p.N + 255) / 256);
    // all tiles of one split on one XCD: they re-read the same token rows of dY and X (one HBM fetch, the rest from that L2)
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = bid % T, split = bid / T;
    const int r0 = (tile / ntc) * 256, c0 = (tile % ntc) * 256;
    const int tsteps = (p.M + 63) / 64;
    const int s_lo = (int)((int64_t)tsteps * split / p.S), s_hi = (int)((int64_t)tsteps * (split + 1) / p.S);
    const int nk = s_hi - s_lo;
    if (nk <= 0) return;                                         // (the launcher keeps S <= tsteps: every split owns a step)

    // ---- staging: this lane's two 1-KB pieces (4 token rows x 256 B each) of a half-tile slot; 16-byte chunk swizzled with the
    //      token row (the image of gemm_tn_kernel)
    uint32_t offY[2][2], offX[2][2];                             // [slot half q][piece i]: byte offsets from the K-tile's first token row
    int trow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int piece = wave * 2 + i, r = piece * 4 + (lane >> 4);
        const int c = (lane & 15) ^ ((r & 3) << 2);
        trow[i] = r;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            offY[q][i] = (uint32_t)(r * (int)p.ldy + min(r0 + 128 * q + c * 8, p.N - 8)) * 2u;
            offX[q][i] = (uint32_t)(r * (int)p.ldx + min(c0 + 128 * q + c * 8, p.K - 8)) * 2u;
        }
    }
    int s_st = s_lo;                                             // the staging cursor: a 64-token step of this split
    uint32_t s_par = 0;
    auto advance = [&]() { s_par ^= 1; if (s_st + 1 < s_hi) ++s_st; };    // (past the last step: duplicates nobody reads)
    auto stage = [&](auto which_) {                              // 0 Y_q0, 1 Y_q1, 2 X_q0, 3 X_q1
        constexpr int which = decltype(which_)::value;
        const char* base = which < 2 ? reinterpret_cast<const char*>(p.dY) + (int64_t)s_st * 64 * p.ldy * 2
                                     : reinterpret_cast<const char*>(p.X) + (int64_t)s_st * 64 * p.ldx * 2;
        const uint32_t dst = lds0 + s_par * KTB + which * SLOT + wave * 2048;
        const uint32_t o0 = which < 2 ? offY[which & 1][0] : offX[which & 1][0], o1 = which < 2 ? offY[which & 1][1] : offX[which & 1][1];
        const int tvalid = p.M - s_st * 64;                      // token rows of this step inside the matrix
        if (tvalid >= 64) {
            nt8_dma2(base, o0, o1, dst);
        } else {                                                 // the last step of an M % 64 != 0: rows beyond M are zeros
            nt8_dma(trow[0] < tvalid ? base + o0 : reinterpret_cast<const char*>(g_nt8_zero), dst);
            nt8_dma(trow[1] < tvalid ? base + o1 : reinterpret_cast<const char*>(g_nt8_zero), dst + 1024);
        }
    };
    using W_Y0 = std::integral_constant<int, 0>; using W_Y1 = std::integral_constant<int, 1>;
    using W_X0 = std::integral_constant<int, 2>; using W_X1 = std::integral_constant<int, 3>;

    // ---- transpose-read fragments (gemm_tn_kernel's): 32 columns from `col`, tokens ms * 16 + 8 (lane >> 5) .. + 7
    const int gi = lane & 15, q4 = lane >> 4;
    auto frag_off = [&](int col) -> uint32_t {
        const int cc = col + 16 * (q4 & 1) + (gi & 3) * 4;       // first of this lane's 4 columns
        const int m = 8 * (q4 >> 1) + (gi >> 2);                 // token row supplied by this lane (first read; + 4 for the second)
        return (uint32_t)(m * 128 + ((((cc >> 3) ^ ((m & 3) << 2)) << 3) | (cc & 7))) * 2u;
    };
    const uint32_t offR = frag_off(32 * wm);                     // block wm of a Y slot
    const uint32_t offC[2] = {frag_off(32 * wn), frag_off(32 * (wn + 2))};   // blocks wn, wn + 2 of an X slot
    auto ldfrag = [&](uint32_t off) -> bf16x8 {                  // (sub-step ms: + ms * 4096 bytes)
        const bf16x4 lo = tr16(reinterpret_cast<const uint16_t*>(smem + off)), hi = tr16(reinterpret_cast<const uint16_t*>(smem + off + 1024));
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };

    f32x16 acc[4][2];                                            // [tn][tm]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    bf16x8 fy0[4], fy1[4], fx[2][4];                             // Y fragments of both row blocks [ms], X fragments [block][ms]
    f32x16 bacc = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const bf16x8 ones = bf16x8{0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    const bool want_bias = BIAS && p.bias_parts != nullptr && c0 == 0;

    // ---- prologue: K-tile 0 whole, K-tile 1's first two half-tiles
    stage(W_Y0{}); stage(W_X0{}); stage(W_Y1{}); stage(W_X1{});
    advance();
    stage(W_Y0{}); stage(W_X0{});
    wait_vmcnt<8>();
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();                  // waves 4-7 run one barrier behind waves 0-3
    __builtin_amdgcn_sched_barrier(0);

#define TN8_PRE()                                                                                \
    do {                                                                                         \
        wait_vmcnt<8>();                                                                         \
        __builtin_amdgcn_s_barrier();                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                       \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    } while (0)
#define TN8_POST()                                                                               \
    do {                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        __builtin_amdgcn_s_barrier();                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    } while (0)
    // quadrant (TM_; column blocks 2 Q_, 2 Q_ + 1), J_ of them multiplied: lane <-> row n of the block (second operand),
    // registers <-> columns k
#define TN8_MMA(TM_, Q_, FY_, J_)                                                                \
    do {                                                                                         \
        _Pragma("unroll") for (int ms = 0; ms < 4; ++ms)                                         \
            _Pragma("unroll") for (int j = 0; j < (J_); ++j)                                     \
                acc[2 * (Q_) + j][TM_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fx[j][ms], FY_[ms], acc[2 * (Q_) + j][TM_], 0, 0, 0); \
    } while (0)

    // How much of the tile lies inside the matrix is the same for the whole workgroup and the whole loop: the K loop exists once per
    // (J0, J1, R1) = column blocks multiplied in the two quadrant columns (per wave: 2 / 1, 2 / 1 / 0) and whether the second row
    // block exists — no branch or register copy between MFMAs.  A wave multiplies a block of its own that sticks out whenever a
    // sibling wave's counterpart is inside (lock-step: the phase lasts as long anyway); what sticks out is never stored.
    uint32_t c_par = 0;
    auto kloop = [&](auto j0_, auto j1_, auto r1_, auto bt_) {
        constexpr int J0 = decltype(j0_)::value, J1 = decltype(j1_)::value;
        constexpr bool R1 = decltype(r1_)::value;
        constexpr int BT = decltype(bt_)::value;                 // the row block whose column sums this wave adds (-1: none)
        for (int kt = 0; kt < nk; ++kt) {
            const uint32_t bufoff = c_par * KTB;
            // ---- phase 1: (tm 0; tn 0, 1)
            stage(W_Y1{});
#pragma unroll
            for (int j = 0; j < J0; ++j)
#pragma unroll
                for (int ms = 0; ms < 4; ++ms) fx[j][ms] = ldfrag(bufoff + S_X0 + ms * 4096 + offC[j]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ms = 0; ms < 4; ++ms) fy0[ms] = ldfrag(bufoff + S_Y0 + ms * 4096 + offR);
            TN8_PRE();
            TN8_MMA(0, 0, fy0, J0);
            TN8_POST();
            // ---- phase 2: (tm 1; tn 0, 1)
            stage(W_X1{});
            if constexpr (R1) {
#pragma unroll
                for (int ms = 0; ms < 4; ++ms) fy1[ms] = ldfrag(bufoff + S_Y1 + ms * 4096 + offR);
            }
            TN8_PRE();
            if constexpr (R1) TN8_MMA(1, 0, fy1, J0);
            TN8_POST();
            // ---- phase 3: (tm 1; tn 2, 3) — the cursor moves on to K-tile t + 2 (this buffer: Y_q0 and X_q0 died in phase 1)
            advance();
            stage(W_Y0{});
#pragma unroll
            for (int j = 0; j < J1; ++j)
#pragma unroll
                for (int ms = 0; ms < 4; ++ms) fx[j][ms] = ldfrag(bufoff + S_X1 + ms * 4096 + offC[j]);
            TN8_PRE();
            if constexpr (R1) TN8_MMA(1, 1, fy1, J1);
            if constexpr (R1 && BT == 1) {
#pragma unroll
                for (int ms = 0; ms < 4; ++ms) bacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, fy1[ms], bacc, 0, 0, 0);
            }
            TN8_POST();
            // ---- phase 4: (tm 0; tn 2, 3)
            stage(W_X0{});
            TN8_PRE();
            TN8_MMA(0, 1, fy0, J1);
            if constexpr (BT == 0) {
#pragma unroll
                for (int ms = 0; ms < 4; ++ms) bacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, fy0[ms], bacc, 0, 0, 0);
            }
            TN8_POST();
            c_par ^= 1;
        }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    const int VN = min(8, (p.K - c0 + 31) / 32), VM = min(8, (p.N - r0 + 31) / 32);   // 32-wide blocks of the tile inside the matrix
    const int j1 = VN >= 7 ? 2 : VN >= 5 ? 1 : 0;
    using BN_ = std::integral_constant<int, -1>; using B0_ = std::integral_constant<int, 0>; using B1_ = std::integral_constant<int, 1>;
    auto by_cols = [&](auto r1_, auto bt_) {
        if (VN >= 3) { if (j1 == 2) kloop(I2{}, I2{}, r1_, bt_); else if (j1 == 1) kloop(I2{}, I1{}, r1_, bt_); else kloop(I2{}, I0{}, r1_, bt_); }
        else kloop(I1{}, I0{}, r1_, bt_);
    };
    auto by_rows = [&](auto bt_) { if (VM >= 5) by_cols(std::true_type{}, bt_); else by_cols(std::false_type{}, bt_); };
    if constexpr (BIAS) {
        if (want_bias) { if (wn == 0) by_rows(B0_{}); else by_rows(B1_{}); }
        else by_rows(BN_{});
    } else {
        by_rows(BN_{});
    }
#undef TN8_PRE
#undef TN8_POST
#undef TN8_MMA
    if (grp == 0) __builtin_amdgcn_s_barrier();                  // matches the last barrier of waves 4-7
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // no DMA may outlive the workgroup's LDS

    // ---- the partial tile: every split's sum over its ~M / S tokens is rounded to bf16 once (the reference's autocast backward
    //      rounds the whole weight gradient to bf16 once); lane <-> row n, registers 4 r4 + e <-> columns 8 r4 + 4 g + e of a block
    if constexpr (BIAS) {
        if (want_bias && g == 0) {                               // every register of bacc holds the full column sum of row n
            const int n = r0 + 32 * (wm + 4 * wn) + c32;         // wave column wn summed row block tm = wn
            if (n < p.N) p.bias_parts[(int64_t)split * p.N + n] = bacc[0];
        }
    }
    uint16_t* const out = p.parts16 + (int64_t)split * p.N * p.K;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        const int n = r0 + 32 * (wm + 4 * tm) + c32;
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
            const int kb = c0 + 32 * (wn + 2 * tn);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x16& a = acc[tn][tm];
                uint32_t a0 = f2bf_pair(a[8 * j], a[8 * j + 1]), a1 = f2bf_pair(a[8 * j + 2], a[8 * j + 3]);
                uint32_t b0 = f2bf_pair(a[8 * j + 4], a[8 * j + 5]), b1 = f2bf_pair(a[8 * j + 6], a[8 * j + 7]);
                // 8-byte runs of r4 = 2 j and 2 j + 1 -> lanes 0-31 hold columns 16 j .. + 7, lanes 32-63 columns 16 j + 8 .. + 15
                auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                const int k = kb + 16 * j + 8 * g;
                if (n < p.N && k < p.K) *reinterpret_cast<u32x4v*>(out + (int64_t)n * p.K + k) = u32x4v{s0[0], s1[0], s0[1], s1[1]};
            }
        }
    }
}

}  // namespace gemm
}  // namespace cream
