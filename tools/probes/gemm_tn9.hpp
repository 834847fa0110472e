// gemm_tn9.hpp — probe kernel (round 5, NOT in the library; profiles/r05_tn9.md): the weight-gradient (TN) product with ONE wave per
// SIMD and a register tile of up to 4 x 6 blocks per wave (gfx950).
//
// Same product, same LDS images and the same bf16 partial tiles as gemm_tn8.hpp.  What changes is the tile: with the accumulators in
// AGPRs a wave holds NRB x NCB 32 x 32 blocks (4 x 4, 4 x 5, 4 x 6, 5 x 4, 6 x 4), four waves make a workgroup tile of 64 NRB (n) x
// 64 NCB (k) — up to 256 x 384 or 384 x 256: the E = 320 / 384 wide operand of a layer in ONE tile instead of a full 256 and a narrow
// one.  The weight gradient is bound by its L2 -> LDS operand stream (profiles/r05_tn9.md: gemm_tn8 and a 256 x 256 build of this
// kernel end up at the same ~20 B/clk per CU); a 256 x 384 tile moves 80 KB per 12.6 MFLOP where 256 x 256 moves 64 KB per 8.4.
//   * ring of 16-token SUB-STEP buffers ([Y slots | X slots] x 16 tokens x 256 B, 8 of them): every wave requests one 1-KB piece of
//     each slot per sub-step, DEPTH sub-steps ahead (counted vmcnt), ONE barrier per sub-step;
//   * no SIMD partner to hide behind: the fragment reads of sub-step s + 1 follow the MFMAs of sub-step s column by column (the X
//     fragment of a column is replaced in place as soon as its MFMAs are issued, the Y fragments are double-buffered);
//   * the MFMAs are inline asm (see tn9_mma);
//   * BIAS: the column sums of dY ride along as v_dot2c_f32_bf16 against (1, 1) on the Y fragments of the first column tile's
//     wave column 0 (fp32 per lane, fixed order).
// Limits: M % 32 == 0 (whole pairs of sub-steps), bf16 partials, N % 8 == 0, K % 8 == 0.
#pragma once
#include "gemm_tn8.hpp"

namespace cream {
namespace gemm {


// one 1-KB piece by LDS-DMA: scalar base + 32-bit lane offset (gemm_nt8's request form), destination in M0
__device__ __forceinline__ void tn9_dma(const void* base, uint32_t off, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(lds_dst), "s"(base) : "memory");
}

// The MFMAs are inline asm: hipcc puts every MFMA destination of a function either in AGPRs or in VGPRs, so more than 16 accumulator
// tiles (256 AGPRs) turn into accvgpr copies around every instruction (1,072 of them in the 4 x 5 build).  Here the first 16 tiles
// are pinned to AGPRs ("+a"), the rest to VGPRs ("+v").  (The compiler does not know these are MFMAs: the epilogue waits out the
// last one's latency itself.)
template <bool AG>
__device__ __forceinline__ void tn9_mma(f32x16& c, const bf16x8& a, const bf16x8& b) {
    if constexpr (AG) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

template <int NRB = 4, int NCB = 4, bool BIAS = false, int DEPTH = 6>
__global__ __launch_bounds__(256, 1) void gemm_tn9_kernel(const TnParams p)
{
    constexpr int TR = 64 * NRB, TC = 64 * NCB;                  // workgroup tile (rows n, columns k)
    constexpr int YS = (TR + 127) / 128, XS = (TC + 127) / 128, NSLOT = YS + XS;
    constexpr uint32_t SLOT = 4096, SUB = NSLOT * SLOT, RING = NSLOT == 4 ? 8 : (163840 / SUB);
    static_assert(DEPTH >= 2 && DEPTH + 2 <= (int)RING, "a buffer is overwritten two sub-steps after its last read at the earliest");
    static_assert(NRB * NCB <= 24, "accumulators");
    extern __shared__ __attribute__((aligned(1024))) char tn9_lds[];
    char* const smem = tn9_lds;

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 5, c32 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>(smem));
    const int ntc = (p.K + TC - 1) / TC, T = ntc * ((p.N + TR - 1) / TR);
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = bid % T, split = bid / T;
    const int r0 = (tile / ntc) * TR, c0 = (tile % ntc) * TC;
    const int tsteps = (p.M + 63) / 64;
    const int s_lo = (int)((int64_t)tsteps * split / p.S), s_hi = (int)((int64_t)tsteps * (split + 1) / p.S);
    if (s_hi <= s_lo) return;
    const int tok0 = s_lo * 64;                                  // first token of the slice
    const int ns = (min(p.M, s_hi * 64) - tok0) / 16;            // sub-steps of this slice, an even number (M % 32 == 0: the caller checks)

    // ---- staging: per sub-step this wave's piece (token rows 4 wave .. + 3 of the 16) of every slot
    const int prow = wave * 4 + (lane >> 4);                     // token row inside a sub-step
    const int pc = (lane & 15) ^ ((prow & 3) << 2);              // source 16-byte chunk of the 256-byte row (swizzled image)
    uint32_t poff[NSLOT];                                        // byte offsets from the sub-step's first token row
#pragma unroll
    for (int q = 0; q < YS; ++q) poff[q] = (uint32_t)(prow * (int)p.ldy + min(r0 + 128 * q + pc * 8, p.N - 8)) * 2u;
#pragma unroll
    for (int q = 0; q < XS; ++q) poff[YS + q] = (uint32_t)(prow * (int)p.ldx + min(c0 + 128 * q + pc * 8, p.K - 8)) * 2u;
    // requests stay DEPTH sub-steps ahead; past the end of the slice the last sub-step is requested again into a buffer nobody reads
    // any more (keeps the vmcnt arithmetic uniform)
    auto request_or_dummy = [&](int sr, uint32_t ring_buf) {     // sub-step sr of the slice -> the buffer at byte ring_buf
        const uint32_t dst = lds0 + ring_buf + wave * 1024;
        const int tok = tok0 + min(sr, ns - 1) * 16;
        const char* by = reinterpret_cast<const char*>(p.dY) + (int64_t)tok * p.ldy * 2;
        const char* bx = reinterpret_cast<const char*>(p.X) + (int64_t)tok * p.ldx * 2;
#pragma unroll
        for (int q = 0; q < NSLOT; ++q) tn9_dma(q < YS ? by : bx, poff[q], dst + q * SLOT);
    };

    // ---- transpose-read fragments (gemm_tn8's, inside one 16-token x 128-column slot)
    const int gi = lane & 15, q4 = lane >> 4;
    auto frag_off = [&](int col) -> uint32_t {                   // col: column inside the operand's part of the buffer
        const int sl = col >> 7, cin = col & 127;
        const int cc = cin + 16 * (q4 & 1) + (gi & 3) * 4;
        const int m = 8 * (q4 >> 1) + (gi >> 2);
        return sl * SLOT + (uint32_t)(m * 128 + ((((cc >> 3) ^ ((m & 3) << 2)) << 3) | (cc & 7))) * 2u;
    };
    uint32_t offY[NRB], offX[NCB];
#pragma unroll
    for (int t = 0; t < NRB; ++t) offY[t] = frag_off(32 * (NRB * wm + t));
#pragma unroll
    for (int t = 0; t < NCB; ++t) offX[t] = YS * SLOT + frag_off(32 * (NCB * wn + t));
    auto ldfrag = [&](uint32_t off) -> bf16x8 {
        const bf16x4 lo = tr16(reinterpret_cast<const uint16_t*>(smem + off)), hi = tr16(reinterpret_cast<const uint16_t*>(smem + off + 1024));
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };

    constexpr int NT = NRB * NCB, NAG = NT < 16 ? NT : 16;       // accumulator tiles [tn * NRB + tm]: the first 16 in AGPRs
    f32x16 accA[NAG], accV[NT > 16 ? NT - 16 : 1];
#pragma unroll
    for (int a = 0; a < NAG; ++a) accA[a] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int a = 0; a < (NT > 16 ? NT - 16 : 1); ++a) accV[a] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    bf16x8 fyA[NRB], fyB[NRB], fx[NCB];                          // Y fragments of the current / the next sub-step; X fragments are replaced
                                                                 // in place, column by column, as soon as their MFMAs are issued

    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    float bsum[NRB];
#pragma unroll
    for (int t = 0; t < NRB; ++t) bsum[t] = 0.f;
    const bool want_bias = BIAS && p.bias_parts != nullptr && c0 == 0 && wn == 0;
    auto colsum = [&](const bf16x8 (&fy)[NRB]) {                 // + the 8 tokens this lane holds of every row block
        const bf2 ones = __builtin_bit_cast(bf2, 0x3F803F80u);
#pragma unroll
        for (int t = 0; t < NRB; ++t) {
            union { bf16x8 f; uint32_t u[4]; } x;
            x.f = fy[t];
#pragma unroll
            for (int e = 0; e < 4; ++e) bsum[t] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, x.u[e]), ones, bsum[t], false);
        }
    };

    // ---- prologue: sub-steps 0 .. DEPTH - 1 requested, sub-step 0 read
    uint32_t rq = 0;                                             // ring buffer (bytes) of the next request
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { request_or_dummy(d, rq); rq = rq + SUB == RING * SUB ? 0u : rq + SUB; }
    wait_vmcnt<NSLOT * (DEPTH - 1)>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int t = 0; t < NRB; ++t) fyA[t] = ldfrag(offY[t]);
#pragma unroll
    for (int t = 0; t < NCB; ++t) fx[t] = ldfrag(offX[t]);
    uint32_t rd = SUB;                                           // ring buffer of sub-step s + 1

    auto substep = [&](int s, bf16x8 (&fy)[NRB], bf16x8 (&ny)[NRB]) {
        request_or_dummy(s + DEPTH, rq);
        rq = rq + SUB == RING * SUB ? 0u : rq + SUB;
        wait_vmcnt<NSLOT * (DEPTH - 1)>();                       // own pieces of sub-step s + 1
        __builtin_amdgcn_s_barrier();                            // ... everyone's
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (BIAS) { if (want_bias) colsum(fy); }
#pragma unroll
        for (int tn = 0; tn < NCB; ++tn) {
#pragma unroll
            for (int tm = 0; tm < NRB; ++tm) {
                constexpr int dummy = 0; (void)dummy;
                const int id = tn * NRB + tm;
                if (id < 16) tn9_mma<true>(accA[id < 16 ? id : 0], fx[tn], fy[tm]);
                else tn9_mma<false>(accV[id >= 16 ? id - 16 : 0], fx[tn], fy[tm]);
            }
            fx[tn] = ldfrag(rd + offX[tn]);                      // the next sub-step's fragment, into the registers just read
            if (tn < NRB) ny[tn] = ldfrag(rd + offY[tn]);
            if (tn + NCB < NRB) ny[tn + NCB] = ldfrag(rd + offY[tn + NCB]);
            __builtin_amdgcn_sched_barrier(0);
        }
        rd = rd + SUB == RING * SUB ? 0u : rd + SUB;
    };
    static_assert(NRB <= 2 * NCB, "Y fragments ride behind the columns");
    int s = 0;
    for (; s + 1 < ns; s += 2) {
        substep(s, fyA, fyB);
        substep(s + 1, fyB, fyA);
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // no DMA may outlive the workgroup's LDS
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");      // the last MFMAs' results (16 passes each)

    if constexpr (BIAS) {
        if (want_bias) {                                         // lanes c32 and c32 + 32 hold the two token halves of a sub-step
#pragma unroll
            for (int t = 0; t < NRB; ++t) {
                const float tot = bsum[t] + __shfl_xor(bsum[t], 32);
                const int n = r0 + 32 * (NRB * wm + t) + c32;
                if (g == 0 && n < p.N) p.bias_parts[(int64_t)split * p.N + n] = tot;
            }
        }
    }
    // ---- the partial tile (gemm_tn8's epilogue on the NRB x NCB block layout)
    uint16_t* const out = p.parts16 + (int64_t)split * p.N * p.K;
#pragma unroll
    for (int tm = 0; tm < NRB; ++tm) {
        const int n = r0 + 32 * (NRB * wm + tm) + c32;
#pragma unroll
        for (int tn = 0; tn < NCB; ++tn) {
            const int kb = c0 + 32 * (NCB * wn + tn);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int id = tn * NRB + tm;
                const f32x16& a = id < 16 ? accA[id < 16 ? id : 0] : accV[id >= 16 ? id - 16 : 0];
                uint32_t a0 = f2bf_pair(a[8 * j], a[8 * j + 1]), a1 = f2bf_pair(a[8 * j + 2], a[8 * j + 3]);
                uint32_t b0 = f2bf_pair(a[8 * j + 4], a[8 * j + 5]), b1 = f2bf_pair(a[8 * j + 6], a[8 * j + 7]);
                auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                const int k = kb + 16 * j + 8 * g;
                if (n < p.N && k < p.K) *reinterpret_cast<u32x4v*>(out + (int64_t)n * p.K + k) = u32x4v{s0[0], s1[0], s0[1], s1[1]};
            }
        }
    }
}

template <int NRB, int NCB> constexpr int tn9_lds_bytes() {
    constexpr int ns = (64 * NRB + 127) / 128 + (64 * NCB + 127) / 128;
    return ns == 4 ? 8 * 16384 : (163840 / (ns * 4096)) * ns * 4096;
}

}  // namespace gemm
}  // namespace cream
