// gemm_nt8.hpp — the NT product of gemm_mfma.hpp on a counted-vmcnt, phase-interleaved ("ping-pong") K loop (gfx950).
//
// Reference semantics (unchanged): LinearSuper.forward / qkv_super.forward = F.linear on the active block W[:out, :in]
// (AutoFormer/model/module/Linear_super.py:38-54, :71-81; qkv_super.py:45-55, :72-83), the Mlp around them
// (supernet_transformer.py:275-285) and what autograd derives (dgrad dx = dy . W through the W^T operand copies).
//     C(M x N) = A(M x K) . B(N x K)^T      both operands K-contiguous, bf16, fp32 accumulate
//
// Why a second kernel: the loop of gemm_nt_kernel is  wait vmcnt(0) -> barrier -> issue the next K-step -> multiply,
// prefetch distance ONE K-step, every wave of the workgroup in lock-step.  On the path's shapes (M = 25,216 against
// 5..28 K-steps) that loop parks the matrix cores on HBM latency: 0.62-0.89x the vendor library on COLD operands
// (profiles/r04_gemm_probe_cold.txt).  This kernel keeps the operand layouts, segment addressing, tile order and
// epilogue semantics of gemm_nt_kernel and replaces the schedule:
//
//   * tile 256 x 256 x 64, 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 of v_mfma_f32_32x32x16_bf16 (swapped
//     product: a lane owns one output row), ONE workgroup per CU, 160 KB of LDS:
//       2 K-tile buffers x 4 half-tile slots of 16 KB  [A_q0 | A_q1 | B_q0 | B_q1]   (128 KB)
//       + 4 KB per wave of epilogue staging                                           ( 32 KB)
//     A slot A_q holds, for BOTH wave rows, the 64 rows a wave multiplies in its quadrant row q (slot row
//     s = wr * 64 + i  <->  tile row wr * 128 + q * 64 + i); B_q likewise for the four wave columns (s = wc * 32 + i
//     <-> tile column wc * 64 + q * 32 + i).  A slot therefore dies for every wave in the same phase.
//   * a K-tile is FOUR phases, one 64 x 32 quadrant of the wave tile each (8 MFMAs = 256 matrix-core cycles):
//       phase 1  C00 = A_q0 . B_q0     reads A_q0 (8 x ds_read_b128), B_q0 (4)
//       phase 2  C01 = A_q0 . B_q1     reads B_q1 (4)
//       phase 3  C11 = A_q1 . B_q1     reads A_q1 (8)
//       phase 4  C10 = A_q1 . B_q0     reads nothing (B_q0 stayed in registers)
//     Every phase is  { stage one half-tile (2 LDS-DMA per wave) | fragment reads | s_waitcnt vmcnt(8) } -> s_barrier ->
//     lgkmcnt(0) -> 8 MFMA -> s_barrier.  The waves of wave row 1 run ONE BARRIER behind
//     those of wave row 0 (waves w and w + 4 share a SIMD): while one wave of a SIMD multiplies, the other one reads
//     fragments and issues loads — the matrix pipe alternates between them instead of idling behind either's memory.
//   * the half-tiles are requested in the order they are needed, one per phase, into slots that died two phases
//     earlier:  phase 1 stages B_q1(t+1), phase 2 A_q1(t+1), phase 3 A_q0(t+2), phase 4 B_q0(t+2)  while K-tile t
//     is multiplied.  Each request is waited for FOUR phases after it was issued and read FIVE phases after
//     (vmcnt(8): the four newest half-tiles — a whole K-tile, 64 KB per CU — stay in flight across every barrier;
//     vmcnt never reaches 0 inside the loop).  Rule of cdna_hip_programming.md ("8-phase template"): the wait sits
//     before the FIRST barrier of phase p, the data is read in phase p + 1 — every wave's DMA share has then landed
//     for both staggered wave rows.
//   * the K-tile stream does not stop at a tile boundary: the persistent workgroup requests the first 1.5 K-tiles of
//     its NEXT output tile during the last six phases of the current one, so an output tile starts on landed data.
//   * epilogue without a workgroup barrier: a wave packs its accumulators to bf16 (bias / GELU applied in fp32 in the
//     accumulator layout first), pairs 8-byte runs into 16-byte chunks with v_permlane32_swap, and turns a 32 x 64
//     sub-tile through ITS OWN 4 KB of LDS so that every global store writes 8 full 128-byte lines.  Stores are newer
//     than the prefetched K-tiles and older than the next requests, and vmcnt retires in order: the first K-tile after
//     an epilogue waits with vmcnt(8 + stores) so that no wave ever waits for its own stores to reach memory.
//
// Limits (the launcher falls back to gemm_nt_kernel otherwise): element offsets of A and B fit 31 bits; N, K % 8 == 0.
#pragma once
#include "gemm_mfma.hpp"

namespace cream {
namespace gemm {

__device__ __attribute__((aligned(16))) const uint32_t g_nt8_zero[4] = {0u, 0u, 0u, 0u};

// phase stamps for tools/probes/gemm_nt_probe.hip (-DGEMM_PROFILE_NT8; compiled out of the library AND of the probe's timing build:
// a stamp is a store behind a pointer load, i.e. a vmcnt(0) in the middle of the counted pipeline)
#ifdef GEMM_PROFILE_NT8
#define NT8_PROF(i) do { if (threadIdx.x == 0 && g_gemm_prof) g_gemm_prof[(long long)blockIdx.x * 8 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define NT8_PROF(i) do {} while (0)
#endif

constexpr int NT8_LDS_BYTES = 2 * 65536 + 8 * 4096;             // 160 KB

// per-phase cycle stamps of every wave (tools/probes/nt8_trace_probe.hip, -DNT8_TRACE): s_memtime values parked in the wave's
// (idle) epilogue staging LDS during the first output tile, dumped to g_nt8_trace[block][wave][512] at the end.  A stamp is an
// SMEM read + lgkmcnt(0): only placed where no ds_read is outstanding.
#ifdef NT8_TRACE
__device__ unsigned long long* g_nt8_trace = nullptr;
#define NT8_STAMP() do { if (tr_on && tr_n < 512) { const unsigned long long t_ = __builtin_readcyclecounter(); if (lane == 0) reinterpret_cast<unsigned long long*>(stg)[tr_n] = t_; ++tr_n; } } while (0)
#else
#define NT8_STAMP() do {} while (0)
#endif

// one 1 KB global -> LDS DMA (16 B per lane, destination = wave-uniform lds_dst + 16 lane) as INLINE ASM: hipcc orders every
// later ds_read behind a builtin LDS-DMA with vmcnt(0) (a pending LDS write); an asm statement is outside its bookkeeping
// (cdna_hip_programming.md 5.7) — the counted waits below are the only ones.  M0 is restored.
__device__ __forceinline__ void nt8_dma(const void* src, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
}

// two 1-KB pieces of one half-tile in ONE statement, scalar base + 32-bit per-lane byte offsets: no VALU on the request path
// (a load segment's VALU runs beside the SIMD partner's MFMAs and is served last, MI355X_MICROARCH.md "Two waves per SIMD")
__device__ __forceinline__ void nt8_dma2(const void* base, uint32_t off0, uint32_t off1, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off0), "v"(off1), "s"(lds_dst), "s"(base) : "memory", "scc");
}

// (Measured and dropped, profiles/r05b_gemm_probe_cold_variants.txt + r05_nt8_phase_trace.txt: s_setprio 1 around every MFMA group and a
//  static s_setprio 1 for wave row 1 — no change; waiting for the fragment reads BEFORE the first barrier of a phase — 2-3 % slower.)
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt8_kernel(const NtParams p)
{
    constexpr bool STORE_AWARE = true;                           // (measured: 2-3 % over waiting for the epilogue's stores)
    constexpr int BM = 256, BN = 256, BK = 64;
    constexpr uint32_t SLOT = 16384, KTB = 65536, EPI_OFF = 2 * KTB;
    constexpr uint32_t S_A0 = 0, S_A1 = SLOT, S_B0 = 2 * SLOT, S_B1 = 3 * SLOT;
    // stores a wave issues in the epilogue of a FULL sub-tile (a lower bound is safe: a smaller vmcnt only waits longer)
    constexpr int NS = EPI == EPI_BIAS_GELU ? 32 : 16;
    static_assert(EPI == EPI_STORE || EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_MUL_COLSUM, "epilogue");
    extern __shared__ __attribute__((aligned(1024))) char nt8_lds[];
    char* const smem = nt8_lds;

    NT8_PROF(0);
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 5, c32 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>(smem));
    const int ntn = (p.N + BN - 1) / BN, ntiles = ntn * ((p.M + BM - 1) / BM);
    const int K = p.K, nk = (K + BK - 1) / BK;
    const int lda = (int)p.lda, ldb = (int)p.ldb, nseg_stride = (int)p.nseg_stride;

    // ---- staging: this lane's two 1-KB pieces of a half-tile slot (piece = wave * 2 + i: slot rows piece * 8 + lane / 8)
    int rA[2], rB[2], lc8[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int srow = (wave * 2 + i) * 8 + (lane >> 3);
        rA[i] = (srow >> 6) * 128 + (srow & 63);                 // tile row of slot row srow in A_q0 (A_q1: + 64)
        rB[i] = (srow >> 5) * 64 + (srow & 31);                  // tile column of slot row srow in B_q0 (B_q1: + 32)
        lc8[i] = ((lane & 7) ^ ((srow >> 1) & 7)) * 8;           // the k-chunk (8 values) this LDS position holds
    }
    uint32_t offA[2][2], offB[2][2];                             // BYTE offsets of the staged OUTPUT tile's rows (from p.A / p.B)
    auto set_offsets = [&](int m0, int n0) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                offA[q][i] = (uint32_t)(min(m0 + rA[i] + 64 * q, p.M - 1) * lda + lc8[i]) * 2u;
                const int n = min(n0 + rB[i] + 32 * q, p.N - 1);
                const int seg = (n >= p.nseg) + (n >= 2 * p.nseg);      // at most 3 row segments (q | k | v)
                offB[q][i] = (uint32_t)(seg * nseg_stride + (n - seg * p.nseg) * ldb + lc8[i]) * 2u;
            }
    };
    // the staging cursor: one K-tile of one output tile (all wave-uniform)
    int s_orig = blockIdx.x, s_kt = 0, s_kin = 0;
    int64_t s_kb = 0;                                            // K offset of the B operand (contraction segments)
    uint32_t s_par = 0;                                          // LDS buffer of the cursor's K-tile
    auto tile_origin = [&](int orig, int& m0, int& n0) {
        const int t = xcd_remap(orig, ntiles);
        m0 = (t / ntn) * BM; n0 = (t % ntn) * BN;
    };
    auto advance = [&]() {
        s_par ^= 1;
        if (s_kt + 1 < nk) {
            ++s_kt; s_kb += BK; s_kin += BK;
            if (s_kin >= p.kseg) { s_kin = 0; s_kb += p.kseg_stride - p.kseg; }
        } else {
            const int next = s_orig + (int)gridDim.x;
            if (next < ntiles) {                                 // (past the last tile the cursor stays: duplicates nobody reads)
                s_orig = next; s_kt = 0; s_kb = 0; s_kin = 0;
                int m0, n0;
                tile_origin(next, m0, n0);
                set_offsets(m0, n0);
            }
        }
    };
    // half-tile `which` (0 A_q0, 1 A_q1, 2 B_q0, 3 B_q1) of the cursor's K-tile -> its slot
    auto stage = [&](auto which_) {
        constexpr int which = decltype(which_)::value;
        const int kvalid = K - s_kt * BK;
        const char* base = which < 2 ? reinterpret_cast<const char*>(p.A) + (int64_t)s_kt * (BK * 2) : reinterpret_cast<const char*>(p.B) + s_kb * 2;
        const uint32_t dst = lds0 + s_par * KTB + which * SLOT + wave * 2048;
        const uint32_t o0 = which < 2 ? offA[which & 1][0] : offB[which & 1][0], o1 = which < 2 ? offA[which & 1][1] : offB[which & 1][1];
        if (kvalid >= BK) {
            nt8_dma2(base, o0, o1, dst);
        } else {                                                 // the last K-tile of a K % 64 != 0: chunks beyond K come from zeros
            nt8_dma(lc8[0] < kvalid ? base + o0 : reinterpret_cast<const char*>(g_nt8_zero), dst);
            nt8_dma(lc8[1] < kvalid ? base + o1 : reinterpret_cast<const char*>(g_nt8_zero), dst + 1024);
        }
    };
    using W_A0 = std::integral_constant<int, 0>; using W_A1 = std::integral_constant<int, 1>;
    using W_B0 = std::integral_constant<int, 2>; using W_B1 = std::integral_constant<int, 3>;

    // ---- fragment reads: row c32 of a 32-row block, 16-byte chunk (2 ks + g) ^ sw of its 128-byte row
    const int sw = (c32 >> 1) & 7;
    uint32_t aoff[4], boff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint32_t ch = (uint32_t)(((ks * 2 + g) ^ sw) << 4);
        aoff[ks] = (uint32_t)(wr * 64 + c32) * 128 + ch;
        boff[ks] = (uint32_t)(wc * 32 + c32) * 128 + ch;
    }
    auto ldfrag = [&](uint32_t off) -> bf16x8 { return *reinterpret_cast<const bf16x8*>(smem + off); };

    f32x16 acc[2][4];                                            // [tn = quadrant column][tm = 2 * quadrant row + tm2]
    bf16x8 fa[2][4], fb0[4], fb1[4];                             // A fragments [tm2][ks], B fragments [ks] of both quadrant columns

    // ---- epilogue geometry
    char* const stg = smem + EPI_OFF + wave * 4096;              // this wave's 32 x 64 bf16 staging tile
    const int rrow = lane >> 3, rlc = lane & 7;                  // after the turn: row rrow + 8 i, 16-byte chunk rlc

    // ---- prologue: K-tile 0 whole, K-tile 1's first two half-tiles (what phases 3, 4 | 1, 2 | 3, 4 would have requested)
    int orig = blockIdx.x, m0, n0;
    tile_origin(orig, m0, n0);
    set_offsets(m0, n0);
    stage(W_A0{}); stage(W_B0{}); stage(W_B1{}); stage(W_A1{});
    advance();
    stage(W_A0{}); stage(W_B0{});
    wait_vmcnt<8>();
    __builtin_amdgcn_s_barrier();
    NT8_PROF(1);
    if (wr == 1) __builtin_amdgcn_s_barrier();                   // wave row 1 runs one barrier behind wave row 0
    __builtin_amdgcn_sched_barrier(0);

    uint32_t c_par = 0;                                          // LDS buffer of the K-tile being multiplied
    // side inputs requested in the last phase of a tile (see phase 4)
    constexpr int NPRE = (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) ? 8 : EPI == EPI_MUL_COLSUM ? 4 : 0;
    u32x2v braw[2][4];
    u32x4v aux0[4];
#ifdef NT8_TRACE
    int tr_n = 0; bool tr_on = true;
#endif
    bool after_full_epi = false;                                 // this wave issued exactly NS stores since its last request

// `counted`: this wave issued exactly NS stores since its last request (first K-tile after the epilogue of a full sub-tile);
// `pre`: NPRE epilogue side-input loads were issued in this phase (phase 4 of a tile's last K-tile).  Both are newer than the
// half-tile being waited for: they join the in-flight allowance, so that neither is ever waited for inside the loop.
#define NT8_PRE(counted, pre)                                                                    \
    do {                                                                                         \
        if (counted) { if (pre) wait_vmcnt<8 + NS + NPRE>(); else wait_vmcnt<8 + NS>(); }        \
        else { if (pre) wait_vmcnt<8 + NPRE>(); else wait_vmcnt<8>(); }                          \
        __builtin_amdgcn_s_barrier();                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                       \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        NT8_STAMP();                                                                             \
    } while (0)
#define NT8_POST()                                                                               \
    do {                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        NT8_STAMP();                                                                             \
        __builtin_amdgcn_s_barrier();                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        NT8_STAMP();                                                                             \
    } while (0)
#define NT8_MMA(TN_, QA_, FB_)                                                                   \
    do {                                                                                         \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                         \
            _Pragma("unroll") for (int t2 = 0; t2 < 2; ++t2)                                     \
                acc[TN_][2 * (QA_) + t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FB_[ks], fa[t2][ks], acc[TN_][2 * (QA_) + t2], 0, 0, 0); \
    } while (0)

    for (;;) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

        // one K-tile; the LAST one of an output tile is its own instantiation (the side-input requests of its phase 4 are then
        // not carried around the loop: they take the registers fb1 has left)
        auto ktile = [&](int kt, auto last_) {
            constexpr bool LAST = decltype(last_)::value;
            const uint32_t bufoff = c_par * KTB;
            const bool counted = after_full_epi && kt == 0;
            // ---- phase 1: C00
            stage(W_B1{});
            NT8_STAMP();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fb0[ks] = ldfrag(bufoff + S_B0 + boff[ks]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) fa[t2][ks] = ldfrag(bufoff + S_A0 + t2 * 4096 + aoff[ks]);
            NT8_PRE(counted, false);
            NT8_MMA(0, 0, fb0);
            NT8_POST();
            // ---- phase 2: C01
            stage(W_A1{});
            NT8_STAMP();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fb1[ks] = ldfrag(bufoff + S_B1 + boff[ks]);
            NT8_PRE(counted, false);
            NT8_MMA(1, 0, fb1);
            NT8_POST();
            // ---- phase 3: C11 (the cursor moves on to K-tile t + 2: the buffer being multiplied, slots dead since phase 1)
            advance();
            stage(W_A0{});
            NT8_STAMP();
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) fa[t2][ks] = ldfrag(bufoff + S_A1 + t2 * 4096 + aoff[ks]);
            NT8_PRE(counted, false);
            NT8_MMA(1, 1, fb1);
            NT8_POST();
            // ---- phase 4: C10
            stage(W_B0{});
            NT8_STAMP();
            // the epilogue's side inputs of this lane (bias in the accumulator layout / the first 32 rows of the element-wise
            // factor) are requested HERE, in the last phase of the tile's last K-tile, into the registers fb1 has left: their
            // latency passes behind the last MFMAs instead of in front of the epilogue.  Always NPRE loads (clamped addresses).
            const bool pre = LAST && NPRE > 0 && (EPI == EPI_MUL_COLSUM || p.bias != nullptr);
            if constexpr (LAST) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) braw[a][b] = u32x2v{0, 0};
            }
            if (pre) {
                if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) {
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4)
                            braw[tn][r4] = *reinterpret_cast<const u32x2v*>(p.bias + min(n0 + wc * 64 + tn * 32 + 8 * r4 + 4 * g, p.N - 4));
                } else if constexpr (EPI == EPI_MUL_COLSUM) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        aux0[i] = *reinterpret_cast<const u32x4v*>(p.aux + (int64_t)min(m0 + wr * 128 + rrow + 8 * i, p.M - 1) * p.ldaux +
                                                                   min(n0 + wc * 64 + rlc * 8, p.N - 8));
                }
            }
            NT8_PRE(counted, pre);
            NT8_MMA(0, 1, fb0);
            NT8_POST();
            c_par ^= 1;
        };
        for (int kt = 0; kt < nk - 1; ++kt) ktile(kt, std::false_type{});
        ktile(nk - 1, std::true_type{});
        if (orig == (int)blockIdx.x) NT8_PROF(2);
#ifdef NT8_TRACE
        if (tr_on && g_nt8_trace) {                              // (before the epilogue reuses the staging LDS)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            unsigned long long* d = g_nt8_trace + ((size_t)blockIdx.x * 8 + wave) * 512;
            for (int i = lane; i < 512; i += 64) d[i] = i < tr_n ? reinterpret_cast<unsigned long long*>(stg)[i] : 0ull;
        }
        tr_on = false;
#endif

        // ---- epilogue (no workgroup barrier: each wave turns its 128 x 64 sub-tile through its own 4 KB of LDS)
        const int wm0 = m0 + wr * 128, wn0 = n0 + wc * 64;       // origin of this wave's sub-tile
        const bool full = wm0 + 128 <= p.M && wn0 + 64 <= p.N;
        // bias of this lane's columns in the accumulator layout (column tn * 32 + 8 r4 + 4 g + e), requested in phase 4
        float bv[2][4][4];
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const u32x2v b2 = braw[tn][r4];
                bv[tn][r4][0] = __uint_as_float(b2[0] << 16); bv[tn][r4][1] = __uint_as_float(b2[0] & 0xFFFF0000u);
                bv[tn][r4][2] = __uint_as_float(b2[1] << 16); bv[tn][r4][3] = __uint_as_float(b2[1] & 0xFFFF0000u);
            }
        const int ncol = wn0 + rlc * 8;                          // first of this lane's 8 columns after the turn
        const bool ncol_ok = ncol < p.N;                         // N % 8 == 0: a chunk is all in or all out
        float cs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        // EPI_MUL_COLSUM: ALL sixteen side-input chunks of this lane are requested here, into the registers the fragments have
        // left (one exposed memory latency per output tile instead of one per 8-row group)
        u32x4v auxv[EPI == EPI_MUL_COLSUM ? 4 : 1][EPI == EPI_MUL_COLSUM ? 4 : 1];
        if constexpr (EPI == EPI_MUL_COLSUM) {
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = wm0 + tm * 32 + rrow + 8 * i;
                    if (tm == 0) auxv[tm][i] = aux0[i];          // (requested in phase 4)
                    else auxv[tm][i] = (full || (m < p.M && ncol_ok)) ? *reinterpret_cast<const u32x4v*>(p.aux + (int64_t)m * p.ldaux + ncol) : u32x4v{0, 0, 0, 0};
                }
        }
        // one packed 32 x 64 sub-tile (pk[tn][2 r4 + half]: columns tn * 32 + 8 r4 + 4 g + 2 half, + 1) -> rows of `dst`
        auto turn_and_store = [&](uint32_t (&pk)[2][8], uint16_t* dst, int tm) {
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    // 8-byte runs of r4 = 2 j (group k) and 2 j + 1 (group k + 1): after the half exchange lanes 0-31 hold columns
                    // 16 j .. + 7, lanes 32-63 columns 16 j + 8 .. + 15 of the 32-column tile (cdna_hip_programming.md T21)
                    uint32_t a0 = pk[tn][4 * j], a1 = pk[tn][4 * j + 1], b0 = pk[tn][4 * j + 2], b1 = pk[tn][4 * j + 3];
                    auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                    auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                    const int chunk = tn * 4 + 2 * j + g;
                    *reinterpret_cast<u32x4v*>(stg + c32 * 128 + ((chunk ^ (c32 & 7)) << 4)) = u32x4v{r0[0], r1[0], r0[1], r1[1]};
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rr = rrow + 8 * i, m = wm0 + tm * 32 + rr;
                u32x4v v = *reinterpret_cast<const u32x4v*>(stg + rr * 128 + ((rlc ^ (rr & 7)) << 4));
                uint16_t* o = dst + (int64_t)m * p.ldo + ncol;
                if constexpr (EPI == EPI_MUL_COLSUM) {
                    // dh = bf16(dy . W2) * gelu'(h): the product is rounded to bf16 first, as the reference's two operators do
                    // (F.linear under autocast, then the GELU backward), then multiplied in fp32 and rounded once more
                    if (full || (m < p.M && ncol_ok)) {
                        const u32x4v fbv = auxv[tm][i];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = f2bf_pair(__uint_as_float(v[e] << 16) * __uint_as_float(fbv[e] << 16),
                                             __uint_as_float(v[e] & 0xFFFF0000u) * __uint_as_float(fbv[e] & 0xFFFF0000u));
                            cs[2 * e] += __uint_as_float(v[e] << 16);              // sums of the ROUNDED values written
                            cs[2 * e + 1] += __uint_as_float(v[e] & 0xFFFF0000u);
                        }
                    }
                }
                if (full) st_out16(o, v);
                else if (m < p.M && ncol_ok) st_out16(o, v);
            }
        };
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
            uint32_t pk[2][8];
            uint32_t pk2[EPI == EPI_BIAS_GELU ? 2 : 1][8];
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float v0 = acc[tn][tm][4 * r4 + 2 * h] + bv[tn][r4][2 * h];
                        const float v1 = acc[tn][tm][4 * r4 + 2 * h + 1] + bv[tn][r4][2 * h + 1];
                        if constexpr (EPI == EPI_BIAS_GELU) {
                            // fc1 under autocast yields bf16 h; gelu runs in fp32 ON that bf16 value and casts back
                            // (supernet_transformer.py:14-16, :276-277); gelu'(h) is written INSTEAD of h
                            const uint32_t hb = f2bf_pair(v0, v1);
                            const float h0 = __uint_as_float(hb << 16), h1 = __uint_as_float(hb & 0xFFFF0000u);
                            float c0, e0, c1, e1;
                            phi_parts(h0, c0, e0);
                            phi_parts(h1, c1, e1);
                            uint32_t gb = f2bf_pair(h0 * c0, h1 * c1);
                            uint32_t pb = f2bf_pair(fmaf(h0 * 0.3989422804014327f, e0, c0), fmaf(h1 * 0.3989422804014327f, e1, c1));
                            pk[tn][2 * r4 + h] = pb;
                            pk2[tn][2 * r4 + h] = gb;
                        } else {
                            pk[tn][2 * r4 + h] = f2bf_pair(v0, v1);
                        }
                    }
            if constexpr (EPI == EPI_BIAS_GELU) {
                if (wn0 + 64 > p.nvalid) {                       // (wave-uniform, rare) padded columns: exact zeros — their gradients vanish
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                        for (int d = 0; d < 8; ++d) {
                            const int n = wn0 + tn * 32 + 8 * (d >> 1) + 4 * g + 2 * (d & 1);
                            const uint32_t keep = (n < p.nvalid ? 0x0000FFFFu : 0u) | (n + 1 < p.nvalid ? 0xFFFF0000u : 0u);
                            pk[tn][d] &= keep; pk2[tn][d] &= keep;
                        }
                }
                if (p.out) turn_and_store(pk, p.out, tm);       // (no gelu' without a backward: inference, frozen teacher)
                turn_and_store(pk2, p.out2, tm);
            } else {
                turn_and_store(pk, p.out, tm);
            }
        }
        if constexpr (EPI == EPI_MUL_COLSUM) {
            // column sums of this wave's 128 rows = one 128-row slab (cream_gemm_rows_per_colsum_slab): lanes of equal
            // chunk rlc hold 16 rows each; fixed-order butterfly over the 8 row lanes
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float s = cs[e];
                s += __shfl_xor(s, 8, 64);
                s += __shfl_xor(s, 16, 64);
                s += __shfl_xor(s, 32, 64);
                cs[e] = s;
            }
            if (rrow == 0 && ncol_ok && wm0 < p.M) {
                float* d = p.colsum + (int64_t)(wm0 / 128) * p.N + ncol;
                *reinterpret_cast<f32x4v*>(d) = f32x4v{cs[0], cs[1], cs[2], cs[3]};
                *reinterpret_cast<f32x4v*>(d + 4) = f32x4v{cs[4], cs[5], cs[6], cs[7]};
            }
        }
        if (orig == (int)blockIdx.x) NT8_PROF(3);
        after_full_epi = full && (EPI != EPI_BIAS_GELU || p.out != nullptr);
        orig += (int)gridDim.x;
        if (orig >= ntiles) break;
        tile_origin(orig, m0, n0);
        __builtin_amdgcn_sched_barrier(0);
    }
#undef NT8_PRE
#undef NT8_POST
#undef NT8_MMA
    NT8_PROF(4);
    if (wr == 0) __builtin_amdgcn_s_barrier();                   // matches wave row 1's last barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // no DMA may outlive the workgroup's LDS
}

}  // namespace gemm
}  // namespace cream
