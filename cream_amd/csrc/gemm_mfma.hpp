// gemm_mfma.hpp — hand-written MFMA GEMMs of the weight-entangled projections (gfx950).
//
// Reference semantics: LinearSuper.forward / qkv_super.forward = F.linear on the active block
// W[:out, :in] of the super weight (AutoFormer/model/module/Linear_super.py:38-54, :71-81;
// qkv_super.py:45-55, :72-83), the Mlp around them (supernet_transformer.py:275-285: fc1 -> gelu in
// fp32 -> fc2) and what autograd derives (dgrad dx = dy . W, wgrad dW = dy^T x).
//
// Shapes of the path: M = B*N tokens = 25,216 (128 images x 197), K and N in 320..1792 — a huge M
// against 5..28 K-steps, so prologue / epilogue / tile quantisation weigh as much as the main loop.
//
// "NT" kernel (forward and, on TRANSPOSED operand copies of the weights, dgrad):
//     C(M x N) = A(M x K) . B(N x K)^T      both operands K-contiguous, bf16, fp32 accumulate
//   * workgroup = 4 waves (2 x 2), tile BM x BN (128x128 / 128x64 / 64x128 / 64x64), BK = 64;
//     wave tile (BM/2) x (BN/2) of v_mfma_f32_32x32x16_bf16;
//   * operands go global -> LDS directly (global_load_lds_dwordx4, 1 KB per wave instruction, no
//     VGPR round trip); the LDS image is lane-linear [row][8 chunks of 16 B], bank conflicts of the
//     ds_read_b128 fragment reads are removed by XOR-swizzling the SOURCE chunk with (row >> 1) & 7
//     (two 128-B rows share one 256-B bank row: with this term the 16 lanes of every b128 lane group
//     hit 16 distinct 16-B slots) and the same XOR on the read;
//   * two LDS stages: the loads of K-step s+1 are in flight while step s is multiplied, one barrier
//     per K-step; 2 (128x128) to 3 (128x64) workgroups per CU overlap each other's barriers,
//     prologues and epilogues;
//   * swapped product D^T = B_tile . A_tile^T: a lane owns ONE output row and 4 runs of 4 columns, so
//     the accumulators leave through LDS as fp32 with 16-byte writes, and every epilogue works on
//     (row, 8 consecutive columns) chunks: 16-byte coalesced loads of the side inputs (bias, the
//     pre-GELU activations) and 16-byte row-contiguous stores;
//   * 1-D grid, XCD-aware tile order (column tiles of one row panel adjacent and on one XCD: the
//     A panel is fetched from HBM once and re-read from that XCD's L2);
//   * B addressing covers the weight-entangled layouts without copies: rows may be split in `nseg`
//     row segments (the q / k / v thirds of the de-interleaved qkv weight) and the contraction in
//     `kseg` segments (the same thirds on the transposed copy used by dgrad); leading dimension =
//     super width (the active block W[:N, :K] is read in place).
// Epilogues (EPI_*): plain store; + bias; + bias and erf-GELU, writing gelu(h) AND gelu'(h) (fc1: the
// derivative shares the expensive terms with the forward value, so the backward never evaluates erf);
// x an element-wise factor (the saved gelu'(h)) with column sums of the result (fc2 dgrad + GELU
// backward + fc1 bias gradient).
//
// M, N arbitrary (N % 8 == 0), K % 8 == 0.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "attn_common.hpp"

namespace cream {
namespace gemm {

enum Epi { EPI_STORE = 0, EPI_BIAS = 1, EPI_BIAS_GELU = 2, EPI_MUL_COLSUM = 3 };

// Phase timestamps for tools/probes/gemm_nt_probe.hip (compiled out of the library)
#ifdef GEMM_PROFILE
__device__ long long* g_gemm_prof = nullptr;
#define GPROF(i) do { if (threadIdx.x == 0 && g_gemm_prof) g_gemm_prof[(long long)blockIdx.x * 8 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define GPROF(i) do {} while (0)
#endif

struct NtParams {
    const uint16_t* A;      // (M x K) bf16, row stride lda
    const uint16_t* B;      // element (n, k) at B[(n / nseg) * nseg_stride + (n % nseg) * ldb + (k / kseg) * kseg_stride + k % kseg]
    int64_t lda, ldb, nseg_stride, kseg_stride;
    int nseg, kseg;         // nseg >= N and kseg >= K for a plain matrix; kseg % 64 == 0 when kseg < K; N <= 3 nseg
    int M, N, K;
    uint16_t* out;          // (M x N) bf16, row stride ldo  (EPI_BIAS_GELU: gelu'(h), h = bf16(x . W^T + bias))
    uint16_t* out2;         // EPI_BIAS_GELU: gelu(h), same layout
    int64_t ldo;
    const uint16_t* bias;   // (N) bf16 or nullptr                      (EPI_BIAS, EPI_BIAS_GELU)
    const uint16_t* aux;    // EPI_MUL_COLSUM: element-wise factor (M x N) bf16, row stride ldaux
    int64_t ldaux;
    float* colsum;          // EPI_MUL_COLSUM: [ceil(M / BM)][N] per-row-tile column sums of `out`
    int nvalid;             // EPI_BIAS_GELU: columns n >= nvalid are written as zeros (N padded up to a multiple of 8)
    int stagger;            // > 0: the second half of the grid starts stagger x 64 clocks late (two workgroups per CU out of phase)
};

// erf-GELU in the epilogues: Phi(x) = 0.5 (1 + erf(x / sqrt 2)) with erf by Abramowitz & Stegun 7.1.26
// (|error| <= 1.5e-7 absolute — three orders below the bf16 resolution of the values written; the
// library erff costs ~3x the instructions and made the fc1 epilogue as long as the product itself).
// e = exp(-x^2 / 2) is shared with the density needed by the derivative.
__device__ __forceinline__ void phi_parts(float x, float& cdf, float& e) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
    float q = fmaf(t, 1.061405429f, -1.453152027f);
    q = fmaf(q, t, 1.421413741f);
    q = fmaf(q, t, -0.284496736f);
    q = fmaf(q, t, 0.254829592f);
    e = __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);
    const float h = 0.5f * q * t * e;                           // 0.5 erfc(z)
    cdf = x < 0.f ? h : 1.f - h;
}
__device__ __forceinline__ float gelu_f(float x) { float c, e; phi_parts(x, c, e); return x * c; }
__device__ __forceinline__ float gelu_grad_f(float x) {
    float c, e;
    phi_parts(x, c, e);
    return fmaf(x * 0.3989422804014327f, e, c);
}

// 16-byte output store of the NT epilogues.  CREAM_NT_OUT_NT (compile-time, measured in profiles/r05_nt_out_stores.md): non-temporal,
// so that the output stream is the first thing the L2 evicts instead of the operand panels its column-tile neighbours still want.
#ifndef CREAM_NT_OUT_NT
#define CREAM_NT_OUT_NT 1
#endif
__device__ __forceinline__ void st_out16(uint16_t* dst, const u32x4v& v) {
#if CREAM_NT_OUT_NT
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4v*>(dst));
#else
    *reinterpret_cast<u32x4v*>(dst) = v;
#endif
}

// XCD-aware tile order (bijective for any grid size): workgroup ids are dealt round-robin to the 8
// XCDs; the remap gives every XCD a CONTIGUOUS range of tiles.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int q = nwg / 8, r = nwg % 8, xcd = orig % 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
}

__host__ __device__ constexpr int nt_lds_bytes(int BM, int BN, int NST) { return NST * (BM + BN) * 64 * 2; }

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---- OPT (round 6, last template parameter of gemm_nt_kernel): the tile epilogue off the memory counters ------------------
// What the ISA of the OPT = 0 kernels shows (tools/isa.sh): the epilogue of a tile is serialised THREE times on memory
// round trips that have nothing to do with it — __syncthreads() drains vmcnt, so (1) the first epilogue barrier waits for
// the NEXT tile's first K-step (requested just before: a full L2 / HBM latency), (2) the barrier between the passes waits
// for the write-back acknowledgement of the first pass's stores, (3) the first K-step of the next tile (vmcnt(0)) waits
// for the second pass's — and under the GELU epilogue every SIMD spends ~2,000 VALU instructions per wave and tile on
// erf / exp.  With OPT & 1:
//   * the LDS-DMA of the operand tiles is an asm statement (the compiler orders every later LDS access behind a BUILTIN
//     LDS-DMA with vmcnt(0)); every read of a stage already sits behind an explicit wait + barrier;
//   * the epilogue's barriers wait for LDS traffic only (lgkmcnt(0) + s_barrier);
//   * the side inputs of the epilogue (bias chunk, the x gelu' factor rows) are requested under the first K-step and are
//     in registers before the next tile's DMA is issued (a compiler-placed wait for them behind that DMA would drain it);
//   * the first K-step after the epilogue of a full tile waits with vmcnt(stores of the epilogue): the stores drain under it.
// With OPT & 2 (EPI_BIAS_GELU): gelu(h) and gelu'(h) of the bf16-ROUNDED h come from a 16 KB LDS table over
// [sign | exponent 115..130 | mantissa] = every bf16 value with 2^-12 <= |h| < 16, filled once per workgroup by the very
// phi_parts() the direct path evaluates (identical bits); |h| < 2^-12: gelu' = 1/2 and gelu = h / 2 exactly (both roundings
// proven by the exhaustive test, tests/test_block_gpu.py); |h| >= 16, inf, nan: the wave falls back to the direct evaluation
// of that chunk.  ~8 VALU + 1 LDS gather per element instead of ~31 VALU.
constexpr int GELU_TAB_BYTES = 16384;
constexpr uint32_t GELU_TAB_LO = 115u << 7;                     // bf16 bits of 2^-12

__device__ __forceinline__ void nt_dma16(const uint16_t* src, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
}
// LDS-only workgroup barrier (global loads / stores / DMA stay in flight across it)
__device__ __forceinline__ void nt_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
typedef short nt_s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short nt_u16x2 __attribute__((ext_vector_type(2)));

// BM x BN tile, WM x WN waves, NST LDS stages (prefetch distance NST - 1 K-steps, counted vmcnt: the
// loads of later steps stay in flight across the per-step barrier), OCC = workgroups per CU wanted.
//
// PERSISTENT tile loop: a workgroup walks tiles blockIdx.x, + gridDim.x, ... (gridDim.x % 8 == 0 or
// gridDim.x == number of tiles, so that a workgroup's tiles stay on its XCD).  Phase stamps of the
// one-tile-per-workgroup version: at K = 384 a workgroup spent a third of its life between
// launch and its first staged tile and a sixth in the epilogue.  Here the first K-step of the NEXT tile is
// requested before the epilogue of the current one — into stage 0, while the epilogue takes the accumulators
// through the LDS of stage 1 in two half-tile passes (XOR-swizzled fp32 rows without padding: 64 rows x BN
// floats are exactly one stage) — so its latency hides behind the epilogue instead of idling the workgroup.
// LDS of a kernel instantiation: a static array up to 64 KB, the dynamic segment above (the 256 x 256 macro tile needs
// 128 KB: the launcher raises hipFuncAttributeMaxDynamicSharedMemorySize and passes the size).  ONE object either way.
template <int BYTES, bool DYN = (BYTES > 65536)> struct LdsBlock {
    static __device__ __forceinline__ char* get() {
        __shared__ __attribute__((aligned(1024))) char s[BYTES];
        return s;
    }
};
template <int BYTES> struct LdsBlock<BYTES, true> {
    static __device__ __forceinline__ char* get() {
        extern __shared__ __attribute__((aligned(1024))) char dyn_lds[];
        return dyn_lds;
    }
};

template <int BM, int BN, int WM, int WN, int NST, int EPI, int OCC, int OPT = 0>
__global__ __launch_bounds__(WM * WN * 64, (OCC * WM * WN + 3) / 4) void gemm_nt_kernel(const NtParams p)
{
    constexpr int BK = 64;
    constexpr bool OPT1 = (OPT & 1) != 0;                       // epilogue off the memory counters (see above)
    constexpr bool TAB = (OPT & 2) != 0 && EPI == EPI_BIAS_GELU;   // gelu / gelu' from the LDS table
    static_assert(!(OPT & 2) || OPT1, "the table epilogue builds on the OPT & 1 schedule");
    constexpr int NW = WM * WN, NT = NW * 64;
    constexpr int WTM = BM / WM, WTN = BN / WN;                 // wave tile
    constexpr int TM = WTM / 32, TN = WTN / 32;                 // MFMA tiles per wave
    constexpr int STAGE = (BM + BN) * BK;                       // bf16 elements per stage
    constexpr int PIECES = (BM + BN) / 8;                       // 1-KB pieces (8 rows) per stage
    constexpr int NPIECE = (PIECES + NW - 1) / NW;              // ... per wave (the last one only for waves < PIECES % NW when ragged)
    constexpr bool RAGGED = PIECES % NW != 0;
    constexpr int NCH = BN / 4;                                 // 4-float chunks per epilogue row
    constexpr bool POW2 = (BN & (BN - 1)) == 0;                 // BN = E / 2 tiles (160 / 192 / 224): rotation instead of XOR swizzle
    // the epilogue takes the fp32 tile through ONE stage's LDS in passes of HM rows: as many 32-row MFMA tiles of a wave
    // row as fit (128 x 128, 128 x 64: 64 rows = a whole wave row, two passes; 256 x 256: 64 of a wave's 128 rows, four), or
    // a whole number of wave rows where a wave row is shorter than what fits (256 x 160 / 224 with eight wave rows of 32: two)
    constexpr int HM_FIT = (STAGE * 2) / (BN * 4) / 32 * 32;
    constexpr int HM = HM_FIT < WTM ? HM_FIT : (HM_FIT / WTM) * WTM > BM ? BM : (HM_FIT / WTM) * WTM;   // rows of one epilogue pass
    constexpr int WR = HM > WTM ? HM / WTM : 1;                 // wave rows that write in one pass
    constexpr int NPASS = BM / HM, TMP = (HM > WTM ? WTM : HM) / 32;   // passes per tile, MFMA row tiles per wave and pass
    constexpr int SLAB = 128;                                   // rows of one column-sum slab (cream_gemm_rows_per_colsum_slab)
    static_assert(WTM % 32 == 0 && WTN % 32 == 0 && (BM + BN) % 8 == 0 && (BM / 8) % NW == 0, "tile");
    static_assert(NST == 2, "the epilogue borrows stage 1 while stage 0 receives the next tile");
    static_assert(HM >= 32 && (WTM % HM == 0 || HM % WTM == 0) && BM % HM == 0 && HM * BN * 4 <= STAGE * 2, "epilogue passes: whole MFMA tiles, inside one stage");
    static_assert(EPI != EPI_MUL_COLSUM || (BM % SLAB == 0 && SLAB % HM == 0), "column sums leave per 128-row slab");
    char* const smem = LdsBlock<NST * STAGE * 2 + (TAB ? GELU_TAB_BYTES : 0)>::get();
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>(smem));
    uint16_t* const lds = reinterpret_cast<uint16_t*>(smem);
    float* const ctile = reinterpret_cast<float*>(smem + STAGE * 2);           // stage 1

    GPROF(0);
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 5, c32 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int ntn = (p.N + BN - 1) / BN, ntiles = ntn * ((p.M + BM - 1) / BM);
    const int K = p.K;

    // ---- per-lane sources of this wave's pieces for a tile (row + this lane's k-chunk; the K offset moves per step)
    int cch[NPIECE];                                            // the k-chunk (8 values) this lane fetches
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
        const int row = (wave + NW * i) * 8 + (lane >> 3);
        cch[i] = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
    }
    auto sources = [&](const uint16_t* (&src)[NPIECE], int m0, int n0) {
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) {
            const int piece = wave + NW * i, row = piece * 8 + (lane >> 3);
            if (row < BM) {
                src[i] = p.A + (int64_t)min(m0 + row, p.M - 1) * p.lda + cch[i];
            } else {
                const int n = min(n0 + min(row - BM, BN - 1), p.N - 1);    // (ragged last piece: past the tile, never issued)
                const int seg = (n >= p.nseg) + (n >= 2 * p.nseg);  // at most 3 row segments (q | k | v): no division
                src[i] = p.B + seg * p.nseg_stride + (int64_t)(n - seg * p.nseg) * p.ldb + cch[i];
            }
        }
    };
    // TAIL: the last K-step of a K that is not a multiple of 64 — chunks beyond K re-read the last
    // valid chunk (their products are zeroed in `step`), so no load leaves the row.
    // kb: wave-uniform K offset of the B operand for the step being issued (contraction segments)
    int64_t kb = 0;
    int kin = 0;                                                // position inside the current segment
    auto issue = [&](const uint16_t* const (&src)[NPIECE], int k0, int buf, auto tail) {
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) {
            const bool isA = i < BM / 8 / NW;                    // pieces wave + NW i < BM / 8 hold A rows
            if constexpr (RAGGED) { if (i == NPIECE - 1 && wave + NW * i >= PIECES) continue; }    // (wave-uniform)
            const uint16_t* s = src[i] + (isA ? (int64_t)k0 : kb);
            if constexpr (decltype(tail)::value) s += min(0, K - 8 - (k0 + cch[i]));
            if constexpr (OPT1) nt_dma16(s, lds0 + (uint32_t)((buf * STAGE + (wave + NW * i) * 8 * BK) * 2));
            else
            __builtin_amdgcn_global_load_lds(
                s, reinterpret_cast<__attribute__((address_space(3))) void*>(
                       reinterpret_cast<uintptr_t>(lds + buf * STAGE + (wave + NW * i) * 8 * BK)), 16, 0, 0);
        }
        kb += BK;
        kin += BK;
        if (kin >= p.kseg) { kin = 0; kb += p.kseg_stride - p.kseg; }
    };
    const int sw = (c32 >> 1) & 7;
    auto frag = [&](const uint16_t* tile, int row, int chunk) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(tile + row * BK + ((chunk ^ sw) << 3));
    };

    f32x16 acc[TN][TM];

    // one K-step from stage `buf`; vc = valid 8-wide chunks (8 unless TAIL).  Fragments are double
    // buffered in registers: the reads of sub-step ks+1 are issued BEFORE the MFMAs of sub-step ks
    // (pinned with sched_group_barrier), so LDS latency hides behind matrix work instead of being
    // paid four times per K-step by an in-order wave.
    auto step_into = [&](f32x16 (&acc)[TN][TM], int buf, int vc, auto tail) {
        constexpr bool T = decltype(tail)::value;
        const uint16_t* At = lds + buf * STAGE;
        const uint16_t* Bt = At + BM * BK;
        bf16x8 fa[2][TM], fb[2][TN];
        auto load = [&](int ks, int slot) {
            const bool dead = T && (ks * 2 + g) >= vc;          // half of a 16-wide sub-step beyond K
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                fa[slot][i] = frag(At, wm * WTM + i * 32 + c32, ks * 2 + g);
                if (T && dead) fa[slot][i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                fb[slot][i] = frag(Bt, wn * WTN + i * 32 + c32, ks * 2 + g);
                if (T && dead) fb[slot][i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        };
        load(0, 0);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            if (T && ks * 2 >= vc) break;
            if (ks + 1 < BK / 16) load(ks + 1, (ks + 1) & 1);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks & 1][tn], fa[ks & 1][tm], acc[tn][tm], 0, 0, 0);
        }
        if constexpr (!T) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TM + TN), 0);      // reads of sub-steps 0 and 1
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);        // MFMAs of sub-step ks
                if (ks + 2 < BK / 16) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);   // reads of ks + 2
            }
        }
    };
    auto step = [&](int buf, int vc, auto tail) { step_into(acc, buf, vc, tail); };
    using No = std::integral_constant<bool, false>;
    using Yes = std::integral_constant<bool, true>;
    const int nfull = K / BK, rem = K % BK, nk = nfull + (rem ? 1 : 0);

    // epilogue geometry: a thread owns an 8-column chunk of rows r0, r0 + RPP, ... of each half tile
    constexpr int CPR = BN / 8;                                 // chunks per tile row
    constexpr int RPP = NT / CPR;                               // rows per pass of the workgroup
    constexpr int NJ = (HM + RPP - 1) / RPP;                    // rows per thread and half
    constexpr bool EXACT = NT % CPR == 0 && HM % RPP == 0;      // every thread has a chunk in every row group (power-of-two tiles)
    const int cc = tid % CPR, r0 = tid / CPR;
    const bool epi_thread = EXACT || tid < CPR * RPP;           // (BN = 160 / 192 / 224: 500 / 504 / 504 of 512 threads)
    // chunk position inside a staged fp32 row: XOR with the row for power-of-two rows, rotation by the row otherwise
    auto cpos = [&](int ch, int row) -> int {
        if constexpr (POW2) return ch ^ (row & (NCH - 1));
        else return (ch + row) % NCH;
    };

    // younger vector-memory instructions every wave is sure to have issued behind the next tile's first K-step when the
    // epilogue of a FULL tile is over (a lower bound keeps the counted wait safe): the row stores
    constexpr int NSTMIN = (EXACT ? NJ : NJ - 1) * NPASS;
    static_assert(!OPT1 || 2 * NSTMIN <= 63, "vmcnt immediate");
    // the x gelu' factor rows of ALL passes requested under the first K-step (up to 32 registers; the larger tiles keep the per-pass loads)
    constexpr bool SIDE_ALL = OPT1 && EPI == EPI_MUL_COLSUM && NPASS * NJ <= 8;
    bool epi_pending = false;                                   // (wave-uniform) the previous tile's stores may still be draining
    bool epi_two = false;
    if constexpr (TAB) {
        // gelu / gelu' of every bf16 h with 2^-12 <= |h| < 16: entry [sign | exponent - 115 | mantissa] = gelu'(h) << 16 | gelu(h)
        uint32_t* tab = reinterpret_cast<uint32_t*>(smem + NST * STAGE * 2);
        for (int i = tid; i < GELU_TAB_BYTES / 4; i += NT) {
            const uint32_t bits = ((uint32_t)(i >> 11) << 15) | (GELU_TAB_LO + (uint32_t)(i & 0x7FF));
            const float h = __uint_as_float(bits << 16);
            float c, e;
            phi_parts(h, c, e);
            const uint32_t g16 = f2bf_pair(h * c, 0.f) & 0xFFFFu, p16 = f2bf_pair(fmaf(h * 0.3989422804014327f, e, c), 0.f) & 0xFFFFu;
            tab[i] = (p16 << 16) | g16;
        }
        // (visible to everybody behind the first K-step's barrier)
    }
    if (p.stagger > 0 && (int)blockIdx.x >= (int)gridDim.x / 2) {
        // co-resident workgroups run equal tiles in the same phases (both in the loop, both in the epilogue): the second
        // half of the grid — the second workgroup of every CU under the dispatcher's round-robin — starts `stagger` x 64 clocks late
        for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(1);
    }

    int orig = blockIdx.x;
    const uint16_t* src[NPIECE];
    int t = xcd_remap(orig, ntiles);
    int m0 = (t / ntn) * BM, n0 = (t % ntn) * BN;
    sources(src, m0, n0);
    if (nk > 0) { if (0 < nfull) issue(src, 0, 0, No{}); else issue(src, 0, 0, Yes{}); }

    for (;;) {
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < TM; ++b) acc[a][b] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

        // ---- main loop: at the top of step s the loads of step s are in flight (the only group); wait for them,
        //      barrier (the other stage is then free for everyone — on the first step of a tile that is the
        //      previous tile's epilogue LDS), issue step s+1 into it, multiply step s.  Raw s_barrier:
        //      __syncthreads() would drain vmcnt.
        // side inputs of this tile's epilogue (OPT & 1): requested under the first K-step, in registers before the next
        // tile's DMA is issued
        u32x4v braw = u32x4v{0, 0, 0, 0};
        u32x4v auxall[SIDE_ALL ? NPASS * NJ : 1];
        auto side_inputs = [&]() {
            const int n_ = n0 + cc * 8;
            const bool ok_ = n_ < p.N && epi_thread;
            if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) {
                if (p.bias && ok_) braw = *reinterpret_cast<const u32x4v*>(p.bias + n_);
            }
            if constexpr (SIDE_ALL) {
#pragma unroll
                for (int h_ = 0; h_ < NPASS; ++h_)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int m = m0 + h_ * HM + r0 + j * RPP;
                        auxall[h_ * NJ + j] = (m < p.M && ok_ && (EXACT || r0 + j * RPP < HM)) ? __builtin_nontemporal_load(reinterpret_cast<const u32x4v*>(p.aux + (int64_t)m * p.ldaux + n_)) : u32x4v{0, 0, 0, 0};
                    }
            }
        };
        auto top_of_step = [&](int s) {
            if constexpr (OPT1) {
                // the first K-step behind an epilogue: only ITS loads (older than the epilogue's stores) are waited for
                if (s == 0 && epi_pending) { if (epi_two) wait_vmcnt<2 * NSTMIN>(); else wait_vmcnt<NSTMIN>(); }
                else wait_vmcnt<0>();
            } else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (s + 1 < nk) { if (s + 1 < nfull) issue(src, (s + 1) * BK, (s + 1) & 1, No{}); else issue(src, (s + 1) * BK, (s + 1) & 1, Yes{}); }
            if constexpr (OPT1) { if (s == 0) side_inputs(); }
        };
        for (int s = 0; s < nfull; ++s) {                       // (the tail step lives outside the loop: one
            top_of_step(s);                                     //  accumulator live range, no phi copies)
            if (s == 0) GPROF(1);
            step(s & 1, 8, No{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (rem) {
            top_of_step(nfull);
            step(nfull & 1, rem >> 3, Yes{});
        }
        GPROF(2);

        // ---- next tile: its first K-step goes to stage 0 now, behind a barrier (every wave is done with the stages)
        const int n = n0 + cc * 8;
        const bool ncol_ok = n < p.N && epi_thread;             // N % 8 == 0: a chunk is all in or all out
        const int cur_m0 = m0, cur_n0 = n0;
        const int next = orig + (int)gridDim.x;
        const bool has_next = next < ntiles;
        if constexpr (OPT1) {
            nt_lds_barrier();
            // the side inputs are used HERE (the compiler places its wait for them in front of the next tile's DMA, where
            // nothing else is outstanding; a wait behind the DMA — which it cannot see — would drain it)
#pragma unroll
            for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(braw[e]));
            if constexpr (SIDE_ALL) {
#pragma unroll
                for (int i = 0; i < NPASS * NJ; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(auxall[i][e]));
            }
        } else __syncthreads();
        if (has_next) {
            orig = next;
            t = xcd_remap(orig, ntiles);
            m0 = (t / ntn) * BM; n0 = (t % ntn) * BN;
            sources(src, m0, n0);
            kb = 0; kin = 0;
            if (0 < nfull) issue(src, 0, 0, No{}); else issue(src, 0, 0, Yes{});
        }

        // ---- epilogue: two passes of HM rows: accumulators -> LDS (fp32, swizzled [HM][BN]) -> (row, 8 columns) chunks
        auto epi_sync = [&]() { if constexpr (OPT1) nt_lds_barrier(); else __syncthreads(); };
        if constexpr (!OPT1 && (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU)) {
            if (p.bias && ncol_ok) braw = *reinterpret_cast<const u32x4v*>(p.bias + n);
        }
        float bv[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bv[2 * e] = __uint_as_float(braw[e] << 16);
            bv[2 * e + 1] = __uint_as_float(braw[e] & 0xFFFF0000u);
        }
        float cs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int half = 0; half < NPASS; ++half) {
            constexpr int dummy_ = 0; (void)dummy_;
            const int wm_of_pass = (half * HM) / WTM, tm0 = WR > 1 ? 0 : ((half * HM) % WTM) / 32;   // which wave row(s) / MFMA tiles hold these rows
            // side inputs of this thread's chunks are requested before the LDS round trip of the accumulators
            u32x4v auxv[EPI == EPI_MUL_COLSUM ? NJ : 1];
            if constexpr (SIDE_ALL) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) auxv[j] = auxall[half * NJ + j];
            } else if constexpr (EPI == EPI_MUL_COLSUM) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int m = cur_m0 + half * HM + r0 + j * RPP;
                    // (read once: non-temporal, like the outputs — profiles/r05_nt_out_stores.md)
                    auxv[j] = (m < p.M && ncol_ok && (EXACT || r0 + j * RPP < HM)) ? __builtin_nontemporal_load(reinterpret_cast<const u32x4v*>(p.aux + (int64_t)m * p.ldaux + n)) : u32x4v{0, 0, 0, 0};
                }
            }
            if (half) epi_sync();                               // the previous pass has been read
            if (wm >= wm_of_pass && wm < wm_of_pass + WR) {
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int tmp = 0; tmp < TMP; ++tmp) {
                        const int ml = (WR > 1 ? (wm - wm_of_pass) * WTM : 0) + tmp * 32 + c32;
                        const f32x16& av = acc[tn][tm0 + tmp];
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) {
                            const int ch = (wn * WTN + tn * 32 + 8 * r4 + 4 * g) >> 2;
                            *reinterpret_cast<f32x4v*>(ctile + ml * BN + (cpos(ch, ml) << 2)) =
                                f32x4v{av[4 * r4], av[4 * r4 + 1], av[4 * r4 + 2], av[4 * r4 + 3]};
                        }
                    }
            }
            epi_sync();
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int row = r0 + j * RPP, m = cur_m0 + half * HM + row;
                if (m >= p.M || !ncol_ok) continue;
                if constexpr (!EXACT) { if (row >= HM) continue; }
                const f32x4v lo = *reinterpret_cast<const f32x4v*>(ctile + row * BN + (cpos(2 * cc, row) << 2));
                const f32x4v hi = *reinterpret_cast<const f32x4v*>(ctile + row * BN + (cpos(2 * cc + 1, row) << 2));
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                uint16_t* o = p.out + (int64_t)m * p.ldo + n;
                if constexpr (EPI == EPI_STORE || EPI == EPI_BIAS) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += bv[e];
                    st_out16(o, u32x4v{f2bf_pair(v[0], v[1]), f2bf_pair(v[2], v[3]), f2bf_pair(v[4], v[5]), f2bf_pair(v[6], v[7])});
                } else if constexpr (EPI == EPI_BIAS_GELU) {
                    // fc1 under autocast yields bf16 h; gelu runs in fp32 ON that bf16 value and casts back
                    // (supernet_transformer.py:14-16, :276-277).  gelu'(h) = Phi(h) + h phi(h) reuses Phi and the
                    // exponential: it is written INSTEAD of h (the backward needs nothing else of h).
                    u32x4v pb, gb;
                    uint32_t hbv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) hbv[e] = f2bf_pair(v[2 * e] + bv[2 * e], v[2 * e + 1] + bv[2 * e + 1]);
                    bool direct = true;
                    if constexpr (TAB) {
                        // packed 16-bit index arithmetic on the two halves of a bf16 pair; table entry = gelu' << 16 | gelu
                        const unsigned char* tabb = reinterpret_cast<const unsigned char*>(smem + NST * STAGE * 2);
                        uint32_t big = 0;
                        uint32_t tl[4], th[4], d2v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const uint32_t abs2 = hbv[e] & 0x7FFF7FFFu;
                            const nt_s16x2 d2 = __builtin_bit_cast(nt_s16x2, abs2) - nt_s16x2{(short)GELU_TAB_LO, (short)GELU_TAB_LO};
                            const nt_s16x2 e2 = __builtin_elementwise_max(d2, nt_s16x2{0, 0});     // |h| < 2^-12 -> the 2^-12 entry (gelu' = 1/2 there too)
                            const uint32_t e2u = __builtin_bit_cast(uint32_t, e2);
                            big |= e2u;
                            const uint32_t idx2 = ((hbv[e] >> 4) & 0x08000800u) | e2u;              // sign -> bit 11
                            const uint32_t a2 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(nt_u16x2, idx2) << nt_u16x2{2, 2});
                            tl[e] = *reinterpret_cast<const uint32_t*>(tabb + (a2 & 0xFFFFu));
                            th[e] = *reinterpret_cast<const uint32_t*>(tabb + (a2 >> 16));
                            d2v[e] = __builtin_bit_cast(uint32_t, d2);
                        }
                        // |h| >= 16, inf, nan anywhere in the wave's chunks: the direct evaluation below (wave-uniform branch)
                        direct = __builtin_amdgcn_ballot_w64((big & 0x78007800u) != 0) != 0;
                        if (!direct) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const uint32_t gt = __builtin_amdgcn_perm(th[e], tl[e], 0x05040100u);      // gelu:  low halves
                                pb[e] = __builtin_amdgcn_perm(th[e], tl[e], 0x07060302u);                  // gelu': high halves
                                // |h| < 2^-12: gelu(h) = h / 2 (exponent - 1; zero stays zero: saturating subtraction)
                                const uint32_t tiny = __builtin_bit_cast(uint32_t, __builtin_bit_cast(nt_s16x2, d2v[e]) >> nt_s16x2{15, 15});
                                const nt_u16x2 habs = __builtin_elementwise_sub_sat(__builtin_bit_cast(nt_u16x2, hbv[e] & 0x7FFF7FFFu), nt_u16x2{0x80, 0x80});
                                const uint32_t half2 = (hbv[e] & 0x80008000u) | __builtin_bit_cast(uint32_t, habs);
                                gb[e] = (half2 & tiny) | (gt & ~tiny);
                            }
                        }
                    }
                    if (direct) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uint32_t hb = hbv[e];
                        const float h0 = __uint_as_float(hb << 16), h1 = __uint_as_float(hb & 0xFFFF0000u);
                        float c0, e0, c1, e1;
                        phi_parts(h0, c0, e0);
                        phi_parts(h1, c1, e1);
                        gb[e] = f2bf_pair(h0 * c0, h1 * c1);
                        pb[e] = f2bf_pair(fmaf(h0 * 0.3989422804014327f, e0, c0), fmaf(h1 * 0.3989422804014327f, e1, c1));
                    }
                    }
                    if (n + 8 > p.nvalid) {                     // padded columns: exact zeros (their gradients vanish)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const uint32_t keep = (n + 2 * e < p.nvalid ? 0x0000FFFFu : 0u) | (n + 2 * e + 1 < p.nvalid ? 0xFFFF0000u : 0u);
                            pb[e] &= keep;
                            gb[e] &= keep;
                        }
                    }
                    if (p.out) st_out16(o, pb);                             // (no gelu' without a backward: inference, frozen teacher)
                    st_out16(p.out2 + (int64_t)m * p.ldo + n, gb);
                } else {   // EPI_MUL_COLSUM
                    const u32x4v fb = auxv[j];
                    u32x4v db;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        db[e] = f2bf_pair(v[2 * e] * __uint_as_float(fb[e] << 16), v[2 * e + 1] * __uint_as_float(fb[e] & 0xFFFF0000u));
                        cs[2 * e] += __uint_as_float(db[e] << 16);              // sums of the ROUNDED values written
                        cs[2 * e + 1] += __uint_as_float(db[e] & 0xFFFF0000u);
                    }
                    st_out16(o, db);
                }
            }
            if constexpr (EPI == EPI_MUL_COLSUM) {
                if (((half + 1) * HM) % SLAB == 0) {            // a 128-row slab is complete: its column sums leave
                    epi_sync();                                 // the fp32 pass has been consumed
                    float* red = ctile;                         // [RPP][BN]
                    if (epi_thread) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) { red[r0 * BN + cc * 8 + e] = cs[e]; cs[e] = 0.f; }
                    }
                    epi_sync();
                    const int slab = (cur_m0 + (half + 1) * HM) / SLAB - 1;
                    if (slab * SLAB < p.M) {
                        for (int c = tid; c < BN; c += NT) {
                            if (cur_n0 + c < p.N) {
                                float sum = 0.f;
#pragma unroll
                                for (int r = 0; r < RPP; ++r) sum += red[r * BN + c];     // fixed order
                                p.colsum[(int64_t)slab * p.N + cur_n0 + c] = sum;
                            }
                        }
                    }
                }
            }
        }
        GPROF(3);
        if (!has_next) break;
        if constexpr (OPT1) {
            // every wave issued all its row stores iff the tile was full (no row past M, no column chunk past N)
            epi_pending = cur_m0 + BM <= p.M && cur_n0 + BN <= p.N;
            epi_two = EPI == EPI_BIAS_GELU && p.out != nullptr;
        }
        // (the first top_of_step of the next tile starts with a barrier: nobody loads into stage 1 — this
        //  epilogue's LDS — before every thread is past its reads)
    }
}


// =================================================================================================
// "TN" kernel — weight gradients:  dW(N x K) = dY(M x N)^T . X(M x K), contraction over the M tokens,
// split S ways over workgroups (fp32 partial tiles; cream_grad_finalize adds them in fixed order —
// no atomics).  Both operands are stored with the CONTRACTION index as the row index, so the MFMA
// fragments (8 consecutive m per lane) are columns of the staged tiles: they are read with
// ds_read_b64_tr_b16 (a 16-lane group fetches a 4(m) x 16(col) block, lane c receives column c — lane
// mapping pinned on the MI355X by tools/probes/gemm_nt_probe.hip), two reads per fragment.
//   * tile 128(n) x 128(k) of dW, 4 waves (2 x 2, wave tile 64 x 64), 64 tokens per step;
//   * operands go global -> LDS directly as full 256-byte rows ([m][128 cols] images); the 16-byte
//     chunk index is XOR-swizzled with (m & 3) << 2 on the source side and on the read, which spreads
//     the four rows of a transpose-read block over distinct bank ranges;
//   * two LDS stages, one barrier per step; rows beyond M (only possible in the last step of the
//     last split) are zeroed in LDS before use;
//   * bias gradient for free: workgroups of the first k-tile also multiply their dY fragments with a
//     ones fragment — the column sums of dY (= F.linear's bias gradient) leave with the partials.
struct TnParams {
    const uint16_t* dY;     // (M x N) bf16, row stride ldy
    const uint16_t* X;      // (M x K) bf16, row stride ldx
    int64_t ldy, ldx;
    int M, N, K, S;         // S splits of the token dimension (in 64-token steps, as even as possible)
    float* parts;           // [S][N][K] fp32
    float* bias_parts;      // [S][N] fp32 or nullptr
    uint16_t* parts16;      // non-null: the partial tiles leave as bf16 [S][N][K] instead (half the partial traffic; `parts` unused)
};

__device__ __forceinline__ bf16x4 tr16(const uint16_t* p) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        reinterpret_cast<__attribute__((address_space(3))) s16x4*>(reinterpret_cast<uintptr_t>(p)));
}

// One 128 x 128 tile of dW over the token steps [s_lo, s_hi): the loop of both TN kernels.
// BM tokens per step, NST LDS stages (NST - 1 steps of loads in flight across the per-step barrier, counted vmcnt).
struct TnTile {
    const uint16_t* dY; const uint16_t* X;
    int64_t ldy, ldx;
    int M, N, K, n0, k0;
};

template <int BM = 64, int NST = 2, bool CAN_BIAS = true>
__device__ __forceinline__ void tn_accumulate(const TnTile& t, int s_lo, int s_hi, bool want_bias, f32x16 (&acc)[2][2],
                                              f32x16 (&bacc)[2], uint16_t* lds, int tid)
{
    constexpr int BT = 128;                                     // output tile edge
    constexpr int STAGE = 2 * BM * BT;                          // bf16 elements per stage ([dY | X] tiles)
    constexpr int NPIECE = 2 * BM / 4 / 4;                      // 1-KB pieces (4 rows of 256 B) per wave and stage
    constexpr int PPO = BM / 4;                                 // pieces per operand tile
    static_assert(BM % 16 == 0 && NPIECE >= 2 && NST >= 2 && NST <= 4, "steps");
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wk = wave & 1;

    // sources: piece j of a stage = rows 4j..4j+3 of [dY tile (pieces 0..15) | X tile (16..31)]
    const uint16_t* src[NPIECE];
    int rowin[NPIECE];
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
        const int piece = wave + 4 * i, r = (piece % PPO) * 4 + (lane >> 4);    // token row inside the step
        const int c = (lane & 15) ^ ((r & 3) << 2);                             // source chunk of this LDS position
        rowin[i] = r;
        if (piece < PPO) src[i] = t.dY + min(t.n0 + c * 8, t.N - 8);            // N, K % 8 == 0: whole chunks
        else src[i] = t.X + min(t.k0 + c * 8, t.K - 8);
    }
    auto issue = [&](int step, int buf) {
        const int m0 = step * BM;
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) {
            const bool isY = i < NPIECE / 2;                                      // pieces wave + 4 i < PPO
            const int m = min(m0 + rowin[i], t.M - 1);
            const uint16_t* s = src[i] + (int64_t)m * (isY ? t.ldy : t.ldx);
            __builtin_amdgcn_global_load_lds(
                s, reinterpret_cast<__attribute__((address_space(3))) void*>(
                       reinterpret_cast<uintptr_t>(lds + buf * STAGE + (wave + 4 * i) * 4 * BT)), 16, 0, 0);
        }
    };
    // fragment of 32 columns starting at `col` for contraction rows mb + 8 g .. + 7 of a [64][128] tile:
    // lane group q = lane >> 4: column block 16 (q & 1), contraction half g = q >> 1
    const int gi = lane & 15, q = lane >> 4;
    auto frag = [&](const uint16_t* tile_, int col, int mb) -> bf16x8 {
        const int cc = col + 16 * (q & 1) + (gi & 3) * 4;                       // first of this lane's 4 columns
        const int m = mb + 8 * (q >> 1) + (gi >> 2);                            // row supplied by this lane (first read)
        // (m & 3) is the same for m and m + 4: one swizzle term for both reads
        const int off = m * BT + ((((cc >> 3) ^ ((m & 3) << 2)) << 3) | (cc & 7));
        const bf16x4 lo = tr16(tile_ + off), hi = tr16(tile_ + off + 4 * BT);
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    const bf16x8 ones = bf16x8{0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};

    // one 64-token step from stage `buf`: fragments double buffered in registers (the transpose reads of
    // sub-step ms+1 are issued before the MFMAs of sub-step ms)
    auto step = [&](int buf, auto with_bias) {
        constexpr bool BIAS = decltype(with_bias)::value;
        const uint16_t* Yt = lds + buf * STAGE;
        const uint16_t* Xt = Yt + BM * BT;
        bf16x8 fy[2][2], fx[2][2];
        auto load = [&](int ms, int slot) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fy[slot][i] = frag(Yt, wn * 64 + i * 32, ms * 16);
                fx[slot][i] = frag(Xt, wk * 64 + i * 32, ms * 16);
            }
        };
        load(0, 0);
#pragma unroll
        for (int ms = 0; ms < BM / 16; ++ms) {
            if (ms + 1 < BM / 16) load(ms + 1, (ms + 1) & 1);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fy[ms & 1][a], fx[ms & 1][b], acc[a][b], 0, 0, 0);
            if constexpr (BIAS) {
                bacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fy[ms & 1][0], ones, bacc[0], 0, 0, 0);
                bacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fy[ms & 1][1], ones, bacc[1], 0, 0, 0);
            }
        }
        constexpr int NM = BIAS ? 6 : 4;
        __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);                     // transpose reads of sub-steps 0 and 1
#pragma unroll
        for (int ms = 0; ms < BM / 16; ++ms) {
            __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
            if (ms + 2 < BM / 16) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        }
    };
    using No = std::integral_constant<bool, false>;
    using Yes = std::integral_constant<bool, true>;
    // the loop exists twice (with / without the bias MFMAs) so that no branch sits between the MFMAs;
    // the token tail (rows beyond M: only the very last step of the last split) is handled after it
    const bool tail = s_hi > s_lo && s_hi * BM > t.M;
    const int s_full = tail ? s_hi - 1 : s_hi;
    constexpr int D = NST - 1;
    auto run = [&](auto with_bias) {
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (s_lo + d < s_hi) issue(s_lo + d, d);
        for (int st = s_lo; st < s_full; ++st) {
            const int buf = (st - s_lo) % NST;
            const int later = min(D - 1, s_hi - 1 - st);            // load groups younger than this step's
            if (later >= 2) wait_vmcnt<2 * NPIECE>();
            else if (later == 1) wait_vmcnt<NPIECE>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();                           // (raw: __syncthreads() would drain vmcnt)
            asm volatile("" ::: "memory");
            if (st + D < s_hi) issue(st + D, (st + D - s_lo) % NST);
            step(buf, with_bias);
        }
        if (tail) {
            const int buf = (s_full - s_lo) % NST;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            uint16_t* Yt = lds + buf * STAGE;                   // zero the token rows beyond M in both tiles
            const int valid = t.M - s_full * BM;
            for (int i = tid; i < 2 * BM * BT / 8; i += 256) {
                const int r = (i / 16) & (BM - 1);
                if (r >= valid) *reinterpret_cast<u32x4v*>(Yt + i * 8) = u32x4v{0, 0, 0, 0};
            }
            __syncthreads();
            step(buf, with_bias);
        }
    };
    if constexpr (CAN_BIAS) {
        if (want_bias && wk == 0) run(Yes{}); else run(No{});
    } else {
        run(No{});
    }
}

__device__ __forceinline__ void tn_zero(f32x16 (&acc)[2][2], f32x16 (&bacc)[2]) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        bacc[a] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    }
}

template <int OCC, int BM = 64, int NST = 2, bool CAN_BIAS = true>
__global__ __launch_bounds__(256, OCC) void gemm_tn_kernel(const TnParams p)
{
    constexpr int BT = 128;
    __shared__ __attribute__((aligned(1024))) uint16_t lds[NST * 2 * BM * BT];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wn = wave >> 1, wk = wave & 1;
    const int ntk = (p.K + BT - 1) / BT, ntn = (p.N + BT - 1) / BT;
    // XCD-aware order: all tiles of one split run on ONE XCD (they re-read the same 64-token rows of dY
    // and X every step: one HBM fetch, the rest from that XCD's L2 — without the remap neighbouring
    // tiles sit on different XCDs and every operand row is fetched ntn / ntk times)
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = bid % (ntk * ntn), split = bid / (ntk * ntn);
    const int n0 = (tile / ntk) * BT, k0 = (tile % ntk) * BT;
    const int tsteps = (p.M + BM - 1) / BM;
    const int s_lo = (int)((int64_t)tsteps * split / p.S), s_hi = (int)((int64_t)tsteps * (split + 1) / p.S);
    const bool want_bias = CAN_BIAS && p.bias_parts && k0 == 0;
    f32x16 acc[2][2], bacc[2];
    tn_zero(acc, bacc);
    const TnTile t{p.dY, p.X, p.ldy, p.ldx, p.M, p.N, p.K, n0, k0};
    tn_accumulate<BM, NST, CAN_BIAS>(t, s_lo, s_hi, want_bias, acc, bacc, lds, threadIdx.x);
    // ---- partial tile: the accumulators (lane = column k, registers = rows n) leave through LDS — the stages are idle —
    //      as [128 n][128 k] fp32, so that the partial is written in 16-byte row chunks (16 stores per thread instead of 64
    //      4-byte ones: the kernel's epilogue was as long as its 25-36 step main loop, profiles/r03_wgrad_group.md)
    const int g = lane >> 5, c32 = lane & 31, tid = threadIdx.x;
    float* ct = reinterpret_cast<float*>(lds);
    __syncthreads();                                                       // every wave is done reading the stages
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) ct[(wn * 64 + a * 32 + acc_row(r, g)) * BT + wk * 64 + b * 32 + c32] = acc[a][b][r];
    if (want_bias && wk == 0 && c32 == 0) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * 64 + a * 32 + acc_row(r, g);
                if (n < p.N) p.bias_parts[(int64_t)split * p.N + n] = bacc[a][r];
            }
    }
    __syncthreads();
    if (p.parts16) {
        // bf16 partials: every split's sum over its ~M / S tokens is rounded once (the reference's autocast backward rounds the
        // whole weight gradient to bf16 once, Linear_super.py:71-81 under torch.autocast); cream_grad_finalize adds them in fp32
        uint16_t* out = p.parts16 + (int64_t)split * p.N * p.K;
        const int c8 = (tid & 15) * 8, r0 = tid >> 4;
        if (k0 + c8 < p.K) {                                               // K % 8 == 0: an 8-chunk is all in or all out
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int n = n0 + r0 + 16 * i;
                if (n < p.N) {
                    const f32x4v lo = *reinterpret_cast<const f32x4v*>(ct + (r0 + 16 * i) * BT + c8);
                    const f32x4v hi = *reinterpret_cast<const f32x4v*>(ct + (r0 + 16 * i) * BT + c8 + 4);
                    *reinterpret_cast<u32x4v*>(out + (int64_t)n * p.K + k0 + c8) =
                        u32x4v{f2bf_pair(lo[0], lo[1]), f2bf_pair(lo[2], lo[3]), f2bf_pair(hi[0], hi[1]), f2bf_pair(hi[2], hi[3])};
                }
            }
        }
        return;
    }
    float* out = p.parts + (int64_t)split * p.N * p.K;
    const int c4 = (tid & 31) * 4, r0 = tid >> 5;
    if (k0 + c4 < p.K) {                                                   // K % 8 == 0: a 4-chunk is all in or all out
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int n = n0 + r0 + 8 * i;
            if (n < p.N)
                *reinterpret_cast<f32x4v*>(out + (int64_t)n * p.K + k0 + c4) = *reinterpret_cast<const f32x4v*>(ct + (r0 + 8 * i) * BT + c4);
        }
    }
}

}  // namespace gemm
}  // namespace cream
