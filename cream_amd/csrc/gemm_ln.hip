// gemm_ln.hip — projections whose output width is the embedding dimension, with the LayerNorm that
// follows them in the SAME kernel (gfx950).
//
// Reference semantics (AutoFormer/model/supernet_transformer.py:262-287, pre-norm block):
//     x1 = x + drop_path(proj(attn))  ;  c = ffn_layer_norm(x1)         (LayerNormSuper: layernorm_super.py:26-37)
// i.e. LinearSuper (Linear_super.py:38-54) -> per-sample scale -> residual add -> LayerNorm.  The
// unfused path runs them as gemm_nt_kernel<EPI_BIAS> + ln_fwd_kernel (csrc/block_ops.hip): the branch
// output crosses HBM twice as bf16 and the LayerNorm is one more launch on the step's critical chain.
//
// Here a workgroup owns 64 COMPLETE output rows (tile 64 x E, E = 64 * TNW <= 512), so the row
// statistics are available in the epilogue:
//   * 4 waves as 2 (rows) x 2 (columns); wave tile 32 x 32*TNW of v_mfma_f32_32x32x16_bf16, swapped
//     product (a lane owns one output row, csrc/gemm_mfma.hpp);
//   * K-steps of 32 through two LDS stages filled by global_load_lds ([row][4 chunks of 16 B], chunk
//     XOR (row >> 2) & 3 on the source side and on the ds_read_b128 fragment reads: the 16 lanes of a
//     read group hit 16 distinct 16-byte slots);  (64 + E) * 64 B per stage: 2 workgroups per CU;
//   * epilogue in two passes of 32 rows: accumulators -> LDS as fp32 (row pitch E + 4 floats: E / 4 + 1
//     is odd mod 16, conflict-free 16-byte writes), then one WAVE per row exactly like ln_fwd_kernel:
//     p = bf16(acc + bias) (the value F.linear yields under autocast), x1 = x + s * p, mean / rstd of
//     x1, c = LN(x1) as bf16 — the same arithmetic in the same order as the two-kernel path, so both
//     give identical bits (tests/test_block_gpu.py).
// HBM traffic per call: A (M x K bf16) + x (M x E f32) in, x1 (f32) + c (bf16) out; the M x E bf16
// branch output never exists.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "attn_common.hpp"
#include "cream_amd.h"

namespace cream {
namespace gemm_ln {          // (named: kernel templates instantiated from a function template need external stubs)

struct RowLnParams {
    const uint16_t* A;      // (M x K) bf16, row stride lda
    const uint16_t* B;      // (E x K) bf16 rows, row stride ldb (the active block of the super weight, read in place)
    int64_t lda, ldb;
    const uint16_t* bias;   // (E) bf16 or nullptr
    int M, K;
    const float* x;         // (M x E) residual stream
    float* xsum;            // (M x E) x + s * p
    uint16_t* y;            // (M x E) LayerNorm(xsum) as bf16
    float* mean; float* rstd;
    const float* gamma; const float* beta;
    const float* sscale;    // per-sample scale of the branch or nullptr
    int rows_per_sample, nsamp;
    float eps;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

constexpr int rowln_lds_bytes(int E) {
    const int stages = 2 * (64 + E) * 32 * 2, ctile = 32 * (E + 4) * 4;
    return stages > ctile ? stages : ctile;
}

template <int TNW>
__global__ __launch_bounds__(256, 2) void gemm_nt_rowln_kernel(const RowLnParams p)
{
    constexpr int BM = 64, BK = 32, E = 64 * TNW, WTN = 32 * TNW;
    constexpr int STAGE = (BM + E) * BK;                        // bf16 elements per stage
    constexpr int NPIECE = (BM + E) / 16 / 4;                   // 1-KB pieces (16 rows of 64 B) per wave and stage
    constexpr int PE = E + 4;                                   // fp32 row pitch of the epilogue tile
    constexpr int NCH = E / 4;                                  // 4-float chunks per row
    static_assert(NPIECE == 1 + TNW && NCH <= 128, "tile");
    __shared__ __attribute__((aligned(1024))) char smem[rowln_lds_bytes(E)];
    uint16_t* const lds = reinterpret_cast<uint16_t*>(smem);
    float* const ctile = reinterpret_cast<float*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 5, c32 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM;

    // ---- sources of this wave's pieces: piece (wave + 4 i) = rows 16 (wave + 4 i) .. + 15 of [A tile (64 rows) | B (E rows)]
    const uint16_t* src[NPIECE];
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
        const int row = (wave + 4 * i) * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);        // (BM % 16 == 0: the same term for tile-local B rows)
        if (i == 0) src[i] = p.A + (int64_t)min(m0 + row, p.M - 1) * p.lda + chunk * 8;
        else src[i] = p.B + (int64_t)(row - BM) * p.ldb + chunk * 8;
    }
    auto issue = [&](int k0, int buf) {
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) {
            const uint16_t* s = src[i] + k0;                    // (a named pointer: the builtin's checks reject the dependent expression)
            __builtin_amdgcn_global_load_lds(
                s, reinterpret_cast<__attribute__((address_space(3))) void*>(
                       reinterpret_cast<uintptr_t>(lds + buf * STAGE + (wave + 4 * i) * 16 * BK)), 16, 0, 0);
        }
    };
    const int sw = (c32 >> 2) & 3;
    auto frag = [&](const uint16_t* tile, int row, int chunk) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(tile + row * BK + ((chunk ^ sw) << 3));
    };

    f32x16 acc[TNW];
#pragma unroll
    for (int i = 0; i < TNW; ++i) acc[i] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    auto step = [&](int buf) {
        const uint16_t* At = lds + buf * STAGE;
        const uint16_t* Bt = At + BM * BK;
        bf16x8 fa[2], fb[2][TNW];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            fa[ks] = frag(At, wm * 32 + c32, ks * 2 + g);
#pragma unroll
            for (int i = 0; i < TNW; ++i) fb[ks][i] = frag(Bt, wn * WTN + i * 32 + c32, ks * 2 + g);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < TNW; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks][i], fa[ks], acc[i], 0, 0, 0);
        // the reads of sub-step 0, then MFMAs interleaved with the reads of sub-step 1
        __builtin_amdgcn_sched_group_barrier(0x100, 1 + TNW, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int i = 1; i < TNW; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, TNW, 0);
    };

    const int nk = p.K / BK;
    issue(0, 0);
    for (int s = 0; s < nk; ++s) {
        wait_vm0();
        __builtin_amdgcn_s_barrier();                           // (raw: __syncthreads() would also drain lgkmcnt — harmless here, but keep the idiom)
        asm volatile("" ::: "memory");
        if (s + 1 < nk) issue((s + 1) * BK, (s + 1) & 1);
        step(s & 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }

    // ---- epilogue: per-lane LayerNorm constants (chunks lane, lane + 64 of a row)
    f32x4v gam[2], bet[2];
    float bia[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = min(lane + 64 * i, NCH - 1);              // unconditional loads (a branch would drain vmcnt): chunks
        gam[i] = *reinterpret_cast<const f32x4v*>(p.gamma + 4 * c);     // beyond the row re-read the last one and are never used
        bet[i] = *reinterpret_cast<const f32x4v*>(p.beta + 4 * c);
        const uint16_t* bsrc = p.bias ? p.bias : reinterpret_cast<const uint16_t*>(p.gamma);    // (any readable address)
        u32x2v braw = *reinterpret_cast<const u32x2v*>(bsrc + 4 * c);
        if (!p.bias) braw = u32x2v{0, 0};
        bia[i][0] = __uint_as_float(braw[0] << 16); bia[i][1] = __uint_as_float(braw[0] & 0xFFFF0000u);
        bia[i][2] = __uint_as_float(braw[1] << 16); bia[i][3] = __uint_as_float(braw[1] & 0xFFFF0000u);
    }
    constexpr int RG = 4;                                       // rows in flight together: inputs requested as a group, the
                                                                // wave reductions of a group interleaved (4 independent chains)
    struct RowIn { f32x4v x[RG][2]; float sc[RG]; };
    auto request = [&](RowIn& in, int rbase) {
        int smp = __builtin_amdgcn_readfirstlane(rbase / p.rows_per_sample);   // one division per group, then counted
        int rem = rbase - smp * p.rows_per_sample;
#pragma unroll
        for (int j = 0; j < RG; ++j) {
            const int m = __builtin_amdgcn_readfirstlane(min(rbase + j, p.M - 1));
            const float* xr = p.x + (int64_t)m * E;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = min(lane + 64 * i, NCH - 1);      // unconditional; masked where it is used
                in.x[j][i] = *reinterpret_cast<const f32x4v*>(xr + 4 * c);
            }
            while (rem >= p.rows_per_sample) { rem -= p.rows_per_sample; ++smp; }
            const float* sp = p.sscale ? p.sscale + min(smp, p.nsamp - 1) : p.gamma;   // (else: any readable address)
            in.sc[j] = *sp;
            if (!p.sscale) in.sc[j] = 1.f;
            ++rem;
        }
    };
    auto wave_sum_group = [&](float (&t)[RG]) {                 // wave_sum of csrc/block_ops.hip (same order), RG rows at once
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            float u[RG];
#pragma unroll
            for (int j = 0; j < RG; ++j) u[j] = __shfl_xor(t[j], o);
#pragma unroll
            for (int j = 0; j < RG; ++j) t[j] += u[j];
        }
    };
    auto rows = [&](RowIn& in, int rb) {                         // rb: first global row of the group (wave-uniform)
        const float* cr = ctile + ((rb - m0) & 31) * PE;
        float s[RG], q[RG];
#pragma unroll
        for (int j = 0; j < RG; ++j) {
            const bool live = rb + j < p.M;
            s[j] = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = lane + 64 * i;
                f32x4v v = c < NCH ? in.x[j][i] : f32x4v{0, 0, 0, 0};
                if (c < NCH) {
                    const f32x4v a = *reinterpret_cast<const f32x4v*>(cr + j * PE + 4 * c);
                    // the branch output as the two-kernel path stores it: bf16(acc + bias)
                    const uint32_t lo = f2bf_pair(a[0] + bia[i][0], a[1] + bia[i][1]);
                    const uint32_t hi = f2bf_pair(a[2] + bia[i][2], a[3] + bia[i][3]);
                    float r[4];
                    r[0] = __uint_as_float(lo << 16); r[1] = __uint_as_float(lo & 0xFFFF0000u);
                    r[2] = __uint_as_float(hi << 16); r[3] = __uint_as_float(hi & 0xFFFF0000u);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += in.sc[j] * r[e];
                    if (live) *reinterpret_cast<f32x4v*>(p.xsum + (int64_t)(rb + j) * E + 4 * c) = v;
                }
                in.x[j][i] = v;
                s[j] += (v[0] + v[1]) + (v[2] + v[3]);
            }
        }
        wave_sum_group(s);
#pragma unroll
        for (int j = 0; j < RG; ++j) {
            s[j] = s[j] / (float)E;                             // mean
            q[j] = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (lane + 64 * i < NCH) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = in.x[j][i][e] - s[j]; q[j] += d * d; }
                }
            }
        }
        wave_sum_group(q);
#pragma unroll
        for (int j = 0; j < RG; ++j) {
            const bool live = rb + j < p.M;
            const float mu = s[j], rs = rsqrtf(q[j] / (float)E + p.eps);
            if (lane == 0 && live) { p.mean[rb + j] = mu; p.rstd[rb + j] = rs; }
            uint16_t* yr = p.y + (int64_t)(rb + j) * E;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = lane + 64 * i;
                if (c < NCH && live) {
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (in.x[j][i][e] - mu) * rs * gam[i][e] + bet[i][e];
                    *reinterpret_cast<u32x2v*>(yr + 4 * c) = u32x2v{f2bf_pair(o[0], o[1]), f2bf_pair(o[2], o[3])};
                }
            }
        }
    };
    __syncthreads();                                            // every wave is done with the stages
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int rb = m0 + half * 32 + wave * 8;               // this wave's 8 rows of the pass
        RowIn ra, rbn;
        request(ra, rb);
        if (half) __syncthreads();                              // pass 0 has been read
        if (wm == half) {
#pragma unroll
            for (int i = 0; i < TNW; ++i)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
                    *reinterpret_cast<f32x4v*>(ctile + c32 * PE + wn * WTN + i * 32 + 8 * r4 + 4 * g) =
                        f32x4v{acc[i][4 * r4], acc[i][4 * r4 + 1], acc[i][4 * r4 + 2], acc[i][4 * r4 + 3]};
        }
        __syncthreads();
        request(rbn, rb + RG);
        rows(ra, rb);
        rows(rbn, rb + RG);
    }
}

}  // namespace gemm_ln
}  // namespace cream

using namespace cream::gemm_ln;

extern "C" {

int cream_linear_add_ln_supported(int E, int K)
{
    return E >= 192 && E <= 512 && E % 64 == 0 && K >= 32 && K % 32 == 0;
}

int cream_linear_add_ln_fwd(float* xsum, void* y, float* mean, float* rstd, const void* a, const void* w, const void* bias,
                            const float* x, const float* sample_scale, int rows_per_sample, const float* gamma,
                            const float* beta, int M, int E, int K, int64_t ldw, float eps, void* stream)
{
    if (M < 0 || rows_per_sample <= 0 || ldw < K || ldw % 8) return CREAM_ERR_BAD_ARG;
    if (!cream_linear_add_ln_supported(E, K)) return CREAM_ERR_TOO_LARGE;
    if (M == 0) return CREAM_OK;
    if (!xsum || !y || !mean || !rstd || !a || !w || !x || !gamma || !beta) return CREAM_ERR_BAD_ARG;
    if (((uintptr_t)xsum | (uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)a | (uintptr_t)w) % 16 ||
        ((uintptr_t)y | (uintptr_t)bias) % 8)
        return CREAM_ERR_BAD_ARG;
    RowLnParams p{};
    p.A = (const uint16_t*)a; p.lda = K;
    p.B = (const uint16_t*)w; p.ldb = ldw;
    p.bias = (const uint16_t*)bias;
    p.M = M; p.K = K;
    p.x = x; p.xsum = xsum; p.y = (uint16_t*)y; p.mean = mean; p.rstd = rstd;
    p.gamma = gamma; p.beta = beta; p.sscale = sample_scale; p.rows_per_sample = rows_per_sample; p.eps = eps;
    p.nsamp = (M + rows_per_sample - 1) / rows_per_sample;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((M + 63) / 64), block(256);
    switch (E / 64) {
        case 3: hipLaunchKernelGGL(gemm_nt_rowln_kernel<3>, grid, block, 0, st, p); break;
        case 4: hipLaunchKernelGGL(gemm_nt_rowln_kernel<4>, grid, block, 0, st, p); break;
        case 5: hipLaunchKernelGGL(gemm_nt_rowln_kernel<5>, grid, block, 0, st, p); break;
        case 6: hipLaunchKernelGGL(gemm_nt_rowln_kernel<6>, grid, block, 0, st, p); break;
        case 7: hipLaunchKernelGGL(gemm_nt_rowln_kernel<7>, grid, block, 0, st, p); break;
        case 8: hipLaunchKernelGGL(gemm_nt_rowln_kernel<8>, grid, block, 0, st, p); break;
        default: return CREAM_ERR_TOO_LARGE;
    }
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

}  // extern "C"
