// gemm_probe.hip — development probe for an own MFMA GEMM of the weight-entangled projections
// (DESIGN.md §9 item 1).  NOT part of the library and NOT yet run on hardware: written at the end of
// round 1 as the starting point of round 2 (the library GEMMs are 7.5 of 12.5 ms per step at ~520
// TFLOP/s; an own kernel is the only way to fuse the LayerNorm / GELU / residual passes into them).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Icream_amd/csrc tools/probes/gemm_probe.hip \
//         -L/opt/rocm/lib -lhipblaslt -o tools/probes/gemm_probe && tools/probes/gemm_probe
//
// Problem ("TN", the forward of LinearSuper): out(M x N) = x(M x K) . W(N x K, ldw)^T + bias(N), bf16
// operands, fp32 accumulation, bf16 output; both operands are K-contiguous, so MFMA fragments are
// 16-byte reads of rows.  Structure (the "step-2/3" rung of cdna_hip_programming.md §5, chosen for
// K = 320..1792, i.e. 5..28 K-steps only — prologue and epilogue matter as much as the main loop):
//   * workgroup = 256 threads = 4 waves (2 x 2), tile BM x BN = 128 x 128, BK = 64;
//   * wave = 64 x 64 of the tile = 2 x 2 MFMA 32x32x16 accumulators (64 VGPRs);
//   * register-staged double buffer: global loads of K-step s+1 are in flight while step s is
//     consumed from LDS; one barrier per K-step;
//   * swapped product D^T = W_tile . x_tile^T: a lane owns ONE output row m and 16 columns n
//     (4 runs of 4 consecutive n), so the epilogue (bias, and later GELU / residual / drop-path scale,
//     which are per-row or per-column) needs no cross-lane traffic; the tile is transposed through LDS
//     for 16-byte row-contiguous stores.
// The probe checks the result against a plain reference kernel and prints TFLOP/s next to the
// library's number for the same problem (hipblasLtMatmul with the heuristic's first algorithm).
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <stdint.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 hwbf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ uint32_t f2bf_pair(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2{lo, hi}), hwbf16x2));
}
__host__ __device__ inline int acc_row(int r, int g) { return (r & 3) + 8 * (r >> 2) + 4 * g; }

constexpr int BM = 128, BN = 128, BK = 64, PITCH = BK + 8;      // LDS rows of 72 bf16 (144 B)
constexpr int CP = BN + 8;                                      // pitch of the transposed output tile

// ---------------------------------------------------------------------------------------------
// out(M x N) = x(M x K) . W(N x K, ldw)^T + bias(N)
// grid = (ceil(N / BN), ceil(M / BM)); M, N arbitrary (rows / columns beyond are masked), K % 64 == 0
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linear_tn_kernel(uint16_t* __restrict__ out, const uint16_t* __restrict__ x,
                                                        const uint16_t* __restrict__ w, const uint16_t* __restrict__ bias,
                                                        int M, int N, int K, int64_t ldw)
{
    __shared__ __attribute__((aligned(16))) uint16_t lds[2][(BM + BN) * PITCH];     // 2 x 36 KB
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 5, c32 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;                      // wave -> (64-row block, 64-column block) of the tile
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    // staging: 128 rows x 64 k = 1024 chunks of 16 B per operand -> 4 per thread and operand
    u32x4 ra[4], rb[4];
    auto issue = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + i * 256, row = c >> 3, cc = c & 7;
            const int m = min(m0 + row, M - 1), n = min(n0 + row, N - 1);          // clamped: masked at the store
            ra[i] = *reinterpret_cast<const u32x4*>(x + (int64_t)m * K + k0 + cc * 8);
            rb[i] = *reinterpret_cast<const u32x4*>(w + (int64_t)n * ldw + k0 + cc * 8);
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + i * 256, row = c >> 3, cc = c & 7;
            *reinterpret_cast<u32x4*>(&lds[buf][row * PITCH + cc * 8]) = ra[i];
            *reinterpret_cast<u32x4*>(&lds[buf][(BM + row) * PITCH + cc * 8]) = rb[i];
        }
    };

    f32x16 acc[2][2] = {};                                        // [n tile][m tile] of the SWAPPED product
    issue(0);
    commit(0);
    __syncthreads();
    const int nk = K / BK;
    for (int s = 0; s < nk; ++s) {
        const int buf = s & 1;
        if (s + 1 < nk) issue((s + 1) * BK);
        const uint16_t* A = &lds[buf][0];                         // x tile rows (m)
        const uint16_t* B = &lds[buf][BM * PITCH];                // W tile rows (n)
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 fx[2], fw[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fx[t] = *reinterpret_cast<const bf16x8*>(A + (wm * 64 + t * 32 + c32) * PITCH + ks * 16 + g * 8);
                fw[t] = *reinterpret_cast<const bf16x8*>(B + (wn * 64 + t * 32 + c32) * PITCH + ks * 16 + g * 8);
            }
            // D^T(n x m) = W_tile(n x k) . x_tile^T: A operand = W rows, B operand = x rows
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[tn], fx[tm], acc[tn][tm], 0, 0, 0);
        }
        if (s + 1 < nk) {
            commit(buf ^ 1);                                       // the other buffer was last read in step s-1
            __syncthreads();
        }
    }

    // ---- epilogue: lane = output row m (column of D^T), registers = 16 columns n; bias; transpose through
    //      LDS (the staging buffers are free after a barrier) for row-contiguous 16-byte stores ---------------
    __syncthreads();
    uint16_t* ct = &lds[0][0];                                    // [BM][CP] bf16 = 34 KB <= 36 KB
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            const int ml = wm * 64 + tm * 32 + c32;                // row of the tile owned by this lane
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int nl = wn * 64 + tn * 32 + 8 * r4 + 4 * g;  // 4 consecutive columns
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = min(n0 + nl + e, N - 1);
                    v[e] = acc[tn][tm][4 * r4 + e] + (bias ? bf2f(bias[n]) : 0.f);
                }
                *reinterpret_cast<u32x2*>(ct + ml * CP + nl) = u32x2{f2bf_pair(v[0], v[1]), f2bf_pair(v[2], v[3])};
            }
        }
    __syncthreads();
    // 128 rows x 128 columns = 2048 chunks of 8 bf16 -> 8 per thread; a row's 16 chunks go to 16 consecutive lanes
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = tid + i * 256, row = c >> 4, cc = c & 15;
        const int m = m0 + row, n = n0 + cc * 8;
        if (m < M && n + 8 <= N)
            *reinterpret_cast<u32x4*>(out + (int64_t)m * N + n) = *reinterpret_cast<const u32x4*>(ct + row * CP + cc * 8);
        else if (m < M)
            for (int e = 0; e < 8 && n + e < N; ++e) out[(int64_t)m * N + n + e] = ct[row * CP + cc * 8 + e];
    }
}

// plain reference: one thread per output element, fp32 accumulation in k order
__global__ void linear_tn_ref(float* out, const uint16_t* x, const uint16_t* w, const uint16_t* bias, int M, int N, int K,
                              int64_t ldw)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= (int64_t)M * N) return;
    const int m = (int)(i / N), n = (int)(i % N);
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += bf2f(x[(int64_t)m * K + k]) * bf2f(w[(int64_t)n * ldw + k]);
    out[i] = s + (bias ? bf2f(bias[n]) : 0.f);
}

static uint16_t f2bf_host(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7FFF + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}

static float time_ms(hipEvent_t e0, hipEvent_t e1) { float t; hipEventElapsedTime(&t, e0, e1); return t; }

int main()
{
    struct Shape { int M, N, K, ldw; const char* what; };
    const Shape shapes[] = {
        {25216, 1152, 384, 384, "qkv fwd  (E=384, H=6)"},  {25216, 384, 384, 448, "proj fwd (E=384, Q=384)"},
        {25216, 1344, 384, 448, "fc1 fwd  (E=384, R=3.5)"}, {25216, 384, 1344, 1792, "fc2 fwd  (E=384, R=3.5)"},
        {25216, 1792, 448, 448, "fc1 fwd  (E=448, R=4)"},   {25216, 320, 320, 448, "proj fwd (E=320, Q=320)"},
        {1000, 200, 128, 136, "ragged edge check"},
    };
    hipblasLtHandle_t lt;
    hipblasLtCreate(&lt);
    void* ws;
    const size_t wsb = 128 << 20;
    hipMalloc(&ws, wsb);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (const Shape& s : shapes) {
        const size_t nx = (size_t)s.M * s.K, nw = (size_t)s.N * s.ldw, no = (size_t)s.M * s.N;
        std::vector<uint16_t> hx(nx), hw(nw), hb(s.N);
        srand(1);
        for (auto& v : hx) v = f2bf_host((rand() / (float)RAND_MAX - 0.5f) * 2.f);
        for (auto& v : hw) v = f2bf_host((rand() / (float)RAND_MAX - 0.5f) * 0.2f);
        for (auto& v : hb) v = f2bf_host((rand() / (float)RAND_MAX - 0.5f));
        uint16_t *dx, *dw, *db, *dout, *dlib;
        float* dref;
        hipMalloc(&dx, nx * 2); hipMalloc(&dw, nw * 2); hipMalloc(&db, s.N * 2);
        hipMalloc(&dout, no * 2); hipMalloc(&dlib, no * 2); hipMalloc(&dref, no * 4);
        hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice);
        hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice);
        hipMemcpy(db, hb.data(), s.N * 2, hipMemcpyHostToDevice);
        const dim3 grid((s.N + BN - 1) / BN, (s.M + BM - 1) / BM);
        // correctness
        linear_tn_ref<<<(unsigned)((no + 255) / 256), 256>>>(dref, dx, dw, db, s.M, s.N, s.K, s.ldw);
        linear_tn_kernel<<<grid, 256>>>(dout, dx, dw, db, s.M, s.N, s.K, s.ldw);
        std::vector<uint16_t> ho(no);
        std::vector<float> hr(no);
        hipMemcpy(ho.data(), dout, no * 2, hipMemcpyDeviceToHost);
        hipMemcpy(hr.data(), dref, no * 4, hipMemcpyDeviceToHost);
        double worst = 0, scale = 0;
        for (size_t i = 0; i < no; ++i) {
            uint32_t u = ((uint32_t)ho[i]) << 16;
            float f;
            memcpy(&f, &u, 4);
            worst = fmax(worst, fabs((double)f - hr[i]));
            scale = fmax(scale, fabs((double)hr[i]));
        }
        // own kernel timing
        for (int i = 0; i < 3; ++i) linear_tn_kernel<<<grid, 256>>>(dout, dx, dw, db, s.M, s.N, s.K, s.ldw);
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) linear_tn_kernel<<<grid, 256>>>(dout, dx, dw, db, s.M, s.N, s.K, s.ldw);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        const double own_us = time_ms(e0, e1) / 20 * 1e3;
        // the library on the same problem: col-major C(N x M) = W('t', lda = ldw) . x('n', ldb = K) + bias
        double lib_us = -1;
        {
            hipblasLtMatmulDesc_t d;
            hipblasLtMatrixLayout_t la, lb, lc;
            hipblasLtMatmulDescCreate(&d, HIPBLAS_COMPUTE_32F, HIP_R_32F);
            const hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
            hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(int32_t));
            hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(int32_t));
            const hipblasLtEpilogue_t ep = HIPBLASLT_EPILOGUE_BIAS;
            const hipDataType bt = HIP_R_16BF;
            const void* bp = db;
            hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_EPILOGUE, &ep, sizeof ep);
            hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof bt);
            hipblasLtMatmulDescSetAttribute(d, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bp, sizeof bp);
            hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, s.K, s.N, s.ldw);
            hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, s.K, s.M, s.K);
            hipblasLtMatrixLayoutCreate(&lc, HIP_R_16BF, s.N, s.M, s.N);
            hipblasLtMatmulPreference_t pref;
            hipblasLtMatmulPreferenceCreate(&pref);
            hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsb, sizeof wsb);
            hipblasLtMatmulHeuristicResult_t hr1[1];
            int n = 0;
            if (hipblasLtMatmulAlgoGetHeuristic(lt, d, la, lb, lc, lc, pref, 1, hr1, &n) == HIPBLAS_STATUS_SUCCESS && n > 0) {
                const float one = 1.f, zero = 0.f;
                for (int i = 0; i < 3; ++i)
                    hipblasLtMatmul(lt, d, &one, dw, la, dx, lb, &zero, dlib, lc, dlib, lc, &hr1[0].algo, ws, wsb, 0);
                hipEventRecord(e0);
                for (int i = 0; i < 20; ++i)
                    hipblasLtMatmul(lt, d, &one, dw, la, dx, lb, &zero, dlib, lc, dlib, lc, &hr1[0].algo, ws, wsb, 0);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                lib_us = time_ms(e0, e1) / 20 * 1e3;
            }
        }
        const double fl = 2.0 * s.M * s.N * s.K;
        printf("%-26s M=%5d N=%4d K=%4d  max|err| %.3g (scale %.3g)  own %7.1f us %6.0f TF/s   library(heuristic) %7.1f us %6.0f TF/s\n",
               s.what, s.M, s.N, s.K, worst, scale, own_us, fl / own_us / 1e6, lib_us, lib_us > 0 ? fl / lib_us / 1e6 : 0.0);
        hipFree(dx); hipFree(dw); hipFree(db); hipFree(dout); hipFree(dlib); hipFree(dref);
    }
    return 0;
}
