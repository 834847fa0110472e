// gemm_probe.hip — development probe for an own MFMA GEMM of the weight-entangled projections
// (DESIGN.md §9 item 1).  NOT part of the library: the starting point of round 2 (the library GEMMs
// are 7.5 of 12.5 ms per step at ~520 TFLOP/s; an own kernel is the only way to fuse the LayerNorm /
// GELU / residual passes into them).  Status at the end of round 1: the first variant (128x128,
// register-staged) has run on the MI355X — correct on every shape, 207-463 TFLOP/s against the
// library's 319-830; the other variants (XCD-aware tile order, 256x128 / 128x256 tiles, direct-to-LDS
// loads with source-side XOR swizzle) compile but have NOT run yet — the probe checks each against
// the reference kernel and flags wrong results, so one run sorts them.
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

constexpr int BK = 64, PITCH = BK + 8;                          // LDS rows of 72 bf16 (144 B), register-staged variants

// XCD-aware tile order (cdna_hip_programming.md T1, bijective form): workgroup ids are dealt round-robin
// to the 8 XCDs; remapping makes each XCD own a CONTIGUOUS range of tiles, and tiles are numbered
// n-fastest, so the column tiles that re-read one x row block share an L2.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int q = nwg / 8, r = nwg % 8, xcd = orig % 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
}

// ---------------------------------------------------------------------------------------------
// out(M x N) = x(M x K) . W(N x K, ldw)^T + bias(N);  tile (WM*64) x (WN*64), WM*WN waves
//   GLDS : operands go global -> LDS directly (global_load_lds_dwordx4, 1 KB per wave instruction);
//          the LDS image is then lane-linear [row][8 chunks of 16 B] WITHOUT padding, and bank conflicts
//          are avoided by XOR-swizzling the SOURCE chunk with the row (chunk c of row r lives at
//          position c ^ (r & 7)); vmcnt(0) + barrier per K-step, two buffers
//   XCD  : 1-D grid with the XCD-aware tile order above
// grid: XCD ? (ntn * ntm) : (ntn, ntm);  M, N arbitrary, K % 64 == 0
// ---------------------------------------------------------------------------------------------
template <int WM, int WN, bool GLDS, bool XCD>
__global__ __launch_bounds__(WM * WN * 64) void linear_tn_kernel(uint16_t* __restrict__ out, const uint16_t* __restrict__ x,
                                                                 const uint16_t* __restrict__ w,
                                                                 const uint16_t* __restrict__ bias, int M, int N, int K,
                                                                 int64_t ldw)
{
    constexpr int BM = WM * 64, BN = WN * 64, NT = WM * WN * 64;
    constexpr int LP = GLDS ? BK : PITCH;                         // row pitch of the staged tiles (elements)
    constexpr int CP = BN + 8;                                    // pitch of the transposed output tile
    constexpr int STAGE = (BM + BN) * LP, OUTT = BM * CP;
    constexpr int LDS_ELEMS = 2 * STAGE > OUTT ? 2 * STAGE : OUTT;
    __shared__ __attribute__((aligned(1024))) uint16_t lds[LDS_ELEMS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 5, c32 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;
    int bx, by;
    if constexpr (XCD) {
        const int ntn = (N + BN - 1) / BN, nwg = gridDim.x;
        const int t = xcd_remap(blockIdx.x, nwg);
        bx = t % ntn;
        by = t / ntn;
    } else {
        bx = blockIdx.x;
        by = blockIdx.y;
    }
    const int m0 = by * BM, n0 = bx * BN;

    constexpr int CA = BM * 8 / NT, CB = BN * 8 / NT;            // 16-byte chunks per thread and K-step
    static_assert((BM * 8) % NT == 0 && (BN * 8) % NT == 0, "tile must split evenly");
    u32x4 ra[GLDS ? 1 : CA], rb[GLDS ? 1 : CB];
    auto issue = [&](int k0, int buf) {
        if constexpr (GLDS) {
            // one wave instruction = 64 lanes x 16 B = 8 rows of 128 B, LDS destination lane-linear
            uint16_t* base = &lds[buf * STAGE];
#pragma unroll
            for (int i = 0; i < (BM + BN) / 8 / (NT / 64); ++i) {
                const int piece = wave + i * (NT / 64);                          // 8-row piece of the [A | B] image
                const int row = piece * 8 + (lane >> 3), cc = lane & 7;
                const int src_c = cc ^ (row & 7);
                const uint16_t* src = row < BM ? x + (int64_t)min(m0 + row, M - 1) * K + k0 + src_c * 8
                                               : w + (int64_t)min(n0 + row - BM, N - 1) * ldw + k0 + src_c * 8;
                __builtin_amdgcn_global_load_lds(src, reinterpret_cast<__attribute__((address_space(3))) void*>(
                                                          reinterpret_cast<uintptr_t>(base + piece * 8 * BK)),
                                                 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < CA; ++i) {
                const int c = tid + i * NT, row = c >> 3, cc = c & 7;
                ra[i] = *reinterpret_cast<const u32x4*>(x + (int64_t)min(m0 + row, M - 1) * K + k0 + cc * 8);
            }
#pragma unroll
            for (int i = 0; i < CB; ++i) {
                const int c = tid + i * NT, row = c >> 3, cc = c & 7;
                rb[i] = *reinterpret_cast<const u32x4*>(w + (int64_t)min(n0 + row, N - 1) * ldw + k0 + cc * 8);
            }
        }
    };
    auto commit = [&](int buf) {
        if constexpr (!GLDS) {
#pragma unroll
            for (int i = 0; i < CA; ++i) {
                const int c = tid + i * NT, row = c >> 3, cc = c & 7;
                *reinterpret_cast<u32x4*>(&lds[buf * STAGE + row * LP + cc * 8]) = ra[i];
            }
#pragma unroll
            for (int i = 0; i < CB; ++i) {
                const int c = tid + i * NT, row = c >> 3, cc = c & 7;
                *reinterpret_cast<u32x4*>(&lds[buf * STAGE + (BM + row) * LP + cc * 8]) = rb[i];
            }
        }
    };
    // position of chunk `c` (8 k-values) of tile row `row`
    auto frag = [&](const uint16_t* tile, int row, int c) -> bf16x8 {
        const int pos = GLDS ? (c ^ (row & 7)) : c;
        return *reinterpret_cast<const bf16x8*>(tile + row * LP + pos * 8);
    };

    f32x16 acc[2][2] = {};                                        // [n tile][m tile] of the SWAPPED product
    issue(0, 0);
    if constexpr (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    commit(0);
    __syncthreads();
    const int nk = K / BK;
    for (int s = 0; s < nk; ++s) {
        const int buf = s & 1;
        if (s + 1 < nk) issue((s + 1) * BK, buf ^ 1);             // (GLDS: the other buffer was last read in step s-1)
        const uint16_t* A = &lds[buf * STAGE];                    // x tile rows (m)
        const uint16_t* B = A + BM * LP;                          // W tile rows (n)
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 fx[2], fw[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fx[t] = frag(A, wm * 64 + t * 32 + c32, ks * 2 + g);
                fw[t] = frag(B, wn * 64 + t * 32 + c32, ks * 2 + g);
            }
            // D^T(n x m) = W_tile(n x k) . x_tile^T: A operand = W rows, B operand = x rows
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[tn], fx[tm], acc[tn][tm], 0, 0, 0);
        }
        if (s + 1 < nk) {
            if constexpr (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            commit(buf ^ 1);
            __syncthreads();
        }
    }

    // ---- epilogue: lane = output row m (column of D^T), registers = 16 columns n; bias; transpose through
    //      LDS (the staging buffers are free after a barrier) for row-contiguous 16-byte stores ---------------
    __syncthreads();
    uint16_t* ct = &lds[0];                                       // [BM][CP] bf16
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
    constexpr int CPR = BN / 8;                                   // 16-byte chunks per output row
#pragma unroll
    for (int i = 0; i < BM * CPR / NT; ++i) {
        const int c = tid + i * NT, row = c / CPR, cc = c % CPR;
        const int m = m0 + row, n = n0 + cc * 8;
        if (m < M && n + 8 <= N)
            *reinterpret_cast<u32x4*>(out + (int64_t)m * N + n) = *reinterpret_cast<const u32x4*>(ct + row * CP + cc * 8);
        else if (m < M)
            for (int e = 0; e < 8 && n + e < N; ++e) out[(int64_t)m * N + n + e] = ct[row * CP + cc * 8 + e];
    }
}

struct Variant {
    const char* name;
    int bm, bn, threads;
    bool xcd;
    void (*kern)(uint16_t*, const uint16_t*, const uint16_t*, const uint16_t*, int, int, int, int64_t);
};
static const Variant VARIANTS[] = {
    {"128x128 reg-staged          ", 128, 128, 256, false, linear_tn_kernel<2, 2, false, false>},
    {"128x128 reg-staged  xcd     ", 128, 128, 256, true, linear_tn_kernel<2, 2, false, true>},
    {"256x128 reg-staged  xcd     ", 256, 128, 512, true, linear_tn_kernel<4, 2, false, true>},
    {"128x128 direct-to-LDS xcd   ", 128, 128, 256, true, linear_tn_kernel<2, 2, true, true>},
    {"256x128 direct-to-LDS xcd   ", 256, 128, 512, true, linear_tn_kernel<4, 2, true, true>},
    {"128x256 direct-to-LDS xcd   ", 128, 256, 512, true, linear_tn_kernel<2, 4, true, true>},
};
static void launch(const Variant& v, uint16_t* out, const uint16_t* x, const uint16_t* w, const uint16_t* b, int M, int N, int K,
                   int64_t ldw)
{
    const int ntn = (N + v.bn - 1) / v.bn, ntm = (M + v.bm - 1) / v.bm;
    const dim3 grid = v.xcd ? dim3(ntn * ntm) : dim3(ntn, ntm);
    hipLaunchKernelGGL(v.kern, grid, dim3(v.threads), 0, 0, out, x, w, b, M, N, K, ldw);
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
        linear_tn_ref<<<(unsigned)((no + 255) / 256), 256>>>(dref, dx, dw, db, s.M, s.N, s.K, s.ldw);
        std::vector<uint16_t> ho(no);
        std::vector<float> hr(no);
        hipMemcpy(hr.data(), dref, no * 4, hipMemcpyDeviceToHost);
        double scale = 0;
        for (size_t i = 0; i < no; ++i) scale = fmax(scale, fabs((double)hr[i]));
        double own_us[8], worst[8];
        int nv = 0;
        for (const Variant& v : VARIANTS) {
            hipMemset(dout, 0xFF, no * 2);                                 // NaN pattern: unwritten outputs are caught
            launch(v, dout, dx, dw, db, s.M, s.N, s.K, s.ldw);
            hipMemcpy(ho.data(), dout, no * 2, hipMemcpyDeviceToHost);
            double wv = 0;
            for (size_t i = 0; i < no; ++i) {
                uint32_t u = ((uint32_t)ho[i]) << 16;
                float f;
                memcpy(&f, &u, 4);
                const double d = fabs((double)f - hr[i]);
                wv = (d == d) ? fmax(wv, d) : 1e30;                        // NaN -> huge
            }
            for (int i = 0; i < 3; ++i) launch(v, dout, dx, dw, db, s.M, s.N, s.K, s.ldw);
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) launch(v, dout, dx, dw, db, s.M, s.N, s.K, s.ldw);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            own_us[nv] = time_ms(e0, e1) / 20 * 1e3;
            worst[nv] = wv;
            ++nv;
        }
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
        printf("%-26s M=%5d N=%4d K=%4d   library(heuristic) %7.1f us %6.0f TF/s   (|ref| max %.3g)\n", s.what, s.M, s.N, s.K, lib_us,
               lib_us > 0 ? fl / lib_us / 1e6 : 0.0, scale);
        for (int i = 0; i < nv; ++i)
            printf("    %s %7.1f us %6.0f TF/s   max|err| %.3g %s\n", VARIANTS[i].name, own_us[i], fl / own_us[i] / 1e6, worst[i],
                   worst[i] < 0.05 * (scale + 1) ? "" : "  <-- WRONG");
        hipFree(dx); hipFree(dw); hipFree(db); hipFree(dout); hipFree(dlib); hipFree(dref);
    }
    return 0;
}
