// nt8_trace_probe.hip — where a K-tile's cycles go in gemm_nt8_kernel (cream_amd/csrc/gemm_nt8.hpp): per-phase s_memtime stamps of
// every wave during a workgroup's first output tile (kernel built with -DNT8_TRACE), averaged over workgroups per phase kind and
// wave row.  Stamps per phase: d (requests issued) | a (past barrier 1, fragments landed) | b (8 MFMAs issued) | c (past barrier 2).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DNT8_TRACE -Iinclude -Icream_amd/csrc tools/probes/nt8_trace_probe.hip -o tools/probes/nt8_trace_probe
//   tools/probes/nt8_trace_probe M N K
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gemm_nt8.hpp"

using namespace cream;
using namespace cream::gemm;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static uint16_t f2bf_host(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

int main(int argc, char** argv)
{
    const int M = argc > 1 ? atoi(argv[1]) : 25216, N = argc > 2 ? atoi(argv[2]) : 448, K = argc > 3 ? atoi(argv[3]) : 1792;
    const size_t nx = (size_t)M * K, nw = (size_t)N * K, no = (size_t)M * N;
    std::vector<uint16_t> hx(nx), hw(nw), hb(N);
    srand(1);
    for (auto& v : hx) v = f2bf_host((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    for (auto& v : hw) v = f2bf_host((rand() / (float)RAND_MAX - 0.5f) * 0.2f);
    for (auto& v : hb) v = f2bf_host((rand() / (float)RAND_MAX - 0.5f));
    const int R = 6;                                            // rotating activation / output sets: cold operands
    uint16_t *dx[R], *dout[R], *dw, *db;
    for (int r = 0; r < R; ++r) { CK(hipMalloc(&dx[r], nx * 2)); CK(hipMalloc(&dout[r], no * 2)); CK(hipMemcpy(dx[r], hx.data(), nx * 2, hipMemcpyHostToDevice)); }
    CK(hipMalloc(&dw, nw * 2)); CK(hipMalloc(&db, N * 2));
    CK(hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), N * 2, hipMemcpyHostToDevice));
    NtParams p{};
    p.A = dx[0]; p.lda = K; p.B = dw; p.ldb = K; p.nseg = N; p.kseg = K; p.M = M; p.N = N; p.K = K; p.nvalid = N; p.out = dout[0]; p.ldo = N; p.bias = db;
    auto kern = gemm_nt8_kernel<EPI_BIAS>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, NT8_LDS_BYTES));
    const int tiles = ((M + 255) / 256) * ((N + 255) / 256), grid = tiles > 256 ? 256 : tiles;
    unsigned long long* dtr; CK(hipMalloc(&dtr, (size_t)grid * 8 * 512 * 8)); CK(hipMemset(dtr, 0, (size_t)grid * 8 * 512 * 8));
    for (int i = 0; i < 4; ++i) { p.A = dx[i % R]; p.out = dout[i % R]; hipLaunchKernelGGL(kern, dim3(grid), dim3(512), NT8_LDS_BYTES, 0, p); }
    CK(hipDeviceSynchronize());
    CK(hipMemcpyToSymbol(HIP_SYMBOL(cream::gemm::g_nt8_trace), &dtr, sizeof(dtr)));
    p.A = dx[4 % R]; p.out = dout[4 % R];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), NT8_LDS_BYTES, 0, p);
    hipEventRecord(e1);
    CK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)grid * 8 * 512);
    CK(hipMemcpy(h.data(), dtr, h.size() * 8, hipMemcpyDeviceToHost));
    const int nk = (K + 63) / 64, nph = nk * 4 < 128 ? nk * 4 : 128;
    printf("M=%d N=%d K=%d  tiles %d grid %d  traced launch %.1f us (stamps cost time)\n", M, N, K, tiles, grid, ms * 1e3);
    // steady state: K-tiles 2 .. nk-1; per phase kind and wave row: stage (c_prev -> d), wait+reads (d -> a), mfma (a -> b), barrier 2 (b -> c)
    double acc[2][4][4] = {}; long cnt[2][4] = {};
    double ktile[2] = {0, 0}; long nkt[2] = {0, 0};
    for (int b = 0; b < grid; ++b)
        for (int w = 0; w < 8; ++w) {
            const unsigned long long* t = &h[((size_t)b * 8 + w) * 512];
            const int wr = w >> 2;
            for (int ph = 8; ph < nph; ++ph) {
                const unsigned long long d = t[ph * 4], a = t[ph * 4 + 1], bb = t[ph * 4 + 2], c = t[ph * 4 + 3], cprev = t[ph * 4 - 1];
                if (!d || !c || !cprev) continue;
                acc[wr][ph & 3][0] += (double)(d - cprev); acc[wr][ph & 3][1] += (double)(a - d);
                acc[wr][ph & 3][2] += (double)(bb - a); acc[wr][ph & 3][3] += (double)(c - bb);
                ++cnt[wr][ph & 3];
            }
            if (nph >= 16 && t[nph * 4 - 1] && t[8 * 4 - 1]) { ktile[wr] += (double)(t[nph * 4 - 1] - t[8 * 4 - 1]) / (nph / 4 - 2); ++nkt[wr]; }
        }
    for (int wr = 0; wr < 2; ++wr) {
        printf("wave row %d: cycles per K-tile (steady state) %.0f\n", wr, nkt[wr] ? ktile[wr] / nkt[wr] : 0.0);
        for (int ph = 0; ph < 4; ++ph) {
            const double n = cnt[wr][ph] ? (double)cnt[wr][ph] : 1.0;
            printf("   phase %d: requests %.0f | reads + vmcnt + barrier 1 + fragments landed %.0f | 8 MFMA issued %.0f | barrier 2 %.0f   = %.0f\n", ph + 1,
                   acc[wr][ph][0] / n, acc[wr][ph][1] / n, acc[wr][ph][2] / n, acc[wr][ph][3] / n,
                   (acc[wr][ph][0] + acc[wr][ph][1] + acc[wr][ph][2] + acc[wr][ph][3]) / n);
        }
    }
    // one workgroup's raw timeline, waves 0 and 4, K-tiles 2 and 3 (relative to wave 0's first stamp of K-tile 2)
    const int bsel = grid / 2;
    const unsigned long long t0 = h[((size_t)bsel * 8 + 0) * 512 + 8 * 4];
    for (int w : {0, 4}) {
        const unsigned long long* t = &h[((size_t)bsel * 8 + w) * 512];
        printf("block %d wave %d:", bsel, w);
        for (int i = 8 * 4; i < 16 * 4 && i < nph * 4; ++i) printf(" %s%lld", (i & 3) == 0 ? "| " : "", (long long)(t[i] - t0));
        printf("\n");
    }
    return 0;
}
