// A/B of the DMA-staged attention forward (csrc/attn_rpe2d_fwd1.hpp) against attn_rpe2d_fwd14_kernel on the same inputs
// (development probe, not part of the library): element-wise comparison of out / lse / bucket sums, then interleaved
// timing of both (HIP events, median of the rounds).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Icream_amd/csrc \
//         tools/probes/attn_fwd1_probe.hip -o tools/probes/attn_fwd1_probe && tools/probes/attn_fwd1_probe
// -DATTN_PROFILE: per-phase cycle stamps of the new kernel (last item of every workgroup; -DATTN_PROFILE_ITEM1: the second)
#include "../../cream_amd/csrc/attn_rpe2d.hip"
namespace cream { thread_local hipEvent_t tl_stop_event = nullptr; thread_local hipEvent_t tl_start_event = nullptr; }   // (block_seq.cpp defines them in the library)
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static float bf(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t tobf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float urand() { return rand() / (float)RAND_MAX * 2.f - 1.f; }

static int run_case(int B, int H, int rounds, float qscale) {
    const int N = 197, gh = 14, gw = 14, mr = 14, NP = 224;
    const int64_t sn = 3 * H * 64, sb = (int64_t)N * sn, sh = 64;
    const size_t nqkv = (size_t)B * N * sn, no = (size_t)B * N * H * 64, nsp = (size_t)B * H * 64 * NP, nl = (size_t)B * H * N;
    std::vector<uint16_t> hq(nqkv);
    for (auto& x : hq) x = tobf(urand() * qscale);
    std::vector<float> ht(4 * 30 * 64);
    for (auto& x : ht) x = urand() * 0.5f;
    uint16_t *dqkv, *dout[2], *dsp[2];
    float *dt, *dlse[2];
    hipMalloc(&dqkv, nqkv * 2); hipMalloc(&dt, ht.size() * 4);
    for (int m = 0; m < 2; ++m) {
        hipMalloc(&dout[m], no * 2); hipMalloc(&dsp[m], nsp * 2); hipMalloc(&dlse[m], nl * 4);
        hipMemset(dout[m], 0xFF, no * 2); hipMemset(dsp[m], 0xFF, nsp * 2); hipMemset(dlse[m], 0xFF, nl * 4);   // NaN poison
    }
    hipMemcpy(dqkv, hq.data(), nqkv * 2, hipMemcpyHostToDevice);
    hipMemcpy(dt, ht.data(), ht.size() * 4, hipMemcpyHostToDevice);
    auto fwd = [&](int mode) {
        cream_attn_rpe2d_fwd_mode(mode);
        return cream_attn_rpe2d_fwd(dout[mode], dlse[mode], dsp[mode], dqkv, dqkv + H * 64, dqkv + 2 * H * 64, sb, sn, sh, dt, dt + 1920,
                                    dt + 3840, dt + 5760, 64, B, H, N, gh, gw, mr, 0.125f, CREAM_BF16, nullptr);
    };
    for (int m = 0; m < 2; ++m) {
        const int rc = fwd(m);
        const hipError_t e = hipDeviceSynchronize();
        if (rc || e != hipSuccess) { printf("fwd mode %d rc=%d hip=%s\n", m, rc, hipGetErrorString(e)); return 1; }
    }
    int fail = 0;
    {
        std::vector<uint16_t> a(no), b(no);
        hipMemcpy(a.data(), dout[0], no * 2, hipMemcpyDeviceToHost); hipMemcpy(b.data(), dout[1], no * 2, hipMemcpyDeviceToHost);
        double md = 0, mr_ = 0; size_t nd = 0, bad = 0;
        for (size_t i = 0; i < no; ++i) {
            const double x = bf(b[i]), y = bf(a[i]);
            if (!(std::isfinite(x) && std::isfinite(y))) { ++bad; continue; }
            nd += a[i] != b[i]; md = std::max(md, std::fabs(x - y)); mr_ = std::max(mr_, std::fabs(y));
        }
        printf("  B=%d H=%d out: %zu of %zu elements differ, max|diff| %.3e (max|ref| %.3e) nonfinite %zu\n", B, H, nd, no, md, mr_, bad);
        if (bad || md > 0.02 * mr_) fail = 1;
    }
    {
        std::vector<float> a(nl), b(nl);
        hipMemcpy(a.data(), dlse[0], nl * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), dlse[1], nl * 4, hipMemcpyDeviceToHost);
        double md = 0; size_t bad = 0;
        for (size_t i = 0; i < nl; ++i) { if (!(std::isfinite(a[i]) && std::isfinite(b[i]))) { ++bad; continue; } md = std::max(md, (double)std::fabs(a[i] - b[i])); }
        printf("  B=%d H=%d lse: max|diff| %.3e nonfinite %zu\n", B, H, md, bad);
        if (bad || md > 1e-4) fail = 1;
    }
    {
        std::vector<uint16_t> a(nsp), b(nsp);
        hipMemcpy(a.data(), dsp[0], nsp * 2, hipMemcpyDeviceToHost); hipMemcpy(b.data(), dsp[1], nsp * 2, hipMemcpyDeviceToHost);
        double md = 0, mr_ = 0; size_t nd = 0, bad = 0;
        for (size_t r = 0; r < (size_t)B * H * 64; ++r)
            for (int q = 0; q < N; ++q) {                                     // (padding queries: any finite value)
                const size_t i = r * NP + q;
                const double x = bf(b[i]), y = bf(a[i]);
                if (!(std::isfinite(x) && std::isfinite(y))) { ++bad; continue; }
                nd += a[i] != b[i]; md = std::max(md, std::fabs(x - y)); mr_ = std::max(mr_, std::fabs(y));
            }
        printf("  B=%d H=%d bucket sums: %zu elements differ, max|diff| %.3e (max|ref| %.3e) nonfinite %zu\n", B, H, nd, md, mr_, bad);
        if (bad || md > 0.02 * mr_) fail = 1;
    }
#ifdef ATTN_PROFILE
    if (rounds > 0) {
        long long* dprof;
        const int grid = cream_attn_rpe2d_dtab_parts(B, H);
        hipMalloc(&dprof, (size_t)grid * 8 * 20 * 8);
        hipMemset(dprof, 0, (size_t)grid * 8 * 20 * 8);
        hipMemcpyToSymbol(HIP_SYMBOL(g_attn_prof), &dprof, sizeof(dprof));
        fwd(1); hipDeviceSynchronize();
        std::vector<long long> hp((size_t)grid * 8 * 20);
        hipMemcpy(hp.data(), dprof, hp.size() * 8, hipMemcpyDeviceToHost);
        long long* nul = nullptr;
        hipMemcpyToSymbol(HIP_SYMBOL(g_attn_prof), &nul, sizeof(nul));
        const char* names[] = {"lookups + extension", "42 score MFMAs", "wait V + barrier A + K' DMA", "softmax", "42 P.V MFMAs",
                               "wait K' + barrier C + V' DMA", "bucket sums + value tables", "store O"};
        const int np = 8;
        std::vector<double> sum(np, 0.0); double tot = 0; int cnt = 0;
        for (int blk = 0; blk < grid; ++blk) for (int w = 0; w < 7; ++w) {
            long long* d = &hp[((size_t)blk * 8 + w) * 20];
            if (d[np] <= d[0]) continue;
            for (int i = 0; i < np; ++i) sum[i] += (double)(d[i + 1] - d[i]);
            tot += (double)(d[np] - d[0]); ++cnt;
        }
        printf("  fwd1 kernel, cycles per item and wave (%d waves):\n", cnt);
        for (int i = 0; i < np; ++i) printf("    %-30s %9.0f\n", names[i], sum[i] / std::max(cnt, 1));
        printf("    %-30s %9.0f\n", "total", tot / std::max(cnt, 1));
        hipFree(dprof);
    }
#endif
    if (rounds > 0) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        std::vector<float> t[2];
        for (int it = 0; it < rounds; ++it)
            for (int mode = 0; mode < 2; ++mode) {
                hipEventRecord(e0);
                fwd(mode);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (it >= 2) t[mode].push_back(ms * 1e3f);
            }
        for (int mode = 0; mode < 2; ++mode) {
            std::sort(t[mode].begin(), t[mode].end());
            printf("  B=%d H=%d %-22s median %.1f us  min %.1f us\n", B, H, mode == 0 ? "fwd14 (register-staged)" : "fwd1 (DMA)",
                   t[mode][t[mode].size() / 2], t[mode][0]);
        }
    }
    hipFree(dqkv); hipFree(dt);
    for (int m = 0; m < 2; ++m) { hipFree(dout[m]); hipFree(dsp[m]); hipFree(dlse[m]); }
    printf("  B=%d H=%d -> %s\n", B, H, fail ? "MISMATCH" : "ok");
    return fail;
}

int main() {
    srand(11);
    int fail = 0;
    fail |= run_case(2, 3, 0, 1.0f);          // fewer items than CUs
    fail |= run_case(3, 5, 0, 2.0f);          // sharper softmax
    fail |= run_case(128, 6, 12, 1.0f);       // the bench shape (3 items per workgroup)
    fail |= run_case(128, 5, 8, 1.0f);
    fail |= run_case(128, 7, 8, 1.0f);
    printf(fail ? "PROBE FAILED\n" : "PROBE OK\n");
    return fail;
}
