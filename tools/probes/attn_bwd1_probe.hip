// A/B of the one-pass attention backward (csrc/attn_rpe2d_bwd1.hpp) against the two-launch backward on the same
// inputs (development probe, not part of the library): element-wise comparison of dq / dk / dv and of the table
// gradients (partials summed on the host), then interleaved timing of both (HIP events, median of the rounds).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Icream_amd/csrc \
//         tools/probes/attn_bwd1_probe.hip -o tools/probes/attn_bwd1_probe && tools/probes/attn_bwd1_probe
// -DATTN_PROFILE: per-phase cycle stamps of the one-pass kernel (last item of every workgroup); -DATTN_PROFILE_STEPS adds
// the sub-phases of step 3
#include "../../cream_amd/csrc/attn_rpe2d.hip"
namespace cream { thread_local hipEvent_t tl_stop_event = nullptr; thread_local hipEvent_t tl_start_event = nullptr; }   // (block_seq.cpp defines them in the library)
namespace cream { int cu_count() { int dev = 0, n = 256; hipGetDevice(&dev); hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n / 8 * 8; } }   // (gemm_mfma.hip in the library)
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static float bf(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t tobf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float urand() { return rand() / (float)RAND_MAX * 2.f - 1.f; }

struct Cmp { double max_abs = 0, max_ref = 0, sum_sq = 0, ref_sq = 0; long bad = 0; };
static void cmp_bf16(Cmp& c, const uint16_t* a, const uint16_t* b, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        const double x = bf(a[i]), y = bf(b[i]);
        if (!(std::isfinite(x) && std::isfinite(y))) { ++c.bad; continue; }
        c.max_abs = std::max(c.max_abs, std::fabs(x - y)); c.max_ref = std::max(c.max_ref, std::fabs(y));
        c.sum_sq += (x - y) * (x - y); c.ref_sq += y * y;
    }
}

static int TM = 1;          // one-pass kernel under test: 1 = bwd1 (7 waves), 2 = bwd2 (12 waves, roles on separate waves); BWD_MODE in the environment
static int run_case(int B, int H, int rounds, float qscale) {
    const int N = 197, gh = 14, gw = 14, mr = 14, NP = 224;
    const int64_t sn = 3 * H * 64, sb = (int64_t)N * sn, sh = 64;
    const size_t nqkv = (size_t)B * N * sn, no = (size_t)B * N * H * 64;
    std::vector<uint16_t> hq(nqkv), hdo(no);
    for (auto& x : hq) x = tobf(urand() * qscale);
    for (auto& x : hdo) x = tobf(urand() * 0.25f);
    std::vector<float> ht(4 * 30 * 64);
    for (auto& x : ht) x = urand() * 0.5f;
    uint16_t *dqkv, *dout, *dsp, *ddo, *g0, *g1, *ddlt, *dqe, *dde;
    float *dt, *dlse, *ddelta, *tab0, *tab1;
    const int parts = cream_attn_rpe2d_dtab_parts(B, H);
    hipMalloc(&dqkv, nqkv * 2); hipMalloc(&dout, no * 2); hipMalloc(&ddo, no * 2);
    hipMalloc(&dsp, (size_t)B * H * 64 * NP * 2); hipMalloc(&dt, ht.size() * 4); hipMalloc(&dlse, (size_t)B * H * N * 4);
    hipMalloc(&g0, nqkv * 2); hipMalloc(&g1, nqkv * 2);
    hipMalloc(&ddlt, (size_t)B * H * 64 * NP * 2); hipMalloc(&dqe, (size_t)B * H * NP * 32 * 2);
    hipMalloc(&dde, (size_t)B * H * NP * 32 * 2); hipMalloc(&ddelta, (size_t)B * H * NP * 4);
    hipMalloc(&tab0, (size_t)parts * 4 * 32 * 64 * 4); hipMalloc(&tab1, (size_t)parts * 4 * 32 * 64 * 4);
    hipMemcpy(dqkv, hq.data(), nqkv * 2, hipMemcpyHostToDevice);
    hipMemcpy(ddo, hdo.data(), no * 2, hipMemcpyHostToDevice);
    hipMemcpy(dt, ht.data(), ht.size() * 4, hipMemcpyHostToDevice);
    hipMemset(g0, 0xFF, nqkv * 2); hipMemset(g1, 0xFF, nqkv * 2);          // NaN poison: every element must be written
    hipMemset(tab0, 0xFF, (size_t)parts * 32768); hipMemset(tab1, 0xFF, (size_t)parts * 32768);
    int rc = cream_attn_rpe2d_fwd(dout, dlse, dsp, dqkv, dqkv + H * 64, dqkv + 2 * H * 64, sb, sn, sh, dt, dt + 1920, dt + 3840,
                                  dt + 5760, 64, B, H, N, gh, gw, mr, 0.125f, CREAM_BF16, nullptr);
    if (rc) { printf("fwd rc=%d\n", rc); return 1; }
    auto bwd = [&](int mode, uint16_t* g, float* tab) {
        cream_attn_rpe2d_bwd_mode(mode);
        return cream_attn_rpe2d_bwd(g, g + H * 64, g + 2 * H * 64, sb, sn, sh, tab, ddlt, dqe, dde, ddelta, ddo, dout, dlse, dsp,
                                    dqkv, dqkv + H * 64, dqkv + 2 * H * 64, sb, sn, sh, dt, dt + 1920, dt + 3840, dt + 5760, 64,
                                    B, H, N, gh, gw, mr, 0.125f, CREAM_BF16, nullptr);
    };
    rc = bwd(0, g0, tab0); hipDeviceSynchronize();
    if (rc || hipGetLastError() != hipSuccess) { printf("bwd two-launch rc=%d\n", rc); return 1; }
    rc = bwd(TM, g1, tab1);
    hipError_t e = hipDeviceSynchronize();
    if (rc || e != hipSuccess) { printf("bwd one-pass rc=%d hip=%s\n", rc, hipGetErrorString(e)); return 1; }
    std::vector<uint16_t> h0(nqkv), h1(nqkv);
    hipMemcpy(h0.data(), g0, nqkv * 2, hipMemcpyDeviceToHost); hipMemcpy(h1.data(), g1, nqkv * 2, hipMemcpyDeviceToHost);
    const char* nm[3] = {"dq", "dk", "dv"};
    int fail = 0;
    for (int part = 0; part < 3; ++part) {
        Cmp c;
        for (int b = 0; b < B; ++b) for (int n = 0; n < N; ++n)
            cmp_bf16(c, h1.data() + (size_t)b * sb + (size_t)n * sn + part * H * 64, h0.data() + (size_t)b * sb + (size_t)n * sn + part * H * 64, (size_t)H * 64);
        const double rel = std::sqrt(c.sum_sq / std::max(c.ref_sq, 1e-30));
        printf("  B=%d H=%d %s: max|diff| %.3e (max|ref| %.3e)  rel-L2 %.3e  nonfinite %ld\n", B, H, nm[part], c.max_abs, c.max_ref, rel, c.bad);
        if (c.bad || rel > 4e-3 || c.max_abs > 0.03 * c.max_ref) fail = 1;
    }
    {
        std::vector<float> t0((size_t)parts * 8192), t1((size_t)parts * 8192);
        hipMemcpy(t0.data(), tab0, t0.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(t1.data(), tab1, t1.size() * 4, hipMemcpyDeviceToHost);
        for (int tab = 0; tab < 4; ++tab) {
            double md = 0, mrf = 0; long bad = 0;
            for (int i = 0; i < 2048; ++i) {
                double a0 = 0, a1 = 0;
                for (int p = 0; p < parts; ++p) { a0 += t0[(size_t)p * 8192 + tab * 2048 + i]; a1 += t1[(size_t)p * 8192 + tab * 2048 + i]; }
                if (!(std::isfinite(a0) && std::isfinite(a1))) { ++bad; continue; }
                md = std::max(md, std::fabs(a0 - a1)); mrf = std::max(mrf, std::fabs(a0));
            }
            printf("  B=%d H=%d dtab[%d]: max|diff| %.3e (max|ref| %.3e) nonfinite %ld\n", B, H, tab, md, mrf, bad);
            if (bad || md > 0.02 * mrf + 1e-6) fail = 1;
        }
    }
    // determinism: a second one-pass run must reproduce the first bit for bit
    {
        uint16_t* g2; float* tab2;
        hipMalloc(&g2, nqkv * 2); hipMalloc(&tab2, (size_t)parts * 32768);
        bwd(TM, g2, tab2); hipDeviceSynchronize();
        std::vector<uint16_t> h2(nqkv);
        hipMemcpy(h2.data(), g2, nqkv * 2, hipMemcpyDeviceToHost);
        size_t diff = 0;
        for (int b = 0; b < B; ++b) for (int n = 0; n < N; ++n)
            diff += memcmp(h2.data() + (size_t)b * sb + (size_t)n * sn, h1.data() + (size_t)b * sb + (size_t)n * sn, (size_t)sn * 2) != 0;
        std::vector<float> t1((size_t)parts * 8192), t2((size_t)parts * 8192);
        hipMemcpy(t1.data(), tab1, t1.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(t2.data(), tab2, t2.size() * 4, hipMemcpyDeviceToHost);
        const int tdiff = memcmp(t1.data(), t2.data(), t1.size() * 4) != 0;
        printf("  B=%d H=%d rerun: %zu rows differ, tables %s\n", B, H, diff, tdiff ? "DIFFER" : "identical");
        if (diff || tdiff) fail = 1;
        hipFree(g2); hipFree(tab2);
    }
#ifdef ATTN_PROFILE
    if (rounds > 0 && TM == 2) {
        long long* dprof;
        const int grid = parts;
        hipMalloc(&dprof, (size_t)grid * 12 * 12 * 8);
        hipMemset(dprof, 0, (size_t)grid * 12 * 12 * 8);
        hipMemcpyToSymbol(HIP_SYMBOL(g_attn_prof), &dprof, sizeof(dprof));
        bwd(2, g1, tab1); hipDeviceSynchronize();
        std::vector<long long> hp((size_t)grid * 12 * 12);
        hipMemcpy(hp.data(), dprof, hp.size() * 8, hipMemcpyDeviceToHost);
        long long* nul = nullptr;
        hipMemcpyToSymbol(HIP_SYMBOL(g_attn_prof), &nul, sizeof(nul));
        const char* pn[7] = {"wait for K, V, Q, dO", "prologue (delta, lookups, shifts)", "7 steps", "K / V requests, wait [C]", "shifts, dq product, dQ rows, dL'", "wait [D]", "wait [E] (key-table jobs)"};
        const char* cn[7] = {"wait for K, V, Q, dO", "value-table job", "7 steps (consume + barriers)", "K / V requests, last consume, [C]", "dK / dV rows", "wait [D]", "key-table job + [E]"};
        for (int role = 0; role < 2; ++role) {
            std::vector<double> sum(7, 0.0); double tot = 0; int cnt = 0;
            for (int blk = 0; blk < grid; ++blk) for (int w = role ? 7 : 0; w < (role ? 12 : 7); ++w) {
                long long* d = &hp[((size_t)blk * 12 + w) * 12];
                if (d[7] <= d[0]) continue;
                for (int i = 0; i < 7; ++i) sum[i] += (double)(d[i + 1] - d[i]);
                tot += (double)(d[7] - d[0]); ++cnt;
            }
            printf("  bwd2 %s, cycles per item and wave (second item of each workgroup, %d waves):\n", role ? "CONSUMERS" : "PRODUCERS", cnt);
            for (int i = 0; i < 7; ++i) printf("    %-36s %9.0f\n", role ? cn[i] : pn[i], sum[i] / std::max(cnt, 1));
            printf("    %-36s %9.0f\n", "total", tot / std::max(cnt, 1));
        }
        hipFree(dprof);
    }
    if (rounds > 0 && TM != 2) {
        long long* dprof;
        const int grid = parts;
        hipMalloc(&dprof, (size_t)grid * 8 * 20 * 8);
        hipMemset(dprof, 0, (size_t)grid * 8 * 20 * 8);
        hipMemcpyToSymbol(HIP_SYMBOL(g_attn_prof), &dprof, sizeof(dprof));
        bwd(1, g1, tab1); hipDeviceSynchronize();
        std::vector<long long> hp((size_t)grid * 8 * 20);
        hipMemcpy(hp.data(), dprof, hp.size() * 8, hipMemcpyDeviceToHost);
        long long* nul = nullptr;
        hipMemcpyToSymbol(HIP_SYMBOL(g_attn_prof), &nul, sizeof(nul));
#ifdef ATTN_PROFILE_STEPS
        const char* names[] = {"load+store K,V,Q,dO", "prologue (delta, lookups)", "steps 0-2", "s3 consume", "s3 S,dP mfma", "s3 exp/dS valu",
                               "s3 dQx mfma", "s3 barrier A", "s3 publish", "s3 barrier B", "steps 4-6", "final consume", "store dk dv",
                               "shift+dq", "barrier C", "table jobs"};
        const int np = 16;
#else
        const char* names[] = {"load+store K,V,Q,dO", "prologue (delta, lookups)", "7 steps", "final consume", "store dk dv", "shift+dq",
                               "barrier C", "table jobs"};
        const int np = 8;
#endif
        std::vector<double> sum(np, 0.0); double tot = 0; int cnt = 0;
        for (int blk = 0; blk < grid; ++blk) for (int w = 0; w < 7; ++w) {
            long long* d = &hp[((size_t)blk * 8 + w) * 20];
            if (d[np] <= d[0]) continue;
            for (int i = 0; i < np; ++i) sum[i] += (double)(d[i + 1] - d[i]);
            tot += (double)(d[np] - d[0]); ++cnt;
        }
        printf("  one-pass kernel, cycles per item and wave (last item of each workgroup, %d waves):\n", cnt);
        for (int i = 0; i < np; ++i) printf("    %-28s %9.0f\n", names[i], sum[i] / std::max(cnt, 1));
        printf("    %-28s %9.0f\n", "total", tot / std::max(cnt, 1));
        hipFree(dprof);
    }
#endif
    if (TM == 2) {          // bwd2 against bwd1: the same bits are expected (same contraction orders, same partial layout)
        uint16_t* g2; float* tab2;
        hipMalloc(&g2, nqkv * 2); hipMalloc(&tab2, (size_t)parts * 32768);
        hipMemset(g2, 0xFF, nqkv * 2);
        bwd(1, g2, tab2); hipDeviceSynchronize();
        std::vector<uint16_t> h2(nqkv);
        hipMemcpy(h2.data(), g2, nqkv * 2, hipMemcpyDeviceToHost);
        size_t diff = 0;
        for (int b = 0; b < B; ++b) for (int n = 0; n < N; ++n)
            diff += memcmp(h2.data() + (size_t)b * sb + (size_t)n * sn, h1.data() + (size_t)b * sb + (size_t)n * sn, (size_t)sn * 2) != 0;
        std::vector<float> t1((size_t)parts * 8192), t2((size_t)parts * 8192);
        hipMemcpy(t1.data(), tab1, t1.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(t2.data(), tab2, t2.size() * 4, hipMemcpyDeviceToHost);
        // (bwd2 adds a workgroup's items in one fp32 chain, bwd1 per item + additions: the partials agree within fp32 rounding)
        double tmd = 0, tmr = 0;
        for (size_t i = 0; i < t1.size(); ++i) { tmd = std::max(tmd, (double)std::fabs(t1[i] - t2[i])); tmr = std::max(tmr, (double)std::fabs(t2[i])); }
        printf("  B=%d H=%d bwd2 vs bwd1: %zu rows differ, table partials max|diff| %.3e (max|ref| %.3e)\n", B, H, diff, tmd, tmr);
        if (diff || !(tmd <= 2e-6 * tmr + 1e-7)) fail = 1;
        hipFree(g2); hipFree(tab2);
    }
    if (rounds > 0) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        std::vector<float> t[4];
        for (int it = 0; it < rounds; ++it)
            for (int mode = 0; mode < 4; ++mode) {
                hipEventRecord(e0);
                if (mode < 3) bwd(mode, mode ? g1 : g0, mode ? tab1 : tab0);
                else cream_attn_rpe2d_fwd(dout, dlse, dsp, dqkv, dqkv + H * 64, dqkv + 2 * H * 64, sb, sn, sh, dt, dt + 1920, dt + 3840,
                                          dt + 5760, 64, B, H, N, gh, gw, mr, 0.125f, CREAM_BF16, nullptr);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (it >= 2) t[mode].push_back(ms * 1e3f);
            }
        for (int mode = 0; mode < 4; ++mode) {
            std::sort(t[mode].begin(), t[mode].end());
            printf("  B=%d H=%d %-22s median %.1f us  min %.1f us\n", B, H, mode == 0 ? "bwd two-launch" : mode == 1 ? "bwd1 one-pass(+images)" : mode == 2 ? "bwd2 one-pass(+images)" : "fwd14",
                   t[mode][t[mode].size() / 2], t[mode][0]);
        }
    }
    hipFree(dqkv); hipFree(dout); hipFree(ddo); hipFree(dsp); hipFree(dt); hipFree(dlse); hipFree(g0); hipFree(g1);
    hipFree(ddlt); hipFree(dqe); hipFree(dde); hipFree(ddelta); hipFree(tab0); hipFree(tab1);
    printf("  B=%d H=%d -> %s\n", B, H, fail ? "MISMATCH" : "ok");
    return fail;
}

int main(int argc, char** argv) {
    srand(7);
    int fail = 0;
    if (getenv("BWD_MODE")) TM = atoi(getenv("BWD_MODE"));
    printf("one-pass mode under test: %d\n", TM);
    fail |= run_case(2, 3, 0, 1.0f);          // fewer items than CUs
    fail |= run_case(3, 5, 0, 2.0f);          // sharper softmax
    fail |= run_case(128, 6, 12, 1.0f);       // the bench shape (3 items per workgroup)
    fail |= run_case(128, 5, 8, 1.0f);
    fail |= run_case(128, 7, 8, 1.0f);
    printf(fail ? "PROBE FAILED\n" : "PROBE OK\n");
    return fail;
}
