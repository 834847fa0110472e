// A/B of the ping-pong online-softmax forward (csrc/attn_rpe2d_fwd2.hpp) against attn_rpe2d_fwd14 on the same inputs
// (development probe, not part of the library): element-wise comparison of out / lse / S'^T, a case that forces the lazy
// maximum to move at a late tile (one key row spiked against one query row), reruns bit-identical, interleaved timing.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Icream_amd/csrc \
//         tools/probes/attn_fwd2_probe.hip -o tools/probes/attn_fwd2_probe && tools/probes/attn_fwd2_probe
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

struct Cmp { double max_abs = 0, max_ref = 0, sum_sq = 0, ref_sq = 0; long bad = 0; };
static void add(Cmp& c, double x, double y) {
    if (!(std::isfinite(x) && std::isfinite(y))) { ++c.bad; return; }
    c.max_abs = std::max(c.max_abs, std::fabs(x - y)); c.max_ref = std::max(c.max_ref, std::fabs(y));
    c.sum_sq += (x - y) * (x - y); c.ref_sq += y * y;
}
static int report(const char* what, int B, int H, const Cmp& c, double tol) {
    const double rel = std::sqrt(c.sum_sq / std::max(c.ref_sq, 1e-30));
    printf("  B=%d H=%d %-5s max|diff| %.3e (max|ref| %.3e)  rel-L2 %.3e  nonfinite %ld\n", B, H, what, c.max_abs, c.max_ref, rel, c.bad);
    return c.bad || rel > tol;
}

static int run_case(int B, int H, int rounds, float qscale, bool spike) {
    const int N = 197, gh = 14, gw = 14, mr = 14, NP = 224;
    const int64_t sn = 3 * H * 64, sb = (int64_t)N * sn, sh = 64;
    const size_t nqkv = (size_t)B * N * sn, no = (size_t)B * N * H * 64, nsp = (size_t)B * H * 64 * NP, nl = (size_t)B * H * N;
    std::vector<uint16_t> hq(nqkv);
    for (auto& x : hq) x = tobf(urand() * qscale);
    if (spike) {
        // query 40 of every (b, h) against key 170 (tile 5): raw q.k far above the row's other scores -> the lazy maximum
        // must move at tile 5, after five tiles were accumulated at the old reference
        for (int b = 0; b < B; ++b) for (int h = 0; h < H; ++h) for (int d = 0; d < 64; ++d) {
            hq[(size_t)b * sb + 40 * sn + 0 * H * 64 + h * 64 + d] = tobf(3.0f);
            hq[(size_t)b * sb + 170 * sn + 1 * H * 64 + h * 64 + d] = tobf(3.0f);
        }
    }
    std::vector<float> ht(4 * 30 * 64);
    for (auto& x : ht) x = urand() * 0.5f;
    uint16_t *dqkv, *o0, *o1, *sp0, *sp1, *img;
    float *dt, *l0, *l1;
    hipMalloc(&dqkv, nqkv * 2); hipMalloc(&o0, no * 2); hipMalloc(&o1, no * 2); hipMalloc(&sp0, nsp * 2); hipMalloc(&sp1, nsp * 2);
    hipMalloc(&dt, ht.size() * 4); hipMalloc(&l0, nl * 4); hipMalloc(&l1, nl * 4); hipMalloc(&img, cream_attn_rpe2d_table_image_bytes());
    hipMemcpy(dqkv, hq.data(), nqkv * 2, hipMemcpyHostToDevice);
    hipMemcpy(dt, ht.data(), ht.size() * 4, hipMemcpyHostToDevice);
    hipMemset(o0, 0xFF, no * 2); hipMemset(o1, 0xFF, no * 2); hipMemset(sp1, 0xFF, nsp * 2); hipMemset(l1, 0xFF, nl * 4);
    int rc = cream_attn_rpe2d_table_images(img, dt, dt + 1920, dt + 3840, dt + 5760, 64, mr, nullptr);
    if (rc) { printf("images rc=%d\n", rc); return 1; }
    auto fwd = [&](int mode, uint16_t* o, float* l, uint16_t* sp) {
        cream_attn_rpe2d_fwd_mode(mode);
        return cream_attn_rpe2d_fwd_img(o, l, sp, dqkv, dqkv + H * 64, dqkv + 2 * H * 64, sb, sn, sh, dt, dt + 1920, dt + 3840, dt + 5760,
                                        64, img, B, H, N, gh, gw, mr, 0.125f, CREAM_BF16, nullptr);
    };
    rc = fwd(0, o0, l0, sp0); hipDeviceSynchronize();
    if (rc) { printf("fwd14 rc=%d\n", rc); return 1; }
    rc = fwd(1, o1, l1, sp1);
    hipError_t e = hipDeviceSynchronize();
    if (rc || e != hipSuccess) { printf("fwd2 rc=%d hip=%s\n", rc, hipGetErrorString(e)); return 1; }
    std::vector<uint16_t> h0(no), h1(no), s0(nsp), s1(nsp);
    std::vector<float> a0(nl), a1(nl);
    hipMemcpy(h0.data(), o0, no * 2, hipMemcpyDeviceToHost); hipMemcpy(h1.data(), o1, no * 2, hipMemcpyDeviceToHost);
    hipMemcpy(s0.data(), sp0, nsp * 2, hipMemcpyDeviceToHost); hipMemcpy(s1.data(), sp1, nsp * 2, hipMemcpyDeviceToHost);
    hipMemcpy(a0.data(), l0, nl * 4, hipMemcpyDeviceToHost); hipMemcpy(a1.data(), l1, nl * 4, hipMemcpyDeviceToHost);
    int fail = 0;
    { Cmp c; for (size_t i = 0; i < no; ++i) add(c, bf(h1[i]), bf(h0[i])); fail |= report("out", B, H, c, 4e-3); }
    if (getenv("PROBE_DUMP")) {
        int shown = 0;
        for (size_t i = 0; i < no && shown < 40; ++i) {
            const float x = bf(h1[i]), y = bf(h0[i]);
            if (!std::isfinite(x) || std::fabs(x - y) > 0.05f) {
                const int d = i % 64, h = (i / 64) % H, n = (i / 64 / H) % N, b = i / 64 / H / N;
                printf("    out[b=%d n=%d h=%d d=%d] = %g (fwd14 %g)\n", b, n, h, d, x, y); ++shown;
            }
        }
    }
    { Cmp c; for (size_t i = 0; i < nl; ++i) add(c, a1[i], a0[i]); fail |= report("lse", B, H, c, 1e-4); }
    {
        Cmp c;
        for (size_t bh = 0; bh < (size_t)B * H; ++bh) for (int u = 0; u < 64; ++u) for (int q = 0; q < N; ++q)
            add(c, bf(s1[(bh * 64 + u) * NP + q]), bf(s0[(bh * 64 + u) * NP + q]));
        fail |= report("sp", B, H, c, 4e-3);
    }
    {   // rerun: bit-identical
        uint16_t* o2; hipMalloc(&o2, no * 2);
        fwd(1, o2, l1, sp1); hipDeviceSynchronize();
        std::vector<uint16_t> h2(no);
        hipMemcpy(h2.data(), o2, no * 2, hipMemcpyDeviceToHost);
        const int d = memcmp(h2.data(), h1.data(), no * 2) != 0;
        printf("  B=%d H=%d rerun: %s\n", B, H, d ? "DIFFERS" : "identical");
        fail |= d;
        hipFree(o2);
    }
    if (rounds > 0) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        std::vector<float> t[2];
        for (int it = 0; it < rounds; ++it)
            for (int mode = 0; mode < 2; ++mode) {
                hipEventRecord(e0);
                fwd(mode, mode ? o1 : o0, mode ? l1 : l0, mode ? sp1 : sp0);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (it >= 2) t[mode].push_back(ms * 1e3f);
            }
        for (int mode = 0; mode < 2; ++mode) {
            std::sort(t[mode].begin(), t[mode].end());
            printf("  B=%d H=%d %-8s median %.1f us  min %.1f us\n", B, H, mode ? "fwd2" : "fwd14", t[mode][t[mode].size() / 2], t[mode][0]);
        }
    }
    hipFree(dqkv); hipFree(o0); hipFree(o1); hipFree(sp0); hipFree(sp1); hipFree(dt); hipFree(l0); hipFree(l1); hipFree(img);
    printf("  B=%d H=%d%s -> %s\n", B, H, spike ? " (spiked)" : "", fail ? "MISMATCH" : "ok");
    return fail;
}

int main(int argc, char** argv) {
    srand(7);
    int fail = 0;
    fail |= run_case(1, 1, 0, 1.0f, false);         // one item: half 1 only requests
    fail |= run_case(1, 2, 0, 1.0f, false);         // one item per half
    fail |= run_case(2, 3, 0, 1.0f, false);         // fewer items than CUs
    fail |= run_case(3, 5, 0, 2.0f, true);          // sharper softmax + the lazy maximum moving at tile 5
    fail |= run_case(128, 6, 14, 1.0f, false);      // the bench shape (3 items per workgroup)
    fail |= run_case(128, 5, 8, 1.0f, false);
    fail |= run_case(128, 7, 8, 1.0f, true);
    fail |= run_case(200, 7, 4, 1.0f, false);       // 5.5 items per workgroup: odd and even item counts per half
    printf(fail ? "PROBE FAILED\n" : "PROBE OK\n");
    return fail;
}
