// Development probe: the rpe_index forward (gather) kernel of config 4 (B=64, H=12, L=577, nb=50) one plane per workgroup
// against several planes per workgroup (vectors per thread NV, planes per group G, workgroup size), every variant compared bit
// for bit with the first.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Icream_amd/csrc tools/probes/rpe_probe.hip -o tools/probes/rpe_probe
#include "../../cream_amd/csrc/rpe_index.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

template <int BYTES>
float run(void* y, const void* in, const int32_t* idx, int BH, int H, int L, int nb, int nblk, int threads, hipEvent_t e0, hipEvent_t e1) {
    using E = typename raw_elem<BYTES>::type;
    const int rpb = (L + nblk - 1) / nblk;
    const size_t lds = (size_t)rpb * nb * BYTES;
    float sum = 0;
    for (int it = 0; it < 7; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((rpe_gather_plane<BYTES>), dim3(nblk, BH), dim3(threads), lds, 0, (E*)y, (const E*)in, idx, H, L, L, nb,
                           (int64_t)H * L * nb, (int64_t)L * nb, (int64_t)nb, (int64_t)1, rpb);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it >= 2) sum += ms;
    }
    return sum / 5;
}

template <int BYTES, int NV>
float run_planes(void* y, const void* in, const int32_t* idx, int BH, int H, int L, int nb, int G, int threads, hipEvent_t e0, hipEvent_t e1,
                 int* nblk_out) {
    using E = typename raw_elem<BYTES>::type;
    constexpr int V = 16 / BYTES;
    int rows = std::min<int64_t>(L, (int64_t)NV * threads * V / L);
    rows = std::min(rows, 4 * threads / nb);
    const int nblk = (L + rows - 1) / rows;
    rows = (L + nblk - 1) / nblk;
    *nblk_out = nblk;
    int period = 1;
    while (((int64_t)period * L * L * BYTES) % 16) ++period;
    const int members = (BH + period - 1) / period, groups = (members + G - 1) / G;
    const size_t lds = 2 * (((size_t)rows * nb * BYTES + 15) / 16 * 16);
    float sum = 0;
    for (int it = 0; it < 7; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((rpe_gather_planes<BYTES, NV>), dim3(nblk, period * groups), dim3(threads), lds, 0, (E*)y, (const E*)in, idx, BH, H, L, L,
                           nb, (int64_t)H * L * nb, (int64_t)L * nb, (int64_t)nb, (int64_t)1, rows, period, G);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it >= 2) sum += ms;
    }
    return sum / 5;
}

template <int BYTES>
void sweep(const char* tag) {
    const int B = 64, H = 12, L = 577, nb = 50, BH = B * H;
    const size_t nin = (size_t)BH * L * nb, nout = (size_t)BH * L * L;
    std::vector<unsigned char> hin(nin * BYTES);
    srand(5);
    for (auto& c : hin) c = rand() & 0xff;
    std::vector<int32_t> hidx((size_t)L * L);
    for (auto& v : hidx) v = rand() % nb;
    void *in, *y, *yref; int32_t* idx;
    hipMalloc(&in, nin * BYTES); hipMalloc(&y, nout * BYTES); hipMalloc(&yref, nout * BYTES); hipMalloc(&idx, hidx.size() * 4);
    hipMemcpy(in, hin.data(), nin * BYTES, hipMemcpyHostToDevice);
    hipMemcpy(idx, hidx.data(), hidx.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = (double)(nin + nout) * BYTES + (double)L * L * 4;
    std::vector<unsigned char> href(nout * BYTES), hgot(nout * BYTES);
    run<BYTES>(yref, in, idx, BH, H, L, nb, 2, 1024, e0, e1);
    hipMemcpy(href.data(), yref, nout * BYTES, hipMemcpyDeviceToHost);
    auto report = [&](int, int nblk, int thr, float ms) {
        hipMemcpy(hgot.data(), y, nout * BYTES, hipMemcpyDeviceToHost);
        const bool same = memcmp(href.data(), hgot.data(), nout * BYTES) == 0;
        printf("%s ONE PLANE blocks/plane=%d threads=%4d  %7.1f us  %.2f TB/s  %s\n", tag, nblk, thr, ms * 1e3, bytes / (ms * 1e-3) / 1e12,
               same ? "same" : "DIFFERENT");
        hipMemset(y, 0xff, nout * BYTES);
    };
    for (int nblk : {2, 4}) report(0, nblk, 1024, run<BYTES>(y, in, idx, BH, H, L, nb, nblk, 1024, e0, e1));
    auto report2 = [&](int nv, int G, int thr, int nblk, float ms) {
        hipMemcpy(hgot.data(), y, nout * BYTES, hipMemcpyDeviceToHost);
        const bool same = memcmp(href.data(), hgot.data(), nout * BYTES) == 0;
        printf("%s PLANES nv=%2d G=%2d threads=%4d blocks/plane=%2d  %7.1f us  %.2f TB/s  %s\n", tag, nv, G, thr, nblk, ms * 1e3,
               bytes / (ms * 1e-3) / 1e12, same ? "same" : "DIFFERENT");
        hipMemset(y, 0xff, nout * BYTES);
    };
    int nblk;
    for (int thr : {1024, 768}) for (int G : {2, 4, 8}) {
        float ms = run_planes<BYTES, 4>(y, in, idx, BH, H, L, nb, G, thr, e0, e1, &nblk); report2(4, G, thr, nblk, ms);
        ms = run_planes<BYTES, 5>(y, in, idx, BH, H, L, nb, G, thr, e0, e1, &nblk); report2(5, G, thr, nblk, ms);
    }
    hipFree(in); hipFree(y); hipFree(yref); hipFree(idx);
}

// ---- backward (scatter): tile width CT, tiles in flight PF, rows per workgroup ------------------------------------
template <typename T, int CT, int PF>
float run_scatter(T* gin, const T* gout, const int32_t* idx, int BH, int L, int nb, int wgs_per_cu, hipEvent_t e0, hipEvent_t e1) {
    using ACC = typename cvt<T>::acc;
    constexpr int V = 16 / (int)sizeof(T);
    const size_t lds = (size_t)WAVE * (CT + V) * sizeof(T) + (size_t)nb * (WAVE + 1) * sizeof(ACC);
    const int gx = (BH + WAVE - 1) / WAVE;
    const int want_y = std::max(1, (256 * wgs_per_cu) / gx);
    const int rpb = std::max(1, (L + want_y - 1) / want_y), gy = (L + rpb - 1) / rpb;
    float sum = 0;
    for (int it = 0; it < 7; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((rpe_scatter_planes<T, CT, PF, false>), dim3(gx, gy), dim3(WAVE), lds, 0, gin, gout, idx, BH, L, L, nb, rpb);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it >= 2) sum += ms;
    }
    return sum / 5;
}

template <typename T>
void sweep_bwd(const char* tag) {
    const int B = 64, H = 12, L = 577, nb = 50, BH = B * H;
    const size_t nin = (size_t)BH * L * nb, nout = (size_t)BH * L * L;
    std::vector<T> hg(nout);
    srand(9);
    for (auto& v : hg) v = T((float)(rand() % 17 - 8) * 0.125f);
    std::vector<int32_t> hidx((size_t)L * L);
    for (auto& v : hidx) v = rand() % nb;
    T *gout, *gin; int32_t* idx;
    hipMalloc(&gout, nout * sizeof(T)); hipMalloc(&gin, nin * sizeof(T)); hipMalloc(&idx, hidx.size() * 4);
    hipMemcpy(gout, hg.data(), nout * sizeof(T), hipMemcpyHostToDevice);
    hipMemcpy(idx, hidx.data(), hidx.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = (double)(nin + nout) * sizeof(T) + (double)L * L * 4;
    std::vector<T> href(nin), hgot(nin);
    bool have_ref = false;
    auto report = [&](int ct, int pf, int w, float ms) {
        hipMemcpy(hgot.data(), gin, nin * sizeof(T), hipMemcpyDeviceToHost);
        if (!have_ref) { href = hgot; have_ref = true; }
        const bool same = memcmp(href.data(), hgot.data(), nin * sizeof(T)) == 0;
        printf("%s SCATTER CT=%3d PF=%d wgs/cu=%2d  %7.1f us  %.2f TB/s  %s\n", tag, ct, pf, w, ms * 1e3, bytes / (ms * 1e-3) / 1e12,
               same ? "same" : "DIFFERENT");
        hipMemset(gin, 0xff, nin * sizeof(T));
    };
    constexpr int C1 = 128 / (int)sizeof(T), C2 = (256 / (int)sizeof(T)) > 64 ? 64 : 256 / (int)sizeof(T);
    for (int w : {8, 6, 12, 16}) {
        report(C1, 2, w, run_scatter<T, C1, 2>(gin, gout, idx, BH, L, nb, w, e0, e1));
        report(C1, 1, w, run_scatter<T, C1, 1>(gin, gout, idx, BH, L, nb, w, e0, e1));
        report(C1, 3, w, run_scatter<T, C1, 3>(gin, gout, idx, BH, L, nb, w, e0, e1));
        report(C1, 4, w, run_scatter<T, C1, 4>(gin, gout, idx, BH, L, nb, w, e0, e1));
        if (C2 != C1) {
            report(C2, 1, w, run_scatter<T, C2, 1>(gin, gout, idx, BH, L, nb, w, e0, e1));
            report(C2, 2, w, run_scatter<T, C2, 2>(gin, gout, idx, BH, L, nb, w, e0, e1));
            report(C2, 3, w, run_scatter<T, C2, 3>(gin, gout, idx, BH, L, nb, w, e0, e1));
        }
    }
    hipFree(gout); hipFree(gin); hipFree(idx);
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "bwd")) {
        sweep_bwd<float>("fp32");
        sweep_bwd<hip_bfloat16>("bf16");
        return 0;
    }
    sweep<4>("fp32");
    sweep<2>("bf16");
    return 0;
}
