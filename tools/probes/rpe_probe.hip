// Development probe: the rpe_index forward (gather) kernel of config 4 (B=64, H=12, L=577, nb=50) one plane per workgroup
// against several planes per workgroup (vectors per thread NV, planes per group G, workgroup size), every variant compared bit
// for bit with the first.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Icream_amd/csrc tools/probes/rpe_probe.hip -o tools/probes/rpe_probe
#include "../../cream_amd/csrc/rpe_index.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <type_traits>

// ---- round-5 experiments (measured, not shipped: profiles/r05_rpe_gather.md) -------------------------------------------------------
namespace {
// Tiles (round 5): the fill's write pattern.  One workgroup = 256 threads = ONE contiguous 16 KB piece (1024 aligned 16-byte vectors)
// of ONE plane, workgroups in memory order, nothing carried from plane to plane.  What made this mapping slow in rounds 1-2 was the id
// stream — 4 bytes from L2 per output element, as much as the fp32 output itself.  Here the ids come from a BYTE copy of the id matrix
// (bucket ids < 256), one copy per 16-byte alignment class of the planes and shifted by the class's offset e0, so that the V ids of an
// aligned output vector are ONE aligned V-byte load: 1 byte per element from a 1.3-2.7 MB array that stays in L2.  Per workgroup:
// ids (NV loads per thread) and the chunk's <= 16 lookup rows (<= NS values per thread) are requested together, one barrier,
// V ds_read + one 16-byte store per vector.  `rpe_ids_u8` rebuilds the byte copies on every call (3 us; the ids are an input).
template <int V>
__global__ __launch_bounds__(256) void rpe_ids_u8(uint8_t* __restrict__ ids8, const int32_t* __restrict__ idx, int64_t plane, int64_t pitch,
                                                   int period)
{
    // class q: e0 = (V - (q * plane) % V) % V elements precede the first aligned vector of its planes (the output base is 16-byte
    // aligned: the launcher checks); ids8[q][s] = idx[e0 + s]
    const int q = blockIdx.y;
    const int e0 = (int)((V - (q * plane) % V) % V);
    const int64_t s = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 4;
    if (s >= pitch) return;
    uint32_t w = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t el = e0 + s + e;
        w |= (el < plane ? (uint32_t)idx[el] & 0xffu : 0u) << (8 * e);
    }
    *reinterpret_cast<uint32_t*>(ids8 + (int64_t)q * pitch + s) = w;
}

template <int BYTES, int NS>
__global__ __launch_bounds__(256) void rpe_gather_tiles(
    typename raw_elem<BYTES>::type* __restrict__ y,
    const typename raw_elem<BYTES>::type* __restrict__ in,
    const int32_t* __restrict__ idx, const uint8_t* __restrict__ ids8, int64_t pitch,
    int BH, int H, int Lq, int Lk, int nb,
    int64_t s0, int64_t s1, int64_t s2, int64_t s3,
    int period, int chunks)
{
    using E = typename raw_elem<BYTES>::type;
    using IDV = typename std::conditional<BYTES == 2, uint64_t, typename std::conditional<BYTES == 4, uint32_t, uint16_t>::type>::type;
    constexpr int V = 16 / BYTES, NT = 256, NV = 4, VPW = NT * NV;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    E* const table = reinterpret_cast<E*>(smem);
    const int tid = threadIdx.x;
    const int p = blockIdx.x / chunks, c = blockIdx.x - p * chunks;
    const int64_t plane = (int64_t)Lq * Lk;
    const int q = p % period;
    const int e0 = (int)((V - ((int64_t)q * plane) % V) % V);
    const int nvec = (int)((plane - e0) / V);
    const int k0 = c * VPW, k1 = min(nvec, k0 + VPW);
    if (c > 0 && k0 >= nvec) return;
    const bool first = c == 0, last = k1 == nvec;
    const int el0 = first ? 0 : e0 + k0 * V, el1 = last ? (int)plane : e0 + k1 * V;
    const int row_lo = el0 / Lk, tvals = ((el1 - 1) / Lk - row_lo + 1) * nb;

    // ---- requests: the chunk's lookup rows, this thread's ids
    const int b = p / H, h = p - b * H;
    const E* src = in + (int64_t)b * s0 + (int64_t)h * s1 + (int64_t)row_lo * s2;
    E stg[NS];
    if (s3 == 1 && s2 == nb) {
#pragma unroll
        for (int u = 0; u < NS; ++u) { const int t = tid + u * NT; if (t < tvals) stg[u] = src[t]; }
    } else {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int t = tid + u * NT;
            if (t < tvals) { const int r = t / nb, cc = t - r * nb; stg[u] = src[(int64_t)r * s2 + (int64_t)cc * s3]; }
        }
    }
    IDV idv[NV];
    const uint8_t* idp = ids8 + (int64_t)q * pitch;
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        const int k = k0 + tid + n * NT;
        idv[n] = k < k1 ? *reinterpret_cast<const IDV*>(idp + (int64_t)k * V) : (IDV)0;
    }
    int edge_el = -1;
    if (first && tid < e0) edge_el = tid;
    else if (last && tid >= 64 && tid - 64 < (int)(plane - e0 - (int64_t)nvec * V)) edge_el = e0 + nvec * V + (tid - 64);
    const int edge_id = edge_el >= 0 ? idx[edge_el] : 0;
#pragma unroll
    for (int u = 0; u < NS; ++u) { const int t = tid + u * NT; if (t < tvals) table[t] = stg[u]; }
    lds_barrier();

    E* out = y + (int64_t)p * plane;
    if (edge_el >= 0) out[edge_el] = table[(edge_el / Lk - row_lo) * nb + edge_id];
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        const int k = k0 + tid + n * NT;
        if (k < k1) {
            const int el = e0 + k * V, i = el / Lk, j = el - i * Lk;
            const int base = (i - row_lo) * nb;
            union { u32x4 vec; E e[V]; } pk;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int id = (int)((idv[n] >> (8 * e)) & 0xff);
                pk.e[e] = table[base + (j + e >= Lk ? nb : 0) + id];
            }
            *reinterpret_cast<u32x4*>(out + el) = pk.vec;      // 16-byte aligned
        }
    }
}

}  // namespace

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
float run_tiles(void* y, const void* in, const int32_t* idx, uint8_t* ids8, int BH, int H, int L, int nb, bool with_ids, hipEvent_t e0, hipEvent_t e1, int* wgs) {
    using E = typename raw_elem<BYTES>::type;
    constexpr int V = 16 / BYTES, VPW = 1024;
    int period = 1;
    while (((int64_t)period * L * L * BYTES) % 16) ++period;
    const int64_t plane = (int64_t)L * L, pitch = (plane + 15) / 16 * 16;
    const int nvec = (int)(plane / V), chunks = (nvec + VPW - 1) / VPW;
    const int max_rows = (VPW * V + V - 2) / L + 2;
    constexpr int NS = BYTES == 2 ? 4 : 2;
    if (max_rows * nb > NS * 256) return -1.f;
    const size_t lds = ((size_t)max_rows * nb * BYTES + 15) / 16 * 16;
    *wgs = chunks * BH;
    float sum = 0;
    for (int it = 0; it < 7; ++it) {
        hipEventRecord(e0);
        if (with_ids || it == 0)
            hipLaunchKernelGGL((rpe_ids_u8<V>), dim3((unsigned)((pitch / 4 + 255) / 256), period), dim3(256), 0, 0, ids8, idx, plane, pitch, period);
        hipLaunchKernelGGL((rpe_gather_tiles<BYTES, NS>), dim3(chunks * BH), dim3(256), lds, 0, (E*)y, (const E*)in, idx, ids8, pitch, BH, H, L, L, nb,
                           (int64_t)H * L * nb, (int64_t)L * nb, (int64_t)nb, (int64_t)1, period, chunks);
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
    auto report3 = [&](int nv, int depth, int qx, int wgs, float ms) {
        hipMemcpy(hgot.data(), y, nout * BYTES, hipMemcpyDeviceToHost);
        const bool same = memcmp(href.data(), hgot.data(), nout * BYTES) == 0;
        printf("%s FRONTIER nv=%d depth=%d planes abreast x%d  workgroups %5d  %7.1f us  %.2f TB/s  %s\n", tag, nv, depth, qx, wgs, ms * 1e3,
               bytes / (ms * 1e-3) / 1e12, same ? "same" : "DIFFERENT");
        hipMemset(y, 0xff, nout * BYTES);
    };
    {
        uint8_t* ids8; hipMalloc(&ids8, 16 * (((size_t)L * L + 15) / 16 * 16));
        int wgs;
        for (int with_ids = 0; with_ids < 2; ++with_ids) {
            const float ms = run_tiles<BYTES>(y, in, idx, ids8, BH, H, L, nb, with_ids != 0, e0, e1, &wgs);
            hipMemcpy(hgot.data(), y, nout * BYTES, hipMemcpyDeviceToHost);
            const bool same = memcmp(href.data(), hgot.data(), nout * BYTES) == 0;
            printf("%s TILES 16 KB per workgroup, byte ids %s  workgroups %6d  %7.1f us  %.2f TB/s  %s\n", tag,
                   with_ids ? "rebuilt every launch" : "built once         ", wgs, ms * 1e3, bytes / (ms * 1e-3) / 1e12, same ? "same" : "DIFFERENT");
            hipMemset(y, 0xff, nout * BYTES);
        }
        hipFree(ids8);
    }
    int nblk;
    for (int thr : {1024, 768}) for (int G : {2, 8}) {
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
