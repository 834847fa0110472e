// host_rpe_index.cpp — HOST-pointer entry points of rpe_index (C ABI).
//
// The reference module exports forward_cpu / backward_cpu next to the GPU functions
// (iRPE/DeiT-with-iRPE/rpe_ops/rpe_index.cpp:8-73, 82-124) and BASELINE config 1
// ("DeiT-tiny + iRPE single-image forward on CPU") runs through them, so the drop-in
// module has to offer them too.  This is its own implementation (row-parallel
// std::thread workers, per-row private accumulation) — it is NOT the parity checker
// (test infrastructure lives outside this package) and is never reached for device tensors.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "cream_amd.h"

namespace {

inline float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {
            int e = -1;
            do { man <<= 1; ++e; } while (!(man & 0x400u));
            bits = sign | ((uint32_t)(112 - e) << 23) | ((man & 0x3ffu) << 13);
        }
    } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
    else bits = sign | ((exp + 112) << 23) | (man << 13);
    float f; memcpy(&f, &bits, 4); return f;
}
inline uint16_t float_to_half(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
    const uint32_t absx = x & 0x7fffffffu;
    if (absx >= 0x7f800000u) return sign | (absx > 0x7f800000u ? 0x7e00u : 0x7c00u);
    if (absx >= 0x477ff000u) return sign | 0x7c00u;                      // overflow -> inf
    if (absx < 0x38800000u) {                                            // subnormal / zero
        if (absx < 0x33000000u) return sign;
        const int shift = 126 - (int)(absx >> 23);
        uint32_t man = (absx & 0x7fffffu) | 0x800000u;
        const uint32_t lsb = 1u << shift, half = lsb >> 1;
        uint32_t r = man >> shift;
        const uint32_t rem = man & (lsb - 1);
        if (rem > half || (rem == half && (r & 1))) ++r;
        return sign | (uint16_t)r;
    }
    uint32_t r = ((absx - 0x38000000u) >> 13);
    const uint32_t rem = absx & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) ++r;
    return sign | (uint16_t)r;
}
inline float bf16_to_float(uint16_t h) {
    const uint32_t bits = (uint32_t)h << 16; float f; memcpy(&f, &bits, 4); return f;
}
inline uint16_t float_to_bf16(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    if ((x & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((x >> 16) | 0x40u);
    x += 0x7fffu + ((x >> 16) & 1u);
    return (uint16_t)(x >> 16);
}

template <typename F>
void parallel_rows(int64_t rows, int64_t work_per_row, F&& fn) {
    unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    int64_t nthr = std::min<int64_t>(hw, std::max<int64_t>(1, rows * work_per_row / 200000));
    nthr = std::min<int64_t>(nthr, rows);
    if (nthr <= 1) { fn(0, rows); return; }
    std::vector<std::thread> pool;
    const int64_t per = (rows + nthr - 1) / nthr;
    for (int64_t t = 0; t < nthr; ++t) {
        const int64_t a = t * per, b = std::min(rows, a + per);
        if (a >= b) break;
        pool.emplace_back([=, &fn] { fn(a, b); });
    }
    for (auto& th : pool) th.join();
}

template <typename E>
void gather_host(E* y, const E* in, const int32_t* idx, int64_t rows, int Lq, int Lk, int nb) {
    parallel_rows(rows, Lk, [&](int64_t a, int64_t b) {
        for (int64_t r = a; r < b; ++r) {
            const E* src = in + r * nb;
            const int32_t* ir = idx + (r % Lq) * (int64_t)Lk;
            E* out = y + r * Lk;
            for (int j = 0; j < Lk; ++j) out[j] = src[ir[j]];
        }
    });
}

// ACC accumulates one row privately in ascending j, seeded with the caller's value.
template <typename E, typename ACC, typename LD, typename ST>
void scatter_host(E* gin, const E* gout, const int32_t* idx, int64_t rows, int Lq, int Lk,
                  int nb, LD ld, ST st) {
    parallel_rows(rows, Lk, [&](int64_t a, int64_t b) {
        std::vector<ACC> acc((size_t)nb);
        for (int64_t r = a; r < b; ++r) {
            E* dst = gin + r * nb;
            const E* g = gout + r * Lk;
            const int32_t* ir = idx + (r % Lq) * (int64_t)Lk;
            for (int u = 0; u < nb; ++u) acc[u] = ld(dst[u]);
            for (int j = 0; j < Lk; ++j) acc[ir[j]] += ld(g[j]);
            for (int u = 0; u < nb; ++u) dst[u] = st(acc[u]);
        }
    });
}

}  // namespace

extern "C" {

const char* cream_version(void) { return "1.2.0"; }

int cream_rpe_index_fwd_host(void* y, const void* in, const int32_t* idx, int B, int H, int Lq,
                             int Lk, int nb, int dtype) {
    if (B < 0 || H < 0 || Lq < 0 || Lk < 0 || nb < 0) return CREAM_ERR_BAD_ARG;
    const int64_t rows = (int64_t)B * H * Lq;
    if (rows * Lk == 0) return CREAM_OK;
    if (!y || !in || !idx || nb == 0) return CREAM_ERR_BAD_ARG;
    switch (dtype) {
        case CREAM_F32: gather_host((uint32_t*)y, (const uint32_t*)in, idx, rows, Lq, Lk, nb); break;
        case CREAM_F16:
        case CREAM_BF16: gather_host((uint16_t*)y, (const uint16_t*)in, idx, rows, Lq, Lk, nb); break;
        case CREAM_F64: gather_host((uint64_t*)y, (const uint64_t*)in, idx, rows, Lq, Lk, nb); break;
        default: return CREAM_ERR_BAD_DTYPE;
    }
    return CREAM_OK;
}

int cream_rpe_index_bwd_host(void* gin, const void* gout, const int32_t* idx, int B, int H,
                             int Lq, int Lk, int nb, int dtype) {
    if (B < 0 || H < 0 || Lq < 0 || Lk < 0 || nb < 0) return CREAM_ERR_BAD_ARG;
    const int64_t rows = (int64_t)B * H * Lq;
    if (rows * Lk == 0 || nb == 0) return CREAM_OK;
    if (!gin || !gout || !idx) return CREAM_ERR_BAD_ARG;
    switch (dtype) {
        case CREAM_F32:
            scatter_host<float, float>((float*)gin, (const float*)gout, idx, rows, Lq, Lk, nb,
                                       [](float x) { return x; }, [](float x) { return x; });
            break;
        case CREAM_F64:
            scatter_host<double, double>((double*)gin, (const double*)gout, idx, rows, Lq, Lk, nb,
                                         [](double x) { return x; }, [](double x) { return x; });
            break;
        case CREAM_F16:
            scatter_host<uint16_t, float>((uint16_t*)gin, (const uint16_t*)gout, idx, rows, Lq, Lk,
                                          nb, half_to_float, float_to_half);
            break;
        case CREAM_BF16:
            scatter_host<uint16_t, float>((uint16_t*)gin, (const uint16_t*)gout, idx, rows, Lq, Lk,
                                          nb, bf16_to_float, float_to_bf16);
            break;
        default: return CREAM_ERR_BAD_DTYPE;
    }
    return CREAM_OK;
}

}  // extern "C"
