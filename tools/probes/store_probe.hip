// Development probe: which workgroup -> address mappings of a pure 16-byte store stream reach the fill rate on the
// rpe_index output of config 4 (64 x 12 planes of 577 x 577 fp32 = 1.02 GB)?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/store_probe.hip -o tools/probes/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k_fill(f4* dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = f4{1, 2, 3, 4};
}
// every workgroup streams through its own contiguous slab of `per` vectors; NT: nontemporal
template <bool NT>
__global__ void k_slab(f4* dst, size_t n, size_t per) {
    const size_t b0 = blockIdx.x * per, b1 = b0 + per < n ? b0 + per : n;
    for (size_t i = b0 + threadIdx.x; i < b1; i += blockDim.x) {
        if (NT) __builtin_nontemporal_store(f4{1, 2, 3, 4}, dst + i);
        else dst[i] = f4{1, 2, 3, 4};
    }
}
// clusters of C workgroups share a region of C * per vectors and interleave chunks of blockDim.x vectors
__global__ void k_interleave(f4* dst, size_t n, size_t per, int C) {
    const size_t cl = blockIdx.x / C, c = blockIdx.x % C;
    const size_t r0 = cl * per * C, r1 = r0 + per * C < n ? r0 + per * C : n;
    for (size_t i = r0 + c * blockDim.x + threadIdx.x; i < r1; i += (size_t)C * blockDim.x) dst[i] = f4{1, 2, 3, 4};
}
// slab walk with U independent stores per thread and iteration
template <int U>
__global__ void k_slab_unroll(f4* dst, size_t n, size_t per) {
    const size_t b0 = blockIdx.x * per, b1 = b0 + per < n ? b0 + per : n;
    size_t i = b0 + threadIdx.x;
    for (; i + (U - 1) * blockDim.x < b1; i += U * blockDim.x) {
#pragma unroll
        for (int u = 0; u < U; ++u) dst[i + u * blockDim.x] = f4{1, 2, 3, 4};
    }
    for (; i < b1; i += blockDim.x) dst[i] = f4{1, 2, 3, 4};
}

int main() {
    const size_t elems = (size_t)64 * 12 * 577 * 577, n = elems / 4;
    f4* a; hipMalloc(&a, n * 16 + 4096); hipMemset(a, 0, n * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto launch) {
        float best = 1e9f, sum = 0;
        for (int it = 0; it < 7; ++it) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (it >= 2) { best = ms < best ? ms : best; sum += ms; }
        }
        printf("%-44s avg %7.1f us  best %7.1f us  %.2f TB/s\n", name, sum / 5 * 1e3, best * 1e3, n * 16.0 / (sum / 5 * 1e-3) / 1e12);
    };
    char nm[96];
    for (int g : {1024, 2048, 4096, 8192}) { snprintf(nm, 96, "fill grid-stride 256thr grid=%d", g); timeit(nm, [&] { k_fill<<<g, 256>>>(a, n); }); }
    for (int thr : {256, 1024}) for (size_t kb : {16, 64, 256, 692, 1331}) {
        const size_t per = kb * 1024 / 16; const int g = (int)((n + per - 1) / per);
        snprintf(nm, 96, "slab %4zu KB  %4d thr  grid=%d", kb, thr, g); timeit(nm, [&] { k_slab<false><<<g, thr>>>(a, n, per); });
    }
    { const size_t per = 692 * 1024 / 16; const int g = (int)((n + per - 1) / per);
      timeit("slab 692 KB 1024 thr nontemporal", [&] { k_slab<true><<<g, 1024>>>(a, n, per); });
      timeit("slab 692 KB 1024 thr unroll 2", [&] { k_slab_unroll<2><<<g, 1024>>>(a, n, per); });
      timeit("slab 692 KB 1024 thr unroll 4", [&] { k_slab_unroll<4><<<g, 1024>>>(a, n, per); });
      for (int C : {2, 4, 8, 16}) { snprintf(nm, 96, "interleave 692 KB x C=%d 1024 thr", C); timeit(nm, [&] { k_interleave<<<g, 1024>>>(a, n, per, C); }); }
      for (int thr : {256, 512}) { snprintf(nm, 96, "slab 692 KB %d thr", thr); timeit(nm, [&] { k_slab<false><<<g, thr>>>(a, n, per); }); }
    }
    return 0;
}
