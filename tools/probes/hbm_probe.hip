// Achievable HBM rates on this MI355X for the access mixes of the rpe_index kernels:
// write-only (the gather's output stream), read-only (the scatter's input stream), copy.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/hbm_probe.hip -o tools/probes/hbm_probe && tools/probes/hbm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k_write(f4* dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = f4{1, 2, 3, 4};
}
__global__ void k_write_nt(f4* dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(f4{1, 2, 3, 4}, dst + i);
}
template <int AUX>
__global__ void k_write_buf(f4* dst, size_t n) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, 0x7fffffff, 0x00020000);
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        __builtin_amdgcn_raw_buffer_store_b128(u4{1, 2, 3, 4}, rs, (int)(i * 16), 0, AUX);
}
template <int U>
__global__ void k_write_unroll(f4* dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) dst[i + u * stride] = f4{1, 2, 3, 4};
    }
    for (; i < n; i += stride) dst[i] = f4{1, 2, 3, 4};
}
template <int U>
__global__ void k_copy_unroll(f4* dst, const f4* src, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) dst[i + u * stride] = v[u];
    }
    for (; i < n; i += stride) dst[i] = src[i];
}
// one workgroup owns a CONTIGUOUS slab (what a memset kernel does) instead of a grid-stride walk
__global__ void k_write_slab(f4* dst, size_t n) {
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t b0 = blockIdx.x * per, b1 = b0 + per < n ? b0 + per : n;
    for (size_t i = b0 + threadIdx.x; i < b1; i += blockDim.x) dst[i] = f4{1, 2, 3, 4};
}
__global__ void k_copy_nt(f4* dst, const f4* src, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
__global__ void k_read(const f4* src, size_t n, float* out) {
    f4 a = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += src[i];
    if (a[0] + a[1] + a[2] + a[3] == 12345.678f) *out = 1.f;
}
__global__ void k_copy(f4* dst, const f4* src, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int main() {
    const size_t bytes = (size_t)1 << 30, n = bytes / 16;
    f4 *a, *b; float* o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grids[] = {256 * 4, 256 * 8, 256 * 16, 256 * 32};
    for (int g : grids) {
        float ms[3];
        for (int k = 0; k < 3; ++k) {
            for (int it = 0; it < 3; ++it) {
                hipEventRecord(e0);
                if (k == 0) k_write<<<g, 256>>>(a, n);
                if (k == 1) k_read<<<g, 256>>>(a, n, o);
                if (k == 2) k_copy<<<g, 256>>>(b, a, n);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms[k], e0, e1);
            }
        }
        printf("grid %5d: write %.0f GB/s  read %.0f GB/s  copy %.0f GB/s (read+write bytes)\n", g, bytes / ms[0] / 1e6,
               bytes / ms[1] / 1e6, 2.0 * bytes / ms[2] / 1e6);
    }
    {
        const size_t nb = (size_t)1 << 30;           // buffer offsets are 32-bit: 1 GiB fits
        for (int k = 0; k < 6; ++k) {
            float ms = 0;
            for (int it = 0; it < 3; ++it) {
                hipEventRecord(e0);
                if (k == 0) k_write_unroll<4><<<2048, 256>>>(a, nb / 16);
                if (k == 1) k_write_unroll<8><<<1024, 256>>>(a, nb / 16);
                if (k == 2) k_write_slab<<<2048, 256>>>(a, nb / 16);
                if (k == 3) k_write_slab<<<8192, 256>>>(a, nb / 16);
                if (k == 4) k_copy_unroll<4><<<2048, 256>>>(b, a, nb / 16);
                if (k == 5) k_write_slab<<<65536, 256>>>(a, nb / 16);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const char* nm[] = {"write unroll4 grid2048", "write unroll8 grid1024", "write slab grid2048", "write slab grid8192",
                                "copy unroll4 (r+w bytes)", "write slab grid65536"};
            printf("%-32s %.0f GB/s\n", nm[k], (k == 4 ? 2.0 : 1.0) * nb / ms / 1e6);
        }
        for (int k = 0; k < 0; ++k) {
            float ms = 0;
            for (int it = 0; it < 3; ++it) {
                hipEventRecord(e0);
                if (k == 0) k_write_nt<<<2048, 256>>>(a, nb / 16);
                if (k == 1) k_write_buf<0><<<2048, 256>>>(a, nb / 16);
                if (k == 2) k_write_buf<2><<<2048, 256>>>(a, nb / 16);
                if (k == 3) k_write_buf<1><<<2048, 256>>>(a, nb / 16);
                if (k == 4) k_write_buf<17><<<2048, 256>>>(a, nb / 16);
                if (k == 5) k_copy_nt<<<2048, 256>>>(b, a, nb / 16);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const char* nm[] = {"nontemporal_store", "buffer_store aux=0", "buffer_store aux=2 (nt)", "buffer_store aux=1 (sc0)",
                                "buffer_store aux=17 (sc0 sc1)", "copy nt load+store (r+w bytes)"};
            printf("%-32s %.0f GB/s\n", nm[k], (k == 5 ? 2.0 : 1.0) * nb / ms / 1e6);
        }
    }
    hipEventRecord(e0); hipMemsetAsync(a, 1, bytes, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float t; hipEventElapsedTime(&t, e0, e1);
    printf("hipMemsetAsync: %.0f GB/s\n", bytes / t / 1e6);
    return 0;
}
