// probe: do 16-byte buffer/global loads & stores honour 2-byte-aligned addresses on gfx950?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(2))) v16a2 { u32x4 v; };

__global__ void k(const uint16_t* in, uint16_t* out_buf, uint16_t* out_glob, uint16_t* st_buf, int n, int shift) {
    const int lane = threadIdx.x;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(in), 0, (n + 16) * 2, 0x00020000);
    union { u32x4 v; uint16_t e[8]; } a, b;
    a.v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + shift * 2, 0, 0);
    b.v = reinterpret_cast<const v16a2*>(in + shift + lane * 8)->v;
    for (int i = 0; i < 8; ++i) { out_buf[lane * 8 + i] = a.e[i]; out_glob[lane * 8 + i] = b.e[i]; }
    __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(st_buf + shift, 0, n * 2, 0x00020000);
    union { u32x4 v; uint16_t e[8]; } c;
    for (int i = 0; i < 8; ++i) c.e[i] = (uint16_t)(1000 + lane * 8 + i);
    __builtin_amdgcn_raw_buffer_store_b128(c.v, rs2, lane * 16, 0, 0);
}
int main() {
    const int n = 64 * 8;
    uint16_t h[n + 16], ob[n], og[n], sb[n + 16];
    for (int i = 0; i < n + 16; ++i) h[i] = i;
    uint16_t *d, *dob, *dog, *dsb;
    hipMalloc(&d, sizeof(h)); hipMalloc(&dob, sizeof(ob)); hipMalloc(&dog, sizeof(og)); hipMalloc(&dsb, sizeof(sb));
    for (int shift = 0; shift < 4; ++shift) {
        hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
        hipMemset(dsb, 0, sizeof(sb));
        k<<<1, 64>>>(d, dob, dog, dsb, n, shift);
        hipMemcpy(ob, dob, sizeof(ob), hipMemcpyDeviceToHost);
        hipMemcpy(og, dog, sizeof(og), hipMemcpyDeviceToHost);
        hipMemcpy(sb, dsb, sizeof(sb), hipMemcpyDeviceToHost);
        int bad_b = 0, bad_g = 0, bad_s = 0;
        for (int i = 0; i < n; ++i) { bad_b += ob[i] != (uint16_t)(i + shift); bad_g += og[i] != (uint16_t)(i + shift); bad_s += sb[i + shift] != (uint16_t)(1000 + i); }
        printf("shift %d elements: buffer_load bad=%d (first: got %d want %d)  global_load bad=%d  buffer_store bad=%d\n", shift, bad_b, ob[0], shift, bad_g, bad_s);
    }
    return 0;
}
