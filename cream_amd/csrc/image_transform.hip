// image_transform.hip — the uint8 -> normalised float input transform of both reference pipelines, on the device, for a whole batch.
//
// Reference: AutoFormer/lib/datasets.py:189-220 (`build_transform`):
//     eval :  Resize(int(256 / 224 * input_size), interpolation=3) -> CenterCrop(input_size) -> ToTensor -> Normalize
//     train:  timm create_transform(is_training=True, interpolation='bicubic') = RandomResizedCropAndInterpolation ->
//             RandomHorizontalFlip -> [RandAugment, host side, out of scope] -> ToTensor -> Normalize -> RandomErasing ('pixel' mode:
//             box from the host, standard-normal noise from a counter-based hash in the vertical-pass kernel)
// On the PIL images of the reference's ImageFolder every resize above is Pillow's `Image.resize(size, BICUBIC)` (third-party, not
// vendored in /root/reference).  These kernels restate Pillow's 8-bit algorithm (libImaging/Resample.c: precompute_coeffs,
// normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc, ImagingResampleVertical_8bpc; bicubic a = -0.5; 22 fixed-point bits) integer
// for integer — the coefficient doubles are evaluated on the device in the same order of IEEE operations as the C code (this library
// is built with -ffp-contract=off) — followed by torchvision's two float32 operations x / 255 and (x - mean) / std.  Byte-exact against
// Pillow itself and bit-exact against torch's CPU float ops (tests/test_image_transform_gpu.py).
//
// One image = { crop box of the decoded HWC uint8 frame (F.crop), size it is resized to (F.resize), window of the resized image that
// becomes the output (CenterCrop; the whole thing for the training crop), mirror flag }.  Three launches for the whole batch
// (the first: the fixed-point coefficient tables of both axes, once per image, into the workspace):
//   H  horizontal pass: for the box rows the window's vertical pass will read, the window's columns -> uint8 rows in the workspace.
//      A workgroup builds the fixed-point coefficient table of the output columns once (LDS, tap-major), then walks its rows R at a
//      time: the source row segments are staged in LDS as one word per pixel (aligned 4-byte loads, bytes scattered), a thread owns
//      one output column of all R rows (3 channels each).
//   V  vertical pass + ToTensor + Normalize (+ mirror): a thread owns 4 consecutive output columns x 3 channels of one output row —
//      three aligned 4-byte loads per tap — and writes three 16-byte vectors of the (B, 3, out_h, out_w) fp32 batch.
// Traffic per image of a 500 x 375 frame -> 224 x 224: 0.56 MB read, 0.25 MB intermediate written + read, 0.6 MB written.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "cream_amd.h"

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int PRECISION_BITS = 32 - 8 - 2;
constexpr int LDS_H_BYTES = 64 * 1024;              // horizontal pass: coefficient table (ksize x out_w int32) + staged row + output row
constexpr int LDS_ROW_BYTES = 14 * 1024;            // widest box: 3 bytes per pixel of a row fit this
constexpr int V_ROWS_MAX = 8;                       // output rows per workgroup of the vertical pass: 256 / (out_w / 4), at most this

typedef cream_image_desc Dev;                       // (planned: row0 / nrows / tmp_off filled by cream_image_batch_plan)

__device__ __forceinline__ double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// precompute_coeffs for ONE output index of an axis (in0 = 0, in1 = in_size)
struct Axis {
    double scale, support, ss;
    int in_size;
    __host__ __device__ Axis(int in, int out) : in_size(in) {
        scale = (double)in / (double)out;
        const double filterscale = scale < 1.0 ? 1.0 : scale;
        support = 2.0 * filterscale;
        ss = 1.0 / filterscale;
    }
    __host__ __device__ int ksize() const {
        int c = (int)support;
        if ((double)c < support) ++c;                // ceil
        return c * 2 + 1;
    }
    __host__ __device__ void bounds(int xx, int& xmin, int& cnt, double& center) const {
        center = 0.0 + (xx + 0.5) * scale;
        xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        cnt = xmax - xmin;
    }
    __device__ double weight(int x, int xmin, double center) const { return bicubic_filter((x + xmin - center + 0.5) * ss); }
    __device__ double norm(int xmin, int cnt, double center) const {
        double ww = 0.0;
        for (int x = 0; x < cnt; ++x) ww += weight(x, xmin, center);
        return ww;
    }
    __device__ int fixed(int x, int xmin, double center, double ww) const { return fixed_of(weight(x, xmin, center), ww); }
    static __device__ int fixed_of(double w, double ww) {
        if (ww != 0.0) w /= ww;
        return w < 0 ? (int)(-0.5 + w * (double)(1 << PRECISION_BITS)) : (int)(0.5 + w * (double)(1 << PRECISION_BITS));
    }
};

// acc + v * k for a pixel value v < 2^8: one full-rate 24-bit multiply-add when |k| < 2^23 (every coefficient of an ordinary
// resize: the normalised weights stay below 2), the 32-bit product (quarter rate) otherwise — the table builders say which
template <bool WIDE> __device__ __forceinline__ int madk(int acc, int v, int k) {
    if constexpr (WIDE) return acc + v * k;
    else return acc + __mul24(v, k);
}
__device__ __forceinline__ bool needs_wide(int k) { return k >= (1 << 23) || k <= -(1 << 23); }

// standard-normal noise of RandomErasing's 'pixel' mode: two 32-bit counter-based hashes of (seed, channel, row, column) through
// Box-Muller (autoformer/data.py: erase_noise_reference restates it)
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float erase_noise(uint32_t seed, int c, int y, int x) {
    const uint32_t key = mix32(seed ^ ((uint32_t)(c + 1) * 0x9E3779B9u));
    const uint32_t h1 = mix32(key ^ (((uint32_t)y << 16) | (uint32_t)x));
    const uint32_t h2 = mix32(h1 ^ 0x85EBCA6Bu);
    const float u1 = ((float)(h1 >> 8) + 1.0f) * (1.0f / 16777216.0f);          // (0, 1]
    const float u2 = (float)(h2 >> 8) * (1.0f / 16777216.0f);                   // [0, 1)
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

__device__ __forceinline__ int clip8(int v) {
    v >>= PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// ---- coefficient tables ----------------------------------------------------------------------------------------------------------
// Per image and axis, ONCE (the resampling kernels read them through L2: a workgroup of the horizontal pass copies its image's
// column table into LDS, the vertical pass reads a row's coefficients as wave-uniform loads).  Region of image b in the workspace
// (ints):  xk [ksx][out_w] | x0 [out_w] | xcnt [out_w] | yk [out_h][ksy] | y0 [out_h] | ycnt [out_h] | wide_x, wide_y, 0, 0
// with ksx / ksy the batch's largest tap counts (taps beyond an index's own count are 0).
struct TabLayout {
    int ksx, ksy, out_h, out_w;
    int64_t stride;                                                 // ints per image (multiple of 4)
    __host__ __device__ TabLayout(int ksx_, int ksy_, int oh, int ow) : ksx(ksx_), ksy(ksy_), out_h(oh), out_w(ow) {
        const int64_t n = (int64_t)ksx * ow + 2 * ow + (int64_t)oh * ksy + 2 * oh + 4;
        stride = (n + 3) & ~(int64_t)3;
    }
    __host__ __device__ int64_t xk() const { return 0; }
    __host__ __device__ int64_t x0() const { return (int64_t)ksx * out_w; }
    __host__ __device__ int64_t xcnt() const { return x0() + out_w; }
    __host__ __device__ int64_t yk() const { return xcnt() + out_w; }
    __host__ __device__ int64_t y0() const { return yk() + (int64_t)out_h * ksy; }
    __host__ __device__ int64_t ycnt() const { return y0() + out_h; }
    __host__ __device__ int64_t meta() const { return ycnt() + out_h; }
};

__global__ __launch_bounds__(256) void image_coeff_kernel(int* __restrict__ tabs, const Dev* __restrict__ descs, int out_h, int out_w,
                                                          int ksx, int ksy)
{
    const TabLayout L(ksx, ksy, out_h, out_w);
    const Dev d = descs[blockIdx.y];
    int* t = tabs + (int64_t)blockIdx.y * L.stride;
    int wide_l = 0;
    if (blockIdx.x == 0) {                                          // columns
        const Axis ax(d.box_w, d.resized_w);
        for (int xo = threadIdx.x; xo < out_w; xo += blockDim.x) {
            int xmin, cnt;
            double center;
            ax.bounds(d.win_left + xo, xmin, cnt, center);
            const double ww = ax.norm(xmin, cnt, center);
            for (int x = 0; x < ksx; ++x) {
                const int k = x < cnt ? ax.fixed(x, xmin, center, ww) : 0;
                t[L.xk() + (int64_t)x * out_w + xo] = k;
                wide_l |= needs_wide(k);
            }
            t[L.x0() + xo] = xmin;
            t[L.xcnt() + xo] = cnt;
        }
    } else {                                                        // rows
        const Axis ay(d.box_h, d.resized_h);
        for (int yo = threadIdx.x; yo < out_h; yo += blockDim.x) {
            int ymin, cnt;
            double center;
            ay.bounds(d.win_top + yo, ymin, cnt, center);
            const double ww = ay.norm(ymin, cnt, center);
            for (int y = 0; y < ksy; ++y) {
                const int k = y < cnt ? ay.fixed(y, ymin, center, ww) : 0;
                t[L.yk() + (int64_t)yo * ksy + y] = k;
                wide_l |= needs_wide(k);
            }
            t[L.y0() + yo] = ymin - d.row0;                         // row of the intermediate
            t[L.ycnt() + yo] = cnt;
        }
    }
    const int wide = __syncthreads_or(wide_l);
    if (threadIdx.x == 0) t[L.meta() + blockIdx.x] = wide != 0;
}

// ---- H ---------------------------------------------------------------------------------------------------------------------------
// R box rows at a time: the rows are staged as one aligned 4-byte word per pixel (R | G << 8 | B << 16, scattered byte-wise from
// aligned 4-byte global loads), a thread owns one output column of ALL R rows — a tap is one coefficient read and R pixel reads
// for 3 R multiply-adds — and the R output rows leave through LDS as whole words.  ITER groups per workgroup amortise the table.
constexpr int H_ITER = 4;
template <int R>
__global__ __launch_bounds__(256) void image_resample_h_kernel(uint8_t* __restrict__ tmp, const uint8_t* __restrict__ pixels,
                                                               const Dev* __restrict__ descs, const int* __restrict__ tabs, int out_h,
                                                               int out_w, int ks_max, int ksy, int box_w_max)
{
    // LDS sized by the launch for the batch's largest table / widest box (typically 25-40 KB: four to six workgroups per CU)
    extern __shared__ __attribute__((aligned(16))) int hlds[];
    int* ktab = hlds;                                                            // [tap][out_w]
    uint32_t* pix = reinterpret_cast<uint32_t*>(ktab + ks_max * out_w);          // [R][box_w_max]
    uint32_t* outbuf = pix + R * box_w_max;                                      // [R][out_w * 3 / 4]
    unsigned short* x0s = reinterpret_cast<unsigned short*>(outbuf + R * ((out_w * 3) / 4));
    unsigned char* cnts = reinterpret_cast<unsigned char*>(x0s + out_w);

    const Dev d = descs[blockIdx.y];
    const int first = blockIdx.x * (R * H_ITER);
    if (first >= d.nrows) return;
    const TabLayout L(ks_max, ksy, out_h, out_w);
    const int* t = tabs + (int64_t)blockIdx.y * L.stride;
    for (int i = threadIdx.x; i < (ks_max * out_w) >> 2; i += blockDim.x)           // the image's column table: L2 -> LDS
        reinterpret_cast<u32x4*>(ktab)[i] = reinterpret_cast<const u32x4*>(t + L.xk())[i];
    for (int xo = threadIdx.x; xo < out_w; xo += blockDim.x) {
        x0s[xo] = (unsigned short)t[L.x0() + xo];
        cnts[xo] = (unsigned char)t[L.xcnt() + xo];
    }
    const bool wide = t[L.meta()] != 0;
    __syncthreads();
    const int seg_bytes = d.box_w * 3;
    const int last = min(first + R * H_ITER, d.nrows);
    const int ow_dwords = (out_w * 3) >> 2;
    for (int r0 = first; r0 < last; r0 += R) {
        const int nr = min(R, last - r0);
        __syncthreads();                                            // the previous group's readers are done with pix / outbuf
        {
            // row segment [box_left, box_left + box_w) of source rows box_top + row0 + r0 + rr.  A thread takes the 12 bytes of four
            // consecutive pixels: four aligned 4-byte loads (the row starts at any byte), three byte-alignments, one 16-byte LDS
            // store of the four pixel words; the R rows' requests of a step are in flight together.
            const int64_t a00 = d.offset + (int64_t)(d.box_top + d.row0 + r0) * d.row_stride + (int64_t)d.box_left * 3;
            const int ngroups = (d.box_w + 3) >> 2;
            for (int gi = threadIdx.x; gi < ngroups; gi += blockDim.x) {
                uint32_t v[R][4];
                int sh[R];
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
                    const int64_t a0 = a00 + (int64_t)rr * d.row_stride;
                    sh[rr] = (int)(a0 & 3);
                    const int nd = (sh[rr] + seg_bytes + 3) >> 2;             // words that hold bytes of this row segment
                    const uint32_t* src = reinterpret_cast<const uint32_t*>(pixels + (a0 - sh[rr])) + 3 * gi;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[rr][e] = (rr < nr && 3 * gi + e < nd) ? src[e] : 0u;
                }
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
                    uint32_t w[3];
#pragma unroll
                    for (int e = 0; e < 3; ++e)                                // bytes [4e + sh, 4e + sh + 4) of the 16 loaded
                        w[e] = sh[rr] ? (v[rr][e] >> (8 * sh[rr])) | (v[rr][e + 1] << (32 - 8 * sh[rr])) : v[rr][e];
                    u32x4 q;
                    q[0] = w[0] & 0xFFFFFFu;
                    q[1] = (w[0] >> 24) | ((w[1] & 0xFFFFu) << 8);
                    q[2] = (w[1] >> 16) | ((w[2] & 0xFFu) << 16);
                    q[3] = w[2] >> 8;
                    *reinterpret_cast<u32x4*>(pix + rr * box_w_max + 4 * gi) = q;
                }
            }
        }
        __syncthreads();
        auto row_pass = [&](auto W) {
            constexpr bool WIDE = decltype(W)::value;
            for (int xo = threadIdx.x; xo < out_w; xo += blockDim.x) {
                const int cnt = cnts[xo];
                int acc[R][3];
#pragma unroll
                for (int rr = 0; rr < R; ++rr) acc[rr][0] = acc[rr][1] = acc[rr][2] = 1 << (PRECISION_BITS - 1);
                const uint32_t* p = pix + x0s[xo];
                const int* kc = ktab + xo;
                for (int x = 0; x < cnt; ++x) {
                    const int k = kc[x * out_w];
#pragma unroll
                    for (int rr = 0; rr < R; ++rr) {
                        const uint32_t v = p[rr * box_w_max + x];
                        acc[rr][0] = madk<WIDE>(acc[rr][0], (int)(v & 255u), k);
                        acc[rr][1] = madk<WIDE>(acc[rr][1], (int)((v >> 8) & 255u), k);
                        acc[rr][2] = madk<WIDE>(acc[rr][2], (int)((v >> 16) & 255u), k);
                    }
                }
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
                    uint8_t* ob = reinterpret_cast<uint8_t*>(outbuf + rr * ow_dwords) + 3 * xo;
                    ob[0] = (uint8_t)clip8(acc[rr][0]);
                    ob[1] = (uint8_t)clip8(acc[rr][1]);
                    ob[2] = (uint8_t)clip8(acc[rr][2]);
                }
            }
        };
        if (wide) row_pass(std::true_type{});
        else row_pass(std::false_type{});
        __syncthreads();
        uint32_t* dst = reinterpret_cast<uint32_t*>(tmp + d.tmp_off + (int64_t)r0 * out_w * 3);     // the nr rows are contiguous
        for (int i = threadIdx.x; i < nr * ow_dwords; i += blockDim.x) dst[i] = outbuf[i];
    }
}

// ---- V + ToTensor + Normalize ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void image_resample_v_kernel(float* __restrict__ out, const uint8_t* __restrict__ tmp,
                                                               const Dev* __restrict__ descs, const int* __restrict__ tabs, int out_h,
                                                               int out_w, int V_ROWS, int ksx, int ksy, float m0, float m1, float m2,
                                                               float s0, float s1, float s2)
{
    const Dev d = descs[blockIdx.y];
    const TabLayout L(ksx, ksy, out_h, out_w);
    const int* t = tabs + (int64_t)blockIdx.y * L.stride;
    const bool wide = t[L.meta() + 1] != 0;
    const int yo0 = blockIdx.x * V_ROWS;
    const int groups = out_w >> 2;                                  // 4 output columns per thread
    const int j = threadIdx.x / groups, gx = threadIdx.x - j * groups;
    if (j >= V_ROWS || yo0 + j >= out_h) return;
    const int row_bytes = out_w * 3;
    const int yo = yo0 + j;
    const int* ky = t + L.yk() + (int64_t)yo * ksy;                 // this row's coefficients: the same address for the row's threads
    const uint8_t* base = tmp + d.tmp_off + (int64_t)t[L.y0() + yo] * row_bytes + gx * 12;
    int acc[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) acc[e] = 1 << (PRECISION_BITS - 1);
    const int cnt = t[L.ycnt() + yo];
    auto col_pass = [&](auto W) {
        constexpr bool WIDE = decltype(W)::value;
#pragma unroll 4
        for (int y = 0; y < cnt; ++y) {
            const int k = ky[y];
            const uint32_t* p = reinterpret_cast<const uint32_t*>(base + (int64_t)y * row_bytes);
            const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[e] = madk<WIDE>(acc[e], (int)((w0 >> (8 * e)) & 255u), k);
                acc[4 + e] = madk<WIDE>(acc[4 + e], (int)((w1 >> (8 * e)) & 255u), k);
                acc[8 + e] = madk<WIDE>(acc[8 + e], (int)((w2 >> (8 * e)) & 255u), k);
            }
        }
    };
    if (wide) col_pass(std::true_type{});
    else col_pass(std::false_type{});
    // torchvision F.to_tensor: float(v) / 255; F.normalize: (x - mean) / std — two IEEE float32 divisions, no contraction
    const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
    float* ob = out + (int64_t)blockIdx.y * 3 * out_h * out_w + (int64_t)yo * out_w;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        f32x4 v;
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            const float x = (float)clip8(acc[3 * px + c]) / 255.f;
            v[d.flip ? 3 - px : px] = (x - mean[c]) / sd[c];
        }
        const int xo = d.flip ? out_w - 4 - 4 * gx : 4 * gx;
        if (d.erase_h > 0 && yo >= d.erase_top && yo < d.erase_top + d.erase_h) {      // RandomErasing, mode 'pixel'
#pragma unroll
            for (int px = 0; px < 4; ++px)
                if (xo + px >= d.erase_left && xo + px < d.erase_left + d.erase_w) v[px] = erase_noise(d.erase_seed, c, yo, xo + px);
        }
        *reinterpret_cast<f32x4*>(ob + (int64_t)c * out_h * out_w + xo) = v;
    }
}

// rows of the box that the vertical pass of the window reads: [first ymin, last ymin + cnt)
void window_rows(const cream_image_desc& d, int out_h, int& row0, int& nrows) {
    const Axis ay(d.box_h, d.resized_h);
    int ymin, cnt, ymin2, cnt2;
    double c;
    ay.bounds(d.win_top, ymin, cnt, c);
    ay.bounds(d.win_top + out_h - 1, ymin2, cnt2, c);
    row0 = ymin;
    nrows = ymin2 + cnt2 - ymin;
}

int pad4(int v) { return (v + 3) & ~3; }
int64_t h_lds_bytes(int ks, int box_w, int out_w, int R = 1) {
    return (int64_t)ks * out_w * 4 + (int64_t)R * pad4(box_w) * 4 + (int64_t)R * out_w * 3 + (int64_t)out_w * 2 + out_w;
}
template <int R>
int launch_h(uint8_t* tmp, const uint8_t* pixels, const cream_image_desc* dd, const int* tabs, int B, int out_h, int out_w, int max_rows,
             int ks, int ksy, int box_w, hipStream_t st) {
    const int per_wg = R * H_ITER;
    hipLaunchKernelGGL(image_resample_h_kernel<R>, dim3((max_rows + per_wg - 1) / per_wg, B), dim3(256),
                       (size_t)h_lds_bytes(ks, box_w, out_w, R), st, tmp, pixels, dd, tabs, out_h, out_w, ks, ksy, pad4(box_w));
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

int v_rows(int out_w) {
    const int r = 256 / (out_w / 4);
    return r > V_ROWS_MAX ? V_ROWS_MAX : r;
}

int check(const cream_image_desc& d, int out_h, int out_w, int64_t pixels_bytes) {
    if (d.height <= 0 || d.width <= 0 || d.row_stride < 3 * d.width || d.offset < 0) return CREAM_ERR_BAD_ARG;
    if (d.box_h <= 0 || d.box_w <= 0 || d.box_top < 0 || d.box_left < 0 || d.box_top + d.box_h > d.height ||
        d.box_left + d.box_w > d.width)
        return CREAM_ERR_BAD_ARG;
    if (d.resized_h <= 0 || d.resized_w <= 0 || d.win_top < 0 || d.win_left < 0 || d.win_top + out_h > d.resized_h ||
        d.win_left + out_w > d.resized_w)
        return CREAM_ERR_BAD_ARG;
    if (d.offset + (int64_t)(d.height - 1) * d.row_stride + 3 * (int64_t)d.width > pixels_bytes) return CREAM_ERR_BAD_ARG;
    if (d.erase_h < 0 || d.erase_w < 0 || (d.erase_h > 0 && (d.erase_top < 0 || d.erase_left < 0 || d.erase_top + d.erase_h > out_h ||
                                                              d.erase_left + d.erase_w > out_w)))
        return CREAM_ERR_BAD_ARG;
    const Axis ax(d.box_w, d.resized_w), ay(d.box_h, d.resized_h);
    if (ax.ksize() > 255 || d.box_w * 3 + 6 > LDS_ROW_BYTES || d.box_w > 65535 || h_lds_bytes(ax.ksize(), d.box_w, out_w) > LDS_H_BYTES)
        return CREAM_ERR_TOO_LARGE;
    if (ay.ksize() > 4096) return CREAM_ERR_TOO_LARGE;
    return CREAM_OK;
}

int64_t align16(int64_t v) { return (v + 15) & ~(int64_t)15; }
bool shape_ok(int B, int out_h, int out_w) { return B > 0 && out_h > 0 && out_w >= 4 && out_w % 4 == 0 && out_w <= 1024; }

// What a batch needs beyond the descriptors themselves.  `fill`: write row0 / nrows / tmp_off (plan); otherwise they must match.
struct Batch {
    int max_rows = 0, ksx = 0, ksy = 0, box_w = 0;
    int64_t tab_bytes = 0, tmp_bytes = 0;
};
int survey(cream_image_desc* descs, const cream_image_desc* cdescs, int B, int out_h, int out_w, int64_t pixels_bytes, Batch& bt) {
    int64_t off = 0;
    for (int b = 0; b < B; ++b) {
        const cream_image_desc& d = cdescs[b];
        const int rc = check(d, out_h, out_w, pixels_bytes);
        if (rc != CREAM_OK) return rc;
        int row0, nrows;
        window_rows(d, out_h, row0, nrows);
        if (descs) { descs[b].row0 = row0; descs[b].nrows = nrows; descs[b].tmp_off = off; }
        else if (d.row0 != row0 || d.nrows != nrows || d.tmp_off != off) return CREAM_ERR_BAD_ARG;     // a stale or hand-made plan
        off += align16((int64_t)nrows * out_w * 3);
        if (nrows > bt.max_rows) bt.max_rows = nrows;
        const int ksy = Axis(d.box_h, d.resized_h).ksize(), ksx = Axis(d.box_w, d.resized_w).ksize();
        if (ksy > bt.ksy) bt.ksy = ksy;
        if (ksx > bt.ksx) bt.ksx = ksx;
        if (d.box_w > bt.box_w) bt.box_w = d.box_w;
    }
    if (h_lds_bytes(bt.ksx, bt.box_w, out_w) > LDS_H_BYTES) return CREAM_ERR_TOO_LARGE;      // (largest table and widest box in ONE batch)
    bt.tab_bytes = align16(TabLayout(bt.ksx, bt.ksy, out_h, out_w).stride * 4 * B);
    bt.tmp_bytes = off;
    return CREAM_OK;
}
}  // namespace

extern "C" int64_t cream_image_batch_plan(cream_image_desc* descs, int B, int out_h, int out_w)
{
    if (!descs || !shape_ok(B, out_h, out_w)) return CREAM_ERR_BAD_ARG;
    Batch bt;
    const int rc = survey(descs, descs, B, out_h, out_w, INT64_MAX, bt);
    return rc != CREAM_OK ? rc : bt.tab_bytes + bt.tmp_bytes;        // [coefficient tables of the B images | intermediate rows]
}

extern "C" int cream_image_batch_transform(float* out, const uint8_t* pixels, int64_t pixels_bytes, const cream_image_desc* descs,
                                           const cream_image_desc* descs_dev, int B, int out_h, int out_w, const float* mean,
                                           const float* stdev, void* workspace, int64_t workspace_bytes, void* stream)
{
    if (B == 0) return CREAM_OK;
    if (!out || !pixels || !descs || !descs_dev || !mean || !stdev || !workspace || !shape_ok(B, out_h, out_w)) return CREAM_ERR_BAD_ARG;
    if (((uintptr_t)out) % 16 || ((uintptr_t)pixels) % 4 || ((uintptr_t)workspace) % 16 || ((uintptr_t)descs_dev) % 8 || pixels_bytes % 4)
        return CREAM_ERR_BAD_ARG;
    Batch bt;
    int rc = survey(nullptr, descs, B, out_h, out_w, pixels_bytes, bt);
    if (rc != CREAM_OK) return rc;
    if (workspace_bytes < bt.tab_bytes + bt.tmp_bytes) return CREAM_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    int* tabs = reinterpret_cast<int*>(workspace);
    uint8_t* tmp = reinterpret_cast<uint8_t*>(workspace) + bt.tab_bytes;
    // 1. the fixed-point coefficient tables of both axes, once per image
    hipLaunchKernelGGL(image_coeff_kernel, dim3(2, B), dim3(256), 0, st, tabs, descs_dev, out_h, out_w, bt.ksx, bt.ksy);
    if (hipGetLastError() != hipSuccess) return CREAM_ERR_LAUNCH;
    // 2. horizontal pass; rows per step: eight where the batch's largest table and widest box leave room for them
    if (h_lds_bytes(bt.ksx, bt.box_w, out_w, 8) <= 40 * 1024)
        rc = launch_h<8>(tmp, pixels, descs_dev, tabs, B, out_h, out_w, bt.max_rows, bt.ksx, bt.ksy, bt.box_w, st);
    else if (h_lds_bytes(bt.ksx, bt.box_w, out_w, 4) <= LDS_H_BYTES)
        rc = launch_h<4>(tmp, pixels, descs_dev, tabs, B, out_h, out_w, bt.max_rows, bt.ksx, bt.ksy, bt.box_w, st);
    else if (h_lds_bytes(bt.ksx, bt.box_w, out_w, 2) <= LDS_H_BYTES)
        rc = launch_h<2>(tmp, pixels, descs_dev, tabs, B, out_h, out_w, bt.max_rows, bt.ksx, bt.ksy, bt.box_w, st);
    else rc = launch_h<1>(tmp, pixels, descs_dev, tabs, B, out_h, out_w, bt.max_rows, bt.ksx, bt.ksy, bt.box_w, st);
    if (rc != CREAM_OK) return rc;
    // 3. vertical pass + ToTensor + Normalize (+ mirror, RandomErasing)
    const int vr = v_rows(out_w);
    hipLaunchKernelGGL(image_resample_v_kernel, dim3((out_h + vr - 1) / vr, B), dim3(256), 0, st, out, tmp, descs_dev, tabs, out_h, out_w,
                       vr, bt.ksx, bt.ksy, mean[0], mean[1], mean[2], stdev[0], stdev[1], stdev[2]);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}
