// block_ops.hip — the HBM-bound passes of one supernet transformer block (gfx950), fused so
// that every activation crosses HBM once per direction.
//
// Reference semantics: AutoFormer/model/supernet_transformer.py:251-287 (pre-norm block):
//     x1 = x  + drop_path(proj(attn(LN1(x))))          LayerNormSuper: layernorm_super.py:26-37
//     x2 = x1 + drop_path(fc2(gelu(fc1(LN2(x1)))))     gelu in fp32: supernet_transformer.py:14-16
// The residual stream stays fp32 (as under the reference's autocast, where LayerNorm and the
// residual adds run in fp32), GEMM operands are bf16.  Kernels here:
//     ln_fwd         x(f32) -> LN(x)(bf16) + (mean, rstd)
//     ln_bwd         dy(bf16), x, stats, gamma, dres(f32) -> dx = dres + LN'(dy)  (f32),
//                    optionally also s_b * dx as bf16 (the gradient entering the previous
//                    branch: drop-path scale fused), per-workgroup partials of dgamma/dbeta
//     gelu_fwd/bwd   exact (erf) GELU evaluated in fp32 on bf16 storage
//     residual_add   x + s_b * y   (y bf16 branch output, s_b per-sample drop-path scale)
//     scale_cast     s_b * dx -> bf16
//     colsum         column sums of a bf16 (M, C) matrix in fp32 (bias gradients), two-stage,
//                    fixed order
// All are one-pass, 16-byte vectorised, no atomics (deterministic).  Rows are contiguous
// (leading dimension = the sampled width E: activations live in our own buffers).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "attn_common.hpp"
#include "cream_amd.h"
#include "launch_ev.hpp"

namespace {
using namespace cream;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

__device__ __forceinline__ void unpack_bf16x4(u32x2v v, float (&f)[4]) {
    f[0] = __uint_as_float(v[0] << 16); f[1] = __uint_as_float(v[0] & 0xFFFF0000u);
    f[2] = __uint_as_float(v[1] << 16); f[3] = __uint_as_float(v[1] & 0xFFFF0000u);
}
__device__ __forceinline__ u32x2v pack_bf16x4(const float (&f)[4]) {
    return u32x2v{f2bf_pair(f[0], f[1]), f2bf_pair(f[2], f[3])};
}

constexpr int LN_MAXC = 5;          // float4 chunks per lane: E <= 64 * 4 * 5 = 1280

// The LayerNorm kernels exist in two instantiations of ONE code path: 16-bit side (uint16_t = bf16 bits: the GEMM-operand
// side of the throughput mode) and fp32 side (float: the parity mode, fp32 in and out like the reference's
// layernorm_super.py:33-37 outside autocast).  Only the pack / unpack of the 4-element chunks differs.
template <typename T> struct Io4;
template <> struct Io4<uint16_t> {
    using raw = u32x2v;
    static __device__ __forceinline__ raw zero() { return u32x2v{0, 0}; }
    static __device__ __forceinline__ raw load(const uint16_t* p) { return *reinterpret_cast<const u32x2v*>(p); }
    static __device__ __forceinline__ void unpack(raw v, float (&f)[4]) { unpack_bf16x4(v, f); }
    // stores the rounded values and returns them (callers that sum what they wrote need the rounding)
    static __device__ __forceinline__ void store(uint16_t* p, float (&f)[4]) {
        const u32x2v pk = pack_bf16x4(f);
        *reinterpret_cast<u32x2v*>(p) = pk;
        unpack_bf16x4(pk, f);
    }
};
template <> struct Io4<float> {
    using raw = f32x4v;
    static __device__ __forceinline__ raw zero() { return f32x4v{0, 0, 0, 0}; }
    static __device__ __forceinline__ raw load(const float* p) { return *reinterpret_cast<const f32x4v*>(p); }
    static __device__ __forceinline__ void unpack(raw v, float (&f)[4]) { f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3]; }
    static __device__ __forceinline__ void store(float* p, float (&f)[4]) { *reinterpret_cast<f32x4v*>(p) = f32x4v{f[0], f[1], f[2], f[3]}; }
};

// ---- LayerNorm forward: one wave per row ------------------------------------------------
// With a branch output `res` (bf16) the kernel first forms the new residual stream
// x1 = x + s_b * res (written to `xsum`, fp32) and normalises that: the residual add and the
// next LayerNorm of the block in one pass.
template <typename TY>
__global__ __launch_bounds__(256) void ln_fwd_kernel(TY* __restrict__ y, float* __restrict__ mean,
                                                     float* __restrict__ rstd, const float* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     int M, int E, float eps, float* __restrict__ xsum,
                                                     const uint16_t* __restrict__ res, const float* __restrict__ sscale,
                                                     int rows_per_sample) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // scalar: row bases in SGPRs
    if (row >= M) return;
    const int nch = E >> 2;
    const float* xr = x + (int64_t)row * E;
    f32x4v v[LN_MAXC];
    float s = 0.f;
    const float sc = (res && sscale) ? sscale[row / rows_per_sample] : 1.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        // the fp32 residual stream is read and written once per pass: non-temporal, so that what the L2 keeps is the bf16 output
        // the next GEMM reads (profiles/r05_nt_out_stores.md)
        v[i] = c < nch ? __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(xr + 4 * c)) : f32x4v{0, 0, 0, 0};
        if (res && c < nch) {
            float r[4];
            unpack_bf16x4(__builtin_nontemporal_load(reinterpret_cast<const u32x2v*>(res + (int64_t)row * E + 4 * c)), r);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][e] += sc * r[e];
            __builtin_nontemporal_store(v[i], reinterpret_cast<f32x4v*>(xsum + (int64_t)row * E + 4 * c));
        }
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mu = wave_sum(s) / (float)E;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        if (lane + 64 * i < nch) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mu; q += d * d; }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)E + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    TY* yr = y + (int64_t)row * E;
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            const f32x4v g = *reinterpret_cast<const f32x4v*>(gamma + 4 * c);
            const f32x4v b = *reinterpret_cast<const f32x4v*>(beta + 4 * c);
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mu) * rs * g[e] + b[e];
            Io4<TY>::store(yr + 4 * c, o);
        }
    }
}

// ---- LayerNorm backward (+ residual gradient, + optional scaled bf16 copy) ------------
// grid = P workgroups of 4 waves; wave w of workgroup p walks rows (p*4 + w), += 4P, ...
// partial[p][0][c] = sum dy*xhat (dgamma), partial[p][1][c] = sum dy (dbeta) over its rows,
// partial[p][2][c] = column sums of the bf16 values written to dxs (the bias gradient of the
// projection whose output gradient dxs is), zero without dxs
// PRE: 0 = loads as they are needed (x, dy; then dres after the row reductions); 1 = all three row operands requested
// before the reductions; 2 = as 1, and the NEXT row's operands are requested before this row's reductions (two rows
// in flight per wave).  Same arithmetic in the same order in all three.
template <int MAXC, typename TD = uint16_t>
struct LnRowIn {
    f32x4v x[MAXC], r[MAXC];
    typename Io4<TD>::raw dy[MAXC];
    float mu, rs, sc;
};

template <int MAXC, bool WITH_RES, typename TD = uint16_t>
__device__ __forceinline__ void ln_row_request(LnRowIn<MAXC, TD>& in, int row, int lane, int nch, int E,
                                               const TD* __restrict__ dy, const float* __restrict__ x,
                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                               const float* __restrict__ dres, const float* __restrict__ sscale,
                                               int rows_per_sample) {
    // row base as an opaque scalar: keeps the accesses in the s[base] + v[lane offset] form (otherwise the loop is
    // strength-reduced into per-array 64-bit vector addresses: +16 VGPRs and VALU adds per row)
    const int64_t ro = (int64_t)__builtin_amdgcn_readfirstlane(row) * E;
    const float* xr = x + ro;
    const TD* dyr = dy + ro;
    const float* rr = dres + ro;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const unsigned c = lane + 64 * i;
        if (c < (unsigned)nch) {
            in.x[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(xr + 4u * c));
            in.dy[i] = Io4<TD>::load(dyr + 4u * c);
            if (WITH_RES) in.r[i] = dres ? __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(rr + 4u * c)) : f32x4v{0, 0, 0, 0};
        }
    }
    in.mu = mean[row];
    in.rs = rstd[row];
    in.sc = sscale ? sscale[row / rows_per_sample] : 1.f;
}

template <int MAXC, int PRE, int WPE = 1, typename TD = uint16_t>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE))) void ln_bwd_kernel(float* __restrict__ dx, TD* __restrict__ dxs,
                                                     float* __restrict__ partial, const TD* __restrict__ dy,
                                                     const float* __restrict__ x, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                     const float* __restrict__ dres, const float* __restrict__ sscale,
                                                     int rows_per_sample, int M, int E) {
    __shared__ float red[4][MAXC * 256 + 4];
    // the wave index as a SCALAR: row offsets then live in SGPRs and every access is base(s) + lane offset(v)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nch = E >> 2;
    f32x4v g[MAXC], ag[MAXC], ab[MAXC], as[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        g[i] = c < nch ? *reinterpret_cast<const f32x4v*>(gamma + 4 * c) : f32x4v{0, 0, 0, 0};
        ag[i] = f32x4v{0, 0, 0, 0};
        ab[i] = f32x4v{0, 0, 0, 0};
        as[i] = f32x4v{0, 0, 0, 0};
    }
    const float invE = 1.f / (float)E;
    const int stride = gridDim.x * 4;
    int row = blockIdx.x * 4 + wave;
    LnRowIn<MAXC, TD> nxt;
    if (PRE == 2 && row < M)
        ln_row_request<MAXC, true, TD>(nxt, row, lane, nch, E, dy, x, mean, rstd, dres, sscale, rows_per_sample);
    for (; row < M; row += stride) {
        LnRowIn<MAXC, TD> in;
        if (PRE == 2) {
            in = nxt;
            if (row + stride < M)
                ln_row_request<MAXC, true, TD>(nxt, row + stride, lane, nch, E, dy, x, mean, rstd, dres, sscale, rows_per_sample);
        } else {
            ln_row_request<MAXC, PRE == 1, TD>(in, row, lane, nch, E, dy, x, mean, rstd, dres, sscale, rows_per_sample);
        }
        const float mu = in.mu, rs = in.rs;
        float xh[MAXC][4], d[MAXC][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                Io4<TD>::unpack(in.dy[i], d[i]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[i][e] = (in.x[i][e] - mu) * rs;
                    ag[i][e] += d[i][e] * xh[i][e];
                    ab[i][e] += d[i][e];
                    const float dg = d[i][e] * g[i][e];
                    d[i][e] = dg;
                    s1 += dg;
                    s2 += dg * xh[i][e];
                }
            }
        }
        s1 = wave_sum(s1) * invE;
        s2 = wave_sum(s2) * invE;
        const float sc = in.sc;
        const int64_t ro = (int64_t)__builtin_amdgcn_readfirstlane(row) * E;
        float* dxr = dx + ro;
        TD* dxsr = dxs + ro;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const unsigned c = lane + 64 * i;
            if (c < (unsigned)nch) {
                f32x4v r;
                if (PRE == 0) r = dres ? __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(dres + ro + 4u * c)) : f32x4v{0, 0, 0, 0};
                else r = in.r[i];
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    r[e] += rs * (d[i][e] - s1 - xh[i][e] * s2);
                    o[e] = r[e] * sc;
                }
                __builtin_nontemporal_store(r, reinterpret_cast<f32x4v*>(dxr + 4u * c));
                if (dxs) {
                    Io4<TD>::store(dxsr + 4u * c, o);                  // (o now holds the values as written)
#pragma unroll
                    for (int e = 0; e < 4; ++e) as[i][e] += o[e];
                }
            }
        }
    }
    // workgroup partials, one plane at a time: the four waves' sums meet in LDS (fixed order)
#pragma unroll
    for (int which = 0; which < 3; ++which) {
#pragma unroll
        for (int i = 0; i < MAXC; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                red[wave][(i * 64 + lane) * 4 + e] = which == 0 ? ag[i][e] : (which == 1 ? ab[i][e] : as[i][e]);
        __syncthreads();
        float* prow = partial + ((int64_t)blockIdx.x * 3 + which) * E;
#pragma nounroll
        for (unsigned c = threadIdx.x; c < (unsigned)E; c += 256)
            prow[c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
        __syncthreads();
    }
}

// ---- GELU -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gelu_fwd_kernel(uint16_t* __restrict__ g, const uint16_t* __restrict__ h,
                                                       int64_t n8) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const u32x4v v = *reinterpret_cast<const u32x4v*>(h + 8 * i);
        float a[4], b[4];
        unpack_bf16x4(u32x2v{v[0], v[1]}, a);
        unpack_bf16x4(u32x2v{v[2], v[3]}, b);
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = gelu_f(a[e]); b[e] = gelu_f(b[e]); }
        const u32x2v pa = pack_bf16x4(a), pb = pack_bf16x4(b);
        *reinterpret_cast<u32x4v*>(g + 8 * i) = u32x4v{pa[0], pa[1], pb[0], pb[1]};
    }
}

__global__ __launch_bounds__(256) void gelu_bwd_kernel(uint16_t* __restrict__ dh, const uint16_t* __restrict__ dg,
                                                       const uint16_t* __restrict__ h, int64_t n8) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const u32x4v hv = *reinterpret_cast<const u32x4v*>(h + 8 * i);
        const u32x4v gv = *reinterpret_cast<const u32x4v*>(dg + 8 * i);
        float a[4], b[4], ga[4], gb[4];
        unpack_bf16x4(u32x2v{hv[0], hv[1]}, a);
        unpack_bf16x4(u32x2v{hv[2], hv[3]}, b);
        unpack_bf16x4(u32x2v{gv[0], gv[1]}, ga);
        unpack_bf16x4(u32x2v{gv[2], gv[3]}, gb);
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = ga[e] * gelu_grad_f(a[e]); b[e] = gb[e] * gelu_grad_f(b[e]); }
        const u32x2v pa = pack_bf16x4(a), pb = pack_bf16x4(b);
        *reinterpret_cast<u32x4v*>(dh + 8 * i) = u32x4v{pa[0], pa[1], pb[0], pb[1]};
    }
}

// ---- residual add: out = x + s_b * y ------------------------------------------------------------
__global__ __launch_bounds__(256) void residual_add_kernel(float* __restrict__ out, const float* __restrict__ x,
                                                           const uint16_t* __restrict__ y, const float* __restrict__ ss,
                                                           int64_t n4, int64_t per_sample4) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4v xv = *reinterpret_cast<const f32x4v*>(x + 4 * i);
        float yv[4];
        unpack_bf16x4(*reinterpret_cast<const u32x2v*>(y + 4 * i), yv);
        const float s = ss ? ss[i / per_sample4] : 1.f;
        *reinterpret_cast<f32x4v*>(out + 4 * i) =
            f32x4v{xv[0] + s * yv[0], xv[1] + s * yv[1], xv[2] + s * yv[2], xv[3] + s * yv[3]};
    }
}

// ---- scale + cast: out(bf16) = s_b * x(f32) ----------------------------------------------------
__global__ __launch_bounds__(256) void scale_cast_kernel(uint16_t* __restrict__ out, const float* __restrict__ x,
                                                         const float* __restrict__ ss, int64_t n4, int64_t per_sample4) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4v xv = *reinterpret_cast<const f32x4v*>(x + 4 * i);
        const float s = ss ? ss[i / per_sample4] : 1.f;
        const float o[4] = {xv[0] * s, xv[1] * s, xv[2] * s, xv[3] * s};
        *reinterpret_cast<u32x2v*>(out + 4 * i) = pack_bf16x4(o);
    }
}

// ---- column sums of a bf16 (M, C) matrix: partial[p][c] over the rows of slab p ---------
// block = 256 threads = 32 column groups of 8 columns x 8 row lanes; grid = (ceil(C/256), P)
__global__ __launch_bounds__(256) void colsum_kernel(float* __restrict__ partial, const uint16_t* __restrict__ a,
                                                     int M, int C, int rows_per_slab) {
    __shared__ float red[8][256 + 8];
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 256 + cg * 8;
    const int r0 = blockIdx.y * rows_per_slab, r1 = min(M, r0 + rows_per_slab);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c0 < C) {
        for (int r = r0 + rl; r < r1; r += 8) {
            const u32x4v v = *reinterpret_cast<const u32x4v*>(a + (int64_t)r * C + c0);
            float f0[4], f1[4];
            unpack_bf16x4(u32x2v{v[0], v[1]}, f0);
            unpack_bf16x4(u32x2v{v[2], v[3]}, f1);
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[e] += f0[e]; acc[4 + e] += f1[e]; }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl][cg * 8 + e] = acc[e];
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < C) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += red[k][threadIdx.x];
        partial[(int64_t)blockIdx.y * C + c] = s;
    }
}

// ---- elementwise pass + column sums of what it writes (bias gradients ride along) ---------
// block = 32 column groups (8 columns, 16 bytes of bf16) x 8 row lanes; grid = (ceil(C/256),
// ceil(M/CS_ROWS)); partial[slab][c] = sum over the slab's rows of the bf16-ROUNDED outputs.
constexpr int CS_ROWS = 128;

__device__ __forceinline__ void colsum_block_reduce(float* __restrict__ partial, const float (&acc)[8], int C) {
    __shared__ float red[8][256 + 8];
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl][cg * 8 + e] = acc[e];
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < C) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += red[k][threadIdx.x];
        partial[(int64_t)blockIdx.y * C + c] = s;
    }
}

// dh = dg * gelu'(h) (bf16), partial = column sums of dh  (fc1 bias gradient)
__global__ __launch_bounds__(256) void gelu_bwd_colsum_kernel(uint16_t* __restrict__ dh, float* __restrict__ partial,
                                                              const uint16_t* __restrict__ dg,
                                                              const uint16_t* __restrict__ h, int M, int C) {
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 256 + cg * 8;
    const int r0 = blockIdx.y * CS_ROWS, r1 = min(M, r0 + CS_ROWS);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c0 < C) {
#pragma unroll 2
        for (int r = r0 + rl; r < r1; r += 8) {
            const int64_t off = (int64_t)r * C + c0;
            const u32x4v hv = *reinterpret_cast<const u32x4v*>(h + off);
            const u32x4v gv = *reinterpret_cast<const u32x4v*>(dg + off);
            float a[4], b[4], ga[4], gb[4];
            unpack_bf16x4(u32x2v{hv[0], hv[1]}, a);
            unpack_bf16x4(u32x2v{hv[2], hv[3]}, b);
            unpack_bf16x4(u32x2v{gv[0], gv[1]}, ga);
            unpack_bf16x4(u32x2v{gv[2], gv[3]}, gb);
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] = ga[e] * gelu_grad_f(a[e]); b[e] = gb[e] * gelu_grad_f(b[e]); }
            const u32x2v pa = pack_bf16x4(a), pb = pack_bf16x4(b);
            *reinterpret_cast<u32x4v*>(dh + off) = u32x4v{pa[0], pa[1], pb[0], pb[1]};
            unpack_bf16x4(pa, a);
            unpack_bf16x4(pb, b);
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[e] += a[e]; acc[4 + e] += b[e]; }
        }
    }
    colsum_block_reduce(partial, acc, C);
}

// out = bf16(s_b * x), partial = column sums of out  (fc2 bias gradient)
__global__ __launch_bounds__(256) void scale_cast_colsum_kernel(uint16_t* __restrict__ out, float* __restrict__ partial,
                                                                const float* __restrict__ x,
                                                                const float* __restrict__ ss, int rows_per_sample,
                                                                int M, int C) {
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 256 + cg * 8;
    const int r0 = blockIdx.y * CS_ROWS, r1 = min(M, r0 + CS_ROWS);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c0 < C) {
#pragma unroll 2
        for (int r = r0 + rl; r < r1; r += 8) {
            const int64_t off = (int64_t)r * C + c0;
            const f32x4v x0 = *reinterpret_cast<const f32x4v*>(x + off);
            const f32x4v x1 = *reinterpret_cast<const f32x4v*>(x + off + 4);
            const float s = ss ? ss[r / rows_per_sample] : 1.f;
            float a[4] = {x0[0] * s, x0[1] * s, x0[2] * s, x0[3] * s};
            float b[4] = {x1[0] * s, x1[1] * s, x1[2] * s, x1[3] * s};
            const u32x2v pa = pack_bf16x4(a), pb = pack_bf16x4(b);
            *reinterpret_cast<u32x4v*>(out + off) = u32x4v{pa[0], pa[1], pb[0], pb[1]};
            unpack_bf16x4(pa, a);
            unpack_bf16x4(pb, b);
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[e] += a[e]; acc[4 + e] += b[e]; }
        }
    }
    colsum_block_reduce(partial, acc, C);
}

// plain column sums at the same slab size (qkv bias gradient: dqkv comes out of the attention kernels)
__global__ __launch_bounds__(256) void colsum128_kernel(float* __restrict__ partial, const uint16_t* __restrict__ a,
                                                        int M, int C) {
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 256 + cg * 8;
    const int r0 = blockIdx.y * CS_ROWS, r1 = min(M, r0 + CS_ROWS);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c0 < C) {
#pragma unroll 4
        for (int r = r0 + rl; r < r1; r += 8) {
            const u32x4v v = *reinterpret_cast<const u32x4v*>(a + (int64_t)r * C + c0);
            float f0[4], f1[4];
            unpack_bf16x4(u32x2v{v[0], v[1]}, f0);
            unpack_bf16x4(u32x2v{v[2], v[3]}, f1);
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[e] += f0[e]; acc[4 + e] += f1[e]; }
        }
    }
    colsum_block_reduce(partial, acc, C);
}

// ---- gradient finalisation: dst[map(r)][c] += sum_p part[p][r][c], many tensors per launch ---
// One workgroup = CL consecutive 4-element chunks x PL part lanes (CL * PL = 256): part lane l
// adds parts l, l + PL, ... in ascending order, the lanes are combined in ascending order through
// LDS — a fixed summation tree (bit-reproducible), no atomics.
struct GradJobs {
    cream_grad_job job[CREAM_MAX_GRAD_JOBS];
    int first_block[CREAM_MAX_GRAD_JOBS + 1];
    int njobs;
};

// W = elements per chunk: 4 (fp32 partials, 16-byte loads) or 8 (bf16 partials with cols % 8 == 0: 16-byte loads too —
// at 4 elements a bf16 load is 8 bytes per lane, which the vector memory path serves at 0.54-0.70x the 16-byte rate)
template <int W> struct PartChunk;
template <> struct PartChunk<4> {
    f32x4v v;
    __device__ __forceinline__ void zero() { v = f32x4v{0, 0, 0, 0}; }
    __device__ __forceinline__ void add(const PartChunk& o) { v += o.v; }
    static __device__ __forceinline__ PartChunk load(const void* src, int64_t idx, bool is_bf16) {
        PartChunk c;
        if (is_bf16) {
            float f[4];
            unpack_bf16x4(*reinterpret_cast<const u32x2v*>(reinterpret_cast<const uint16_t*>(src) + idx), f);
            c.v = f32x4v{f[0], f[1], f[2], f[3]};
        } else {
            c.v = *reinterpret_cast<const f32x4v*>(reinterpret_cast<const float*>(src) + idx);
        }
        return c;
    }
    __device__ __forceinline__ void add_into(float* d) const {
        f32x4v o = *reinterpret_cast<f32x4v*>(d);
        o += v;
        *reinterpret_cast<f32x4v*>(d) = o;
    }
    __device__ __forceinline__ void store(float* d) const { *reinterpret_cast<f32x4v*>(d) = v; }
};
template <> struct PartChunk<8> {                                  // bf16 sources only
    f32x4v lo, hi;
    __device__ __forceinline__ void zero() { lo = f32x4v{0, 0, 0, 0}; hi = lo; }
    __device__ __forceinline__ void add(const PartChunk& o) { lo += o.lo; hi += o.hi; }
    static __device__ __forceinline__ PartChunk load(const void* src, int64_t idx, bool) {
        const u32x4v r = *reinterpret_cast<const u32x4v*>(reinterpret_cast<const uint16_t*>(src) + idx);
        float a[4], b[4];
        unpack_bf16x4(u32x2v{r[0], r[1]}, a);
        unpack_bf16x4(u32x2v{r[2], r[3]}, b);
        PartChunk c;
        c.lo = f32x4v{a[0], a[1], a[2], a[3]};
        c.hi = f32x4v{b[0], b[1], b[2], b[3]};
        return c;
    }
    __device__ __forceinline__ void add_into(float* d) const {
        f32x4v o0 = *reinterpret_cast<f32x4v*>(d), o1 = *reinterpret_cast<f32x4v*>(d + 4);
        o0 += lo;
        o1 += hi;
        *reinterpret_cast<f32x4v*>(d) = o0;
        *reinterpret_cast<f32x4v*>(d + 4) = o1;
    }
    __device__ __forceinline__ void store(float* d) const {
        *reinterpret_cast<f32x4v*>(d) = lo;
        *reinterpret_cast<f32x4v*>(d + 4) = hi;
    }
};

// part lanes per chunk: a lane walks its parts four loads at a time, so the dependent chain of a job is nparts / (4 PL)
// round trips — the LayerNorm jobs (1024 slab partials of a few hundred columns) set the kernel's duration at PL = 16
// (16 round trips on 6 workgroups while the chip idles): 64 lanes there
__host__ __device__ __forceinline__ int grad_part_lanes(int nparts) { return nparts <= 16 ? 4 : (nparts <= 64 ? 16 : 64); }
__host__ __device__ __forceinline__ int grad_chunk_width(const cream_grad_job& b) { return b.src_bf16 && b.cols % 8 == 0 ? 8 : 4; }

template <int W>
__device__ __forceinline__ void grad_finalize_job(const cream_grad_job& jb, int block_in_job, PartChunk<W>* red) {
    const int PL = grad_part_lanes(jb.nparts), CL = 256 / PL;
    const int cl = threadIdx.x % CL, pl = threadIdx.x / CL;
    const int cpr = jb.cols / W;                               // chunks per row
    const int64_t nchunks = (int64_t)jb.rows * cpr;
    const int64_t chunk = (int64_t)block_in_job * CL + cl;
    PartChunk<W> acc;
    acc.zero();
    if (chunk < nchunks) {
        const bool bf = jb.src_bf16 != 0;
        int p = pl;
        for (; p + 3 * PL < jb.nparts; p += 4 * PL) {          // four loads in flight
            const PartChunk<W> a0 = PartChunk<W>::load(jb.src, (int64_t)p * jb.pstride + chunk * W, bf);
            const PartChunk<W> a1 = PartChunk<W>::load(jb.src, (int64_t)(p + PL) * jb.pstride + chunk * W, bf);
            const PartChunk<W> a2 = PartChunk<W>::load(jb.src, (int64_t)(p + 2 * PL) * jb.pstride + chunk * W, bf);
            const PartChunk<W> a3 = PartChunk<W>::load(jb.src, (int64_t)(p + 3 * PL) * jb.pstride + chunk * W, bf);
            acc.add(a0); acc.add(a1); acc.add(a2); acc.add(a3);
        }
        for (; p < jb.nparts; p += PL) acc.add(PartChunk<W>::load(jb.src, (int64_t)p * jb.pstride + chunk * W, bf));
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (PL == 64) {                                            // two levels (fixed order): 8 groups of 8 lanes, then the 8 sums
        if (pl < 8) {
            PartChunk<W> s = red[(8 * pl) * CL + cl];
            for (int l = 1; l < 8; ++l) s.add(red[(8 * pl + l) * CL + cl]);
            acc = s;
        }
        __syncthreads();
        if (pl < 8) red[pl * CL + cl] = acc;
        __syncthreads();
    }
    const int nl = PL == 64 ? 8 : PL;
    if (pl == 0 && chunk < nchunks) {
        PartChunk<W> s = red[cl];
        for (int l = 1; l < nl; ++l) s.add(red[l * CL + cl]);
        const int r = (int)(chunk / cpr), c = (int)(chunk - (int64_t)r * cpr) * W;
        const int rr = jb.interleave > 0 ? 3 * (r % jb.interleave) + r / jb.interleave : r;
        if (jb.overwrite) s.store(jb.dst + (int64_t)rr * jb.ld + c);      // dst = the sum (a gradient that did not exist yet: no zero fill)
        else s.add_into(jb.dst + (int64_t)rr * jb.ld + c);
    }
}

__global__ __launch_bounds__(256) void grad_finalize_kernel(const GradJobs J) {
    __shared__ __attribute__((aligned(16))) char red[256 * sizeof(PartChunk<8>)];
    int j = 0;
    while (j + 1 < J.njobs && (int)blockIdx.x >= J.first_block[j + 1]) ++j;
    const cream_grad_job& jb = J.job[j];
    const int b = (int)blockIdx.x - J.first_block[j];
    if (grad_chunk_width(jb) == 8) grad_finalize_job<8>(jb, b, reinterpret_cast<PartChunk<8>*>(red));
    else grad_finalize_job<4>(jb, b, reinterpret_cast<PartChunk<4>*>(red));
}

int grid_for(int64_t n_items, int per_block) {
    const int64_t b = (n_items + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > 256 * 16 ? 256 * 16 : b));
}

}  // namespace

extern "C" {

int cream_ln_partials(void)
{
    // workgroups (= partial rows) of the LayerNorm backward (256 .. 1024 measured in the step: 1024 stays)
    return 1024;
}

int cream_ln_fwd(void* y, float* mean, float* rstd, const float* x, const float* gamma, const float* beta,
                 int M, int E, float eps, void* stream)
{
    if (M < 0 || E <= 0) return CREAM_ERR_BAD_ARG;
    if (M == 0) return CREAM_OK;
    if (!y || !mean || !rstd || !x || !gamma || !beta) return CREAM_ERR_BAD_ARG;
    if (E % 4 || E > 64 * 4 * LN_MAXC) return CREAM_ERR_TOO_LARGE;
    if (((uintptr_t)y | (uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta) % 16) return CREAM_ERR_BAD_ARG;
    CREAM_LAUNCH(ln_fwd_kernel<uint16_t>, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, (uint16_t*)y, mean, rstd,
                       x, gamma, beta, M, E, eps, (float*)nullptr, (const uint16_t*)nullptr, (const float*)nullptr, 1);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

int cream_ln_f32_fwd(float* y, float* mean, float* rstd, const float* x, const float* gamma, const float* beta,
                     int M, int E, float eps, void* stream)
{
    if (M < 0 || E <= 0) return CREAM_ERR_BAD_ARG;
    if (M == 0) return CREAM_OK;
    if (!y || !mean || !rstd || !x || !gamma || !beta) return CREAM_ERR_BAD_ARG;
    if (E % 4 || E > 64 * 4 * LN_MAXC) return CREAM_ERR_TOO_LARGE;
    if (((uintptr_t)y | (uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta) % 16) return CREAM_ERR_BAD_ARG;
    hipLaunchKernelGGL(ln_fwd_kernel<float>, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, y, mean, rstd,
                       x, gamma, beta, M, E, eps, (float*)nullptr, (const uint16_t*)nullptr, (const float*)nullptr, 1);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

int cream_ln_f32_bwd(float* dx, float* partial, const float* dy, const float* x, const float* mean, const float* rstd,
                     const float* gamma, int M, int E, void* stream)
{
    if (M <= 0 || E <= 0) return CREAM_ERR_BAD_ARG;
    if (!dx || !partial || !dy || !x || !mean || !rstd || !gamma) return CREAM_ERR_BAD_ARG;
    if (E % 4 || E > 64 * 4 * LN_MAXC) return CREAM_ERR_TOO_LARGE;
    if (((uintptr_t)dx | (uintptr_t)dy | (uintptr_t)x | (uintptr_t)gamma) % 16) return CREAM_ERR_BAD_ARG;
    auto kern = E <= 512 ? ln_bwd_kernel<2, 0, 1, float> : (E <= 768 ? ln_bwd_kernel<3, 0, 1, float> : ln_bwd_kernel<LN_MAXC, 0, 1, float>);
    hipLaunchKernelGGL(kern, dim3(cream_ln_partials()), dim3(256), 0, (hipStream_t)stream, dx, (float*)nullptr, partial, dy, x,
                       mean, rstd, gamma, (const float*)nullptr, (const float*)nullptr, 1, M, E);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

int cream_add_ln_fwd(float* xsum, void* y, float* mean, float* rstd, const float* x, const void* res,
                     const float* sample_scale, int rows_per_sample, const float* gamma, const float* beta,
                     int M, int E, float eps, void* stream)
{
    if (M < 0 || E <= 0 || rows_per_sample <= 0) return CREAM_ERR_BAD_ARG;
    if (M == 0) return CREAM_OK;
    if (!xsum || !y || !mean || !rstd || !x || !res || !gamma || !beta) return CREAM_ERR_BAD_ARG;
    if (E % 4 || E > 64 * 4 * LN_MAXC) return CREAM_ERR_TOO_LARGE;
    if (((uintptr_t)xsum | (uintptr_t)y | (uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta) % 16 || (uintptr_t)res % 8)
        return CREAM_ERR_BAD_ARG;
    CREAM_LAUNCH(ln_fwd_kernel<uint16_t>, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, (uint16_t*)y, mean, rstd,
                       x, gamma, beta, M, E, eps, xsum, (const uint16_t*)res, sample_scale, rows_per_sample);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

int cream_ln_bwd(float* dx, void* dx_scaled, float* partial, const void* dy, const float* x, const float* mean,
                 const float* rstd, const float* gamma, const float* dres, const float* sample_scale,
                 int rows_per_sample, int M, int E, void* stream)
{
    if (M <= 0 || E <= 0 || rows_per_sample <= 0) return CREAM_ERR_BAD_ARG;
    if (!dx || !partial || !dy || !x || !mean || !rstd || !gamma) return CREAM_ERR_BAD_ARG;
    if (E % 4 || E > 64 * 4 * LN_MAXC) return CREAM_ERR_TOO_LARGE;
    if (((uintptr_t)dx | (uintptr_t)dx_scaled | (uintptr_t)dy | (uintptr_t)x | (uintptr_t)gamma | (uintptr_t)dres) % 16)
        return CREAM_ERR_BAD_ARG;
    // register footprint follows the row width: 2 chunks per lane cover E <= 512, 3 cover E <= 768
    // E <= 512: two rows in flight per wave (106 VGPRs, still 4 waves per SIMD = the grid's 4 workgroups per CU).  Alone
    // on the chip all variants run at 4.6-4.8 TB/s (tools/probes/ln_probe.hip); next to the weight-gradient GEMMs of
    // the side stream the deeper request queue is worth 1.0-1.2 % of the training step (same-box A/B, twice).
    auto kern = E <= 512 ? ln_bwd_kernel<2, 2> : (E <= 768 ? ln_bwd_kernel<3, 0> : ln_bwd_kernel<LN_MAXC, 0>);
    CREAM_LAUNCH(kern, dim3(cream_ln_partials()), dim3(256), 0, (hipStream_t)stream, dx,
                       (uint16_t*)dx_scaled, partial, (const uint16_t*)dy, x, mean, rstd, gamma, dres, sample_scale,
                       rows_per_sample, M, E);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

int cream_gelu_fwd(void* g, const void* h, int64_t n, void* stream)
{
    if (n < 0 || n % 8) return CREAM_ERR_BAD_ARG;
    if (n == 0) return CREAM_OK;
    if (!g || !h || ((uintptr_t)g | (uintptr_t)h) % 16) return CREAM_ERR_BAD_ARG;
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (uint16_t*)g,
                       (const uint16_t*)h, n / 8);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

int cream_gelu_bwd(void* dh, const void* dg, const void* h, int64_t n, void* stream)
{
    if (n < 0 || n % 8) return CREAM_ERR_BAD_ARG;
    if (n == 0) return CREAM_OK;
    if (!dh || !dg || !h || ((uintptr_t)dh | (uintptr_t)dg | (uintptr_t)h) % 16) return CREAM_ERR_BAD_ARG;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (uint16_t*)dh,
                       (const uint16_t*)dg, (const uint16_t*)h, n / 8);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

int cream_residual_add(float* out, const float* x, const void* y, const float* sample_scale,
                       int64_t n, int64_t per_sample, void* stream)
{
    if (n < 0 || n % 4 || per_sample <= 0 || per_sample % 4) return CREAM_ERR_BAD_ARG;
    if (n == 0) return CREAM_OK;
    if (!out || !x || !y || ((uintptr_t)out | (uintptr_t)x) % 16 || (uintptr_t)y % 8) return CREAM_ERR_BAD_ARG;
    hipLaunchKernelGGL(residual_add_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, out, x,
                       (const uint16_t*)y, sample_scale, n / 4, per_sample / 4);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

int cream_scale_cast(void* out, const float* x, const float* sample_scale, int64_t n, int64_t per_sample,
                     void* stream)
{
    if (n < 0 || n % 4 || per_sample <= 0 || per_sample % 4) return CREAM_ERR_BAD_ARG;
    if (n == 0) return CREAM_OK;
    if (!out || !x || (uintptr_t)x % 16 || (uintptr_t)out % 8) return CREAM_ERR_BAD_ARG;
    hipLaunchKernelGGL(scale_cast_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, (uint16_t*)out,
                       x, sample_scale, n / 4, per_sample / 4);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

int cream_colsum_slabs(int M) { return M <= 0 ? 0 : (M + 511) / 512; }

int cream_colsum(float* partial, const void* a, int M, int C, void* stream)
{
    if (M < 0 || C <= 0 || C % 8) return CREAM_ERR_BAD_ARG;
    if (M == 0) return CREAM_OK;
    if (!partial || !a || (uintptr_t)a % 16) return CREAM_ERR_BAD_ARG;
    hipLaunchKernelGGL(colsum_kernel, dim3((C + 255) / 256, cream_colsum_slabs(M)), dim3(256), 0, (hipStream_t)stream,
                       partial, (const uint16_t*)a, M, C, 512);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}


int cream_colsum128_slabs(int M) { return M <= 0 ? 0 : (M + CS_ROWS - 1) / CS_ROWS; }

int cream_colsum128(float* partial, const void* a, int M, int C, void* stream)
{
    if (M < 0 || C <= 0 || C % 8) return CREAM_ERR_BAD_ARG;
    if (M == 0) return CREAM_OK;
    if (!partial || !a || (uintptr_t)a % 16) return CREAM_ERR_BAD_ARG;
    hipLaunchKernelGGL(colsum128_kernel, dim3((C + 255) / 256, cream_colsum128_slabs(M)), dim3(256), 0,
                       (hipStream_t)stream, partial, (const uint16_t*)a, M, C);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

int cream_gelu_bwd_colsum(void* dh, float* partial, const void* dg, const void* h, int M, int C, void* stream)
{
    if (M < 0 || C <= 0 || C % 8) return CREAM_ERR_BAD_ARG;
    if (M == 0) return CREAM_OK;
    if (!dh || !partial || !dg || !h || ((uintptr_t)dh | (uintptr_t)dg | (uintptr_t)h) % 16) return CREAM_ERR_BAD_ARG;
    hipLaunchKernelGGL(gelu_bwd_colsum_kernel, dim3((C + 255) / 256, cream_colsum128_slabs(M)), dim3(256), 0,
                       (hipStream_t)stream, (uint16_t*)dh, partial, (const uint16_t*)dg, (const uint16_t*)h, M, C);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

int cream_scale_cast_colsum(void* out, float* partial, const float* x, const float* sample_scale,
                            int rows_per_sample, int M, int C, void* stream)
{
    if (M < 0 || C <= 0 || C % 8 || rows_per_sample <= 0) return CREAM_ERR_BAD_ARG;
    if (M == 0) return CREAM_OK;
    if (!out || !partial || !x || ((uintptr_t)out | (uintptr_t)x) % 16) return CREAM_ERR_BAD_ARG;
    hipLaunchKernelGGL(scale_cast_colsum_kernel, dim3((C + 255) / 256, cream_colsum128_slabs(M)), dim3(256), 0,
                       (hipStream_t)stream, (uint16_t*)out, partial, x, sample_scale, rows_per_sample, M, C);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

int cream_grad_finalize(const cream_grad_job* jobs, int njobs, void* stream)
{
    if (njobs < 0 || njobs > CREAM_MAX_GRAD_JOBS) return CREAM_ERR_BAD_ARG;
    if (njobs == 0) return CREAM_OK;
    if (!jobs) return CREAM_ERR_BAD_ARG;
    GradJobs J;
    J.njobs = njobs;
    int blocks = 0;
    for (int j = 0; j < njobs; ++j) {
        const cream_grad_job& b = jobs[j];
        if (!b.dst || !b.src || b.nparts <= 0 || b.rows <= 0 || b.cols <= 0 || b.cols % 4 || b.ld % 4 ||
            b.pstride % 4 || b.interleave < 0 || (b.interleave > 0 && b.rows != 3 * b.interleave))
            return CREAM_ERR_BAD_ARG;
        const int W = grad_chunk_width(b);
        if ((uintptr_t)b.dst % 16 || (uintptr_t)b.src % (b.src_bf16 && W == 4 ? 8 : 16)) return CREAM_ERR_BAD_ARG;
        if (W == 8 && b.pstride % 8) return CREAM_ERR_BAD_ARG;
        J.job[j] = b;
        J.first_block[j] = blocks;
        const int CL = 256 / grad_part_lanes(b.nparts);
        const int64_t nchunks = (int64_t)b.rows * (b.cols / W);
        blocks += (int)((nchunks + CL - 1) / CL);
    }
    J.first_block[njobs] = blocks;
    CREAM_LAUNCH(grad_finalize_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, J);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

}  // extern "C"
