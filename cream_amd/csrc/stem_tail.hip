// stem_tail.hip — the two ends of the supernet around the block stack (gfx950), HBM-bound passes.
//
// Reference semantics (AutoFormer/model/supernet_transformer.py:147-172, `forward_features`):
//     x = patch_embed(img)                                  embedding_super.py:27-40 (stride = kernel conv)
//     x = cat(cls_token[..., :E], x) + pos_embed[..., :E]   :150-155
//     ... blocks ...
//     x = norm(x);  return mean(x[:, 1:], dim=1)            :166-170  (pre_norm, gp)
// Stem: `im2patch` unfolds the image into bf16 GEMM rows in one pass (the framework's permute copy + cast
// were two); `stem_assemble` writes the fp32 residual stream from the GEMM's bf16 output, the class token
// and the position embedding; `stem_bwd` splits the stream's gradient into the GEMM's bf16 output
// gradient and per-chunk partial sums over the batch for pos_embed / cls_token.
// Tail: the last block hands over its output PENDING (x1 + s_b * f, as between blocks); `tail_fwd` forms it,
// normalises every row and accumulates the token mean of the normalised rows (the affine map commutes with
// the mean); `tail_bwd` is the LayerNorm backward of the broadcast gradient and emits exactly what the last
// block's backward wants: the residual-stream gradient (fp32), the drop-path-scaled bf16 copy and the
// column sums of that copy (fc2 bias gradient) — replacing residual_add, LayerNorm, mean, their autograd
// and scale_cast_colsum.  One wave per row, 16-byte vectors, no atomics, fixed summation order.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "attn_common.hpp"
#include "cream_amd.h"

namespace {
using namespace cream;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ void unpack_bf16x4(u32x2v v, float (&f)[4]) {
    f[0] = __uint_as_float(v[0] << 16); f[1] = __uint_as_float(v[0] & 0xFFFF0000u);
    f[2] = __uint_as_float(v[1] << 16); f[3] = __uint_as_float(v[1] & 0xFFFF0000u);
}
__device__ __forceinline__ u32x2v pack_bf16x4(const float (&f)[4]) {
    return u32x2v{f2bf_pair(f[0], f[1]), f2bf_pair(f[2], f[3])};
}

constexpr int MAXC_ALL = 5;          // float4 chunks per lane: E <= 1280
constexpr int TAIL_ROWS = 32;        // rows of one image per workgroup in tail_fwd

// row of the pending stream: x1 + s * f
template <int MAXC>
__device__ __forceinline__ void load_row(f32x4v (&v)[MAXC], const float* x1r, const uint16_t* fr, float sc, int nch, int lane) {
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < nch ? *reinterpret_cast<const f32x4v*>(x1r + 4 * c) : f32x4v{0, 0, 0, 0};
        if (fr && c < nch) {
            float r[4];
            unpack_bf16x4(*reinterpret_cast<const u32x2v*>(fr + 4 * c), r);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][e] += sc * r[e];
        }
    }
}

// grid (chunks of TAIL_ROWS rows, B); part[b][chunk][c] = sum over the chunk's rows n >= 1 of xhat
template <int MAXC>
__global__ __launch_bounds__(256) void tail_fwd_kernel(float* __restrict__ part, float* __restrict__ mean,
                                                       float* __restrict__ rstd, const float* __restrict__ x1,
                                                       const uint16_t* __restrict__ f, const float* __restrict__ sscale,
                                                       int N, int E, float eps) {
    __shared__ float red[4][MAXC * 256 + 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y, n0 = blockIdx.x * TAIL_ROWS, n1 = min(N, n0 + TAIL_ROWS);
    const int nch = E >> 2;
    const float sc = (f && sscale) ? sscale[b] : 1.f;
    f32x4v acc[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) acc[i] = f32x4v{0, 0, 0, 0};
    for (int n = n0 + wave; n < n1; n += 4) {
        const int64_t row = (int64_t)b * N + n;
        f32x4v v[MAXC];
        load_row<MAXC>(v, x1 + row * E, f ? f + row * E : nullptr, sc, nch, lane);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        const float mu = wave_sum(s) / (float)E;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i)
            if (lane + 64 * i < nch) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mu; q += d * d; }
            }
        const float rs = rsqrtf(wave_sum(q) / (float)E + eps);
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
        if (n >= 1) {
#pragma unroll
            for (int i = 0; i < MAXC; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][e] += (v[i][e] - mu) * rs;
        }
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave][(i * 64 + lane) * 4 + e] = acc[i][e];
    __syncthreads();
    float* dst = part + ((int64_t)b * gridDim.x + blockIdx.x) * E;
    for (int c = threadIdx.x; c < E; c += 256) dst[c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// pooled[b][c] = gamma[c] * xm[b][c] + beta[c],  xm = (sum of the chunks) / (N - 1)
__global__ void tail_pool_kernel(float* __restrict__ pooled, float* __restrict__ xm, const float* __restrict__ part,
                                 const float* __restrict__ gamma, const float* __restrict__ beta, int chunks, int E, int N,
                                 int total) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int b = idx / E, c = idx - b * E;
    float s = 0.f;
    for (int r = 0; r < chunks; ++r) s += part[((int64_t)b * chunks + r) * E + c];
    const float m = s / (float)(N - 1);
    xm[idx] = m;
    pooled[idx] = m * gamma[c] + beta[c];
}

// grid = P workgroups of 4 waves walking rows; partial[p][c] = column sums of the bf16 values written to dxs
template <int MAXC>
__global__ __launch_bounds__(256) void tail_bwd_kernel(float* __restrict__ dx, uint16_t* __restrict__ dxs,
                                                       float* __restrict__ partial, const float* __restrict__ g,
                                                       const float* __restrict__ x1, const uint16_t* __restrict__ f,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ sscale,
                                                       int N, int M, int E) {
    __shared__ float red[4][MAXC * 256 + 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = E >> 2;
    const float invE = 1.f / (float)E, invN = 1.f / (float)(N - 1);
    f32x4v gm[MAXC], as[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        gm[i] = c < nch ? *reinterpret_cast<const f32x4v*>(gamma + 4 * c) : f32x4v{0, 0, 0, 0};
        as[i] = f32x4v{0, 0, 0, 0};
    }
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        const int b = row / N, n = row - b * N;
        const float sc = sscale ? sscale[b] : 1.f;
        if (n == 0) {                                   // the class token does not reach the pooled output
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
                const int c = lane + 64 * i;
                if (c < nch) {
                    *reinterpret_cast<f32x4v*>(dx + (int64_t)row * E + 4 * c) = f32x4v{0, 0, 0, 0};
                    *reinterpret_cast<u32x2v*>(dxs + (int64_t)row * E + 4 * c) = u32x2v{0, 0};
                }
            }
            continue;
        }
        const float mu = mean[row], rs = rstd[row];
        f32x4v v[MAXC];
        load_row<MAXC>(v, x1 + (int64_t)row * E, f ? f + (int64_t)row * E : nullptr, sc, nch, lane);
        float d[MAXC][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                const f32x4v gv = *reinterpret_cast<const f32x4v*>(g + (int64_t)b * E + 4 * c);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[i][e] = (v[i][e] - mu) * rs;
                    d[i][e] = gv[e] * gm[i][e] * invN;
                    s1 += d[i][e];
                    s2 += d[i][e] * v[i][e];
                }
            }
        }
        s1 = wave_sum(s1) * invE;
        s2 = wave_sum(s2) * invE;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                f32x4v r;
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    r[e] = rs * (d[i][e] - s1 - v[i][e] * s2);
                    o[e] = r[e] * sc;
                }
                *reinterpret_cast<f32x4v*>(dx + (int64_t)row * E + 4 * c) = r;
                const u32x2v pk = pack_bf16x4(o);
                *reinterpret_cast<u32x2v*>(dxs + (int64_t)row * E + 4 * c) = pk;
                unpack_bf16x4(pk, o);
#pragma unroll
                for (int e = 0; e < 4; ++e) as[i][e] += o[e];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave][(i * 64 + lane) * 4 + e] = as[i][e];
    __syncthreads();
    for (int c = threadIdx.x; c < E; c += 256)
        partial[(int64_t)blockIdx.x * E + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// ---- stem ---------------------------------------------------------------------------------------------
// patches[(b, gi, gj)][(c, i, j)] = bf16(img[b][c][gi*ph + i][gj*pw + j]); one thread per (patch, c, i) run of pw pixels
__global__ __launch_bounds__(256) void im2patch_kernel(uint16_t* __restrict__ out, const float* __restrict__ img, int C,
                                                       int H, int W, int ph, int pw, int64_t total) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int segs = C * ph;
    const int64_t patch = t / segs;
    const int seg = (int)(t - patch * segs), c = seg / ph, i = seg - c * ph;
    const int gw = W / pw, gh = H / ph;
    const int gj = (int)(patch % gw);
    const int64_t bg = patch / gw;
    const int gi = (int)(bg % gh);
    const int64_t b = bg / gh;
    const float* src = img + ((b * C + c) * H + (int64_t)gi * ph + i) * W + (int64_t)gj * pw;
    uint16_t* dst = out + patch * ((int64_t)segs * pw) + (int64_t)seg * pw;
    for (int j = 0; j < pw; j += 8) {
        const f32x4v a = *reinterpret_cast<const f32x4v*>(src + j);
        const f32x4v b4 = *reinterpret_cast<const f32x4v*>(src + j + 4);
        *reinterpret_cast<u32x4v*>(dst + j) = u32x4v{f2bf_pair(a[0], a[1]), f2bf_pair(a[2], a[3]), f2bf_pair(b4[0], b4[1]),
                                                     f2bf_pair(b4[2], b4[3])};
    }
}

// x0[b][0] = cls + pos[0];  x0[b][n] = y[b][n-1] + pos[n]      (4 columns per thread)
__global__ __launch_bounds__(256) void stem_assemble_kernel(float* __restrict__ x0, const uint16_t* __restrict__ y,
                                                            const float* __restrict__ cls, const float* __restrict__ pos,
                                                            int64_t ld_pos, int N, int E, int64_t total4) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total4) return;
    const int nch = E >> 2;
    const int64_t row = t / nch;
    const int c = (int)(t - row * nch) * 4;
    const int64_t b = row / N;
    const int n = (int)(row - b * N);
    f32x4v v = pos ? *reinterpret_cast<const f32x4v*>(pos + (int64_t)n * ld_pos + c) : f32x4v{0, 0, 0, 0};
    if (n == 0) {
        const f32x4v k = *reinterpret_cast<const f32x4v*>(cls + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += k[e];
    } else {
        float r[4];
        unpack_bf16x4(*reinterpret_cast<const u32x2v*>(y + (b * (N - 1) + (n - 1)) * E + c), r);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += r[e];
    }
    *reinterpret_cast<f32x4v*>(x0 + row * E + c) = v;
}

// dy[b][n-1] = bf16(dx0[b][n]) (n >= 1);  psum[chunk][n][c] = sum over the chunk's images of dx0[b][n][c]
// grid (N * E/4 / 256, chunks): a thread owns 4 columns of one token and walks the chunk's images
__global__ __launch_bounds__(256) void stem_bwd_kernel(uint16_t* __restrict__ dy, float* __restrict__ psum,
                                                       const float* __restrict__ dx0, int B, int N, int E, int per_chunk) {
    const int nch = E >> 2;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * nch) return;
    const int n = t / nch, c = (t - n * nch) * 4;
    const int b0 = blockIdx.y * per_chunk, b1 = min(B, b0 + per_chunk);
    f32x4v acc = {0, 0, 0, 0};
#pragma unroll 4
    for (int b = b0; b < b1; ++b) {
        const f32x4v v = *reinterpret_cast<const f32x4v*>(dx0 + ((int64_t)b * N + n) * E + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += v[e];
        if (n >= 1) {
            const float o[4] = {v[0], v[1], v[2], v[3]};
            *reinterpret_cast<u32x2v*>(dy + ((int64_t)b * (N - 1) + (n - 1)) * E + c) = pack_bf16x4(o);
        }
    }
    *reinterpret_cast<f32x4v*>(psum + ((int64_t)blockIdx.y * N + n) * E + c) = acc;
}

bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

// ---- soft-target cross entropy: loss rows AND the logit gradient in one pass -----------------------
// timm.loss.SoftTargetCrossEntropy as the step uses it (supernet_engine.py:60-66: criterion(outputs, targets) with the
// Mixup / label-smoothing soft targets): loss_b = sum_c -t[b,c] log_softmax(x[b,:])[c], mean over the batch.  One
// workgroup per row: max, log-sum-exp, the row loss and d loss / d x[b,c] = (softmax(x)[c] sum_c t - t[c]) * gscale —
// what autograd derives, in fp32 (the caller scales it by the incoming gradient and rounds once to the bf16 operand type
// of the head's dgrad / wgrad GEMMs, like the cast at the autocast boundary).  The framework spent ~10 launches on this (log_softmax, mul, neg, two reductions, their backward).
template <typename TX>
__global__ __launch_bounds__(256) void soft_ce_kernel(float* __restrict__ loss_rows, float* __restrict__ dlogits,
                                                      const TX* __restrict__ logits, const float* __restrict__ target, int C,
                                                      float gscale)
{
    __shared__ float red[3][4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const TX* x = logits + (int64_t)row * C;
    const float* t = target + (int64_t)row * C;
    constexpr int PER = 8;                                    // classes per thread: C <= 2048
    float xv[PER], tv[PER];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = tid + 256 * i;
        const bool ok = c < C;
        float v = -INFINITY;
        if (ok) {
            if constexpr (sizeof(TX) == 2) v = bf2f((short)x[c]); else v = x[c];
        }
        xv[i] = v;
        tv[i] = ok ? t[c] : 0.f;
        mx = fmaxf(mx, v);
    }
    auto block_reduce = [&](float v, int slot, bool is_max) -> float {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { const float u = __shfl_xor(v, o); v = is_max ? fmaxf(v, u) : v + u; }
        if (lane == 0) red[slot][wave] = v;
        __syncthreads();
        const float a = red[slot][0], b = red[slot][1], c = red[slot][2], d = red[slot][3];
        return is_max ? fmaxf(fmaxf(a, b), fmaxf(c, d)) : (a + b) + (c + d);          // fixed order
    };
    mx = block_reduce(mx, 0, true);
    float se = 0.f, st = 0.f, stx = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const float e = xv[i] == -INFINITY ? 0.f : __expf(xv[i] - mx);
        se += e;
        st += tv[i];
        stx += tv[i] == 0.f ? 0.f : tv[i] * (xv[i] - mx);
        xv[i] = e;
    }
    se = block_reduce(se, 1, false);
    // pack the two target sums into one more pass through the same slot array (after everyone has read slot 1)
    st = block_reduce(st, 2, false);
    __syncthreads();
    stx = block_reduce(stx, 0, false);
    const float lse = __logf(se);
    if (tid == 0) loss_rows[row] = st * lse - stx;            // sum_c t (lse - (x - mx))
    const float inv = 1.f / se;
    float* d = dlogits + (int64_t)row * C;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = tid + 256 * i;
        if (c < C) d[c] = (xv[i] * inv * st - tv[i]) * gscale;
    }
}

extern "C" {

int cream_soft_ce(float* loss_rows, float* dlogits, const void* logits, const float* target, int B, int C, int logits_dtype,
                  float grad_scale, void* stream)
{
    if (B <= 0 || C <= 0 || C > 2048) return B == 0 ? CREAM_OK : CREAM_ERR_TOO_LARGE;
    if (!loss_rows || !dlogits || !logits || !target) return CREAM_ERR_BAD_ARG;
    if (logits_dtype == CREAM_BF16)
        hipLaunchKernelGGL(soft_ce_kernel<uint16_t>, dim3(B), dim3(256), 0, (hipStream_t)stream, loss_rows, dlogits,
                           (const uint16_t*)logits, target, C, grad_scale);
    else if (logits_dtype == CREAM_F32)
        hipLaunchKernelGGL(soft_ce_kernel<float>, dim3(B), dim3(256), 0, (hipStream_t)stream, loss_rows, dlogits,
                           (const float*)logits, target, C, grad_scale);
    else
        return CREAM_ERR_BAD_DTYPE;
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

int cream_tail_chunks(int N) { return N <= 0 ? 0 : (N + TAIL_ROWS - 1) / TAIL_ROWS; }

int cream_tail_fwd(float* pooled, float* xm, float* part, float* mean, float* rstd, const float* x1, const void* f,
                   const float* sample_scale, const float* gamma, const float* beta, int B, int N, int E, float eps,
                   void* stream)
{
    if (B <= 0 || N <= 1 || E <= 0) return CREAM_ERR_BAD_ARG;
    if (!pooled || !xm || !part || !mean || !rstd || !x1 || !gamma || !beta) return CREAM_ERR_BAD_ARG;
    if (E % 4 || E > 64 * 4 * MAXC_ALL) return CREAM_ERR_TOO_LARGE;
    if (!al16(x1) || !al16(gamma) || !al16(beta) || ((uintptr_t)f & 7)) return CREAM_ERR_BAD_ARG;
    const int chunks = cream_tail_chunks(N);
    hipStream_t st = (hipStream_t)stream;
    auto kern = E <= 512 ? tail_fwd_kernel<2> : (E <= 768 ? tail_fwd_kernel<3> : tail_fwd_kernel<MAXC_ALL>);
    hipLaunchKernelGGL(kern, dim3(chunks, B), dim3(256), 0, st, part, mean, rstd, x1, (const uint16_t*)f, sample_scale, N, E,
                       eps);
    const int total = B * E;
    hipLaunchKernelGGL(tail_pool_kernel, dim3((total + 255) / 256), dim3(256), 0, st, pooled, xm, part, gamma, beta, chunks,
                       E, N, total);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

int cream_tail_bwd(float* dx, void* dx_scaled, float* partial, const float* g, const float* x1, const void* f,
                   const float* mean, const float* rstd, const float* gamma, const float* sample_scale, int B, int N,
                   int E, void* stream)
{
    if (B <= 0 || N <= 1 || E <= 0) return CREAM_ERR_BAD_ARG;
    if (!dx || !dx_scaled || !partial || !g || !x1 || !mean || !rstd || !gamma) return CREAM_ERR_BAD_ARG;
    if (E % 4 || E > 64 * 4 * MAXC_ALL) return CREAM_ERR_TOO_LARGE;
    if (!al16(dx) || !al16(g) || !al16(x1) || !al16(gamma) || ((uintptr_t)f & 7) || ((uintptr_t)dx_scaled & 7))
        return CREAM_ERR_BAD_ARG;
    auto kern = E <= 512 ? tail_bwd_kernel<2> : (E <= 768 ? tail_bwd_kernel<3> : tail_bwd_kernel<MAXC_ALL>);
    hipLaunchKernelGGL(kern, dim3(cream_ln_partials()), dim3(256), 0, (hipStream_t)stream, dx, (uint16_t*)dx_scaled, partial,
                       g, x1, (const uint16_t*)f, mean, rstd, gamma, sample_scale, N, B * N, E);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

int cream_im2patch(void* patches, const float* img, int B, int C, int H, int W, int ph, int pw, void* stream)
{
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || ph <= 0 || pw <= 0 || H % ph || W % pw || pw % 8) return CREAM_ERR_BAD_ARG;
    if (!patches || !img || !al16(patches) || !al16(img) || W % 4) return CREAM_ERR_BAD_ARG;
    const int64_t total = (int64_t)B * (H / ph) * (W / pw) * C * ph;
    hipLaunchKernelGGL(im2patch_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (uint16_t*)patches, img, C, H, W, ph, pw, total);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

int cream_stem_assemble(float* x0, const void* y, const float* cls, const float* pos, int64_t ld_pos, int B, int N, int E,
                        void* stream)
{
    if (B <= 0 || N <= 1 || E <= 0 || E % 4) return CREAM_ERR_BAD_ARG;
    if (!x0 || !y || !cls || !al16(x0) || !al16(cls) || ((uintptr_t)y & 7) || (pos && (!al16(pos) || ld_pos % 4)))
        return CREAM_ERR_BAD_ARG;
    const int64_t total4 = (int64_t)B * N * (E / 4);
    hipLaunchKernelGGL(stem_assemble_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x0,
                       (const uint16_t*)y, cls, pos, ld_pos, N, E, total4);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

int cream_stem_bwd_chunks(int B) { return B <= 0 ? 0 : (B + 15) / 16; }

int cream_stem_bwd(void* dy, float* psum, const float* dx0, int B, int N, int E, void* stream)
{
    if (B <= 0 || N <= 1 || E <= 0 || E % 4) return CREAM_ERR_BAD_ARG;
    if (!dy || !psum || !dx0 || !al16(psum) || !al16(dx0) || ((uintptr_t)dy & 7)) return CREAM_ERR_BAD_ARG;
    const int chunks = cream_stem_bwd_chunks(B);
    hipLaunchKernelGGL(stem_bwd_kernel, dim3((N * (E / 4) + 255) / 256, chunks), dim3(256), 0, (hipStream_t)stream,
                       (uint16_t*)dy, psum, dx0, B, N, E, 16);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

}  // extern "C"
