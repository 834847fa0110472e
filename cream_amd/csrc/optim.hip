// optim.hip — the parameter update of the supernet step and the operand copies of the GEMM kernels in
// ONE launch over every tensor of the model.
//
// Reference: timm's create_optimizer(args, model) -> torch.optim.AdamW over add_weight_decay's two
// parameter groups (AutoFormer/supernet_train.py:296, lr = args.lr * batch * world / 512 at :294),
// stepped once per iteration by loss_scaler(...) (supernet_engine.py:96).  timm is not vendored in the
// reference: the update rule is restated from torch.optim.AdamW (decoupled weight decay, bias-corrected
// moments, no amsgrad) — parity unpinned by the reference, pinned here against torch.optim.AdamW.
//
// Why one kernel: per step the framework path cost ~3 ms of HOST time (foreach moment update over 232
// tensors + a second pass that re-converts every weight to its bf16 operand copy) for ~0.25 ms of HBM
// traffic.  Here every element is read once (p, g, m, v) and written once (p, m, v, the bf16 copy
// W[:, :] in the layout the forward GEMM reads, and the TRANSPOSED bf16 copy the dgrad GEMM reads —
// see csrc/gemm_mfma.hip), the qkv weight is de-interleaved into its [q | k | v] matrices on the way
// (qkv_super.py:75: row 3 i + j of the super weight is output i of part j).
//
// The job table (one entry per tensor) and its tile prefix sums live in device memory: pointers never
// change between steps, so the host builds and uploads them once.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "attn_common.hpp"
#include "cream_amd.h"

namespace {
using namespace cream;

constexpr int TR = 96, TC = 64;                                 // tile: 96 rows (32 x the 3 qkv parts) x 64 columns
constexpr int LP = TC + 2;                                      // LDS pitch (bf16) of the staged tile

__global__ __launch_bounds__(256) void adamw_mirror_kernel(const cream_param_job* __restrict__ jobs,
                                                           const int32_t* __restrict__ first_tile, int njobs, int update,
                                                           float lr, float beta1, float beta2, float omb1, float omb2, float eps,
                                                           float inv_bc1, float inv_sqrt_bc2)
{
    __shared__ uint16_t tile[TR * LP];
    // job of this workgroup: last j with first_tile[j] <= blockIdx.x
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (first_tile[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const cream_param_job jb = jobs[lo];
    const int t = (int)blockIdx.x - first_tile[lo];
    const int tcols = (jb.cols + TC - 1) / TC;
    const int r0 = (t / tcols) * TR, c0 = (t % tcols) * TC;
    const int tid = threadIdx.x, cq = tid & 15, rr = tid >> 4;  // 16 threads x 4 columns per row, 16 rows per pass
    const int c = c0 + cq * 4;
    const bool upd = update && jb.g != nullptr;
    const float decay = 1.f - lr * jb.weight_decay, step = lr * inv_bc1;
    // three passes at a time: all twelve 16-byte loads of the three row groups are in flight before the first store (the parameter,
    // moment and gradient pointers may alias as far as the compiler knows, so it would not move a load across a store itself)
    const bool vec = (jb.ld & 3) == 0;
#pragma unroll
    for (int pass0 = 0; pass0 < TR / 16; pass0 += 3) {
        f32x4v pv[3], gv[3], mv[3], vv[3];
        bool fast[3], in[3];
        int64_t o[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int r = r0 + (pass0 + u) * 16 + rr;
            in[u] = r < jb.rows && c < jb.cols;
            o[u] = (int64_t)r * jb.ld + c;
            fast[u] = in[u] && vec && c + 4 <= jb.cols;
            pv[u] = f32x4v{0, 0, 0, 0};
            if (fast[u]) {
                pv[u] = *reinterpret_cast<const f32x4v*>(jb.p + o[u]);
                if (upd) {
                    gv[u] = *reinterpret_cast<const f32x4v*>(jb.g + o[u]);
                    mv[u] = *reinterpret_cast<const f32x4v*>(jb.m + o[u]);
                    vv[u] = *reinterpret_cast<const f32x4v*>(jb.v + o[u]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int lr_ = (pass0 + u) * 16 + rr;
            float v4[4] = {0, 0, 0, 0};
            if (fast[u]) {
                if (upd) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        pv[u][e] *= decay;
                        mv[u][e] = mv[u][e] + (gv[u][e] - mv[u][e]) * omb1;
                        vv[u][e] = beta2 * vv[u][e] + omb2 * gv[u][e] * gv[u][e];
                        pv[u][e] -= step * mv[u][e] / (sqrtf(vv[u][e]) * inv_sqrt_bc2 + eps);
                    }
                    *reinterpret_cast<f32x4v*>(jb.p + o[u]) = pv[u];
                    *reinterpret_cast<f32x4v*>(jb.m + o[u]) = mv[u];
                    *reinterpret_cast<f32x4v*>(jb.v + o[u]) = vv[u];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v4[e] = pv[u][e];
            } else if (in[u]) {
                const int nv = min(4, jb.cols - c);
                for (int e = 0; e < nv; ++e) {
                    float p1 = jb.p[o[u] + e];
                    if (upd) {
                        const float g1 = jb.g[o[u] + e];
                        float m1 = jb.m[o[u] + e], v1 = jb.v[o[u] + e];
                        p1 *= decay;
                        m1 = m1 + (g1 - m1) * omb1;
                        v1 = beta2 * v1 + omb2 * g1 * g1;
                        p1 -= step * m1 / (sqrtf(v1) * inv_sqrt_bc2 + eps);
                        jb.p[o[u] + e] = p1; jb.m[o[u] + e] = m1; jb.v[o[u] + e] = v1;
                    }
                    v4[e] = p1;
                }
            }
            if (jb.mir) {
                // LDS row: parts grouped (row 3 i + j -> j * 32 + i) so that the transposed copy below writes
                // 8 consecutive outputs of ONE part
                const int srow = jb.deinterleave ? (lr_ % 3) * (TR / 3) + lr_ / 3 : lr_;
                uint16_t* d = tile + srow * LP + cq * 4;
                *reinterpret_cast<uint32_t*>(d) = f2bf_pair(v4[0], v4[1]);
                *reinterpret_cast<uint32_t*>(d + 2) = f2bf_pair(v4[2], v4[3]);
            }
        }
    }
    if (!jb.mir) return;
    __syncthreads();
    uint16_t* mir = reinterpret_cast<uint16_t*>(jb.mir);
    // ---- row-major copy: LDS row s -> (part, row) of the copy -------------------------------------
    for (int i = tid; i < TR * (TC / 4); i += 256) {
        const int s = i / (TC / 4), q4 = (i % (TC / 4)) * 4;
        int part = 0, row;
        if (jb.deinterleave) { part = s / (TR / 3); row = r0 / 3 + s % (TR / 3); if (3 * row + part >= jb.rows) continue; }
        else { row = r0 + s; if (row >= jb.rows) continue; }
        const int cc = c0 + q4;
        if (cc >= jb.cols) continue;
        uint16_t* d = mir + (int64_t)part * jb.seg_stride + (int64_t)row * jb.ld_mir + cc;
        const uint16_t* sp = tile + s * LP + q4;
        if (cc + 4 <= jb.cols && (jb.ld_mir & 3) == 0) *reinterpret_cast<u32x2v*>(d) = u32x2v{*reinterpret_cast<const uint32_t*>(sp), *reinterpret_cast<const uint32_t*>(sp + 2)};
        else for (int e = 0; e < 4 && cc + e < jb.cols; ++e) d[e] = sp[e];
    }
    if (!jb.mir_t) return;
    // ---- transposed copy: column cc of the tile -> row cc of W^T, 8 consecutive outputs per store --
    uint16_t* mt = reinterpret_cast<uint16_t*>(jb.mir_t);
    for (int i = tid; i < TC * (TR / 8); i += 256) {
        const int col = i % TC, s8 = (i / TC) * 8;              // LDS rows s8 .. s8 + 7 (one part when de-interleaved)
        const int cc = c0 + col;
        if (cc >= jb.cols) continue;
        int part = 0, row;
        int nrows;                                              // rows of the copy in this part
        if (jb.deinterleave) { part = s8 / (TR / 3); row = r0 / 3 + s8 % (TR / 3); nrows = (jb.rows - part + 2) / 3; }
        else { row = r0 + s8; nrows = jb.rows; }
        if (row >= nrows) continue;
        uint16_t* d = mt + (int64_t)part * jb.seg_stride_t + (int64_t)cc * jb.ld_mir_t + row;
        uint16_t e8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) e8[e] = tile[(s8 + e) * LP + col];
        if (row + 8 <= nrows && (jb.ld_mir_t & 7) == 0 && (((uintptr_t)d) & 15) == 0)
            *reinterpret_cast<u32x4v*>(d) = u32x4v{(uint32_t)e8[0] | ((uint32_t)e8[1] << 16), (uint32_t)e8[2] | ((uint32_t)e8[3] << 16),
                                                   (uint32_t)e8[4] | ((uint32_t)e8[5] << 16), (uint32_t)e8[6] | ((uint32_t)e8[7] << 16)};
        else for (int e = 0; e < 8 && row + e < nrows; ++e) d[e] = e8[e];
    }
}

}  // namespace

extern "C" {

int cream_param_job_tiles(int rows, int cols)
{
    if (rows <= 0 || cols <= 0) return 0;
    return ((rows + TR - 1) / TR) * ((cols + TC - 1) / TC);
}

int cream_adamw_step(const cream_param_job* jobs_dev, const int32_t* first_tile_dev, int njobs, int total_tiles, int update,
                     double lr, double beta1, double beta2, double eps, int64_t step, void* stream)
{
    if (njobs < 0 || total_tiles < 0 || (update && step < 1)) return CREAM_ERR_BAD_ARG;
    if (njobs == 0 || total_tiles == 0) return CREAM_OK;
    if (!jobs_dev || !first_tile_dev) return CREAM_ERR_BAD_ARG;
    double bc1 = 1.0, bc2 = 1.0;
    if (update) {
        bc1 = 1.0 - pow(beta1, (double)step);
        bc2 = 1.0 - pow(beta2, (double)step);
    }
    hipLaunchKernelGGL(adamw_mirror_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, jobs_dev, first_tile_dev, njobs,
                       update, (float)lr, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)(1.0 / bc1),
                       (float)(1.0 / sqrt(bc2)));
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

}  // extern "C"
