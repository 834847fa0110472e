// slices.hip — pack / unpack the ACTIVE slices of the super-weight gradients (HBM-bound copies).
//
// Reference semantics: DistributedDataParallel all-reduces every parameter's full gradient
// (AutoFormer/supernet_train.py:286-289, find_unused_parameters=True); a sampled sub-network only ever
// writes W.grad[:out, :in] of each super weight (Linear_super.py:71-81, qkv_super.py:72-83 — the qkv rows
// 3 i + j, i < Q are exactly rows [0, 3 Q)), and every rank samples the SAME sub-network
// (supernet_engine.py:36), so the bytes outside the slices are zeros on every rank.  The reducer sends the
// slices only: one job table per (bucket, sampled configuration), one launch to gather them into a
// contiguous message and one to scatter the averaged message back.  16-byte vectors where the slice
// geometry allows (always for the AutoFormer search spaces), grid-stride over rows.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cream_amd.h"

namespace {

__global__ __launch_bounds__(256) void slices_copy_kernel(const cream_slice_job* __restrict__ jobs, float* __restrict__ packed,
                                                          int to_packed)
{
    const cream_slice_job j = jobs[blockIdx.y];
    float* msg = packed + j.packed_off;
    const bool vec = (j.cols % 4 == 0) && (j.ld % 4 == 0) && (j.packed_off % 4 == 0) && ((reinterpret_cast<uintptr_t>(j.full) & 15) == 0);
    for (int r = blockIdx.x; r < j.rows; r += gridDim.x) {
        float* f = j.full + (int64_t)r * j.ld;
        float* m = msg + (int64_t)r * j.cols;
        if (vec) {
            const int n4 = j.cols >> 2;
            for (int c = threadIdx.x; c < n4; c += 256) {
                if (to_packed) reinterpret_cast<float4*>(m)[c] = reinterpret_cast<const float4*>(f)[c];
                else reinterpret_cast<float4*>(f)[c] = reinterpret_cast<const float4*>(m)[c];
            }
        } else {
            for (int c = threadIdx.x; c < j.cols; c += 256) {
                if (to_packed) m[c] = f[c];
                else f[c] = m[c];
            }
        }
    }
}

}  // namespace

extern "C" int cream_slices_copy(const cream_slice_job* jobs_dev, int njobs, float* packed, int max_rows, int to_packed, void* stream)
{
    if (njobs < 0 || max_rows < 0 || (njobs > 0 && (!jobs_dev || !packed))) return CREAM_ERR_BAD_ARG;
    if (njobs == 0 || max_rows == 0) return CREAM_OK;
    const int gx = max_rows < 128 ? max_rows : 128;
    hipLaunchKernelGGL(slices_copy_kernel, dim3(gx, njobs), dim3(256), 0, (hipStream_t)stream, jobs_dev, packed, to_packed);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}
