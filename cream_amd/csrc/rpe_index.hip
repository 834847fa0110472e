// rpe_index.hip — bucketed relative-position gather (fwd) / scatter-add (bwd) for gfx950.
//
// What it computes (reference: iRPE/DeiT-with-iRPE/rpe_ops/rpe_index_cuda.cu:24-52):
//   fwd  Y[b,h,i,j]   = in[b,h,i, idx[i,j]]
//   bwd  gin[b,h,i,u] += sum_{j: idx[i,j]==u} gout[b,h,i,j]
// Both are zero-FLOP, HBM-bound passes over the (B,H,Lq,Lk) tensor.  The design is
// not a translation of the reference's one-thread-per-element grid-stride loops:
//
//   fwd  one WORKGROUP owns a run of query rows of ONE (b,h) plane.  Its lookup rows (a
//        contiguous rows x nb block of the input, <= 60 KB) are staged in LDS with one
//        coalesced read; the block's output — one contiguous slab of the (B,H,Lq,Lk)
//        tensor — is then streamed as a FLAT array: every lane owns 16-byte ALIGNED
//        vectors of consecutive elements (a vector may straddle two query rows), reads
//        their bucket ids straight from the equally flat (Lq,Lk) index matrix (L2
//        resident), gathers from the LDS block and issues one aligned dwordx4 store.
//        No per-row ragged tails, no scattered 200-byte lookup reads, full-line writes.
//
//   bwd  the reference's global atomics contend on <= nb addresses per 577 adds and
//        give a run-to-run different fp sum.  Here the observation is that idx[i,j]
//        does not depend on (b,h): with LANE <-> PLANE the bucket id of step j is
//        WAVE-UNIFORM.  A wave owns 64 planes; tiles of gout are transposed through
//        LDS (coalesced 16-byte HBM reads along j, conflict-free ds_read_b128 along
//        planes), every lane adds its value into its private column of an LDS bin
//        array with a conflict-free read-modify-write — no cross-lane reduction,
//        no global atomics, and a FIXED ascending-j summation order (bit-identical to
//        a sequential CPU loop for f32/f64).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bfloat16.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>

#include "cream_amd.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// 16-byte vector that is only guaranteed 4-/2-/8-byte aligned in memory: rows of an
// odd Lk (197, 577) start at arbitrary element offsets.  gfx950 global memory accepts
// element-aligned dwordx4 accesses; HBM traffic is unchanged.
struct __attribute__((packed, aligned(4))) vec16_a4 { u32x4 v; };
struct __attribute__((packed, aligned(2))) vec16_a2 { u32x4 v; };
struct __attribute__((packed, aligned(8))) vec16_a8 { u32x4 v; };

template <int BYTES> struct raw_elem;
template <> struct raw_elem<2> { using type = uint16_t; using vec = vec16_a2; };
template <> struct raw_elem<4> { using type = uint32_t; using vec = vec16_a4; };
template <> struct raw_elem<8> { using type = uint64_t; using vec = vec16_a8; };

constexpr int WAVE = 64;

// Workgroup barrier for LDS hand-offs only.  __syncthreads() is a release fence for global memory as well: hipcc puts
// s_waitcnt vmcnt(0) in front of it, so every barrier of a store loop would wait until all stores issued so far are ACKNOWLEDGED —
// the write stream of the gather kernels drained once per plane.  Nothing here ever reads what the kernel itself stores.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ------------------------------------------------------------------------------------
// forward: gather
// ------------------------------------------------------------------------------------
// grid = (row blocks, planes) — consecutive workgroups write consecutive memory; 1024 threads.
// Measured evolution on config 4 (B=64, H=12, L=577, nb=50; 1.11 GB fp32 / 0.56 GB bf16 algorithmic;
// profiles/r02_rpe_index.md): the round-1 kernel (one wave per query row walking the planes, ids in
// registers, element-aligned 2.3 KB rows) ran at 3.4 (fp32) / 2.9 (bf16) TB/s; rocprofv3 showed the
// memory pipeline saturated (TA busy 98 %, 74 % of wave cycles stalled on instruction issue), and
// ablations put 0.07 of its 0.31 ms on the scattered 200-byte lookup-row reads.  This kernel:
// 3.8-4.2 / 3.7-3.8 TB/s.  What was tried and did NOT pay on top: shifting lanes to aligned stores
// inside the row-per-wave kernel (DPP + alignbyte: slower), small row blocks in memory order (8-64
// rows: slower than one large block per workgroup), 256/512-thread workgroups (slower).  A pure-store
// ablation of this mapping tops out at ~4.4 TB/s against 6.9 TB/s of a framework fill of the same
// buffer.  Round 2 (tools/probes/store_probe.hip, rpe_probe.hip): a pure 16-byte store stream with THIS mapping (one
// 692 KB slab per 1024-thread workgroup) reaches 5.8 TB/s, so the mapping is not the limit; requesting the ids of the
// next 1-4 vectors ahead of the store (software prefetch) changes nothing either.  What costs is the id stream itself:
// 4 bytes of L2 traffic per output element for every plane -> rpe_gather_planes below.  This kernel remains the
// fallback for rows too long for that kernel's per-thread vector budget.
template <int BYTES>
__global__ __launch_bounds__(1024) void rpe_gather_plane(
    typename raw_elem<BYTES>::type* __restrict__ y,
    const typename raw_elem<BYTES>::type* __restrict__ in,
    const int32_t* __restrict__ idx,
    int H, int Lq, int Lk, int nb,
    int64_t s0, int64_t s1, int64_t s2, int64_t s3,
    int rows_per_block)
{
    using E = typename raw_elem<BYTES>::type;
    constexpr int V = 16 / BYTES;
    const int NT = blockDim.x;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    E* table = reinterpret_cast<E*>(smem);                     // [rows of the block][nb]
    const int tid = threadIdx.x;
    const int p = blockIdx.y;                                  // x = row block (fastest): consecutive workgroups write
    const int r0 = blockIdx.x * rows_per_block, r1 = min(Lq, r0 + rows_per_block);   // consecutive memory
    if (r0 >= r1) return;
    const int b = p / H, h = p - b * H;
    const E* src = in + (int64_t)b * s0 + (int64_t)h * s1 + (int64_t)r0 * s2;
    const int total = (r1 - r0) * nb;
    if (s3 == 1 && s2 == nb) {                                  // the block is contiguous in memory
        for (int t = tid; t < total; t += NT) table[t] = src[t];
    } else {
        for (int t = tid; t < total; t += NT) {
            const int r = t / nb, u = t - r * nb;
            table[t] = src[(int64_t)r * s2 + (int64_t)u * s3];
        }
    }
    __syncthreads();

    // flat element range of this block inside the plane, and its 16-byte aligned interior
    E* out = y + (int64_t)p * Lq * Lk;
    const int32_t* fidx = idx;                                 // same flat layout as one output plane
    const int f0 = r0 * Lk, f1 = r1 * Lk;
    const int mis = (int)((reinterpret_cast<uintptr_t>(out + f0) & 15) / BYTES);   // elements above alignment
    const int fa = min(f1, f0 + ((V - mis) % V));              // first aligned element
    const int nvec = (f1 - fa) / V;
    const int ft = fa + nvec * V;                              // tail start
    auto gather1 = [&](int f) -> E {
        const int i = f / Lk;                                  // (only on the <= 2 (V - 1) edge elements)
        return table[(i - r0) * nb + fidx[f]];
    };
    if (tid < fa - f0) out[f0 + tid] = gather1(f0 + tid);
    if (tid < f1 - ft) out[ft + tid] = gather1(ft + tid);

    // per-thread walk: vector q = tid, tid + NT, ...; (i, j) advanced without divisions
    const int stepq = (NT * V) / Lk, stepr = (NT * V) % Lk;
    int f = fa + tid * V;
    int i = f / Lk, j = f - i * Lk;
    const __amdgpu_buffer_rsrc_t rs_idx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<int32_t*>(idx), 0, (int)((int64_t)Lq * Lk * 4), 0x00020000);
    constexpr int NW = V / 4 + (V < 4);                        // 16-byte id loads per output vector
    for (int q = tid; q < nvec; q += NT) {
        int32_t ids[V];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const u32x4 t4 = __builtin_amdgcn_raw_buffer_load_b128(rs_idx, (f + 4 * w) * 4, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * w + e < V) ids[4 * w + e] = (int32_t)t4[e];
        }
        union { u32x4 vec; E e[V]; } pk;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const int over = (j + e >= Lk) ? 1 : 0;            // the vector may run into the next query row
            pk.e[e] = table[(i + over - r0) * nb + ids[e]];
        }
        *reinterpret_cast<u32x4*>(out + f) = pk.vec;           // 16-byte aligned
        f += NT * V;
        i += stepq;
        j += stepr;
        if (j >= Lk) { j -= Lk; ++i; }
    }
}

// Several planes per workgroup (the measured path).  The bucket ids depend on (i, j) only, yet the kernel above
// fetches them once per PLANE: 4 bytes of L2 traffic per output element — as many bytes as the fp32 output itself and
// twice the bf16 output — competing with the store stream for the same fabric.  Here a workgroup owns a block of query
// rows for G planes of the same 16-byte alignment class (plane p + k * period: identical vector partition of the flat
// output range), turns every id ONCE into the LDS address of its lookup value (row * nb + id, held in registers:
// NV vectors x V elements per thread), and then for each plane only stages that plane's lookup rows (double-buffered,
// one barrier per plane) and runs  V x ds_read + one aligned 16-byte store  per vector: no id traffic, no address
// arithmetic in the per-plane loop.
template <int BYTES, int NV>
__global__ __launch_bounds__(1024) void rpe_gather_planes(
    typename raw_elem<BYTES>::type* __restrict__ y,
    const typename raw_elem<BYTES>::type* __restrict__ in,
    const int32_t* __restrict__ idx,
    int BH, int H, int Lq, int Lk, int nb,
    int64_t s0, int64_t s1, int64_t s2, int64_t s3,
    int rows_per_block, int period, int G)
{
    using E = typename raw_elem<BYTES>::type;
    constexpr int V = 16 / BYTES;
    const int NT = blockDim.x;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    // grid = (row blocks, plane groups): workgroups that are neighbours in launch order write neighbouring memory (whole
    // planes fill up one after the other).  The transposed order (groups fastest: neighbours share their slice of the id
    // matrix) brings the HBM reads down from 282 MB to 97 MB per launch (algorithmic: 90) but is SLOWER, 207 -> 238 us:
    // the resident workgroups then write to all 768 planes at once.
    const int r0 = blockIdx.x * rows_per_block, r1 = min(Lq, r0 + rows_per_block);
    const int cls = blockIdx.y % period, m0 = (blockIdx.y / period) * G;
    const int p0 = cls + period * m0;                          // first plane of the group
    if (r0 >= r1 || p0 >= BH) return;
    const int total = (r1 - r0) * nb;
    const int tstride = (rows_per_block * nb * BYTES + 15) / 16 * 16;          // bytes of one table buffer
    const int64_t plane = (int64_t)Lq * Lk;

    // lookup rows of one plane: requested into registers (NS per thread: the host keeps rows * nb <= NS * threads),
    // written to the other LDS buffer after the gathers of the current plane — the round trip hides behind them
    constexpr int NS = 4;
    E stg[NS];
    auto request = [&](int p) {
        const int b = p / H, h = p - b * H;
        const E* src = in + (int64_t)b * s0 + (int64_t)h * s1 + (int64_t)r0 * s2;
        if (s3 == 1 && s2 == nb) {
#pragma unroll
            for (int u = 0; u < NS; ++u) { const int t = tid + u * NT; if (t < total) stg[u] = src[t]; }
        } else {
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                const int t = tid + u * NT;
                if (t < total) { const int r = t / nb, c = t - r * nb; stg[u] = src[(int64_t)r * s2 + (int64_t)c * s3]; }
            }
        }
    };
    auto commit = [&](int buf) {
        E* table = reinterpret_cast<E*>(smem + buf * tstride);
#pragma unroll
        for (int u = 0; u < NS; ++u) { const int t = tid + u * NT; if (t < total) table[t] = stg[u]; }
    };
    request(p0);
    commit(0);

    // flat element range of this row block inside a plane and its 16-byte aligned interior (same for the whole group)
    const int f0 = r0 * Lk, f1 = r1 * Lk;
    const int mis = (int)((reinterpret_cast<uintptr_t>(y + (int64_t)p0 * plane + f0) & 15) / BYTES);
    const int fa = min(f1, f0 + ((V - mis) % V));
    const int nvec = (f1 - fa) / V;
    const int ft = fa + nvec * V;
    const __amdgpu_buffer_rsrc_t rs_idx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<int32_t*>(idx), 0, (int)((int64_t)Lq * Lk * 4), 0x00020000);
    // LDS byte offsets (inside one table buffer) of the values of this thread's vectors
    uint32_t toff[NV][V];
    {
        constexpr int NW = V / 4 + (V < 4);
        const int stepq = (NT * V) / Lk, stepr = (NT * V) % Lk;
        int f = fa + tid * V;
        int i = f / Lk, j = f - i * Lk;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            int32_t ids[V];
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const u32x4 t4 = __builtin_amdgcn_raw_buffer_load_b128(rs_idx, (f + 4 * w) * 4, 0, 0);   // 0 past the end
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * w + e < V) ids[4 * w + e] = (int32_t)t4[e];
            }
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int over = (j + e >= Lk) ? 1 : 0;        // the vector may run into the next query row
                toff[k][e] = (uint32_t)(((i + over - r0) * nb + ids[e]) * BYTES);
            }
            f += NT * V;
            i += stepq;
            j += stepr;
            if (j >= Lk) { j -= Lk; ++i; }
        }
    }
    // the <= 2 (V - 1) edge elements of the block: ids kept as table indices too
    const int nhead = fa - f0, ntail = f1 - ft;
    int edge = -1, edge_f = 0;
    if (tid < nhead) edge_f = f0 + tid;
    else if (tid - nhead < ntail) edge_f = ft + (tid - nhead);
    if (tid < nhead + ntail) edge = (edge_f / Lk - r0) * nb + idx[edge_f];

    for (int g = 0; g < G; ++g) {
        const int p = p0 + period * g;
        if (p >= BH) break;
        lds_barrier();                                         // table g staged; table g - 1 no longer read
        const bool more = g + 1 < G && p + period < BH;
        if (more) request(p + period);
        const unsigned char* tb = smem + (g & 1) * tstride;
        E* out = y + (int64_t)p * plane;
        if (edge >= 0) out[edge_f] = reinterpret_cast<const E*>(tb)[edge];
        E* ov = out + fa + tid * V;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (tid + k * NT < nvec) {
                union { u32x4 vec; E e[V]; } pk;
#pragma unroll
                for (int e = 0; e < V; ++e) pk.e[e] = *reinterpret_cast<const E*>(tb + toff[k][e]);
                // 16-byte aligned; nt: +5 % for bf16 (4.29 -> 4.50 TB/s), neutral for fp32; sc1 / sc0 sc1 (line dropped from L2 at once,
                // which would keep the id matrix resident) run at 2.7 TB/s — profiles/r05_rpe_gather_policy_sweep.txt
                __builtin_nontemporal_store(pk.vec, reinterpret_cast<u32x4*>(ov + (int64_t)k * NT * V));
            }
        }
        if (more) commit((g + 1) & 1);
    }
}

// (Round 5, measured and dropped — profiles/r05_rpe_gather.md: a "frontier" kernel in which every workgroup owns a fixed slice of
//  EVERY plane and the whole chip writes `period` neighbouring planes at a time, ids once per launch in registers: 3.4-4.1 TB/s; with
//  the gathers and the table staging compiled out its bare store stream — 1 KB per wave and plane, then a jump of `period` planes —
//  still needs 228-267 us where a fill of the buffer needs 149.  What the fill has is not the frontier but short-lived workgroups that
//  each write one contiguous 16 KB piece and leave.)
//
// (Also measured and dropped: "tiles" — the fill's own pattern, one short-lived 256-thread workgroup per contiguous 16 KB piece of one
//  plane, ids from a byte copy of the id matrix (1 byte per element from L2, one aligned V-byte load per vector): bit-identical,
//  4.1 TB/s fp32 / 3.2 bf16.  A workgroup that has to LOAD before it can store lives ~9 us; the fill's lives ~1.  Both kernels are kept
//  in tools/probes/rpe_probe.hip.)
//
// Shape-generic fallback (very long rows or very many buckets): one thread per output
// element, index re-read from L2.  Correct for every shape; not the measured path.
template <int BYTES>
__global__ __launch_bounds__(256) void rpe_gather_generic(
    typename raw_elem<BYTES>::type* __restrict__ y,
    const typename raw_elem<BYTES>::type* __restrict__ in,
    const int32_t* __restrict__ idx,
    int64_t rows, int H, int Lq, int Lk,
    int64_t s0, int64_t s1, int64_t s2, int64_t s3)
{
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const int i = (int)(r % Lq);
        const int64_t p = r / Lq;
        const int h = (int)(p % H);
        const int64_t b = p / H;
        const auto* src = in + b * s0 + (int64_t)h * s1 + (int64_t)i * s2;
        const int32_t* irow = idx + (int64_t)i * Lk;
        auto* out = y + r * Lk;
        for (int j = threadIdx.x; j < Lk; j += blockDim.x)
            out[j] = src[(int64_t)irow[j] * s3];
    }
}

// ------------------------------------------------------------------------------------
// backward: deterministic scatter-add, lane <-> (b,h) plane
// ------------------------------------------------------------------------------------
template <typename T> struct cvt;
template <> struct cvt<float> {
    using acc = float;
    static __device__ __forceinline__ float load(float x) { return x; }
    static __device__ __forceinline__ float store(float x) { return x; }
};
template <> struct cvt<double> {
    using acc = double;
    static __device__ __forceinline__ double load(double x) { return x; }
    static __device__ __forceinline__ double store(double x) { return x; }
};
template <> struct cvt<__half> {
    using acc = float;
    static __device__ __forceinline__ float load(__half x) { return __half2float(x); }
    static __device__ __forceinline__ __half store(float x) { return __float2half_rn(x); }
};
template <> struct cvt<hip_bfloat16> {
    using acc = float;
    static __device__ __forceinline__ float load(hip_bfloat16 x) { return static_cast<float>(x); }
    static __device__ __forceinline__ hip_bfloat16 store(float x) { return hip_bfloat16(x); }
};

// element-sized buffer load/store helpers (bounds-checked by the descriptor)
template <int BYTES> struct bufop;
template <> struct bufop<2> {
    static __device__ __forceinline__ uint16_t ld(__amdgpu_buffer_rsrc_t r, int off) {
        return (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(r, off, 0, 0);
    }
    static __device__ __forceinline__ void st(uint16_t v, __amdgpu_buffer_rsrc_t r, int off) {
        __builtin_amdgcn_raw_buffer_store_b16((short)v, r, off, 0, 0);
    }
};
template <> struct bufop<4> {
    static __device__ __forceinline__ uint32_t ld(__amdgpu_buffer_rsrc_t r, int off) {
        return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0);
    }
    static __device__ __forceinline__ void st(uint32_t v, __amdgpu_buffer_rsrc_t r, int off) {
        __builtin_amdgcn_raw_buffer_store_b32((int)v, r, off, 0, 0);
    }
};
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <> struct bufop<8> {
    static __device__ __forceinline__ uint64_t ld(__amdgpu_buffer_rsrc_t r, int off) {
        union { u32x2 v; uint64_t x; } c;
        c.v = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
        return c.x;
    }
    static __device__ __forceinline__ void st(uint64_t v, __amdgpu_buffer_rsrc_t r, int off) {
        union { u32x2 v; uint64_t x; } c;
        c.x = v;
        __builtin_amdgcn_raw_buffer_store_b64(c.v, r, off, 0, 0);
    }
};

// One 64-thread workgroup = one wave = 64 consecutive planes x a range of query rows.
// CT = keys per LDS tile (128-byte tile rows).  Tile rows are padded by 16 bytes so that
// both the row-wise 16-byte writes and the plane-wise 16-byte reads are bank-conflict-free;
// bins use a 65-column pitch so that the accumulate (lane-contiguous) and the transposed
// init/flush (bucket-contiguous) are conflict-free too.  The bucket ids of a tile are
// fetched with ONE coalesced load and broadcast per key with v_readlane (wave-uniform).
// All HBM traffic goes through two per-(wave,row) buffer descriptors whose extent ends
// at the last valid plane: planes past BH read as zero / drop their stores in hardware,
// so the loops carry no per-lane bounds branches.  PF tiles are kept in flight in
// registers (the wave is the only latency-hiding unit it has: LDS limits a CU to ~7 of
// these waves).
template <typename T, int CT, int PF, bool ACCUM>
__global__ __launch_bounds__(64) void rpe_scatter_planes(
    T* __restrict__ gin, const T* __restrict__ gout, const int32_t* __restrict__ idx,
    int BH, int Lq, int Lk, int nb, int rows_per_block)
{
    using ACC = typename cvt<T>::acc;
    constexpr int BYTES = sizeof(T);
    using E = typename raw_elem<BYTES>::type;
    constexpr int V = 16 / BYTES;                 // elements per 16-byte vector
    constexpr int LPR = CT / V;                   // lanes that cover one tile row (8)
    constexpr int RPI = WAVE / LPR;               // tile rows per wave-instruction (8)
    constexpr int NLD = WAVE / RPI;               // wave-instructions per tile (8)
    constexpr int TP = CT + V;                    // tile pitch in elements (16-byte pad)
    constexpr int BP = WAVE + 1;                  // bin pitch
    static_assert(CT <= WAVE, "one index load per tile");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* tile = reinterpret_cast<T*>(smem);                                   // [64][TP]
    ACC* bins = reinterpret_cast<ACC*>(smem + (size_t)WAVE * TP * BYTES);   // [nb][BP]

    const int lane = threadIdx.x;
    const int pbase = blockIdx.x * WAVE;           // first plane of this wave
    const int nplanes = min(WAVE, BH - pbase);
    const int i0 = blockIdx.y * rows_per_block;
    const int i1 = min(Lq, i0 + rows_per_block);

    const int lr = lane / LPR;                     // tile row this lane loads (per instruction)
    const int lc = (lane % LPR) * V;               // first key of this lane's vector
    const int nchunk = (Lk + CT - 1) / CT;
    const int64_t plane_go = (int64_t)Lq * Lk;     // plane pitch of gout (elements)
    const int64_t plane_gi = (int64_t)Lq * nb;     // plane pitch of gin

    union Pack { u32x4 vec; T e[V]; E raw[V]; };
    struct Stage { Pack p[NLD]; int32_t ids; };

    for (int i = i0; i < i1; ++i) {
        // descriptors: base = (first plane of the wave, row i); extent = EXACTLY up to the
        // end of row i of the last valid plane (never over-reads the tensor).  Planes past
        // BH are out of range: their loads return 0 and their stores are dropped.
        const T* go_base = gout + ((int64_t)pbase * Lq + i) * Lk;
        T* gi_base = gin + ((int64_t)pbase * Lq + i) * nb;
        const int go_bytes = (int)(((int64_t)(nplanes - 1) * plane_go + Lk) * BYTES);
        const int gi_bytes = (int)(((int64_t)(nplanes - 1) * plane_gi + nb) * BYTES);
        const __amdgpu_buffer_rsrc_t rs_go = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<T*>(go_base), 0, go_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_gi = __builtin_amdgcn_make_buffer_rsrc(
            gi_base, 0, gi_bytes, 0x00020000);

        auto load_tile = [&](Stage& st, int ch) {
            const int c0 = ch * CT;
            st.ids = (lane < CT && c0 + lane < Lk) ? idx[(int64_t)i * Lk + c0 + lane] : 0;
            if (c0 + CT <= Lk) {
#pragma unroll
                for (int t = 0; t < NLD; ++t)
                    st.p[t].vec = __builtin_amdgcn_raw_buffer_load_b128(
                        rs_go, (int)(((int64_t)(t * RPI + lr) * plane_go + c0 + lc) * BYTES), 0, 0);
            } else {
                // tail tile of the row: 16-byte loads only where the whole vector lies inside
                // the row; the straddling lane reads element-wise.  (The hardware range check
                // is per dword: a dword crossing the end of the descriptor is dropped whole,
                // which would lose the last 2-byte element of the last plane.)
#pragma unroll
                for (int t = 0; t < NLD; ++t) {
                    const int off = (int)(((int64_t)(t * RPI + lr) * plane_go + c0 + lc) * BYTES);
                    if (c0 + lc + V <= Lk) {
                        st.p[t].vec = __builtin_amdgcn_raw_buffer_load_b128(rs_go, off, 0, 0);
                    } else {
#pragma unroll
                        for (int v = 0; v < V; ++v)
                            st.p[t].raw[v] = (c0 + lc + v < Lk)
                                ? bufop<BYTES>::ld(rs_go, off + v * BYTES) : E(0);
                    }
                }
            }
        };

        auto consume = [&](const Stage& st, int ch) {
            const int c0 = ch * CT;
#if defined(RPE_SCATTER_ABLATE) && RPE_SCATTER_ABLATE >= 2   // probe switch (tools/probes/scatter_probe.hip): loads only
            {
                ACC* slot = bins + lane;
                ACC sum = *slot;
#pragma unroll
                for (int t = 0; t < NLD; ++t)
#pragma unroll
                    for (int v = 0; v < V; ++v) sum += (ACC)cvt<T>::load(st.p[t].e[v]);
                *slot = sum + (ACC)st.ids;
                return;
            }
#endif
            // registers -> LDS tile (row-wise, 16-byte writes); keys past the row end were
            // loaded as zero
#pragma unroll
            for (int t = 0; t < NLD; ++t)
                *reinterpret_cast<u32x4*>(tile + (t * RPI + lr) * TP + lc) = st.p[t].vec;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // plane-wise read-back: lane = plane, ascending key order, G keys per group.
            // Each lane owns column `lane` of the bins, so a plain read-modify-write is
            // race-free; the G bin reads of a group are issued together (one LDS round
            // trip per group) and a key whose bucket already occurred in the group takes
            // the running sum of that earlier key instead of the stale read — the compare
            // is wave-uniform.  The writes retire in order, so the last sum of a bucket
            // wins.  Net effect: exactly bin = ((bin + g[j0]) + g[j1]) + ... in ascending j.
            // (Keys past Lk in the last tile are zero with bucket id 0; x + 0.0 is exact.)
#pragma unroll
            for (int q = 0; q < LPR; ++q) {
                Pack pk;
                pk.vec = *reinterpret_cast<const u32x4*>(tile + lane * TP + q * V);
#ifdef RPE_SCATTER_GROUP
                constexpr int G = V < RPE_SCATTER_GROUP ? V : RPE_SCATTER_GROUP;       // probe (tools/probe_irpe_variants.sh)
#else
                // keys per read-modify-write group.  Same-call A/B at config 4 (round 6): 8 / 4 / 2 / 1 -> bf16 230 / 203 / 195 / 203 us,
                // fp32 272 / 268 / 275 / 263 (noise): the kernel is not bound by this chain — two for 16-bit elements, four otherwise
                constexpr int G = BYTES == 2 ? 2 : (V < 4 ? V : 4);
#endif
#pragma unroll
                for (int g0 = 0; g0 < V; g0 += G) {
                    int u[G];
                    ACC* slot[G];
                    ACC sum[G];
#pragma unroll
                    for (int v = 0; v < G; ++v) {
                        u[v] = __builtin_amdgcn_readlane(st.ids, q * V + g0 + v);
                        slot[v] = bins + u[v] * BP + lane;
                        sum[v] = *slot[v];
                    }
#pragma unroll
                    for (int v = 0; v < G; ++v) {
                        ACC base = sum[v];
#pragma unroll
                        for (int w = 0; w < v; ++w)
                            base = (u[w] == u[v]) ? sum[w] : base;  // latest earlier match wins
                        sum[v] = base + (ACC)cvt<T>::load(pk.e[g0 + v]);
                    }
#pragma unroll
                    for (int v = 0; v < G; ++v) *slot[v] = sum[v];
                }
            }
            __builtin_amdgcn_wave_barrier();
        };

        // first tiles of this row start flying before the bins are (re)initialised
        Stage stage[PF];
#pragma unroll
        for (int k = 0; k < PF; ++k)
            if (k < nchunk) load_tile(stage[k], k);

        if (ACCUM) {
            // bins <- current gin rows (the reference accumulates INTO grad_input).
            // Columns of planes beyond BH read as zero and only ever receive zeros.
            for (int ub = 0; ub < nb; ub += WAVE) {
                const int u = ub + lane;
#pragma unroll 16
                for (int r = 0; r < WAVE; ++r) {
                    union { E raw; T val; } c;
                    c.raw = (u < nb) ? bufop<BYTES>::ld(rs_gi, (int)((r * plane_gi + u) * BYTES)) : E(0);
                    if (u < nb) bins[u * BP + r] = cvt<T>::load(c.val);
                }
            }
        } else {
            for (int u = lane; u < nb * BP; u += WAVE) bins[u] = ACC(0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        for (int ch = 0; ch < nchunk; ch += PF) {
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                if (ch + k < nchunk) {
                    consume(stage[k], ch + k);
                    if (ch + k + PF < nchunk) load_tile(stage[k], ch + k + PF);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // flush: one coalesced nb-element row per plane
        for (int ub = 0; ub < nb; ub += WAVE) {
            const int u = ub + lane;
            if (u < nb) {
#if defined(RPE_SCATTER_ABLATE) && RPE_SCATTER_ABLATE == 3   // probe switch: loads only and (almost) no flush
                for (int r = 0; r < 1; ++r) {
#else
#pragma unroll 8
                for (int r = 0; r < WAVE; ++r) {
#endif
                    union { E raw; T val; } c;
                    c.val = cvt<T>::store(bins[u * BP + r]);
                    bufop<BYTES>::st(c.raw, rs_gi, (int)((r * plane_gi + u) * BYTES));
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Fallback when nb is too large for the LDS bin array: one thread per (row, bucket),
// sequential over j (same fixed order).  O(nb*Lk) per row — exotic shapes only.
template <typename T>
__global__ __launch_bounds__(256) void rpe_scatter_generic(
    T* __restrict__ gin, const T* __restrict__ gout, const int32_t* __restrict__ idx,
    int64_t rows, int Lq, int Lk, int nb)
{
    using ACC = typename cvt<T>::acc;
    const int64_t total = rows * nb;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int u = (int)(t % nb);
        const int64_t r = t / nb;
        const int i = (int)(r % Lq);
        const int32_t* irow = idx + (int64_t)i * Lk;
        const T* g = gout + r * Lk;
        ACC acc = cvt<T>::load(gin[t]);
        for (int j = 0; j < Lk; ++j)
            if (irow[j] == u) acc += cvt<T>::load(g[j]);
        gin[t] = cvt<T>::store(acc);
    }
}

// ------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------
inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// hipGetLastError() is sticky per host thread: drop whatever an earlier, unrelated HIP
// call left behind before launching, so that the code we return is about OUR launch.
inline void clear_stale_error() { (void)hipGetLastError(); }
inline int launch_status() { return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH; }

template <int BYTES>
int launch_gather(void* y, const void* in, const int32_t* idx, int B, int H, int Lq, int Lk,
                  int nb, int64_t s0, int64_t s1, int64_t s2, int64_t s3, hipStream_t st)
{
    using E = typename raw_elem<BYTES>::type;
    constexpr int V = 16 / BYTES;
    clear_stale_error();
    const int BH = B * H;
    // rows of one plane per workgroup: as many as fit 60 KB of LDS, fewer when there are few planes
    int rpb = std::min<int64_t>(Lq, 61440 / ((int64_t)nb * BYTES));
    if (rpb < 1 || Lk < V) {
        const int64_t rows = (int64_t)BH * Lq;
        const int grid = (int)std::min<int64_t>(rows, 256 * 16);
        hipLaunchKernelGGL((rpe_gather_generic<BYTES>), dim3(grid), dim3(256), 0, st,
                           (E*)y, (const E*)in, idx, rows, H, Lq, Lk, s0, s1, s2, s3);
        return launch_status();
    }
    // measured path: several planes per workgroup (ids fetched once per group), tools/probes/rpe_probe.hip on config 4:
    //   fp32  NV = 4, G = 2, 1024 threads  5.4 TB/s   (one plane per workgroup: 4.2-4.6)
    //   bf16  NV = 4, G = 8,  768 threads  4.5 TB/s   (3.3)
    {
        constexpr int NV = 4, NS = 4;
        // (round 5 re-sweep in the product path, profiles/r05_rpe_gather_sweep.txt: G 2..32 x 512 / 768 / 1024 threads — fp32
        //  0.557-0.604 of the HBM peak with G = 2, 1024 threads at 0.596; bf16 best at G = 8, 768 threads: the choice stands)
        const int thr = BYTES == 2 ? 768 : 1024, G = BYTES == 2 ? 8 : 2;
        int64_t rows = std::min<int64_t>(Lq, (int64_t)NV * thr * V / Lk);      // NV vectors per thread cover the row block
        rows = std::min<int64_t>(rows, (int64_t)NS * thr / nb);                 // NS staged lookup values per thread
        rows = std::min<int64_t>(rows, 30720 / ((int64_t)nb * BYTES));          // two table buffers in 60 KB
        if (rows >= 1) {
            int nblk = ceil_div(Lq, rows);
            rows = ceil_div(Lq, nblk);
            int period = 1;                                     // planes p, p + period, ... share their 16-byte alignment
            while (((int64_t)period * Lq * Lk * BYTES) % 16) ++period;
            const int members = ceil_div(BH, period), groups = ceil_div(members, G);
            const size_t lds = 2 * (((size_t)rows * nb * BYTES + 15) / 16 * 16);
            if ((int64_t)period * groups <= 65535) {
                hipLaunchKernelGGL((rpe_gather_planes<BYTES, NV>), dim3(nblk, period * groups), dim3(thr), lds, st, (E*)y,
                                   (const E*)in, idx, BH, H, Lq, Lk, nb, s0, s1, s2, s3, (int)rows, period, G);
                return launch_status();
            }
        }
    }
    const int nthreads = 1024;
    int nblk = ceil_div(Lq, rpb);
    rpb = ceil_div(Lq, nblk);                                   // equal blocks
    const size_t lds = (size_t)rpb * nb * BYTES;
    hipLaunchKernelGGL((rpe_gather_plane<BYTES>), dim3(nblk, BH), dim3(nthreads), lds, st, (E*)y, (const E*)in, idx, H, Lq, Lk, nb,
                       s0, s1, s2, s3, rpb);
    return launch_status();
}

template <typename T>
int launch_scatter(void* gin, const void* gout, const int32_t* idx, int B, int H, int Lq,
                   int Lk, int nb, bool accumulate, hipStream_t st)
{
    using ACC = typename cvt<T>::acc;
    constexpr int CT = 128 / (int)sizeof(T);     // 128-byte tile rows for every element size
    clear_stale_error();
    constexpr int V = 16 / (int)sizeof(T);
    const int BH = B * H;
    const size_t lds = (size_t)WAVE * (CT + V) * sizeof(T) + (size_t)nb * (WAVE + 1) * sizeof(ACC);
    // the fast path addresses 64 planes through 32-bit buffer offsets
    const bool fits32 = (int64_t)WAVE * Lq * std::max(Lk, nb) * (int64_t)sizeof(T) < (int64_t)INT32_MAX;
    if (lds > 64 * 1024 || !fits32) {
        const int64_t rows = (int64_t)BH * Lq;
        const int grid = (int)std::min<int64_t>(ceil_div(rows * nb, 256), 256 * 32);
        if (!accumulate) {
            if (hipMemsetAsync(gin, 0, (size_t)rows * nb * sizeof(T), st) != hipSuccess)
                return CREAM_ERR_LAUNCH;
        }
        hipLaunchKernelGGL((rpe_scatter_generic<T>), dim3(grid), dim3(256), 0, st, (T*)gin,
                           (const T*)gout, idx, rows, Lq, Lk, nb);
        return launch_status();
    }
    const int gx = ceil_div(BH, WAVE);
    // enough workgroups to give every CU several waves
    int want_y = std::max(1, (256 * 8) / gx);
    int rpb = std::max(1, ceil_div(Lq, want_y));
    const int gy = ceil_div(Lq, rpb);
    constexpr int PF = 2;
    if (accumulate)
        hipLaunchKernelGGL((rpe_scatter_planes<T, CT, PF, true>), dim3(gx, gy), dim3(WAVE), lds, st,
                           (T*)gin, (const T*)gout, idx, BH, Lq, Lk, nb, rpb);
    else
        hipLaunchKernelGGL((rpe_scatter_planes<T, CT, PF, false>), dim3(gx, gy), dim3(WAVE), lds, st,
                           (T*)gin, (const T*)gout, idx, BH, Lq, Lk, nb, rpb);
    return launch_status();
}

}  // namespace

extern "C" {

int cream_rpe_index_fwd(void* y, const void* in, const int32_t* idx, int B, int H, int Lq,
                        int Lk, int nb, int64_t s0, int64_t s1, int64_t s2, int64_t s3,
                        int dtype, void* stream)
{
    if (B < 0 || H < 0 || Lq < 0 || Lk < 0 || nb < 0) return CREAM_ERR_BAD_ARG;
    if ((int64_t)B * H * Lq * Lk == 0) return CREAM_OK;            // empty output
    if (!y || !in || !idx || nb == 0) return CREAM_ERR_BAD_ARG;
    if ((int64_t)B * H > INT32_MAX) return CREAM_ERR_TOO_LARGE;
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case CREAM_F32: return launch_gather<4>(y, in, idx, B, H, Lq, Lk, nb, s0, s1, s2, s3, st);
        case CREAM_F16:
        case CREAM_BF16: return launch_gather<2>(y, in, idx, B, H, Lq, Lk, nb, s0, s1, s2, s3, st);
        case CREAM_F64: return launch_gather<8>(y, in, idx, B, H, Lq, Lk, nb, s0, s1, s2, s3, st);
        default: return CREAM_ERR_BAD_DTYPE;
    }
}

int cream_rpe_index_bwd(void* gin, const void* gout, const int32_t* idx, int B, int H, int Lq,
                        int Lk, int nb, int dtype, int accumulate, void* stream)
{
    if (B < 0 || H < 0 || Lq < 0 || Lk < 0 || nb < 0) return CREAM_ERR_BAD_ARG;
    if ((int64_t)B * H * Lq * nb == 0) return CREAM_OK;            // empty grad_input
    if (!gin) return CREAM_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if ((int64_t)Lk == 0) {                                         // nothing to add
        if (!accumulate) {
            const size_t es = dtype == CREAM_F64 ? 8 : dtype == CREAM_F32 ? 4 : 2;
            if (hipMemsetAsync(gin, 0, (size_t)B * H * Lq * nb * es, st) != hipSuccess)
                return CREAM_ERR_LAUNCH;
        }
        return CREAM_OK;
    }
    if (!gout || !idx) return CREAM_ERR_BAD_ARG;
    if ((int64_t)B * H > INT32_MAX) return CREAM_ERR_TOO_LARGE;
    const bool acc = accumulate != 0;
    switch (dtype) {
        case CREAM_F32: return launch_scatter<float>(gin, gout, idx, B, H, Lq, Lk, nb, acc, st);
        case CREAM_F16: return launch_scatter<__half>(gin, gout, idx, B, H, Lq, Lk, nb, acc, st);
        case CREAM_BF16: return launch_scatter<hip_bfloat16>(gin, gout, idx, B, H, Lq, Lk, nb, acc, st);
        case CREAM_F64: return launch_scatter<double>(gin, gout, idx, B, H, Lq, Lk, nb, acc, st);
        default: return CREAM_ERR_BAD_DTYPE;
    }
}

}  // extern "C"
