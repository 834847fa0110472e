// gemm_mfma.hip — C entry points of the hand-written MFMA GEMMs (kernels: gemm_mfma.hpp).
//
// Reference semantics: LinearSuper.forward / qkv_super.forward = F.linear on the active block
// W[:out, :in] of the super weight (AutoFormer/model/module/Linear_super.py:38-54, :71-81;
// qkv_super.py:45-55, :72-83), erf-GELU between fc1 and fc2 (supernet_transformer.py:14-16,
// :275-285) and what autograd derives for them.  The weight operands are bf16 COPIES of the fp32
// master weights laid out for these kernels by cream_adamw_step (csrc/optim.hip):
//   * W    (out x in, ld = super in)        forward operand, read in place as W[:N, :K];
//   * W^T  (in x out, ld = super out)       dgrad operand (makes dgrad the same K-contiguous product);
//   * qkv: the interleaved super weight (row 3 i + j = output i of q/k/v part j, qkv_super.py:75)
//     de-interleaved ONCE per optimizer step into three (Qmax x in) matrices [q | k | v] and their
//     transposes — the sampled row gather of every forward becomes plain segment addressing.
//
// Tile choice (measured on the MI355X, profiles/r02_gemm_probe.txt): 128x128 tiles (2 workgroups
// per CU) for wide outputs, 128x64 (3 per CU) for N < 640 where 128-wide tiles leave CUs idle in
// the last round (M = 197 x 128 rows against 256 CUs).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>

#include "cream_amd.h"
#include "gemm_mfma.hpp"
#include "gemm_nt8.hpp"
#include "gemm_tn8.hpp"
#include "launch_ev.hpp"
#include "cu_budget.hpp"

namespace {
std::atomic<int> g_cu_reserve{0};
std::atomic<int> g_layout_epoch{0};      // bumped whenever a switch changes cream_linear_wgrad_splits_bf16 (= the backward workspace layout)
int physical_cus()
{
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}
}  // namespace
namespace cream {
int cu_count()
{
    int c = (physical_cus() - g_cu_reserve.load(std::memory_order_relaxed)) / 8 * 8;
    return c < 8 ? 8 : c;
}
}  // namespace cream

namespace {
using namespace cream;
using namespace cream::gemm;

int num_cus() { return cream::cu_count(); }

// persistent launch: as many workgroups as the chip holds at once (OCC per CU, a multiple of 8 so that a
// workgroup's tiles stay on its XCD), or one per tile if there are fewer tiles.  Same-box A/B of the step with
// the same kernel launched one tile per workgroup: 10.97 -> 10.86 ms; per GEMM against the previous kernel
// 3-8 % (profiles/r02_gemm_probe.txt)
// 256 x 256 macro tile (8 waves, 128 x 64 wave tiles, 128 KB of LDS, one persistent workgroup per CU) for the wide
// outputs (qkv, fc1, fc2 dgrad: N >= 960): half the L2 -> LDS bytes per flop of the 128 x 128 tile.
// CREAM_GEMM_NT256 in the environment / cream_gemm_nt256() switch it (A/B runs); default: see nt256_mode().
std::atomic<int> g_nt256{-1};
int nt256_mode()
{
    int m = g_nt256.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = getenv("CREAM_GEMM_NT256");
        // default 2: with COLD operands (rotating buffer sets, profiles/r04_gemm_probe_cold.txt) the macro tile is 6-14 % faster
        // than the 128-wide tiles where the contraction is long (fc2, fc1 dgrad, qkv dgrad: K = 1152 .. 1792, N = E) and
        // equal or slower on the wide-output shapes (mode 1) — the warm-buffer probe of the same kernels had it the other way
        // round.  Same-box A/B of the step, alternating, twice: 9.455 / 9.502 -> 9.351 / 9.402 ms (-1.0 %).
        m = e ? atoi(e) : 2;
        g_nt256.store(m, std::memory_order_relaxed);
    }
    return m;
}

// Round 6: the two-stage kernels with their tile epilogue off the memory counters (gemm_mfma.hpp, "OPT"): asm LDS-DMA, LDS-only
// epilogue barriers, side inputs requested under the first K-step, counted wait for the first K-step behind an epilogue
// (bit 0), and gelu / gelu' of the fc1 epilogue from a 16 KB LDS table instead of ~31 VALU instructions per element (bit 1).
// Outputs are bit-identical to the OPT = 0 kernels (test_nt_epilogue_variants_are_bit_identical, exhaustive over the bf16
// values for the table).  CREAM_GEMM_NTOPT in the environment / cream_gemm_ntopt(): 0 = off, 1 = bit 0, 3 = both (default).
std::atomic<int> g_ntopt{-1};
int ntopt_mode()
{
    int m = g_ntopt.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = getenv("CREAM_GEMM_NTOPT");
        m = e ? atoi(e) : 3;
        g_ntopt.store(m, std::memory_order_relaxed);
    }
    return m;
}
// CREAM_GEMM_STAGGER / cream_gemm_stagger(): the second half of a two-workgroups-per-CU grid starts n x 64 clocks late
std::atomic<int> g_stagger{-1};
int stagger_mode()
{
    int m = g_stagger.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = getenv("CREAM_GEMM_STAGGER");
        m = e ? atoi(e) : 0;
        g_stagger.store(m, std::memory_order_relaxed);
    }
    return m;
}

template <int EPI>
int launch_nt256(const NtParams& p, hipStream_t st)
{
    constexpr int BM = 256, BN = 256;
    // (bit 2: the macro tiles too — cold probe: 256 x 192 +-1 %, 256 x 256 +12 % SLOWER on fc2 E448: not in the default)
    auto kern = (ntopt_mode() & 4) ? gemm_nt_kernel<BM, BN, 2, 4, 2, EPI, 1, 1> : gemm_nt_kernel<BM, BN, 2, 4, 2, EPI, 1>;
    constexpr int lds = nt_lds_bytes(BM, BN, 2);
    if (!cream::raise_dynamic_lds(kern, lds)) return CREAM_ERR_LAUNCH;
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN), slots = num_cus() / 8 * 8;
    CREAM_LAUNCH(kern, dim3(tiles < slots ? tiles : slots), dim3(512), lds, st, p);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

// the counted-vmcnt, phase-interleaved kernel (gemm_nt8.hpp): 256 x 256 tiles, one persistent 8-wave workgroup per CU.
// CREAM_GEMM_NT8 in the environment / cream_gemm_nt8(): 0 = never, 1 = wherever its limits allow, 2 = by shape (nt8_wanted)
std::atomic<int> g_nt8{-1};
int nt8_mode()
{
    int m = g_nt8.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = getenv("CREAM_GEMM_NT8");
        m = e ? atoi(e) : 4;
        g_nt8.store(m, std::memory_order_relaxed);
    }
    return m;
}

bool nt8_fits(const NtParams& p)
{
    // 31-bit element offsets inside the kernel
    const int64_t amax = (int64_t)p.M * p.lda, bmax = (int64_t)2 * p.nseg_stride + (int64_t)p.nseg * p.ldb + p.ldb;
    const int64_t bplain = (int64_t)p.N * p.ldb;
    return amax < ((int64_t)1 << 31) && bmax < ((int64_t)1 << 31) && bplain < ((int64_t)1 << 31) && p.lda < (1 << 24) && p.ldb < (1 << 24);
}

template <int EPI>
int launch_nt8(const NtParams& p, hipStream_t st)
{
    auto kern = gemm_nt8_kernel<EPI>;
    if (!cream::raise_dynamic_lds(kern, NT8_LDS_BYTES)) return CREAM_ERR_LAUNCH;
    // one persistent workgroup per CU (one tile per workgroup instead: 9.45 vs 9.45 ms per step, 192 workgroups: 10.2 —
    // profiles/r05_nt8_step_ab.txt)
    const int slots = num_cus() / 8 * 8;
    const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    CREAM_LAUNCH(kern, dim3(tiles < slots ? tiles : slots), dim3(512), NT8_LDS_BYTES, st, p);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

// Round 6: a 256 x 192 tile (4 x 2 waves of 64 x 96, 112 KB of LDS, one workgroup per CU) for the products whose output is E = 384 or
// 320 wide (proj, fc2, the fc1 / qkv / proj input gradients): two column tiles, 198 workgroups like the 256-wide tile, but 0 / 17 %
// padding instead of 25 / 37 %, five fragment reads per six MFMAs, 96 accumulator registers.  Cold probe
// (profiles/r06_nt_256x192.md): 14-21 % shorter than the kernel chosen before on every such shape, 0.78-1.01 x the vendor library.
// Measured next to it and dropped: 256 x 160 as 8 x 1 waves of 32 x 160 (80-column wave tiles are no multiple of the 32-wide MFMA
// block): six reads per five MFMAs want 300 B/clk from a 256 B/clk LDS — slower than every other tile; 256 x 224 the same way needs
// 98 more registers than a wave has.  E = 448 stays on the 256-wide tiles (12 % padding).
// Same contraction order per output element as every other tile: bit-identical results.
// CREAM_GEMM_NTHALF in the environment / cream_gemm_nthalf(): 0 = off, 1 = on (plain and bias epilogues, M >= 1024).
std::atomic<int> g_nthalf{-1};
int nthalf_mode()
{
    int m = g_nthalf.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = getenv("CREAM_GEMM_NTHALF");
        // default 1: same-call step A/B x3 (profiles/r06_nt_256x192.md): 8.945 / 8.952 / 8.903 -> 8.763 / 8.770 / 8.770 ms (-2.0 %)
        m = e ? atoi(e) : 1;
        g_nthalf.store(m, std::memory_order_relaxed);
    }
    return m;
}

template <int EPI, int BN, int WM, int WN>
int launch_nt_half(const NtParams& p, hipStream_t st)
{
    constexpr int BM = 256;
    auto kern = (ntopt_mode() & 4) ? gemm_nt_kernel<BM, BN, WM, WN, 2, EPI, 1, 1> : gemm_nt_kernel<BM, BN, WM, WN, 2, EPI, 1>;
    constexpr int lds = nt_lds_bytes(BM, BN, 2);
    if (!cream::raise_dynamic_lds(kern, lds)) return CREAM_ERR_LAUNCH;
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN), slots = num_cus() / 8 * 8;
    CREAM_LAUNCH(kern, dim3(tiles < slots ? tiles : slots), dim3(WM * WN * 64), lds, st, p);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

template <int EPI>
int launch_nt(const NtParams& p, hipStream_t st)
{
    if constexpr (EPI == EPI_STORE || EPI == EPI_BIAS) {
        if (nthalf_mode() > 0 && p.M >= 1024) {
            if (p.N == 384 || p.N == 320) return launch_nt_half<EPI, 192, 4, 2>(p, st);
        }
    }
    {
        // mode 2: where the cold probe has it ahead (profiles/r05_gemm_probe_cold.txt): every plain / bias product; the two
        // epilogues with element-wise work on the whole tile (GELU, x gelu') stay on the kernels whose second workgroup per CU
        // multiplies while the first one is in its epilogue — except on the long contractions, where the loop outweighs it
        const int m8 = nt8_mode();
        const bool light = EPI == EPI_STORE || EPI == EPI_BIAS;
        // mode 3: light epilogues, and only where at most 1/8 of the 256-wide column tiles is padding (CU-time, not wall time,
        // is what the two-stream step pays for)
        const int npad = (p.N + 255) / 256 * 256;
        // mode 4: the forward products only (bias epilogue: no weight-gradient stream runs beside them), same padding bound
        if ((m8 == 1 || (m8 == 2 && (light || p.K >= 1024)) || (m8 == 3 && light && (npad - p.N) * 8 <= npad) ||
             (m8 == 4 && EPI == EPI_BIAS && (npad - p.N) * 8 <= npad) ||
             // ... and every epilogue where BOTH the output and the contraction are at least 640 wide — no AutoFormer search space has
             // such a product (one side is always the embedding width, <= 624); DeiT-base / CLIP ViT-B blocks (768 / 2304 / 3072) are
             // nothing else: same-call A/B of the native DeiT-base-384 + iRPE step, mode 4 / 1: 36.7 / 35.4 ms (round 6)
             (m8 == 4 && p.K >= 640 && p.N >= 640)) && nt8_fits(p))
            return launch_nt8<EPI>(p, st);
    }
    {
        // mode 1: the wide outputs (N >= 960); mode 2: the LONG contractions (K >= 1152: fc2, fc1 dgrad, qkv dgrad at E >= 384 — N = E); mode 3: K >= 960
        const int m256 = nt256_mode();
        // (also measured: K >= 1152 plus the K = 448 shapes and qkv at E = 384, where the cold probe has the macro tile ahead: 9.47 / 9.45 ->
        //  9.50 / 9.51 ms per step — the step does not follow the probe there)
        if ((m256 == 1 && p.N >= 960) || (m256 == 2 && p.K >= 1152) || (m256 == 3 && p.K >= 960)) return launch_nt256<EPI>(p, st);
    }
    // bit 0 serves the FORWARD epilogues (bias, bias + GELU); the backward's (plain store, x gelu') only with bit 3: in the two-stream
    // backward a shorter main chain buys 0.4 % of the step and stretches the in-step durations of both streams' kernels by a fifth
    // (profiles/r06k_step_ab_kernel_timing.txt) — kept as a switch
    const int mode_ = ntopt_mode();
    const bool bwd_epi = EPI == EPI_MUL_COLSUM || EPI == EPI_STORE;
    const int opt = (bwd_epi && !(mode_ & 8)) ? 0 : mode_;
    if (p.N >= 640) {
        constexpr int BM = 128, BN = 128, OCC = 2;
        const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN), slots = OCC * num_cus() / 8 * 8;
        NtParams q = p;
        q.stagger = tiles > slots / 2 ? stagger_mode() : 0;
        if (EPI == EPI_BIAS_GELU && (opt & 3) == 3) {
            // + the 16 KB table: 2 x 80 KB = the CU's whole LDS, still two workgroups per CU
            auto kern = gemm_nt_kernel<BM, BN, 2, 2, 2, EPI, OCC, 3>;
            constexpr int lds = nt_lds_bytes(BM, BN, 2) + GELU_TAB_BYTES;
            if (!cream::raise_dynamic_lds(kern, lds)) return CREAM_ERR_LAUNCH;
            CREAM_LAUNCH(kern, dim3(tiles < slots ? tiles : slots), dim3(256), lds, st, q);
        } else if (opt & 1) {
            CREAM_LAUNCH((gemm_nt_kernel<BM, BN, 2, 2, 2, EPI, OCC, 1>), dim3(tiles < slots ? tiles : slots), dim3(256), 0, st, q);
        } else {
            CREAM_LAUNCH((gemm_nt_kernel<BM, BN, 2, 2, 2, EPI, OCC>), dim3(tiles < slots ? tiles : slots), dim3(256), 0, st, q);
        }
    } else {
        constexpr int BM = 128, BN = 64, OCC = 3;
        const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN), slots = OCC * num_cus() / 8 * 8;
        if (opt & 1) {
            CREAM_LAUNCH((gemm_nt_kernel<BM, BN, 2, 2, 2, EPI, OCC, 1>), dim3(tiles < slots ? tiles : slots), dim3(256), 0, st, p);
        } else {
            CREAM_LAUNCH((gemm_nt_kernel<BM, BN, 2, 2, 2, EPI, OCC>), dim3(tiles < slots ? tiles : slots), dim3(256), 0, st, p);
        }
    }
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// common argument checks of the NT products: C(M x N) = A(M x K) . B(N x K)^T
int check_nt(const void* out, const void* a, const void* b, int M, int N, int K, int64_t ldb, int kmin_ld)
{
    if (M < 0 || N <= 0 || K <= 0 || ldb < kmin_ld) return CREAM_ERR_BAD_ARG;
    if (N % 8 || K % 8 || ldb % 8) return CREAM_ERR_BAD_ARG;
    if (M == 0) return 1;                                       // empty problem: nothing to do
    if (!out || !a || !b || !aligned16(out) || !aligned16(a) || !aligned16(b)) return CREAM_ERR_BAD_ARG;
    return CREAM_OK;
}

NtParams plain(void* out, const void* a, const void* b, int M, int N, int K, int64_t ldb)
{
    NtParams p{};
    p.A = (const uint16_t*)a; p.lda = K;
    p.B = (const uint16_t*)b; p.ldb = ldb;
    p.nseg = N; p.kseg = K; p.nseg_stride = 0; p.kseg_stride = 0;
    p.M = M; p.N = N; p.K = K;
    p.out = (uint16_t*)out; p.ldo = N;
    return p;
}

}  // namespace

extern "C" {

int cream_gemm_rows_per_colsum_slab(void) { return 128; }

int cream_gemm_nt256(int on)
{
    const int prev = nt256_mode();
    if (on >= 0) g_nt256.store(on, std::memory_order_relaxed);
    return prev;
}

int cream_gemm_nthalf(int on)
{
    const int prev = nthalf_mode();
    if (on >= 0) g_nthalf.store(on != 0, std::memory_order_relaxed);
    return prev;
}

int cream_gemm_ntopt(int mode)
{
    const int prev = ntopt_mode();
    if (mode >= 0) g_ntopt.store(mode & 15, std::memory_order_relaxed);
    return prev;
}

int cream_gemm_stagger(int n)
{
    const int prev = stagger_mode();
    if (n >= 0) g_stagger.store(n, std::memory_order_relaxed);
    return prev;
}

int cream_gemm_nt8(int mode)
{
    const int prev = nt8_mode();
    if (mode >= 0) g_nt8.store(mode, std::memory_order_relaxed);
    return prev;
}

int cream_linear_fwd(void* out, const void* x, const void* w, const void* bias, int M, int N, int K, int64_t ldw,
                     void* stream)
{
    const int rc = check_nt(out, x, w, M, N, K, ldw, K);
    if (rc) return rc < 0 ? rc : CREAM_OK;
    NtParams p = plain(out, x, w, M, N, K, ldw);
    p.bias = (const uint16_t*)bias;
    return launch_nt<EPI_BIAS>(p, (hipStream_t)stream);
}

int cream_linear_fwd_seg(void* out, const void* x, const void* w, const void* bias, int M, int N, int K, int64_t ldw,
                         int nseg, int64_t nseg_stride, void* stream)
{
    const int rc = check_nt(out, x, w, M, N, K, ldw, K);
    if (rc) return rc < 0 ? rc : CREAM_OK;
    if (nseg <= 0 || nseg % 8 || (int64_t)3 * nseg < N || nseg_stride % 8) return CREAM_ERR_BAD_ARG;
    NtParams p = plain(out, x, w, M, N, K, ldw);
    p.bias = (const uint16_t*)bias;
    p.nseg = nseg; p.nseg_stride = nseg_stride;
    return launch_nt<EPI_BIAS>(p, (hipStream_t)stream);
}

int cream_linear_gelu_fwd_pad(void* gp, void* g, const void* x, const void* w, const void* bias, int M, int N, int Nvalid,
                              int K, int64_t ldw, void* stream)
{
    // gp == NULL: forward without a backward (evaluation, a frozen teacher): only gelu(h) is written — half the output bytes
    const int rc = check_nt(gp ? gp : g, x, w, M, N, K, ldw, K);
    if (rc) return rc < 0 ? rc : CREAM_OK;
    if (!g || !bias || !aligned16(g) || Nvalid <= 0 || Nvalid > N) return CREAM_ERR_BAD_ARG;
    NtParams p = plain(gp, x, w, M, N, K, ldw);
    p.bias = (const uint16_t*)bias;
    p.out2 = (uint16_t*)g;
    p.nvalid = Nvalid;
    return launch_nt<EPI_BIAS_GELU>(p, (hipStream_t)stream);
}

int cream_linear_gelu_fwd(void* gp, void* g, const void* x, const void* w, const void* bias, int M, int N, int K,
                          int64_t ldw, void* stream)
{
    return cream_linear_gelu_fwd_pad(gp, g, x, w, bias, M, N, N, K, ldw, stream);
}

int cream_linear_dgrad(void* dx, const void* dy, const void* wt, int M, int N, int K, int64_t ldwt, void* stream)
{
    // dx(M x K) = dy(M x N) . W(N x K)  ==  NT product with B = W^T (K rows, N contiguous)
    const int rc = check_nt(dx, dy, wt, M, K, N, ldwt, N);
    if (rc) return rc < 0 ? rc : CREAM_OK;
    const NtParams p = plain(dx, dy, wt, M, K, N, ldwt);
    return launch_nt<EPI_STORE>(p, (hipStream_t)stream);
}

int cream_linear_dgrad_seg(void* dx, const void* dy, const void* wt, int M, int N, int K, int64_t ldwt, int kseg,
                           int64_t kseg_stride, void* stream)
{
    const int rc = check_nt(dx, dy, wt, M, K, N, ldwt, kseg);
    if (rc) return rc < 0 ? rc : CREAM_OK;
    if (kseg <= 0 || kseg % 64 || N % kseg || kseg_stride % 8) return CREAM_ERR_BAD_ARG;
    NtParams p = plain(dx, dy, wt, M, K, N, ldwt);
    p.kseg = kseg; p.kseg_stride = kseg_stride;
    return launch_nt<EPI_STORE>(p, (hipStream_t)stream);
}

int cream_linear_dgrad_mul(void* dh, float* colsum_parts, const void* dy, const void* wt, const void* factor, int M, int N,
                             int K, int64_t ldwt, void* stream)
{
    const int rc = check_nt(dh, dy, wt, M, K, N, ldwt, N);
    if (rc) return rc < 0 ? rc : CREAM_OK;
    if (!colsum_parts || !factor || !aligned16(factor)) return CREAM_ERR_BAD_ARG;
    NtParams p = plain(dh, dy, wt, M, K, N, ldwt);
    p.aux = (const uint16_t*)factor; p.ldaux = K;
    p.colsum = colsum_parts;
    return launch_nt<EPI_MUL_COLSUM>(p, (hipStream_t)stream);
}

}  // extern "C"

namespace {
// the weight-gradient product on the macro tile of gemm_tn8.hpp (bf16 partial tiles, no bias partials).
// CREAM_GEMM_TN8 in the environment / cream_gemm_tn8(): 0 = never, 1 = problems of at least six 256 x 256 tiles, 2 = every problem (default)
std::atomic<int> g_tn8{-1};
int tn8_mode()
{
    int m = g_tn8.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = getenv("CREAM_GEMM_TN8");
        m = e ? atoi(e) : 2;
        g_tn8.store(m, std::memory_order_relaxed);
    }
    return m;
}
int tn8_tiles(int N, int K) { return ((N + 255) / 256) * ((K + 255) / 256); }
// one workgroup per CU: as many token slices as fit the chip once
int tn8_splits(int M, int N, int K)
{
    const int T = tn8_tiles(N, K), steps = (M + 63) / 64;
    // workgroups per launch: HALF the CUs.  In the step the weight gradients share the chip with the main chain and every slice is
    // another partial tile through HBM (same-call A/B, profiles/r05_tn8_step_ab.txt: 256 / 192 / 128 workgroups -> 9.62 / 9.48 /
    // 9.41 ms per step; another box: 128 / 96 / 64 -> 9.69 / 9.77 / 10.1)
    // CREAM_WGRAD_CUS in the environment (read once per process: the backward workspace layout depends on it) overrides the share
    static const int share = getenv("CREAM_WGRAD_CUS") ? atoi(getenv("CREAM_WGRAD_CUS")) : 0;
    int slots = num_cus() / 2 / 8 * 8;
    if (share >= 8) slots = (share < num_cus() ? share : num_cus()) / 8 * 8;
    int s = slots / T;
    if (s > steps) s = steps;
    return s < 1 ? 1 : s;
}
bool tn8_wanted(int M, int N, int K)
{
    const int mode = tn8_mode();
    if (mode <= 0 || M <= 0 || N % 8 || K % 8) return false;
    const int T = tn8_tiles(N, K);
    if ((int64_t)64 * (N > K ? N : K) * 2 + 4096 >= ((int64_t)1 << 31)) return false;     // 32-bit lane offsets inside a K-tile
    return mode >= 2 ? T <= num_cus() : (T >= 6 && T <= num_cus());
}
}  // namespace

extern "C" {

int cream_gemm_tn8(int mode)
{
    const int prev = tn8_mode();
    if (mode >= 0 && mode != prev) {
        g_tn8.store(mode, std::memory_order_relaxed);
        g_layout_epoch.fetch_add(1, std::memory_order_relaxed);
    }
    return prev;
}

int cream_block_layout_epoch(void) { return g_layout_epoch.load(std::memory_order_relaxed); }

int cream_cu_reserve(int reserve)
{
    const int prev = g_cu_reserve.load(std::memory_order_relaxed);
    if (reserve >= 0 && reserve != prev) {
        g_cu_reserve.store(reserve, std::memory_order_relaxed);
        g_layout_epoch.fetch_add(1, std::memory_order_relaxed);       // (the weight gradients' token slices follow the CU count)
    }
    return prev;
}

int cream_cu_count(void) { return cream::cu_count(); }

int cream_linear_wgrad_splits_bf16(int M, int N, int K)
{
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return tn8_wanted(M, N, K) ? tn8_splits(M, N, K) : cream_linear_wgrad_splits(M, N, K);
}

int cream_linear_wgrad_splits(int M, int N, int K)
{
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int tiles = ((N + 127) / 128) * ((K + 127) / 128), steps = (M + 63) / 64;
    // all workgroups resident at once (2 per CU: 512 slots — one more workgroup than slots costs a whole
    // extra round), at most 16 splits: every split is another fp32 partial tile through HBM (written
    // here, read by cream_grad_finalize).  Measured A/B in one call, 3 runs each: 256 / 384 / 512 slots ->
    // 11.60 / 11.60 / 11.65 ms per step (fewer partial bytes vs. a 1.5x slower kernel: 78 vs 53 us standalone);
    // 512 keeps the kernel itself at its best rate
    const int slots = 512;
    int s = slots / tiles;
    if (s > 16) s = 16;                                         // (32 for the small proj gradient: 10.64 vs 10.60 ms per step, A/B x3)
    if (s > steps) s = steps;
    return s < 1 ? 1 : s;
}

namespace {
int wgrad_parts_any(float* parts, uint16_t* parts16, float* bias_parts, const void* dy, const void* x, int M, int N, int K, int S, void* stream);
}

int cream_linear_wgrad_parts(float* parts, float* bias_parts, const void* dy, const void* x, int M, int N, int K, int S,
                             void* stream)
{
    if (!parts || !aligned16(parts)) return CREAM_ERR_BAD_ARG;
    return wgrad_parts_any(parts, nullptr, bias_parts, dy, x, M, N, K, S, stream);
}

int cream_linear_wgrad_parts_bf16(void* parts_bf16, float* bias_parts, const void* dy, const void* x, int M, int N, int K, int S,
                                  void* stream)
{
    if (!parts_bf16 || !aligned16(parts_bf16)) return CREAM_ERR_BAD_ARG;
    return wgrad_parts_any(nullptr, (uint16_t*)parts_bf16, bias_parts, dy, x, M, N, K, S, stream);
}

}  // extern "C"

namespace {
int wgrad_parts_any(float* parts, uint16_t* parts16, float* bias_parts, const void* dy, const void* x, int M, int N, int K, int S,
                    void* stream)
{
    if (M <= 0 || N <= 0 || K <= 0 || S <= 0 || N % 8 || K % 8) return CREAM_ERR_BAD_ARG;
    if (!dy || !x || !aligned16(dy) || !aligned16(x)) return CREAM_ERR_BAD_ARG;
    TnParams p{(const uint16_t*)dy, (const uint16_t*)x, N, K, M, N, K, S, parts, bias_parts, parts16};
    if (parts16 && tn8_wanted(M, N, K) && S == tn8_splits(M, N, K)) {
        if (bias_parts) {
            auto kern = gemm_tn8_kernel<true>;
            if (!cream::raise_dynamic_lds(kern, TN8_LDS_BYTES)) return CREAM_ERR_LAUNCH;
            CREAM_LAUNCH(kern, dim3(tn8_tiles(N, K) * S), dim3(512), TN8_LDS_BYTES, (hipStream_t)stream, p);
        } else {
            auto kern = gemm_tn8_kernel<false>;
            if (!cream::raise_dynamic_lds(kern, TN8_LDS_BYTES)) return CREAM_ERR_LAUNCH;
            CREAM_LAUNCH(kern, dim3(tn8_tiles(N, K) * S), dim3(512), TN8_LDS_BYTES, (hipStream_t)stream, p);
        }
        return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
    }
    const int grid = ((N + 127) / 128) * ((K + 127) / 128) * S;
    // the bias-free instantiation has no bias accumulators (126 instead of 165 VGPRs: room for two more waves of the main chain's
    // kernels per SIMD next to two of these — measured neutral on the step, 10.76 vs 10.75 ms in a same-box A/B x3: the two
    // streams share throughput, not register space)
    if (bias_parts) CREAM_LAUNCH((gemm_tn_kernel<2, 64, 2, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    else CREAM_LAUNCH((gemm_tn_kernel<2, 64, 2, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? CREAM_OK : CREAM_ERR_LAUNCH;
}
}  // namespace
