/*
 * rpe_index_oracle.c — TEST INFRASTRUCTURE, not product code.
 *
 * Plain-C, single-threaded restatement of the reference's rpe_index operator, used only
 * by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker for
 * the HIP kernels.  Nothing under cream_amd/ may link, import or call it.
 *
 * Pinned against: the reference's own self-test (rpe_ops/rpe_index.py:59-100, restated
 * in tests/test_oracle.py), the compiled reference extension oracle/_ref (when built)
 * and the golden vectors in tests/golden/ that were produced by the reference's Python.
 *
 *   forward  (iRPE/DeiT-with-iRPE/rpe_ops/rpe_index.cpp:38-44, the commented canonical
 *            form of the loop at :46-69):
 *                Y[i] = input[i / L_key * num_buckets + index[i % L_qk]]
 *   backward (rpe_index.cpp:116-122 with cpuAtomicAdd :75-80, executed in order):
 *                grad_input[i / L_key * num_buckets + index[i % L_qk]] += grad_output[i]
 *            i ascending, i.e. within a (b,h,i) row the addends arrive in ascending j.
 *            (The reference's at::parallel_for + omp critical visits the same addends
 *            in a thread-dependent order; ascending i is its 1-thread order.)
 */
#include <stddef.h>
#include <stdint.h>

#define DEFINE_FWD(NAME, T)                                                          \
    void NAME(T* y, const T* in, const int32_t* idx, int64_t B, int64_t H,           \
              int64_t Lq, int64_t Lk, int64_t nb) {                                  \
        const int64_t Lqk = Lq * Lk, n = B * H * Lqk;                                \
        for (int64_t i = 0; i < n; ++i) y[i] = in[i / Lk * nb + idx[i % Lqk]];       \
    }

DEFINE_FWD(oracle_rpe_index_fwd_u16, uint16_t) /* half / bfloat16 bit patterns */
DEFINE_FWD(oracle_rpe_index_fwd_u32, uint32_t) /* float bit patterns */
DEFINE_FWD(oracle_rpe_index_fwd_u64, uint64_t) /* double bit patterns */

/* strided variant used to check the non-contiguous (transposed view) input path of
 * rpe_index_cuda.cu:30-38: ind = b*s0 + h*s1 + q*s2 + index[..]*s3 */
void oracle_rpe_index_fwd_strided_u32(uint32_t* y, const uint32_t* in, const int32_t* idx,
                                      int64_t B, int64_t H, int64_t Lq, int64_t Lk,
                                      int64_t s0, int64_t s1, int64_t s2, int64_t s3) {
    const int64_t Lqk = Lq * Lk, n = B * H * Lqk;
    for (int64_t i = 0; i < n; ++i) {
        int64_t gi = i / Lk;
        const int64_t qi = gi % Lq; gi /= Lq;
        const int64_t hi = gi % H;  gi /= H;
        y[i] = in[gi * s0 + hi * s1 + qi * s2 + (int64_t)idx[i % Lqk] * s3];
    }
}

void oracle_rpe_index_bwd_f32(float* gin, const float* gout, const int32_t* idx, int64_t B,
                              int64_t H, int64_t Lq, int64_t Lk, int64_t nb) {
    const int64_t Lqk = Lq * Lk, n = B * H * Lqk;
    for (int64_t i = 0; i < n; ++i) {
        /* volatile store keeps gcc from re-associating or widening the float adds */
        volatile float* p = gin + i / Lk * nb + idx[i % Lqk];
        *p = *p + gout[i];
    }
}

void oracle_rpe_index_bwd_f64(double* gin, const double* gout, const int32_t* idx, int64_t B,
                              int64_t H, int64_t Lq, int64_t Lk, int64_t nb) {
    const int64_t Lqk = Lq * Lk, n = B * H * Lqk;
    for (int64_t i = 0; i < n; ++i) {
        volatile double* p = gin + i / Lk * nb + idx[i % Lqk];
        *p = *p + gout[i];
    }
}
