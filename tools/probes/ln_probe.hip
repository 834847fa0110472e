// Development probe (not part of the library): ln_bwd operand-request variants and grid sizes, timed on rotating
// buffer sets larger than the 256 MB Infinity Cache, results compared bit for bit with variant 0.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Icream_amd/csrc tools/probes/ln_probe.hip -o tools/probes/ln_probe
#include "../../cream_amd/csrc/block_ops.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef void (*kern_t)(float*, uint16_t*, float*, const uint16_t*, const float*, const float*, const float*, const float*,
                       const float*, const float*, int, int, int);

int main() {
    const int M = 128 * 197, NSET = 4;
    const int Es[] = {384, 448};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int E : Es) {
        const size_t n = (size_t)M * E;
        std::vector<float> hx(n); std::vector<uint16_t> hdy(n);
        srand(3);
        for (size_t i = 0; i < n; ++i) {
            hx[i] = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
            float f = (rand() / (float)RAND_MAX - 0.5f); uint32_t u; memcpy(&u, &f, 4); hdy[i] = u >> 16;
        }
        float *x[NSET], *dres[NSET], *dx[NSET], *mean, *rstd, *gamma, *ss, *partial; uint16_t *dy[NSET], *dxs[NSET];
        for (int s = 0; s < NSET; ++s) {
            hipMalloc(&x[s], n * 4); hipMalloc(&dres[s], n * 4); hipMalloc(&dx[s], n * 4); hipMalloc(&dy[s], n * 2); hipMalloc(&dxs[s], n * 2);
            hipMemcpy(x[s], hx.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dres[s], hx.data(), n * 4, hipMemcpyHostToDevice);
            hipMemcpy(dy[s], hdy.data(), n * 2, hipMemcpyHostToDevice);
        }
        std::vector<float> hm(M, 0.1f), hr(M, 0.9f), hg(E, 1.1f), hs(128, 1.f / 0.9f);
        hipMalloc(&mean, M * 4); hipMalloc(&rstd, M * 4); hipMalloc(&gamma, E * 4); hipMalloc(&ss, 128 * 4);
        hipMalloc(&partial, (size_t)4096 * 3 * E * 4);
        hipMemcpy(mean, hm.data(), M * 4, hipMemcpyHostToDevice); hipMemcpy(rstd, hr.data(), M * 4, hipMemcpyHostToDevice);
        hipMemcpy(gamma, hg.data(), E * 4, hipMemcpyHostToDevice); hipMemcpy(ss, hs.data(), 128 * 4, hipMemcpyHostToDevice);
        kern_t kerns[] = {ln_bwd_kernel<2, 0>, ln_bwd_kernel<2, 1>, ln_bwd_kernel<2, 2>, ln_bwd_kernel<2, 0, 6>, ln_bwd_kernel<2, 1, 6>,
                          ln_bwd_kernel<2, 0, 8>, ln_bwd_kernel<2, 1, 8>, ln_bwd_kernel<2, 2, 4>};
        const char* names[] = {"pre0", "pre1", "pre2", "pre0/w6", "pre1/w6", "pre0/w8", "pre1/w8", "pre2/w4"};
        const int grids[] = {512, 1024, 1536, 2048, 3072};
        std::vector<float> ref(n), got(n); std::vector<float> pref, pgot;
        for (int v = 0; v < 8; ++v) for (int grid : grids) {
            float best = 1e9f, sum = 0.f; const int reps = 12;
            for (int it = 0; it < reps + 2; ++it) {
                const int s = it % NSET;
                hipEventRecord(e0);
                hipLaunchKernelGGL(kerns[v], dim3(grid), dim3(256), 0, 0, dx[s], dxs[s], partial, dy[s], x[s], mean, rstd, gamma,
                                   dres[s], ss, 197, M, E);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (it >= 2) { best = ms < best ? ms : best; sum += ms; }
            }
            hipMemcpy(got.data(), dx[0], n * 4, hipMemcpyDeviceToHost);
            pgot.resize((size_t)grid * 3 * E);
            hipMemcpy(pgot.data(), partial, pgot.size() * 4, hipMemcpyDeviceToHost);
            double colsum = 0; for (float f : pgot) colsum += f;
            bool same = true;
            if (v == 0 && grid == 1024) ref = got;
            else if (grid == 1024) same = memcmp(ref.data(), got.data(), n * 4) == 0;
            const double bytes = 16.0 * n;
            printf("E=%d %-8s grid=%4d  avg %.1f us  best %.1f us  %.2f TB/s  dx %s  partial-sum %.6e\n", E, names[v], grid,
                   sum / reps * 1e3, best * 1e3, bytes / (sum / reps * 1e-3) / 1e12, same ? "same" : "DIFFERENT", colsum);
        }
        for (int s = 0; s < NSET; ++s) { hipFree(x[s]); hipFree(dres[s]); hipFree(dx[s]); hipFree(dy[s]); hipFree(dxs[s]); }
        hipFree(mean); hipFree(rstd); hipFree(gamma); hipFree(ss); hipFree(partial);
    }
    return 0;
}
