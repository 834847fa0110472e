// timing probe for the rpe_index scatter kernel with parts of it switched off (RPE_SCATTER_ABLATE in rpe_index.hip):
//   0 = the kernel, 1 = loads + LDS transpose + one add per key (no bins), 2 = loads only.
//   for v in 0 1 2; do hipcc --offload-arch=gfx950 -O3 -Iinclude -DRPE_SCATTER_ABLATE=$v tools/probes/scatter_probe.hip -o /tmp/sp$v && /tmp/sp$v; done
#include "../../cream_amd/csrc/rpe_index.hip"
#include <stdio.h>
#include <stdlib.h>
int main() {
    const int B = 64, H = 12, L = 577, nb = 50;
    const size_t n_go = (size_t)B * H * L * L, n_gi = (size_t)B * H * L * nb;
    for (int dt = 0; dt < 2; ++dt) {
        const int es = dt == 0 ? 4 : 2;
        void *go, *gi; int32_t* idx;
        hipMalloc(&go, n_go * es); hipMalloc(&gi, n_gi * es); hipMalloc(&idx, (size_t)L * L * 4);
        hipMemset(go, 0, n_go * es);
        int32_t* h = (int32_t*)malloc((size_t)L * L * 4);
        for (int i = 0; i < L * L; ++i) h[i] = rand() % nb;
        hipMemcpy(idx, h, (size_t)L * L * 4, hipMemcpyHostToDevice);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        const int code = dt == 0 ? CREAM_F32 : CREAM_BF16;
        for (int it = 0; it < 3; ++it) cream_rpe_index_bwd(gi, go, idx, B, H, L, L, nb, code, 0, 0);
        hipDeviceSynchronize();
        hipEventRecord(a, 0);
        const int N = 10;
        for (int it = 0; it < N; ++it) cream_rpe_index_bwd(gi, go, idx, B, H, L, L, nb, code, 0, 0);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double bytes = (double)(n_go + n_gi) * es + (double)L * L * 4;
#ifndef RPE_SCATTER_ABLATE
#define RPE_SCATTER_ABLATE 0
#endif
        printf("ablate %d %s: %.1f us per bwd  %.2f TB/s\n", RPE_SCATTER_ABLATE, dt == 0 ? "fp32" : "bf16", ms / N * 1e3, bytes / (ms / N * 1e-3) / 1e12);
        hipFree(go); hipFree(gi); hipFree(idx); free(h);
    }
    return 0;
}
