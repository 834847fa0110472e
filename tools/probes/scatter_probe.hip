// timing probe for the rpe_index scatter kernel variants (see CREAM_PROBE_VARIANT in rpe_index.hip)
#include "../../cream_amd/csrc/rpe_index.hip"
#include <stdio.h>
#include <stdlib.h>
int main() {
    const int B = 64, H = 12, L = 577, nb = 50;
    const size_t n_go = (size_t)B * H * L * L, n_gi = (size_t)B * H * L * nb;
    float *go, *gi; int32_t* idx;
    hipMalloc(&go, n_go * 4); hipMalloc(&gi, n_gi * 4); hipMalloc(&idx, (size_t)L * L * 4);
    hipMemset(go, 0, n_go * 4);
    int32_t* h = (int32_t*)malloc((size_t)L * L * 4);
    for (int i = 0; i < L * L; ++i) h[i] = rand() % nb;
    hipMemcpy(idx, h, (size_t)L * L * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int it = 0; it < 3; ++it) cream_rpe_index_bwd(gi, go, idx, B, H, L, L, nb, CREAM_F32, 0, 0);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    const int N = 10;
    for (int it = 0; it < N; ++it) cream_rpe_index_bwd(gi, go, idx, B, H, L, L, nb, CREAM_F32, 0, 0);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("variant %d: %.3f ms per bwd\n",
#ifdef CREAM_PROBE_VARIANT
           CREAM_PROBE_VARIANT,
#else
           0,
#endif
           ms / N);
    return 0;
}
