// Micro-benchmark of the decode-time cross-attention kernel at the bench shape (development tool, not part of the library).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -c tools/attn_bench.hip -o /tmp/ab.o && hipcc --offload-arch=gfx950 /tmp/ab.o asr_amd/csrc/attention.o -o tools/attn_bench
#include "../asr_amd/csrc/asr_common.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 50;
    const int B = argc > 2 ? atoi(argv[2]) : 64, T = argc > 3 ? atoi(argv[3]) : 334, K = argc > 4 ? atoi(argv[4]) : 10;
    const int heads = 16, H = heads * 64, R = B * K, L = 6;
    // L layers of K/V so that back-to-back launches stream from HBM like the decode step does (6 x 175 MB > every cache)
    std::vector<float*> kc(L), vc(L);
    const size_t kvn = (size_t)B * T * H;
    std::vector<float> h(kvn);
    srand(1);
    for (auto& x : h) x = rand() / (float)RAND_MAX - 0.5f;
    for (int l = 0; l < L; ++l) {
        CK(hipMalloc(&kc[l], kvn * 4 + ASR_XATT_SLACK_ROWS * 256)); CK(hipMalloc(&vc[l], kvn * 4 + ASR_XATT_SLACK_ROWS * 256));
        CK(hipMemcpy(kc[l], h.data(), kvn * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(vc[l], h.data(), kvn * 4, hipMemcpyHostToDevice));
    }
    float *q, *qb, *ctx; int32_t *pad, *done;
    CK(hipMalloc(&q, (size_t)2 * R * H * 4)); CK(hipMalloc(&qb, H * 4)); CK(hipMalloc(&ctx, (size_t)R * H * 4));
    CK(hipMalloc(&pad, (size_t)B * T * 4 + ASR_XATT_SLACK_ROWS * 4)); CK(hipMalloc(&done, 4));
    CK(hipMemcpy(q, h.data(), (size_t)2 * R * H * 4, hipMemcpyHostToDevice));
    CK(hipMemset(qb, 0, H * 4)); CK(hipMemset(pad, 0, (size_t)B * T * 4)); CK(hipMemset(done, 0, 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int nsplit = 1; nsplit <= 2; ++nsplit) {
        CrossAttnArgs a{};
        a.qpart = q; a.nsplit = nsplit; a.split_stride = (size_t)R * H; a.qbias = qb; a.pad2 = pad; a.done = done; a.ctx = ctx;
        a.b = B; a.k = K; a.heads = heads; a.t = T;
        for (int i = 0; i < 6; ++i) { a.kc = kc[i % L]; a.v = vc[i % L]; CK(launch_decoder_cross_attention(a, st)); }
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) { a.kc = kc[i % L]; a.v = vc[i % L]; (void)launch_decoder_cross_attention(a, st); }
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        printf("cross-attention B=%d T=%d K=%d nsplit=%d: %.1f us  %.2f TB/s (K/V bytes)\n", B, T, K, nsplit, us, 2.0 * kvn * 4 / us * 1e-6);
    }
    return 0;
}
