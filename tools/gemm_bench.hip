// Micro-benchmark of the decode / encoder GEMM shapes (development tool, not part of the library).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/gemm_bench.hip asr_amd/csrc/gemm_f32.o -o tools/gemm_bench
#include "../asr_amd/csrc/asr_common.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Shape { const char* name; int M, N, K, epi; };

int main(int argc, char** argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 30;
    const char* only = argc > 2 ? argv[2] : nullptr;          // shape name prefix
    const int only_tile = argc > 3 ? atoi(argv[3]) : 0, only_split = argc > 4 ? atoi(argv[4]) : 0;
    std::vector<Shape> shapes = {
        {"qkv   ", 640, 3072, 1024, EPI_PARTIAL}, {"proj  ", 640, 1024, 1024, EPI_PARTIAL},
        {"glu   ", 640, 8192, 1024, EPI_GLU},     {"ffn2  ", 640, 1024, 4096, EPI_PARTIAL},
        {"logits", 640, 3900, 1024, EPI_PARTIAL}, {"encqkv", 21376, 3072, 1024, EPI_BIAS},
        {"encglu", 21376, 8192, 1024, EPI_GLU},   {"encff2", 21376, 1024, 4096, EPI_PARTIAL},
        {"encprj", 21376, 1024, 1024, EPI_PARTIAL},
        {"fixk32", 640, 1024, 32, EPI_PARTIAL},   {"fixk64", 640, 1024, 64, EPI_PARTIAL},
        {"fixk128", 640, 1024, 128, EPI_PARTIAL}, {"fixk256", 640, 1024, 256, EPI_PARTIAL},
        {"fixk512", 640, 1024, 512, EPI_PARTIAL}, {"fix8k32", 640, 8192, 32, EPI_PARTIAL},
        {"fix8k256", 640, 8192, 256, EPI_PARTIAL},
    };
    size_t maxA = 0, maxW = 0, maxC = 0;
    for (auto& s : shapes) {
        maxA = std::max(maxA, (size_t)s.M * s.K); maxW = std::max(maxW, (size_t)s.N * s.K);
        maxC = std::max(maxC, (size_t)s.M * s.N * (s.M <= 640 ? ASR_MAX_SPLITK : 1));
    }
    float *A, *W, *C, *bias;
    CK(hipMalloc(&A, maxA * 4)); CK(hipMalloc(&W, maxW * 4)); CK(hipMalloc(&C, maxC * 4)); CK(hipMalloc(&bias, 65536 * 4));
    std::vector<float> h(std::max(maxA, maxW));
    srand(1);
    for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    CK(hipMemcpy(A, h.data(), maxA * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, h.data(), maxW * 4, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, 65536 * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& s : shapes) {
        if (only && strncmp(only, s.name, strlen(only)) != 0) continue;
        const bool dec = s.M <= 640;
        for (int tile = 1; tile <= 5; ++tile) {
            for (int split = 1; split <= (dec && s.epi == EPI_PARTIAL ? 16 : 1); split *= 2) {
                if (split > 1 && s.K / split < 128) continue;
                if ((only_tile && tile != only_tile) || (only_split && split != only_split)) continue;
                GemmArgs a{};
                a.A = A; a.W = W; a.bias = s.epi == EPI_PARTIAL ? nullptr : bias; a.C = C;
                a.M = s.M; a.N = s.N; a.K = s.K; a.lda = s.K; a.ldw = s.K; a.ldc = s.epi == EPI_GLU ? s.N / 2 : s.N;
                a.split_k = split; a.epilogue = s.epi;
                hipError_t e = launch_gemm_f32(a, tile, st);
                if (e != hipSuccess) { (void)hipGetLastError(); continue; }
                for (int i = 0; i < 3; ++i) launch_gemm_f32(a, tile, st);
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < reps; ++i) launch_gemm_f32(a, tile, st);
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double us = ms * 1e3 / reps, tf = 2.0 * s.M * s.N * s.K / us * 1e-6;
                printf("%s M=%5d N=%5d K=%5d tile=%d split=%2d  %8.1f us  %6.1f TF  %.2f\n", s.name, s.M, s.N, s.K, tile, split, us, tf, tf / 157.3);
            }
        }
    }
    return 0;
}
