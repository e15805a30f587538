// Micro-benchmark of the bf16-plane GEMM (gemm_split.hip) on the decode / encoder shapes, every tile variant, checked
// against the exact-f32 kernel (development tool, not part of the library).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/gemm_split_bench.hip asr_amd/csrc/gemm_f32.o asr_amd/csrc/gemm_split.o -o tools/gemm_split_bench
// run:   tools/gemm_split_bench [reps] [shape prefix] [nprod] [tile] [split] [cold]
//        cold = 1: every timed launch reads another copy of the weights (a 640 MB pool, beyond the 256 MB Infinity Cache),
//        as in the decode pipeline where a step streams 470 MB of weights between two uses of the same matrix
#include "../asr_amd/csrc/asr_common.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Shape { const char* name; int M, N, K, epi; };

int main(int argc, char** argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 30;
    const char* only = argc > 2 && argv[2][0] != '-' ? argv[2] : nullptr;
    const int nprod = argc > 3 ? atoi(argv[3]) : 6;
    const int only_tile = argc > 4 ? atoi(argv[4]) : 0, only_split = argc > 5 ? atoi(argv[5]) : 0;
    const bool cold = argc > 6 && atoi(argv[6]) != 0;
    const size_t POOL = (size_t)640 << 20;
    std::vector<Shape> shapes = {
        {"qkv   ", 640, 3072, 1024, EPI_PARTIAL}, {"proj  ", 640, 1024, 1024, EPI_PARTIAL},
        {"glu   ", 640, 8192, 1024, EPI_GLU},     {"xgluP ", 640, 8192, 1024, EPI_PARTIAL}, {"xgluB ", 640, 8192, 1024, EPI_BIAS},     {"ffn2  ", 640, 1024, 4096, EPI_PARTIAL},
        {"logits", 640, 3900, 1024, EPI_PARTIAL},
        {"dqkv  ", 1280, 3072, 1024, EPI_PARTIAL}, {"dproj ", 1280, 1024, 1024, EPI_PARTIAL}, {"dglu  ", 1280, 8192, 1024, EPI_GLU},
        {"dffn2 ", 1280, 1024, 4096, EPI_PARTIAL}, {"dlogit", 1280, 3900, 1024, EPI_PARTIAL},
        {"hqkv  ", 320, 3072, 1024, EPI_PARTIAL}, {"hproj ", 320, 1024, 1024, EPI_PARTIAL},
        {"hglu  ", 320, 8192, 1024, EPI_GLU},     {"hffn2 ", 320, 1024, 4096, EPI_PARTIAL},
        // fixed cost vs cost per k-tile: the same N at shorter K
        {"kglu256", 320, 8192, 256, EPI_GLU}, {"kglu512", 320, 8192, 512, EPI_GLU}, {"kgluP256", 320, 8192, 256, EPI_PARTIAL},
        {"kgluP1024", 320, 8192, 1024, EPI_PARTIAL}, {"kproj256", 320, 1024, 256, EPI_PARTIAL}, {"kproj512", 320, 1024, 512, EPI_PARTIAL},
        {"kGglu256", 640, 8192, 256, EPI_GLU}, {"kGglu512", 640, 8192, 512, EPI_GLU},
        {"encqkv", 21376, 3072, 1024, EPI_BIAS},
        {"encglu", 21376, 8192, 1024, EPI_GLU},   {"encff2", 21376, 1024, 4096, EPI_BIAS},
        {"encprj", 21376, 1024, 1024, EPI_BIAS},
    };
    size_t maxA = 0, maxW = 0, maxC = 0;
    for (auto& s : shapes) {
        maxA = std::max(maxA, (size_t)s.M * s.K); maxW = std::max(maxW, (size_t)s.N * s.K);
        maxC = std::max(maxC, (size_t)s.M * s.N * (s.M <= 1280 ? ASR_MAX_SPLITK : 1));
    }
    const int npl = gemm_split_planes(nprod);
    float *A, *W, *C, *Cref, *bias;
    uint16_t* Wp;
    CK(hipMalloc(&A, maxA * 4)); CK(hipMalloc(&W, maxW * 4)); CK(hipMalloc(&C, maxC * 4)); CK(hipMalloc(&Cref, maxC * 4));
    CK(hipMalloc(&bias, 65536 * 4)); CK(hipMalloc(&Wp, std::max((maxW + 32 * 4096) * 2 * npl, cold ? POOL : (size_t)0)));
    std::vector<float> hA(maxA), hW(maxW), hC(maxC), hR(maxC);
    srand(1);
    for (auto& v : hA) v = (rand() / (float)RAND_MAX - 0.5f) * 2.0f;
    for (auto& v : hW) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    CK(hipMemcpy(A, hA.data(), maxA * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, hW.data(), maxW * 4, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, 65536 * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<uint16_t> planes;
    for (auto& s : shapes) {
        if (only && strncmp(only, s.name, strlen(only)) != 0) continue;
        const size_t cnt = (size_t)s.N * s.K;
        (void)cnt;
        planes.resize(gemm_split_weight_elems(s.N, s.K, npl));
        gemm_split_weights_host(hW.data(), s.N, s.K, nprod, planes.data());
        const size_t wbytes = (planes.size() * 2 + 255) / 256 * 256;
        const int copies = cold ? (int)std::max<size_t>(1, POOL / wbytes) : 1;
        for (int c = 0; c < copies; ++c) CK(hipMemcpy(reinterpret_cast<char*>(Wp) + c * wbytes, planes.data(), planes.size() * 2, hipMemcpyHostToDevice));
        const bool dec = s.M <= 1280;
        const int ldc = s.epi == EPI_GLU ? s.N / 2 : s.N;
        // reference: exact-f32 kernel, unsplit
        GemmArgs r{};
        r.A = A; r.W = W; r.bias = s.epi == EPI_PARTIAL ? nullptr : bias; r.C = Cref;
        r.M = s.M; r.N = s.N; r.K = s.K; r.lda = s.K; r.ldw = s.K; r.ldc = ldc; r.split_k = 1; r.epilogue = s.epi;
        CK(launch_gemm_f32(r, 0, st));
        CK(hipStreamSynchronize(st));
        const size_t nout = (size_t)s.M * ldc;
        CK(hipMemcpy(hR.data(), Cref, nout * 4, hipMemcpyDeviceToHost));
        {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < reps; ++i) launch_gemm_f32(r, 0, st);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s M=%5d N=%5d K=%5d exact f32 (unsplit)      %8.1f us\n", s.name, s.M, s.N, s.K, ms * 1e3 / reps);
        }
        for (int tile : {2, 6, 9, 10, 11, 12}) {
            for (int split = 1; split <= (dec && s.epi == EPI_PARTIAL ? 16 : 1); split *= 2) {
                if (split > 1 && s.K / split < 128) continue;
                if ((only_tile && tile != only_tile) || (only_split && split != only_split)) continue;
                GemmArgs a = r;
                a.Wp = Wp; a.C = C; a.split_k = split;
                CK(hipMemsetAsync(C, 0xff, nout * 4 * split, st));
                hipError_t e = launch_gemm_split(a, nprod, tile, st);
                if (e != hipSuccess) { (void)hipGetLastError(); continue; }
                CK(hipStreamSynchronize(st));
                CK(hipMemcpy(hC.data(), C, nout * 4 * split, hipMemcpyDeviceToHost));
                double maxerr = 0, maxref = 0;
                for (size_t i = 0; i < nout; ++i) {
                    double v = 0;
                    for (int zz = 0; zz < split; ++zz) v += hC[(size_t)zz * nout + i];
                    maxerr = std::max(maxerr, fabs(v - hR[i])); maxref = std::max(maxref, (double)fabs(hR[i]));
                }
                for (int i = 0; i < 3; ++i) launch_gemm_split(a, nprod, tile, st);
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < reps; ++i) {
                    a.Wp = reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(Wp) + (size_t)(i % copies) * wbytes);
                    launch_gemm_split(a, nprod, tile, st);
                }
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double us = ms * 1e3 / reps, tf = 2.0 * s.M * s.N * s.K / us * 1e-6;
                printf("%s M=%5d N=%5d K=%5d nprod=%d tile=%d split=%2d  %8.1f us  %6.1f TF(alg)  maxerr %.2e (ref max %.2e)\n", s.name, s.M,
                       s.N, s.K, nprod, tile, split, us, tf, maxerr, maxref);
            }
        }
    }
    return 0;
}
