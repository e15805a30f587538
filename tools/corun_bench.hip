// Co-run micro-benchmark (development tool, not part of the library): does the HBM-bound decode-time cross-attention
// overlap with the matrix-bound decode-row GEMMs when the two run on two streams, as they do with several engine handles
// in flight?  Times each kernel alone (back-to-back launches on one stream), then both together on two streams for the
// same number of launches, and prints   together / (alone_a + alone_b)   (1.0 = they queue, max/sum = perfect overlap).
// A third, synthetic kernel - a plain 16-byte-per-lane streaming read of the same K/V bytes with a handful of registers
// and no LDS - takes the attention kernel's place to tell CU footprint from memory-side contention.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -c tools/corun_bench.hip -o /tmp/corun.o && hipcc --offload-arch=gfx950 /tmp/corun.o asr_amd/csrc/attention.o asr_amd/csrc/gemm_split.o asr_amd/csrc/gemm_f32.o -o tools/corun_bench
// run:   GPU_MAX_HW_QUEUES=8 tools/corun_bench [reps] [gemm: glu|qkv|ffn2|proj] [attention variant: 0 exact f32, 1 fp16x3, 2 bf16 K/V]
#include "../asr_amd/csrc/asr_common.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// streaming read of `bytes` (16 B per lane per load, `unroll` loads in flight per lane), result folded into one float per
// workgroup so that nothing is optimised away
template <int UNROLL>
__global__ __launch_bounds__(256) void stream_read_kernel(const f32x4* __restrict__ src, size_t n16, float* __restrict__ out)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        f32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
    for (; i < n16; i += stride) acc += __builtin_nontemporal_load(src + i);
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[blockIdx.x] = acc[0];
}

// nothing but matrix instructions on register operands (random, non-zero: the power of a real GEMM's inner loop without any
// of its memory traffic): `iters` x 8 v_mfma_f32_32x32x16_f16 per wave, 8 waves per workgroup
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(512) void mfma_only_kernel(const float* __restrict__ seed, int iters, float* __restrict__ out)
{
    f16x8_t a[4], b[2];
    const float s0 = seed ? seed[threadIdx.x & 255] : 0.0f;      // null: all-zero operands (same instruction stream, far less switching power)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) a[i][j] = (_Float16)(s0 * (float)(i + 1) + (seed ? 0.01f * (float)j : 0.0f));
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) b[i][j] = (_Float16)(s0 * 0.5f - (seed ? 0.02f * (float)(i + j) : 0.0f));
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += acc[i][0] + acc[i][7];
    if (t == 12345.678f) out[blockIdx.x] = t;
}

__global__ void empty_kernel(float* out) { if (out == nullptr) out[0] = 0.f; }

struct Timed { float ms_a = 0.f, ms_b = 0.f, ms_all = 0.f; };

int main(int argc, char** argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 60;
    const char* gname = argc > 2 ? argv[2] : "glu";
    const int variant = argc > 3 ? atoi(argv[3]) : 1;
    const bool hot_w = argc > 4 && atoi(argv[4]) != 0;          // 1: one copy of the weights (they stay in the Infinity Cache)
    const int B = 64, T = 334, K = 10, heads = 16, H = heads * 64, R = B * K, L = 6;

    // ---- cross-attention operands: L layers of K/V so that back-to-back launches stream from HBM like the decode step
    const size_t kvn = (size_t)B * T * H, esz = variant == 2 ? 2 : 4;
    std::vector<void*> kc(L), vc(L);
    std::vector<float> h(std::max(kvn, (size_t)2 * R * H));
    srand(1);
    for (auto& x : h) x = rand() / (float)RAND_MAX - 0.5f;
    for (int l = 0; l < L; ++l) {
        CK(hipMalloc(&kc[l], kvn * 4 + ASR_XATT_SLACK_ROWS * 256)); CK(hipMalloc(&vc[l], kvn * 4 + ASR_XATT_SLACK_ROWS * 256));
        CK(hipMemcpy(kc[l], h.data(), kvn * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(vc[l], h.data(), kvn * 4, hipMemcpyHostToDevice));
    }
    float *q, *qb, *ctx, *sink;
    int32_t *pad, *flags;
    CK(hipMalloc(&q, (size_t)2 * R * H * 4)); CK(hipMalloc(&qb, H * 4)); CK(hipMalloc(&ctx, (size_t)R * H * 4));
    CK(hipMalloc(&sink, 65536 * 4));
    CK(hipMalloc(&pad, (size_t)B * T * 4 + ASR_XATT_SLACK_ROWS * 4)); CK(hipMalloc(&flags, 16));
    CK(hipMemcpy(q, h.data(), (size_t)2 * R * H * 4, hipMemcpyHostToDevice));
    CK(hipMemset(qb, 0, H * 4)); CK(hipMemset(pad, 0, (size_t)B * T * 4 + ASR_XATT_SLACK_ROWS * 4)); CK(hipMemset(flags, 0, 16));
    CrossAttnArgs ca{};
    ca.qpart = q; ca.nsplit = 1; ca.split_stride = (size_t)R * H; ca.qbias = qb; ca.pad2 = pad; ca.done = flags; ca.ctx = ctx;
    ca.b = B; ca.k = K; ca.heads = heads; ca.t = T; ca.kv_bf16 = variant == 2; ca.f16x3 = variant == 1; ca.range_flag = variant == 1 ? flags + 1 : nullptr;

    // ---- GEMM operands (fp16x3 planes, a pool of weight copies beyond the Infinity Cache)
    int M = 640, N = 8192, Kd = 1024, epi = EPI_GLU, tile = 9, split = 1;
    if (!strcmp(gname, "qkv")) { N = 3072; epi = EPI_PARTIAL; }
    else if (!strcmp(gname, "ffn2")) { N = 1024; Kd = 4096; epi = EPI_PARTIAL; split = 2; }
    else if (!strcmp(gname, "proj")) { N = 1024; epi = EPI_PARTIAL; tile = 2; }
    const int nprod = 13;
    std::vector<float> hW((size_t)N * Kd), hA((size_t)M * Kd);
    for (auto& v : hW) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    for (auto& v : hA) v = (rand() / (float)RAND_MAX - 0.5f) * 2.0f;
    std::vector<uint16_t> planes(gemm_split_weight_elems(N, Kd, 2));
    gemm_split_weights_host(hW.data(), N, Kd, nprod, planes.data());
    const size_t wbytes = (planes.size() * 2 + 255) / 256 * 256, POOL = (size_t)640 << 20;
    const int copies = hot_w ? 1 : (int)std::max<size_t>(1, POOL / wbytes);
    char* Wp;
    float *A, *C, *bias;
    CK(hipMalloc(&Wp, (size_t)copies * wbytes)); CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&C, (size_t)M * N * 4 * split));
    CK(hipMalloc(&bias, (size_t)N * 4)); CK(hipMemset(bias, 0, (size_t)N * 4));
    for (int c = 0; c < copies; ++c) CK(hipMemcpy(Wp + (size_t)c * wbytes, planes.data(), planes.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    GemmArgs ga{};
    ga.A = A; ga.bias = epi == EPI_PARTIAL ? nullptr : bias; ga.C = C; ga.M = M; ga.N = N; ga.K = Kd; ga.lda = Kd; ga.ldw = Kd;
    ga.ldc = epi == EPI_GLU ? N / 2 : N; ga.split_k = split; ga.epilogue = epi; ga.range_flag = flags + 2;

    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    CK(hipMemsetAsync(flags, 0, 16, sa)); CK(hipMemsetAsync(flags, 0, 16, sb));        // bind each stream to its hardware queue now
    CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
    hipEvent_t a0, a1, b0, b1;
    CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));

    auto launch_a = [&](int kind, int i) -> hipError_t {
        if (kind == 0) { ca.kc = kc[i % L]; ca.v = vc[i % L]; return launch_decoder_cross_attention(ca, sa); }
        // synthetic stream of the same bytes: K then V of layer i % L; the kinds differ in how many bytes they keep in flight
        // (workgroups x 256 lanes x loads per lane x 16 B): 1: 1024 x 8 = 32 MiB, 2: 512 x 8 = 16 MiB, 3: 256 x 8 = 8 MiB,
        // 4: 256 x 4 = 4 MiB, 5: 256 x 2 = 2 MiB, 6: 128 x 4 = 2 MiB on half the CUs
        const int grid = kind == 1 ? 1024 : kind == 2 ? 512 : kind == 6 ? 128 : 256;
        const f32x4* ks = static_cast<const f32x4*>(kc[i % L]);
        const f32x4* vs = static_cast<const f32x4*>(vc[i % L]);
        const size_t n16 = kvn * esz / 16;
        if (kind == 9 || kind == 10) {
            // nothing but kernel boundaries: 175 launches of a kernel that does nothing, one workgroup (9) or 256 (10)
            for (int j = 0; j < 175; ++j) hipLaunchKernelGGL(empty_kernel, dim3(kind == 9 ? 1 : 256), dim3(64), 0, sa, sink);
            return hipGetLastError();
        }
        if (kind == 7 || kind == 8) {
            // the same number of bytes, but re-reading a small buffer: 7 = 2 MiB (stays in every XCD's L2), 8 = 64 MiB (Infinity
            // Cache): no HBM traffic, the same load on the CUs' memory pipelines
            const size_t small16 = (kind == 7 ? (size_t)2 << 20 : (size_t)64 << 20) / 16;
            for (size_t done16 = 0; done16 < 2 * n16; done16 += small16)
                hipLaunchKernelGGL(stream_read_kernel<8>, dim3(256), dim3(256), 0, sa, ks, small16, sink);
            return hipGetLastError();
        }
        if (kind <= 3) {
            hipLaunchKernelGGL(stream_read_kernel<8>, dim3(grid), dim3(256), 0, sa, ks, n16, sink);
            hipLaunchKernelGGL(stream_read_kernel<8>, dim3(grid), dim3(256), 0, sa, vs, n16, sink);
        } else if (kind == 5) {
            hipLaunchKernelGGL(stream_read_kernel<2>, dim3(grid), dim3(256), 0, sa, ks, n16, sink);
            hipLaunchKernelGGL(stream_read_kernel<2>, dim3(grid), dim3(256), 0, sa, vs, n16, sink);
        } else {
            hipLaunchKernelGGL(stream_read_kernel<4>, dim3(grid), dim3(256), 0, sa, ks, n16, sink);
            hipLaunchKernelGGL(stream_read_kernel<4>, dim3(grid), dim3(256), 0, sa, vs, n16, sink);
        }
        return hipGetLastError();
    };
    const bool mfma_only = !strcmp(gname, "mfma");            // the GEMM's side replaced by bare matrix instructions (no memory traffic)
    auto launch_b = [&](int i) -> hipError_t {
        if (mfma_only) {
            // as many matrix instructions as the GLU GEMM executes: 3 x 2 x 640 x 8192 x 1024 flop / (2 x 32 x 32 x 16 per MFMA) over 2048 waves
            hipLaunchKernelGGL(mfma_only_kernel, dim3(256), dim3(512), 0, sb, hot_w ? static_cast<const float*>(nullptr) : q, 60, sink);
            return hipGetLastError();
        }
        ga.Wp = reinterpret_cast<const uint16_t*>(Wp + (size_t)(i % copies) * wbytes);
        return launch_gemm_split(ga, nprod, tile, sb);
    };
    // run na launches of kernel A (kind) and nb of the GEMM, each stream back to back; 0 = that side is idle
    auto run = [&](int kind, int na, int nb, Timed& t) -> int {
        for (int i = 0; i < 3; ++i) { if (na) CK(launch_a(kind, i)); if (nb) CK(launch_b(i)); }
        CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
        if (na) CK(hipEventRecord(a0, sa));
        if (nb) CK(hipEventRecord(b0, sb));
        {                                                      // one host thread per stream, so that neither queue waits for the host
            std::thread ta([&] { for (int i = 0; i < na; ++i) (void)launch_a(kind, i); });
            std::thread tb([&] { for (int i = 0; i < nb; ++i) (void)launch_b(i); });
            ta.join(); tb.join();
        }
        if (na) CK(hipEventRecord(a1, sa));
        if (nb) CK(hipEventRecord(b1, sb));
        CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
        if (na) CK(hipEventElapsedTime(&t.ms_a, a0, a1));
        if (nb) CK(hipEventElapsedTime(&t.ms_b, b0, b1));
        if (na && nb) {
            float x0, x1;                                      // wall span of both: from the earlier start to the later end
            CK(hipEventElapsedTime(&x0, a0, b0)); CK(hipEventElapsedTime(&x1, a0, b1));
            t.ms_all = std::max(t.ms_a, x1) - std::min(0.0f, x0);
        }
        return 0;
    };
    const char* kinds[11] = {"cross-attention", "stream 32 MiB in flight", "stream 16 MiB in flight", "stream 8 MiB in flight", "stream 4 MiB in flight",
                            "stream 2 MiB in flight", "stream 2 MiB, 128 wg", "L2-resident re-read", "Infinity-Cache re-read", "175 empty kernels (1 wg)", "175 empty kernels (256 wg)"};
    printf("GEMM %s %dx%dx%d fp16x3 tile %d split %d; attention variant %d (0 exact f32, 1 fp16x3, 2 bf16 K/V); %d launches each\n", gname, M, N, Kd,
           tile, split, variant, reps);
    Timed g;
    if (run(0, 0, reps, g)) return 1;
    printf("GEMM alone                 %7.1f us per launch\n", g.ms_b * 1e3 / reps);
    for (int kind = 0; kind < 11; ++kind) {
        if (kind == 2 || kind == 3 || kind == 5 || kind == 6) continue;
        Timed a, ab;
        if (run(kind, reps, 0, a)) return 1;
        // equalise the two sides' solo durations so that neither runs alone for long
        const double ta = a.ms_a / reps, tg = g.ms_b / reps;
        const int na = ta >= tg ? reps : (int)(reps * tg / ta + 0.5), nb = ta >= tg ? (int)(reps * ta / tg + 0.5) : reps;
        Timed a2, g2;
        if (run(kind, na, 0, a2) || run(0, 0, nb, g2) || run(kind, na, nb, ab)) return 1;
        const double sum = a2.ms_a + g2.ms_b, mx = std::max(a2.ms_a, g2.ms_b);
        printf("%-22s alone %7.1f us (%.2f TB/s of K/V); %d x it + %d x GEMM: alone %.2f + %.2f = %.2f ms, together %.2f ms = %.2f of the sum "
               "(perfect overlap would be %.2f); in the mix: %.1f us per attention-side launch, %.1f us per GEMM\n",
               kinds[kind], ta * 1e3, 2.0 * kvn * esz / (ta * 1e-3) * 1e-12, na, nb, a2.ms_a, g2.ms_b, sum, ab.ms_all, ab.ms_all / sum, mx / sum,
               ab.ms_a * 1e3 / na, ab.ms_b * 1e3 / nb);
    }
    return 0;
}
