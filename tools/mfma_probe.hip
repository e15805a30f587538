// MFMA f32 issue-rate probes (development tool).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// MODE 0: registers only; 1: + 4 ds_read_b128 per 16 MFMAs; 2: + barrier every 64 MFMAs; 3: + 8 ds_write + barrier
template <int MODE, int NACC>
__global__ __launch_bounds__(256) void probe(float* out, int iters)
{
    __shared__ f32x4 lds[4096];
    const int tid = threadIdx.x;
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x4 a = {1.f + tid, 2.f, 3.f, 4.f}, b = {0.5f, 0.25f, 0.125f, 1.f};
    for (int i = tid; i < 4096; i += 256) lds[i] = a;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 fa = a, fb = b;
            if (MODE >= 1) { fa = lds[(tid + q * 256 + it) & 4095]; fb = lds[(tid * 3 + q * 64 + it) & 4095]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[n % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j], fb[(j + n) & 3], acc[n % NACC], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE >= 3) { for (int w = 0; w < 8; ++w) lds[(tid + w * 256 + it * 7) & 4095] = b; }
        if (MODE >= 2) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE, int NACC>
int run(const char* name, int wgs, float* out, int iters)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((probe<MODE, NACC>), dim3(wgs), dim3(256), 0, 0, out, iters);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((probe<MODE, NACC>), dim3(wgs), dim3(256), 0, 0, out, iters);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)wgs * 4 * iters * 64 * 4096.0;
    printf("%-28s wgs=%5d  %8.1f us  %6.1f TF\n", name, wgs, ms * 1e3, flops / ms * 1e-9);
    return 0;
}

int main()
{
    float* out; CK(hipMalloc(&out, 8192 * 256 * 4));
    const int it = 2000;
    for (int wgs : {256, 512, 1024, 2048}) {
        run<0, 4>("regs, 4 acc", wgs, out, it);
        run<0, 1>("regs, 1 acc", wgs, out, it);
        run<1, 4>("lds reads, 4 acc", wgs, out, it);
        run<2, 4>("lds reads+barrier, 4 acc", wgs, out, it);
        run<3, 4>("lds rd+wr+barrier, 4 acc", wgs, out, it);
        run<3, 1>("lds rd+wr+barrier, 1 acc", wgs, out, it);
    }
    return 0;
}
