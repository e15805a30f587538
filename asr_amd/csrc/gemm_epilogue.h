// Shared GEMM epilogue of libasr_amd's matrix-core kernels (gemm_f32.hip, gemm_split.hip): both produce
// 32x32 f32 accumulator fragments in the same register layout.
#pragma once
#include "asr_common.h"
#include <math.h>

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// acc[i][n] = fragment (i, n) of the wave's WM x WN = (32*MI) x (32*NI) tile whose top-left element is
// C[m0 + wm0][n0 + wn0]; z = split-K slice of this workgroup.
template <int EPI, int MI, int NI>
__device__ __forceinline__ void gemm_store_tile(const GemmArgs& p, const f32x16 (&acc)[MI][NI], int m0, int n0, int wm0, int wn0,
                                                int z, int lane)
{
    const int l31 = lane & 31, lhalf = lane >> 5;
    // ---- epilogue: acc reg r of lane l is C[row (r&3)+8*(r>>2)+4*(l>>5)][col l&31] of its 32x32 fragment
    if constexpr (EPI == EPI_GLU) {
        // W rows are packed in groups of 32: 16 value rows | 16 gate rows, so every 32-column fragment
        // holds its own pairs: lanes 0-15 the value, lanes 16-31 the gate of the same 16 output columns.
        const int nout = p.N >> 1;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int n = 0; n < NI; ++n) {
                const int pcol = n0 + wn0 + n * 32 + l31;            // packed column
                const int col = ((n0 + wn0 + n * 32) >> 1) + (l31 & 15);
                const float bb = (p.bias && pcol < p.N) ? p.bias[pcol] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float mine = acc[i][n][r] + bb;
                    const float gate = __shfl_down(mine, 16);
                    const int gm = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                    if (l31 < 16 && col < nout && gm < p.M) p.C[(size_t)gm * p.ldc + col] = mine * sigmoidf_(gate);
                }
            }
    } else {
        float* C = p.C;
        if constexpr (EPI == EPI_PARTIAL) C += (size_t)z * p.M * p.ldc;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int n = 0; n < NI; ++n) {
                const int col = n0 + wn0 + n * 32 + l31;
                if (col < p.N) {
                    float b = 0.f;
                    if constexpr (EPI != EPI_PARTIAL) b = p.bias ? p.bias[col] : 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int gm = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                        if (gm < p.M) {
                            float v = acc[i][n][r] + b;
                            if constexpr (EPI == EPI_RELU) v = fmaxf(v, 0.f);
                            if constexpr (EPI == EPI_SIGMOID) v = sigmoidf_(v);
                            if constexpr (EPI == EPI_TANH) v = tanhf(v);
                            if constexpr (EPI == EPI_SWISH) v = v * sigmoidf_(v);
                            C[(size_t)gm * p.ldc + col] = v;
                        }
                    }
                }
            }
    }
}

}  // namespace
