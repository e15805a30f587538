// Shared GEMM epilogue of libasr_amd's matrix-core kernels (gemm_f32.hip, gemm_split.hip): both produce
// 32x32 f32 accumulator fragments in the same register layout.
#pragma once
#include "asr_common.h"
#include <math.h>

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// sigmoid for the GLU epilogue (the hot one): e = exp(-|x|) from v_exp_f32 with the argument's rounding error carried
// separately (as attention.hip's exp_neg), 1 / (1 + e) from v_rcp_f32 + one Newton step; x >= 0: 1 / (1 + e), x < 0:
// e / (1 + e).  ~2 ulp, a dozen instructions instead of the ~40 of expf + an IEEE division (measured: the GLU epilogue cost
// 2.4 us of a 13 us launch at K = 256).
__device__ __forceinline__ float sigmoid_fast(float x)
{
    const float ax = -fabsf(x);
    const float t = ax * 1.44269502e+00f;
    float lo = fmaf(ax, 1.44269502e+00f, -t);
    lo = fmaf(ax, 1.92596303e-08f, lo);
    const float p = __builtin_amdgcn_exp2f(t);
    const float e = (ax < -100.0f) ? 0.0f : fmaf(p, lo * 6.93147182e-01f, p);          // exp(-|x|) in [0, 1] (-inf: no inf - inf)
    const float d = 1.0f + e;
    float r = __builtin_amdgcn_rcpf(d);
    r = fmaf(fmaf(-d, r, 1.0f), r, r);
    return x >= 0.0f ? r : e * r;
}

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
                const bool is_value = l31 < 16;
                // The value lane finishes registers 0..7 of the pair, the gate lane registers 8..15: one exchange per
                // register pair serves both directions and every lane evaluates 8 sigmoids instead of 16.
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float lo = acc[i][n][r] + bb, hi = acc[i][n][r + 8] + bb;
                    const float got = __shfl_xor(is_value ? hi : lo, 16);     // value lane: gate of reg r; gate lane: value of reg r + 8
                    const float out = is_value ? lo * sigmoid_fast(got) : got * sigmoid_fast(hi);
                    const int rr = is_value ? r : r + 8;
                    const int gm = m0 + wm0 + i * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * lhalf;
                    if (col < nout && gm < p.M) p.C[(size_t)gm * p.ldc + col] = out;
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
