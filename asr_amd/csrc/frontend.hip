// Feature front end either side of the frame stacking: delta / delta-delta, CMVN and the general splice
// (reference: third_party/feat_convert/io/fea.py).  All of it is HBM-bound element work on [frames, dim]
// float32 matrices: one thread per output element (coalesced along dim), column statistics by a
// workgroup per 64 columns.  The file is compiled with floating-point contraction off, so every product and
// sum is rounded on its own as numpy / TensorFlow do: the delta and global-CMVN outputs are bit-identical to
// numpy's float32 evaluation of the reference's expressions.
#include "asr_common.h"

#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
// plain operators, compiled under the pragma above (the __fmul_rn / __fadd_rn wrappers of the HIP headers are
// compiled with contraction allowed and do fuse)
__device__ __forceinline__ float mul(float a, float b) { return a * b; }
__device__ __forceinline__ float add(float a, float b) { return a + b; }
__device__ __forceinline__ float sub(float a, float b) { return a - b; }

// np_fea_add_delt / tf_fea_add_delt (fea.py:61-72,117-128): out [n, 3*dim] = x | delta1 | delta2.
// The reference pads "symmetric" by one frame, repeatedly (fea.py:21-22,34-37,76-77,89-92): padding an already
// padded matrix mirrors the copy of the edge frame, so frame t+o reads x[clamp(t+o, 0, n-1)].
//   delta1 (fea.py:20-31):  (r1 - l1) * 0.1 + (r2 - l2) * 0.2
//   delta2 (fea.py:33-52):  -0.1*c - 0.04*(l1 + r1) + 0.01*(l2 + r2) + 0.04*(l3 + l4 + r4 + r3)
__global__ void fea_add_delt_kernel(const float* __restrict__ x, int n, int dim, float* __restrict__ out)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)n * dim) return;
    const int t = (int)(idx / dim), d = (int)(idx - (long)t * dim);
    auto at = [&](int o) { return x[(size_t)clampi(t + o, 0, n - 1) * dim + d]; };
    const float c = at(0), l1 = at(-1), l2 = at(-2), l3 = at(-3), l4 = at(-4), r1 = at(1), r2 = at(2), r3 = at(3), r4 = at(4);
    const float d1 = add(mul(sub(r1, l1), 0.1f), mul(sub(r2, l2), 0.2f));
    float d2 = sub(mul(-0.1f, c), mul(0.04f, add(l1, r1)));
    d2 = add(d2, mul(0.01f, add(l2, r2)));
    d2 = add(d2, mul(0.04f, add(add(add(l3, l4), r4), r3)));
    float* o = out + (size_t)t * 3 * dim + d;
    o[0] = c;
    o[dim] = d1;
    o[2 * dim] = d2;
}

// tf_fea_cmvn_global (fea.py:130-132): (feature - mean) / var -- the reference divides by `var` as given.
__global__ void fea_cmvn_global_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                       const float* __restrict__ var, int n, int dim, float* __restrict__ out)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)n * dim) return;
    const int d = (int)(idx % dim);
    out[idx] = sub(x[idx], mean[d]) / var[d];      // IEEE division (hipcc's default: correctly rounded)
}

// tf_fea_cmvn_utt (fea.py:134-140): mean and E[x^2] over frames per column, var = E[x^2] - mean^2,
// out = (x - mean) * rsqrt(var + 1e-12).  One workgroup per 64 columns: 4 waves stride the frames (256-byte
// coalesced rows), sums in double, then the same workgroup normalises its columns.
__global__ __launch_bounds__(256) void fea_cmvn_utt_kernel(const float* __restrict__ x, int n, int dim, float* __restrict__ out)
{
    __shared__ double s1[4][64], s2[4][64];
    __shared__ float smean[64], sistd[64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const bool ok = c < dim;
    double a = 0.0, b = 0.0;
    if (ok)
        for (int t = g; t < n; t += 4) {
            const double v = (double)x[(size_t)t * dim + c];
            a += v;
            b += v * v;
        }
    s1[g][threadIdx.x & 63] = a;
    s2[g][threadIdx.x & 63] = b;
    __syncthreads();
    if (g == 0) {
        const int l = threadIdx.x;
        const double m = (s1[0][l] + s1[1][l] + s1[2][l] + s1[3][l]) / (double)n;
        const double q = (s2[0][l] + s2[1][l] + s2[2][l] + s2[3][l]) / (double)n;
        const float mf = (float)m, varf = (float)(q - m * m);
        smean[l] = mf;
        sistd[l] = 1.0f / sqrtf(varf + 1e-12f);
    }
    __syncthreads();
    if (ok) {
        const float m = smean[threadIdx.x & 63], is = sistd[threadIdx.x & 63];
        for (int t = g; t < n; t += 4) out[(size_t)t * dim + c] = mul(sub(x[(size_t)t * dim + c], m), is);
    }
}

// General splice: tf_fea_splice (fea.py:142-149) with `edge` = 0 (context frames outside the utterance are
// zeros), or Kaldi splice-feats' behaviour with `edge` = 1 (they repeat the first / last frame); keeps frames
// offset, offset+sub, offset+2*sub, ...
__global__ void fea_splice_kernel(const float* __restrict__ x, int n, int dim, int left, int right, int sub, int offset,
                                  int edge, int t_out, float* __restrict__ out)
{
    const int width = dim * (left + right + 1);
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)t_out * width) return;
    const int tt = (int)(idx / width), c = (int)(idx - (long)tt * width);
    const int ctx = c / dim, d = c - ctx * dim;
    int f = offset + tt * sub + ctx - left;
    float v = 0.f;
    if (edge) f = clampi(f, 0, n - 1);
    if (f >= 0 && f < n) v = x[(size_t)f * dim + d];
    out[idx] = v;
}

inline unsigned blocks_for(long elems) { return (unsigned)((elems + 255) / 256); }

}  // namespace

hipError_t launch_fea_add_delt(const float* x, int n, int dim, float* out, hipStream_t s)
{
    if (n <= 0 || dim <= 0) return hipSuccess;
    hipLaunchKernelGGL(fea_add_delt_kernel, dim3(blocks_for((long)n * dim)), dim3(256), 0, s, x, n, dim, out);
    return hipGetLastError();
}

hipError_t launch_fea_cmvn_global(const float* x, const float* mean, const float* var, int n, int dim, float* out, hipStream_t s)
{
    if (n <= 0 || dim <= 0) return hipSuccess;
    hipLaunchKernelGGL(fea_cmvn_global_kernel, dim3(blocks_for((long)n * dim)), dim3(256), 0, s, x, mean, var, n, dim, out);
    return hipGetLastError();
}

hipError_t launch_fea_cmvn_utt(const float* x, int n, int dim, float* out, hipStream_t s)
{
    if (n <= 0 || dim <= 0) return hipSuccess;
    hipLaunchKernelGGL(fea_cmvn_utt_kernel, dim3((dim + 63) / 64), dim3(256), 0, s, x, n, dim, out);
    return hipGetLastError();
}

hipError_t launch_fea_splice(const float* x, int n, int dim, int left, int right, int sub, int offset, int edge, float* out,
                             hipStream_t s)
{
    if (n <= 0 || dim <= 0 || offset >= n) return hipSuccess;
    const int t_out = (n - offset + sub - 1) / sub;
    hipLaunchKernelGGL(fea_splice_kernel, dim3(blocks_for((long)t_out * dim * (left + right + 1))), dim3(256), 0, s, x, n, dim,
                       left, right, sub, offset, edge, t_out, out);
    return hipGetLastError();
}
