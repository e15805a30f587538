// Memory-bound row kernels of the inference path (wave64 reductions, 16-byte accesses):
// frame stacking, padding mask, input layer norm + timing signal, residual add + layer norm,
// embedding lookup, cross-K/V packing, fused log-softmax + top-k, and the beam pruning step.
// Reference citations are relative to /root/reference/transformer/.
#include "asr_common.h"
#include <math.h>

namespace {

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---- frame stacking: third_party/feat_convert/io/fea.py:142-149 + every `sub`-th frame ----------
// raw [n][raw_frames][dim] -> out [n][t][dim*(left+right+1)], t = ceil(raw_frames/sub);
// context frames outside [0, len_u) read as zero; stacked frames t >= ceil(len_u/sub) are zero rows.
__global__ void frame_stack_kernel(const float* __restrict__ raw, const int32_t* __restrict__ lens, int raw_frames,
                                   int dim, int left, int right, int sub, int t, float* __restrict__ out)
{
    const int u = blockIdx.y, tt = blockIdx.x;
    const int len = lens ? lens[u] : raw_frames;
    const int tlen = (len + sub - 1) / sub;
    const int width = dim * (left + right + 1);
    float* o = out + ((size_t)u * t + tt) * width;
    const float* r = raw + (size_t)u * raw_frames * dim;
    for (int c = threadIdx.x; c < width; c += blockDim.x) {
        const int ctx = c / dim, d = c - ctx * dim;
        const int f = tt * sub + ctx - left;
        float v = 0.f;
        if (tt < tlen && f >= 0 && f < len) v = r[(size_t)f * dim + d];
        o[c] = v;
    }
}

// ---- encoder_padding = (sum_d |X| == 0)  (model.py:450) -----------------------------------------
__global__ void pad_mask_kernel(const float* __restrict__ X, int rows, int d, int32_t* __restrict__ pad)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* x = X + (size_t)row * d;
    float s = 0.f;
    for (int c = lane; c < d; c += 64) s += fabsf(x[c]);
    s = wave_sum(s);
    if (lane == 0) pad[row] = (s == 0.0f) ? 1 : 0;
}

// ---- tf.contrib.layers.layer_norm over (t, hidden) + timing signal (model.py:453,456; Q1) -------
constexpr int IN_CHUNKS = 64;     // partial sums per utterance

template <int PASS>   // 0: sum x    1: sum (x - mean)^2
__global__ __launch_bounds__(256) void input_norm_stats_kernel(const float* __restrict__ x, size_t per_utt,
                                                                double* __restrict__ scratch)
{
    // scratch layout per utterance u: [0..63] sums, [64..127] squared-deviation sums
    const int u = blockIdx.y, c = blockIdx.x;
    double* su = scratch + (size_t)u * (2 * IN_CHUNKS);
    float mean = 0.f;
    if (PASS == 1) {
        double tot = 0.0;
        for (int i = 0; i < IN_CHUNKS; ++i) tot += su[i];
        mean = (float)(tot / (double)per_utt);
    }
    const size_t n4 = per_utt / 4;
    const size_t per_chunk = (n4 + IN_CHUNKS - 1) / IN_CHUNKS;
    const size_t beg = (size_t)c * per_chunk, end = min(n4, beg + per_chunk);
    const f32x4* xp = reinterpret_cast<const f32x4*>(x + (size_t)u * per_utt);
    float acc = 0.f;
    for (size_t i = beg + threadIdx.x; i < end; i += 256) {
        f32x4 v = xp[i];
        if (PASS == 0) acc += (v[0] + v[1]) + (v[2] + v[3]);
        else {
            const float a = v[0] - mean, b = v[1] - mean, cc = v[2] - mean, d = v[3] - mean;
            acc += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    __shared__ double red[4];
    double dsum = wave_sum_d((double)acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dsum;
    __syncthreads();
    if (threadIdx.x == 0) su[PASS * IN_CHUNKS + c] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void input_norm_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ pe,
                                                                int t, int hidden, const double* __restrict__ scratch,
                                                                float* __restrict__ y)
{
    const int u = blockIdx.y;
    const size_t per_utt = (size_t)t * hidden;
    const double* su = scratch + (size_t)u * (2 * IN_CHUNKS);
    double s0 = 0.0, s1 = 0.0;
    for (int i = 0; i < IN_CHUNKS; ++i) { s0 += su[i]; s1 += su[IN_CHUNKS + i]; }
    const float mean = (float)(s0 / (double)per_utt);
    const float var = (float)(s1 / (double)per_utt);
    const float rstd = 1.0f / sqrtf(var + 1e-12f);
    const int h4 = hidden / 4;
    const size_t n4 = per_utt / 4;
    const f32x4* xp = reinterpret_cast<const f32x4*>(x + (size_t)u * per_utt);
    f32x4* yp = reinterpret_cast<f32x4*>(y + (size_t)u * per_utt);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(gamma);
    const f32x4* b4 = reinterpret_cast<const f32x4*>(beta);
    const f32x4* pe4 = reinterpret_cast<const f32x4*>(pe);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int hc = (int)(i % h4);
        const f32x4 v = xp[i], g = g4[hc], b = b4[hc], p = pe4[i];   // pe is [t][hidden], same linear index
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float inv = rstd * g[j];               // tf.nn.batch_normalization: x*inv + (beta - mean*inv)
            o[j] = (v[j] * inv + (b[j] - mean * inv)) + p[j];
        }
        yp[i] = o;
    }
}

// ---- utils.residual: layer_norm(x + y), eps 1e-6 (utils.py:494-507; common_layers.py:333-338) ---
// One wave per row.  Input row = resid + dbias + sum_z part[z] (split-K slabs reduced here, in z
// order, so the projection GEMM needs no reduction pass of its own).
template <int NCH>   // 16-byte chunks per lane: hidden <= NCH*256
__global__ __launch_bounds__(256) void add_layer_norm_kernel(const float* __restrict__ resid, const float* __restrict__ part,
                                                              int nsplit, size_t split_stride, const float* __restrict__ dbias,
                                                              const float* __restrict__ scale, const float* __restrict__ bias,
                                                              const int32_t* __restrict__ mask, int32_t* __restrict__ pad2_out,
                                                              int rows, int hidden, float* __restrict__ y,
                                                              const int32_t* __restrict__ done)
{
    if (done && *done) return;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int h4 = hidden >> 2;
    const size_t base = (size_t)row * hidden;
    f32x4 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ch = lane + 64 * c;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (ch < h4) {
            a = reinterpret_cast<const f32x4*>(resid + base)[ch];
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
            for (int z = 0; z < nsplit; ++z) d += reinterpret_cast<const f32x4*>(part + z * split_stride + base)[ch];
            if (dbias) d += reinterpret_cast<const f32x4*>(dbias)[ch];
            a += d;                                         // inputs + outputs (utils.py:505)
            s += (a[0] + a[1]) + (a[2] + a[3]);
        }
        v[c] = a;
    }
    const float mean = wave_sum(s) / (float)hidden;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (lane + 64 * c < h4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float dlt = v[c][j] - mean; q += dlt * dlt; }
        }
    }
    const float var = wave_sum(q) / (float)hidden;
    const float rstd = 1.0f / sqrtf(var + 1e-6f);
    const bool zero_row = mask && mask[row];
    float asum = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ch = lane + 64 * c;
        if (ch < h4) {
            const f32x4 g = reinterpret_cast<const f32x4*>(scale)[ch], b = reinterpret_cast<const f32x4*>(bias)[ch];
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = ((v[c][j] - mean) * rstd) * g[j] + b[j];
                if (zero_row) o[j] = o[j] * 0.0f;            // encoder_output *= (1 - padding) (model.py:489)
                asum += fabsf(o[j]);
            }
            reinterpret_cast<f32x4*>(y + base)[ch] = o;
        }
    }
    if (pad2_out) {                                          // decoder re-detects padding (model.py:566)
        asum = wave_sum(asum);
        if (lane == 0) pad2_out[row] = (asum == 0.0f) ? 1 : 0;
    }
}

// Same operation for the decode-time row counts (a few hundred rows): one 256-thread workgroup per row so
// that the launch covers the chip, and the split-K slabs are fetched four at a time (independent loads in
// flight) but still summed in slab order.
template <int NCH>   // 16-byte chunks per thread: hidden <= NCH*1024
__global__ __launch_bounds__(256) void add_layer_norm_row_kernel(const float* __restrict__ resid, const float* __restrict__ part,
                                                                  int nsplit, size_t split_stride, const float* __restrict__ dbias,
                                                                  const float* __restrict__ scale, const float* __restrict__ bias,
                                                                  const int32_t* __restrict__ mask, int32_t* __restrict__ pad2_out,
                                                                  int hidden, float* __restrict__ y, const int32_t* __restrict__ done)
{
    if (done && *done) return;
    __shared__ float red[3][4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h4 = hidden >> 2;
    const size_t base = (size_t)row * hidden;
    f32x4 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ch = tid + 256 * c;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (ch < h4) {
            a = reinterpret_cast<const f32x4*>(resid + base)[ch];
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
            for (int z0 = 0; z0 < nsplit; z0 += 4) {
                f32x4 t4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int z = min(z0 + j, nsplit - 1);
                    t4[j] = reinterpret_cast<const f32x4*>(part + z * split_stride + base)[ch];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (z0 + j < nsplit) d += t4[j];
            }
            if (dbias) d += reinterpret_cast<const f32x4*>(dbias)[ch];
            a += d;
            s += (a[0] + a[1]) + (a[2] + a[3]);
        }
        v[c] = a;
    }
    s = wave_sum(s);
    if (lane == 0) red[0][wave] = s;
    __syncthreads();
    const float mean = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)hidden;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (tid + 256 * c < h4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float dlt = v[c][j] - mean; q += dlt * dlt; }
        }
    }
    q = wave_sum(q);
    if (lane == 0) red[1][wave] = q;
    __syncthreads();
    const float var = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (float)hidden;
    const float rstd = 1.0f / sqrtf(var + 1e-6f);
    const bool zero_row = mask && mask[row];
    float asum = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ch = tid + 256 * c;
        if (ch < h4) {
            const f32x4 g = reinterpret_cast<const f32x4*>(scale)[ch], b = reinterpret_cast<const f32x4*>(bias)[ch];
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = ((v[c][j] - mean) * rstd) * g[j] + b[j];
                if (zero_row) o[j] = o[j] * 0.0f;
                asum += fabsf(o[j]);
            }
            reinterpret_cast<f32x4*>(y + base)[ch] = o;
        }
    }
    if (pad2_out) {
        asum = wave_sum(asum);
        if (lane == 0) red[2][wave] = asum;
        __syncthreads();
        if (tid == 0) pad2_out[row] = (((red[2][0] + red[2][1]) + (red[2][2] + red[2][3])) == 0.0f) ? 1 : 0;
    }
}

// ---- embedding * sqrt(H) + timing signal of the current position (model.py:569-576) ------------
__global__ __launch_bounds__(256) void embed_pe_kernel(const float* __restrict__ E, const int32_t* __restrict__ preds0,
                                                        const int32_t* __restrict__ preds1, int preds_stride,
                                                        const int32_t* __restrict__ pos_dev, const float* __restrict__ pe,
                                                        float mult, int hidden, float* __restrict__ y,
                                                        const int32_t* __restrict__ done)
{
    if (*done) return;
    const int r = blockIdx.x;
    const int pos = *pos_dev;
    const int32_t* preds = (pos & 1) ? preds1 : preds0;
    const int tok = preds[(size_t)r * preds_stride + pos];
    const f32x4* e = reinterpret_cast<const f32x4*>(E + (size_t)tok * hidden);
    const f32x4* p = reinterpret_cast<const f32x4*>(pe + (size_t)pos * hidden);
    f32x4* o = reinterpret_cast<f32x4*>(y + (size_t)r * hidden);
    for (int c = threadIdx.x; c < hidden / 4; c += 256) {
        f32x4 v = e[c];
        if (mult != 1.0f) v = v * mult;
        o[c] = v + p[c];
    }
}

// ---- teacher-forced decoder input (model.py:171 shift_right + :500-508): row r = (utterance, position p) gets
// E[p == 0 ? 2 : Y[u][p-1]] * mult + timing signal of position p ------------------------------------------------
__global__ __launch_bounds__(256) void embed_shifted_kernel(const float* __restrict__ E, const int32_t* __restrict__ Y, int ty,
                                                             int vocab, const float* __restrict__ pe, float mult, int hidden,
                                                             float* __restrict__ y)
{
    const int r = blockIdx.x, p = r % ty;
    int tok = (p == 0) ? 2 : Y[r - 1];                       // utils.shift_right: <S> then the targets but the last
    tok = min(max(tok, 0), vocab - 1);
    const f32x4* e = reinterpret_cast<const f32x4*>(E + (size_t)tok * hidden);
    const f32x4* pp = reinterpret_cast<const f32x4*>(pe + (size_t)p * hidden);
    f32x4* o = reinterpret_cast<f32x4*>(y + (size_t)r * hidden);
    for (int c = threadIdx.x; c < hidden / 4; c += 256) {
        f32x4 v = e[c];
        if (mult != 1.0f) v = v * mult;
        o[c] = v + pp[c];
    }
}

// ---- model.test_loss (model.py:350-360): per row softmax_cross_entropy_with_logits against one_hot(Y), times
// (Y != 0); the rows are summed by the host in float64 ------------------------------------------------------------
__global__ __launch_bounds__(256) void xent_rows_kernel(const float* __restrict__ logits, const int32_t* __restrict__ Y, int vocab,
                                                         float* __restrict__ row_loss)
{
    __shared__ float red[4];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* x = logits + (size_t)r * vocab;
    float m = -INFINITY;
    for (int i = tid; i < vocab; i += 256) m = fmaxf(m, x[i]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int i = tid; i < vocab; i += 256) s += expf(x[i] - m);
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) {
        const int yv = Y[r];
        const float lse = logf((red[0] + red[1]) + (red[2] + red[3]));
        const float xy = (yv >= 0 && yv < vocab) ? x[yv] : 0.0f;      // one_hot of an out-of-range label is all zeros
        const float l = (yv >= 0 && yv < vocab) ? (lse - (xy - m)) : 0.0f;
        row_loss[r] = (yv != 0) ? l : 0.0f;
    }
}

// ---- cross-attention K/V layouts: kv [n*t][2H] -> k, v [n][heads][t][64] (head-major, so that one
// (utterance, head) workgroup of the decode-time cross-attention streams two contiguous arrays) ----
template <bool KV16>
__global__ __launch_bounds__(256) void pack_cross_kv_kernel(const float* __restrict__ kv, int t, int heads,
                                                             void* __restrict__ kout, void* __restrict__ vout,
                                                             int32_t* __restrict__ range_flag)
{
    bool fits = true;                                          // every value below fp16's largest (attention.hip F16X3)
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    const int t0 = blockIdx.x * 64, h = blockIdx.y, u = blockIdx.z;
    const int hidden = heads * 64;
    for (int i = threadIdx.x; i < 64 * 16; i += 256) {
        const int tt = i >> 4, c = i & 15;
        if (t0 + tt < t) {
            const float* row = kv + ((size_t)u * t + t0 + tt) * (2 * hidden);
            const size_t o = (((size_t)u * heads + h) * t + t0 + tt) * 64;
            const f32x4 k4 = reinterpret_cast<const f32x4*>(row + h * 64)[c], v4 = reinterpret_cast<const f32x4*>(row + hidden + h * 64)[c];
#pragma unroll
            for (int e = 0; e < 4; ++e) fits = fits && fabsf(k4[e]) < 65504.0f && fabsf(v4[e]) < 65504.0f;   // (false for NaN)
            if constexpr (KV16) {
                reinterpret_cast<u32x2_t*>(static_cast<uint16_t*>(kout) + o)[c] = __builtin_bit_cast(u32x2_t, __builtin_convertvector(k4, bf16x4_t));
                reinterpret_cast<u32x2_t*>(static_cast<uint16_t*>(vout) + o)[c] = __builtin_bit_cast(u32x2_t, __builtin_convertvector(v4, bf16x4_t));
            } else {
                reinterpret_cast<f32x4*>(static_cast<float*>(kout) + o)[c] = k4;
                reinterpret_cast<f32x4*>(static_cast<float*>(vout) + o)[c] = v4;
            }
        }
    }
    if (!fits && range_flag) atomicOr(range_flag, 1);
}

// A logit that is NaN or infinite (garbage in: NaN features, an overflowed layer) must not reach the selection code:
// with NaN every comparison is false, no candidate would be chosen and the beam search would index memory with
// whatever the output buffers held.  Finite values pass unchanged.
__device__ __forceinline__ float finite_logit(float x) { return (x == x) ? fminf(fmaxf(x, -3.0e38f), 3.0e38f) : -3.0e38f; }

// ---- model.test_output: log_softmax + top_k (model.py:345-347) -----------------------------------
// One workgroup per row; the row lives in LDS (vocab*4 bytes).  Candidate order: value descending,
// ties -> lower index (Q5).  z = (x - max) - log(sum exp(x - max)) exactly as tf.nn.log_softmax.
// The logits may arrive as split-K slabs (summed here in slab order).  Top-k: every wave extracts the k best of
// its own quarter of the row with wave-level arg-max rounds (each lane caches the best of the elements it owns
// and rescans only after one of them was taken), then the 4k candidates are ranked by counting.
__global__ __launch_bounds__(256) void log_softmax_topk_kernel(const float* __restrict__ logits, int nsplit, size_t split_stride,
                                                                int vocab, int k, float* __restrict__ values,
                                                                int32_t* __restrict__ indices, const int32_t* __restrict__ done)
{
    if (done && *done) return;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* z = reinterpret_cast<float*>(smem_raw);
    __shared__ float redv[4];
    __shared__ float candv[4 * ASR_MAX_BEAM];
    __shared__ int candi[4 * ASR_MAX_BEAM];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* x = logits + (size_t)r * vocab;
    float m = -INFINITY;
    for (int i = tid; i < vocab; i += 256) {
        float v = x[i];
        for (int zz = 1; zz < nsplit; ++zz) v += x[zz * split_stride + i];
        v = finite_logit(v);
        z[i] = v;
        m = fmaxf(m, v);
    }
    m = wave_max(m);
    if (lane == 0) redv[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(redv[0], redv[1]), fmaxf(redv[2], redv[3]));
    __syncthreads();
    float s = 0.f;
    for (int i = tid; i < vocab; i += 256) { const float sh = z[i] - m; z[i] = sh; s += expf(sh); }   // own elements only
    s = wave_sum(s);
    if (lane == 0) redv[wave] = s;
    __syncthreads();
    const float logz = logf((redv[0] + redv[1]) + (redv[2] + redv[3]));
    // from here on a thread touches only the elements it owns (i = tid + 256 j): no workgroup barrier needed
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < vocab; i += 256) { const float v = z[i] - logz; z[i] = v; if (v > bv) { bv = v; bi = i; } }   // first max kept
    for (int round = 0; round < k; ++round) {
        float wv = bv;
        int wi = bi;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(wv, o);
            const int oi = __shfl_xor(wi, o);
            if (ov > wv || (ov == wv && oi < wi)) { wv = ov; wi = oi; }
        }
        if (lane == 0) { candv[wave * k + round] = wv; candi[wave * k + round] = wi; }
        if (wi == bi && bi != 0x7fffffff) {                   // this lane owned the winner: retire it, rescan own elements
            z[bi] = -INFINITY;
            bv = -INFINITY; bi = 0x7fffffff;
            for (int i = tid; i < vocab; i += 256) { const float v = z[i]; if (v > bv) { bv = v; bi = i; } }
        }
    }
    if (tid < k) { values[(size_t)r * k + tid] = -INFINITY; indices[(size_t)r * k + tid] = 0; }   // ranks no candidate reaches (vocab < k only)
    __syncthreads();
    // rank the 4k candidates (value desc, index asc); entries with index 0x7fffffff are exhausted quarters
    const int nc = 4 * k;
    if (tid < nc) {
        const float v = candv[tid];
        const int idx = candi[tid];
        int rank = 0;
        for (int o = 0; o < nc; ++o) {
            const float w = candv[o];
            const int wi = candi[o];
            rank += (w > v || (w == v && wi < idx)) ? 1 : 0;
        }
        if (rank < k && idx != 0x7fffffff) { values[(size_t)r * k + rank] = v; indices[(size_t)r * k + rank] = idx; }
    }
}

// Same result for rows of at most 1024*NV elements, held in REGISTERS (thread t owns the 16-byte groups t, t+256, ...):
// no LDS image of the row, 16-byte loads of the slabs, and the arg-max rounds rescan registers.
template <int NV, int NW>   // NW waves per row (4 or 16): thread t owns the 16-byte groups t, t + 64 NW, ...
__global__ __launch_bounds__(64 * NW) void log_softmax_topk_reg_kernel(const float* __restrict__ logits, int nsplit, size_t split_stride,
                                                                    int vocab, int k, float* __restrict__ values,
                                                                    int32_t* __restrict__ indices, const int32_t* __restrict__ done)
{
    if (done && *done) return;
    constexpr int NT = 64 * NW;
    __shared__ float redv[2][NW];
    __shared__ float candv[NW * ASR_MAX_BEAM];
    __shared__ int candi[NW * ASR_MAX_BEAM];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const f32x4* x = reinterpret_cast<const f32x4*>(logits + (size_t)r * vocab);
    const size_t stride4 = split_stride / 4;
    const int groups = vocab >> 2;
    float v[NV * 4];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int g = tid + NT * j;
        const bool ok = g < groups;
        const int gc = ok ? g : 0;                                   // branch-free, clamped address
        f32x4 a = x[gc];
        for (int zz = 1; zz < nsplit; ++zz) {
            const f32x4 b = x[zz * stride4 + gc];
            a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) v[j * 4 + c] = ok ? finite_logit(a[c]) : -INFINITY;
    }
    float m = -INFINITY;
#pragma unroll
    for (int e = 0; e < NV * 4; ++e) m = fmaxf(m, v[e]);
    m = wave_max(m);
    if (lane == 0) redv[0][wave] = m;
    __syncthreads();
    m = redv[0][0];
#pragma unroll
    for (int w = 1; w < NW; ++w) m = fmaxf(m, redv[0][w]);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < NV * 4; ++e) { v[e] -= m; s += expf(v[e]); }   // exp(-inf) = 0 for the slots past the row
    s = wave_sum(s);
    if (lane == 0) redv[1][wave] = s;
    __syncthreads();
    float zsum = 0.f;                                                // pairs of pairs, in wave order (NW = 4: (s0 + s1) + (s2 + s3))
#pragma unroll
    for (int w = 0; w < NW; w += 4) {
        const float q4 = (redv[1][w] + redv[1][w + 1]) + (redv[1][w + 2] + redv[1][w + 3]);
        zsum = (w == 0) ? q4 : zsum + q4;
    }
    const float logz = logf(zsum);
    // The lane's elements are kept in groups of 16 with the best of every group cached, so that after one of them has been
    // taken only its group is rescanned (a 28444-word row is 112 elements per lane: rescanning them all cost 190 us a step).
    // Ascending element order = ascending index, strict comparisons: the first maximum is kept everywhere.
    constexpr int NE = NV * 4, GS = NE < 16 ? NE : 16, NG = NE / GS;
    float gv[NG];
    int gi[NG];
    auto idx_of = [&](int e) { return 4 * (tid + NT * (e >> 2)) + (e & 3); };
    auto scan_group = [&](int g) {
        float bvg = -INFINITY;
        int big = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < GS; ++j) {
            const int e = g * GS + j;
            if (v[e] > bvg) { bvg = v[e]; big = idx_of(e); }
        }
        gv[g] = bvg; gi[g] = big;
    };
#pragma unroll
    for (int e = 0; e < NE; ++e) v[e] -= logz;
#pragma unroll
    for (int g = 0; g < NG; ++g) scan_group(g);
    float bv;
    int bi;
    auto best_of_groups = [&]() {
        bv = -INFINITY; bi = 0x7fffffff;
#pragma unroll
        for (int g = 0; g < NG; ++g)
            if (gv[g] > bv) { bv = gv[g]; bi = gi[g]; }
    };
    best_of_groups();
    for (int round = 0; round < k; ++round) {
        float wv = bv;
        int wi = bi;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(wv, o);
            const int oi = __shfl_xor(wi, o);
            if (ov > wv || (ov == wv && oi < wi)) { wv = ov; wi = oi; }
        }
        if (lane == 0) { candv[wave * k + round] = wv; candi[wave * k + round] = wi; }
        if (wi == bi && bi != 0x7fffffff) {                         // this lane owned the winner: retire it, rescan its group
            const int taken = bi;
            const int esel = (((taken >> 2) - tid) / NT) * 4 + (taken & 3);
#pragma unroll
            for (int g = 0; g < NG; ++g)
                if (esel / GS == g) {
#pragma unroll
                    for (int j = 0; j < GS; ++j)
                        if (idx_of(g * GS + j) == taken) v[g * GS + j] = -INFINITY;
                    scan_group(g);
                }
            best_of_groups();
        }
    }
    if (tid < k) { values[(size_t)r * k + tid] = -INFINITY; indices[(size_t)r * k + tid] = 0; }   // ranks no candidate reaches
    __syncthreads();
    const int nc = NW * k;
    if (tid < nc) {
        const float cv = candv[tid];
        const int idx = candi[tid];
        int rank = 0;
        for (int o = 0; o < nc; ++o) {
            const float w = candv[o];
            const int wi = candi[o];
            rank += (w > cv || (w == cv && wi < idx)) ? 1 : 0;
        }
        if (rank < k && idx != 0x7fffffff) { values[(size_t)r * k + rank] = cv; indices[(size_t)r * k + rank] = idx; }
    }
}

// ---- beam pruning (model.py:280-301) ------------------------------------------------
// One workgroup per utterance.  K*K candidates are ranked by counting (value desc, ties -> lower
// flat index), which reproduces a sorted tf.nn.top_k.
struct BeamLds {
    float tot[ASR_MAX_BEAM * ASR_MAX_BEAM];
    int ctok[ASR_MAX_BEAM * ASR_MAX_BEAM];
    float nscore[ASR_MAX_BEAM];
    int ntok[ASR_MAX_BEAM];
    int npar[ASR_MAX_BEAM];      // parent beam, local to the utterance
};

__device__ void beam_prune_block(const float* scores, const int32_t* finished, const float* cand_scores,
                                 const int32_t* cand_tokens, int k, int u, BeamLds& L)
{
    const int kk = k * k, tid = threadIdx.x, nth = blockDim.x;
    for (int c = tid; c < kk; c += nth) {
        const int beam = c / k, j = c - beam * k;
        const int r = u * k + beam;
        const bool f = finished[r] != 0;
        // scores*(1-f) + b*f with b = [0, -1e10, ...]  (model.py:238-240); tokens*(1-f) + 3*f (:253)
        const float cs = f ? (j == 0 ? 0.0f : -1e10f) : cand_scores[(size_t)r * k + j];
        L.ctok[c] = f ? 3 : cand_tokens[(size_t)r * k + j];
        const float tot = scores[r] + cs;
        L.tot[c] = (tot == tot) ? tot : -INFINITY;          // NaN has no rank: the counting below needs a total order
    }
    __syncthreads();
    for (int c = tid; c < kk; c += nth) {
        const float v = L.tot[c];
        int rank = 0;
        for (int o = 0; o < kk; ++o) { const float w = L.tot[o]; rank += (w > v || (w == v && o < c)) ? 1 : 0; }
        if (rank < k) { L.nscore[rank] = v; L.ntok[rank] = L.ctok[c]; L.npar[rank] = c / k; }
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void beam_prune_kernel(const float* scores, const int32_t* finished, const float* cand_scores,
                                                          const int32_t* cand_tokens, int k, float* new_scores,
                                                          int32_t* tokens, int32_t* parents, int32_t* finished_out)
{
    __shared__ BeamLds L;
    const int u = blockIdx.x;
    beam_prune_block(scores, finished, cand_scores, cand_tokens, k, u, L);
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        const int r = u * k + i;
        new_scores[r] = L.nscore[i]; tokens[r] = L.ntok[i]; parents[r] = u * k + L.npar[i];
        finished_out[r] = (L.ntok[i] == 3) ? 1 : 0;
    }
}

__global__ void beam_init_kernel(BeamState st, int rows, int k)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r == 0) { *st.pos = 0; *st.done = 0; *st.unfinished = 0; st.pos[3] = 0; }
    if (r >= rows) return;
    if (r % k == 0) st.live_cnt[r / k] = 0;                  // no history before the first step
    st.scores[0][r] = (r % k == 0) ? 0.0f : -1e10f;          // model.py:264-265
    st.finished[0][r] = 0;                                   // :266
    st.preds[0][(size_t)r * st.width] = 2;                   // <S>  (:263)
}

// Also does what two more launches per step used to do: the embedding of the tokens just chosen (the next step's decoder
// input, model.py:569-576) for the utterance's own K rows, and - in the workgroup that finishes last - the loop condition
// of model.py:305-312 (beam_loop_ctl_kernel's arithmetic).
__global__ __launch_bounds__(256) void beam_step_kernel(BeamState st, const float* cand_scores, const int32_t* cand_tokens, int k,
                                                         const float* __restrict__ E, const float* __restrict__ pe, float mult,
                                                         int hidden, float* __restrict__ y_next)
{
    if (*st.done) return;
    __shared__ BeamLds L;
    const int u = blockIdx.x, tid = threadIdx.x;
    const int pos = *st.pos, cur = pos & 1, nxt = cur ^ 1;
    // joint decode of groups: this utterance's own loop (model.py:305-312 with ITS T) ends after limit_u + 1 steps
    const bool stopped = st.limit_u && pos + 1 > st.limit_u[u];
    beam_prune_block(st.scores[cur], st.finished[cur], cand_scores, cand_tokens, k, u, L);
    // preds = concat(gather(preds, parent), token); ancestry table gathered the same way (model.py:296-298)
    const int w = st.width;
    for (int i = tid; i < k * (pos + 1); i += blockDim.x) {
        const int beam = i / (pos + 1), j = i - beam * (pos + 1);
        const size_t dst = (size_t)(u * k + beam) * w + j, src = (size_t)(u * k + L.npar[beam]) * w + j;
        st.preds[nxt][dst] = st.preds[cur][src];
        st.anc[nxt][dst] = (j < pos) ? st.anc[cur][src] : L.npar[beam];   // position `pos` was computed in slot `parent`
    }
    if (tid < k) {
        const int r = u * k + tid;
        st.scores[nxt][r] = L.nscore[tid];
        st.finished[nxt][r] = (L.ntok[tid] == 3 || stopped) ? 1 : 0;      // :301
        st.preds[nxt][(size_t)r * w + pos + 1] = L.ntok[tid];
        st.tokens[r] = L.ntok[tid];
        st.parents[r] = u * k + L.npar[tid];
    }
    // decoder input of the next step: E[token] * mult + timing signal of position pos + 1, rows of this utterance
    {
        const f32x4* p4 = reinterpret_cast<const f32x4*>(pe + (size_t)(pos + 1) * hidden);
        const int h4 = hidden >> 2;
        for (int i = tid; i < k * h4; i += blockDim.x) {
            const int beam = i / h4, c4 = i - beam * h4;
            f32x4 v = reinterpret_cast<const f32x4*>(E + (size_t)L.ntok[beam] * hidden)[c4];
            if (mult != 1.0f) v = v * mult;
            reinterpret_cast<f32x4*>(y_next + (size_t)(u * k + beam) * hidden)[c4] = v + p4[c4];
        }
    }
    // Live rows of the K/V cache for the next step's self-attention: (position j, slot) pairs that at least one
    // of the new beams descends from, j = 0..pos, ordered by j then slot.  Each thread owns a contiguous run of
    // positions; a workgroup-wide exclusive scan of the per-thread counts gives the write offsets.
    {
        __shared__ int scan[256];
        const int npos = pos + 1, chunk = (npos + 255) / 256;
        const int j0 = min(tid * chunk, npos), j1 = min(j0 + chunk, npos);
        auto slots_of = [&](int j) {
            unsigned m = 0;
            for (int b = 0; b < k; ++b) {
                const int par = L.npar[b];
                const int sl = (j < pos) ? st.anc[cur][(size_t)(u * k + par) * w + j] : par;
                m |= 1u << sl;
            }
            return m;
        };
        int cnt = 0;
        for (int j = j0; j < j1; ++j) cnt += __popc(slots_of(j));
        scan[tid] = cnt;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {            // Hillis-Steele inclusive scan
            const int v = (tid >= off) ? scan[tid - off] : 0;
            __syncthreads();
            scan[tid] += v;
            __syncthreads();
        }
        int o = scan[tid] - cnt;
        int32_t* dst = st.live + (size_t)u * k * w;
        for (int j = j0; j < j1; ++j) {
            unsigned m = slots_of(j);
            while (m) { const int sl = __ffs(m) - 1; m &= m - 1; dst[o++] = (j << 5) | sl; }
        }
        if (tid == 255) st.live_cnt[u] = scan[255];
    }
    // Loop condition, by whichever workgroup arrives last.  The counters are device-scope atomics; pos / done are plain
    // words that only later kernels of the stream read (every workgroup of this launch has read them before it arrives).
    if (tid == 0) {
        int any = 0;
        for (int i = 0; i < k; ++i) any |= (L.ntok[i] != 3);
        if (any && !stopped) atomicAdd(st.unfinished, 1);
        __threadfence();                                      // the count before the ticket
        int32_t* arrive = st.pos + 3;
        if (atomicAdd(arrive, 1) == (int)gridDim.x - 1) {
            const int any_total = atomicExch(st.unfinished, 0);
            atomicExch(arrive, 0);
            const int np = pos + 1;
            *st.pos = np;
            if (!(any_total > 0 && np <= st.limit)) *st.done = 1;
        }
    }
}

}  // namespace

hipError_t launch_frame_stack_lens(const float* raw, const int32_t* lens, int n, int raw_frames, int raw_dim, int left,
                                   int right, int sub, float* out, hipStream_t s)
{
    if (n <= 0 || raw_frames <= 0) return hipSuccess;
    const int t = (raw_frames + sub - 1) / sub;
    hipLaunchKernelGGL(frame_stack_kernel, dim3(t, n), dim3(128), 0, s, raw, lens, raw_frames, raw_dim, left, right, sub, t, out);
    return hipGetLastError();
}
hipError_t launch_frame_stack(const float* raw, int n, int raw_frames, int raw_dim, int left, int right, int sub,
                              float* out, hipStream_t s)
{
    return launch_frame_stack_lens(raw, nullptr, n, raw_frames, raw_dim, left, right, sub, out, s);
}

hipError_t launch_pad_mask(const float* X, int rows, int d, int32_t* pad, hipStream_t s)
{
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(pad_mask_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, X, rows, d, pad);
    return hipGetLastError();
}

size_t input_norm_scratch_doubles(int n, int, int) { return (size_t)n * 2 * IN_CHUNKS; }

hipError_t launch_input_norm_timing(const float* x, const float* gamma, const float* beta, const float* pe, int n, int t,
                                    int hidden, double* scratch, float* y, hipStream_t s)
{
    if (n <= 0 || t <= 0) return hipSuccess;
    if (hidden & 3) return hipErrorInvalidValue;
    const size_t per_utt = (size_t)t * hidden;
    hipLaunchKernelGGL(input_norm_stats_kernel<0>, dim3(IN_CHUNKS, n), dim3(256), 0, s, x, per_utt, scratch);
    hipLaunchKernelGGL(input_norm_stats_kernel<1>, dim3(IN_CHUNKS, n), dim3(256), 0, s, x, per_utt, scratch);
    const int gx = (int)min((size_t)64, (per_utt / 4 + 255) / 256);
    hipLaunchKernelGGL(input_norm_apply_kernel, dim3(gx, n), dim3(256), 0, s, x, gamma, beta, pe, t, hidden, scratch, y);
    return hipGetLastError();
}

hipError_t launch_add_layer_norm_done(const float* resid, const float* part, int nsplit, size_t split_stride,
                                      const float* dbias, const float* scale, const float* bias, const int32_t* mask,
                                      int32_t* pad2_out, int rows, int hidden, float* y, const int32_t* done, hipStream_t s)
{
    if (rows <= 0) return hipSuccess;
    if ((hidden & 3) || hidden > 2048) return hipErrorInvalidValue;
    if (rows <= 4096) {                                      // decode-time rows: one workgroup per row
        if (hidden <= 1024)
            hipLaunchKernelGGL(add_layer_norm_row_kernel<1>, dim3(rows), dim3(256), 0, s, resid, part, nsplit, split_stride, dbias,
                               scale, bias, mask, pad2_out, hidden, y, done);
        else
            hipLaunchKernelGGL(add_layer_norm_row_kernel<2>, dim3(rows), dim3(256), 0, s, resid, part, nsplit, split_stride, dbias,
                               scale, bias, mask, pad2_out, hidden, y, done);
        return hipGetLastError();
    }
    dim3 grid((rows + 3) / 4), block(256);
#define ALN(N) hipLaunchKernelGGL(add_layer_norm_kernel<N>, grid, block, 0, s, resid, part, nsplit, split_stride, dbias, \
                                  scale, bias, mask, pad2_out, rows, hidden, y, done)
    if (hidden <= 256) ALN(1);
    else if (hidden <= 512) ALN(2);
    else if (hidden <= 1024) ALN(4);
    else ALN(8);
#undef ALN
    return hipGetLastError();
}
hipError_t launch_add_layer_norm(const float* resid, const float* part, int nsplit, size_t split_stride, const float* dbias,
                                 const float* scale, const float* bias, const int32_t* mask, int32_t* pad2_out, int rows,
                                 int hidden, float* y, hipStream_t s)
{
    return launch_add_layer_norm_done(resid, part, nsplit, split_stride, dbias, scale, bias, mask, pad2_out, rows, hidden, y,
                                      nullptr, s);
}

hipError_t launch_embed_pe(const float* E, const int32_t* preds0, const int32_t* preds1, int preds_stride,
                           const int32_t* pos_dev, const float* pe, float mult, int rows, int hidden, float* y,
                            const int32_t* done, hipStream_t s)
{
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(embed_pe_kernel, dim3(rows), dim3(256), 0, s, E, preds0, preds1, preds_stride, pos_dev, pe, mult,
                       hidden, y, done);
    return hipGetLastError();
}

hipError_t launch_embed_shifted(const float* E, const int32_t* Y, int rows, int ty, int vocab, const float* pe, float mult,
                                int hidden, float* y, hipStream_t s)
{
    if (rows <= 0) return hipSuccess;
    if (ty < 1 || (hidden & 3)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(embed_shifted_kernel, dim3(rows), dim3(256), 0, s, E, Y, ty, vocab, pe, mult, hidden, y);
    return hipGetLastError();
}

hipError_t launch_xent_rows(const float* logits, const int32_t* Y, int rows, int vocab, float* row_loss, hipStream_t s)
{
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(xent_rows_kernel, dim3(rows), dim3(256), 0, s, logits, Y, vocab, row_loss);
    return hipGetLastError();
}

hipError_t launch_pack_cross_kv(const float* kv, int n, int t, int heads, void* k, void* v, int kv_bf16, int32_t* range_flag, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    if (kv_bf16) hipLaunchKernelGGL(pack_cross_kv_kernel<true>, dim3((t + 63) / 64, heads, n), dim3(256), 0, s, kv, t, heads, k, v, range_flag);
    else hipLaunchKernelGGL(pack_cross_kv_kernel<false>, dim3((t + 63) / 64, heads, n), dim3(256), 0, s, kv, t, heads, k, v, range_flag);
    return hipGetLastError();
}

hipError_t launch_log_softmax_topk(const float* logits, int nsplit, size_t split_stride, int rows, int vocab, int k, float* values,
                                   int32_t* indices, const int32_t* done, hipStream_t s)
{
    if (rows <= 0) return hipSuccess;
    const size_t lds = (size_t)vocab * sizeof(float);
    if (lds > 150 * 1024 || k < 1 || k > vocab || k > ASR_MAX_BEAM || nsplit < 1) return hipErrorInvalidValue;
    // rows of up to 32768 elements live in registers (16-byte slab loads, no LDS image); the LDS variant (one 4-byte load
    // per element, one workgroup per CU, arg-max rescans through LDS) took 368 us per step at the word vocabulary (28444)
    // against 18 us at 3900
    if ((vocab & 3) == 0 && (split_stride & 3) == 0 && vocab <= 32768 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0) {
#define TOPK_REG(NV, NW) hipLaunchKernelGGL((log_softmax_topk_reg_kernel<NV, NW>), dim3(rows), dim3(64 * NW), 0, s, logits, nsplit, \
                                            split_stride, vocab, k, values, indices, done)
        if (vocab <= 2048) TOPK_REG(2, 4);
        else if (vocab <= 4096) TOPK_REG(4, 4);
        else if (vocab <= 8192) TOPK_REG(2, 16);          // long rows: 1024 threads a row keep the per-thread chains short
        else if (vocab <= 16384) TOPK_REG(4, 16);
        else TOPK_REG(8, 16);
#undef TOPK_REG
        return hipGetLastError();
    }
    static std::atomic<bool> attr_done[64];      // zero-initialised; several handles' host threads may launch first at once (the call is idempotent)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_done[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(log_softmax_topk_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return e;
        attr_done[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(log_softmax_topk_kernel, dim3(rows), dim3(256), lds, s, logits, nsplit, split_stride, vocab, k, values, indices, done);
    return hipGetLastError();
}

static bool beam_state_ok(const BeamState& st, int b, int k)
{
    if (b < 1 || k < 1 || k > ASR_MAX_BEAM || st.width < 2) return false;
    for (int i = 0; i < 2; ++i)
        if (!st.scores[i] || !st.finished[i] || !st.preds[i] || !st.anc[i]) return false;
    return st.live && st.live_cnt && st.pos && st.done && st.unfinished && st.tokens && st.parents;
}

hipError_t launch_beam_init(const BeamState& st, int b, int k, hipStream_t s)
{
    if (!beam_state_ok(st, b, k)) return hipErrorInvalidValue;      // a null table would fault the card
    const int rows = b * k;
    hipLaunchKernelGGL(beam_init_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, st, rows, k);
    return hipGetLastError();
}

hipError_t launch_beam_step(const BeamState& st, const float* cand_scores, const int32_t* cand_tokens, int b, int k,
                            const float* E, const float* pe, float mult, int hidden, float* y_next, hipStream_t s)
{
    if (!beam_state_ok(st, b, k) || !cand_scores || !cand_tokens || !E || !pe || !y_next || (hidden & 3)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(beam_step_kernel, dim3(b), dim3(256), 0, s, st, cand_scores, cand_tokens, k, E, pe, mult, hidden, y_next);
    return hipGetLastError();
}

hipError_t launch_beam_prune(const float* scores, const int32_t* finished, const float* cand_scores,
                             const int32_t* cand_tokens, int b, int k, float* new_scores, int32_t* tokens,
                             int32_t* parents, int32_t* finished_out, hipStream_t s)
{
    hipLaunchKernelGGL(beam_prune_kernel, dim3(b), dim3(256), 0, s, scores, finished, cand_scores, cand_tokens, k,
                       new_scores, tokens, parents, finished_out);
    return hipGetLastError();
}
