// Attention kernels of the inference path (d_head = 64, fp32):
//   * encoder self-attention: flash-style, exact-f32 MFMA, key-padding mask, K/V tiles staged in LDS
//   * decoder self-attention: one query per row against the per-beam K/V cache in HBM
//     (ancestry-table indirection instead of copying the cache on every beam reorder)
//   * decoder cross-attention: the K beams of an utterance share one pass over that utterance's K/V
// Replaces utils.multihead_attention's core (transformer/utils.py:621-631) and
// common_attention.dot_product_attention (third_party/tensor2tensor/common_attention.py:295-334).
#include "asr_common.h"
#include <type_traits>
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

// ------------------------------------------------------------------------------------------------
// Many-query attention (encoder self-attention; causal self- and cross-attention of the teacher-forced
// decoder).  grid (ceil(tq/128), heads, n), 4 waves x 32 query rows.
// S^T = K.Q^T is computed (keys on the MFMA row index, queries on the lane), so each lane owns one
// query: row max / sum are in-lane over 16 registers plus one exchange with lane^32, and the S^T
// accumulator is already the B operand of O^T += V^T.P^T (no LDS round trip for P).
// Masks, exactly as the reference: -1e9 added to logits of padded keys (common_attention.py:202) and of
// keys after the query when `causal` (attention_bias_lower_triangle, common_attention.py:179-190; key
// tiles wholly after the workgroup's last query are skipped - their weights are exactly 0 in f32);
// keys beyond tk (tile padding only) are excluded with -inf.
// ------------------------------------------------------------------------------------------------
struct MhaArgs {
    const float* q; const float* k; const float* v;   // row pointers of utterance 0, head 0
    int ldq, ldk, ldv;                                // floats between consecutive rows
    const int32_t* pad;                               // [n][tk] key padding (1 = padded) or null
    int tq, tk, heads, causal;
    float* ctx;                                       // [n][tq][heads*64]
};

__global__ __launch_bounds__(256) void encoder_attention_kernel(MhaArgs a)
{
    __shared__ f32x4 Ks[32 * 16];        // [key][16-B chunk ^ (key & 15)]
    __shared__ float Vs[32 * 64];        // [key][d]
    __shared__ float Bias[32];
    __shared__ float Os[4][32][65];

    const int qt = blockIdx.x, h = blockIdx.y, u = blockIdx.z;
    const int heads = a.heads, hidden = heads * 64;
    const int t = a.tq, tk = a.tk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int q0w = qt * 128 + wave * 32;
    const int q = q0w + l31;
    const float* qbase = a.q + (size_t)u * t * a.ldq;
    const float* kbase = a.k + (size_t)u * tk * a.ldk;
    const float* vbase = a.v + (size_t)u * tk * a.ldv;
    const int32_t* pad = a.pad;
    float* ctx = a.ctx;

    f32x4 qf[8];
#pragma unroll
    for (int qq = 0; qq < 8; ++qq) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (q < t) v = *reinterpret_cast<const f32x4*>(qbase + (size_t)q * a.ldq + h * 64 + (2 * qq + half) * 4);
        qf[qq] = v * 0.125f;                                   // q *= key_depth_per_head ** -0.5 (utils.py:628)
    }

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    int nkt = (tk + 31) / 32;
    if (a.causal) nkt = min(nkt, (min(qt * 128 + 127, t - 1) >> 5) + 1);
    for (int kt = 0; kt < nkt; ++kt) {
        const int k0 = kt * 32;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i, row = idx >> 4, c = idx & 15;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (k0 + row < tk) {
                kv = *reinterpret_cast<const f32x4*>(kbase + (size_t)(k0 + row) * a.ldk + h * 64 + c * 4);
                vv = *reinterpret_cast<const f32x4*>(vbase + (size_t)(k0 + row) * a.ldv + h * 64 + c * 4);
            }
            Ks[row * 16 + (c ^ (row & 15))] = kv;
            reinterpret_cast<f32x4*>(Vs)[row * 16 + c] = vv;
        }
        if (tid < 32) {
            const int kk = k0 + tid;
            Bias[tid] = (kk >= tk) ? -INFINITY : ((pad && pad[(size_t)u * tk + kk]) ? -1e9f : 0.0f);
        }
        __syncthreads();

        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int qq = 0; qq < 8; ++qq) {
            const f32x4 a = Ks[l31 * 16 + ((2 * qq + half) ^ (l31 & 15))];
#pragma unroll
            for (int j = 0; j < 4; ++j) s = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], qf[qq][j], s, 0, 0, 0);
        }
        // s[r] = S[query l31][key k0 + (r&3) + 8*(r>>2) + 4*half]
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kl = (r & 3) + 8 * (r >> 2) + 4 * half;
            s[r] = s[r] + Bias[kl];
            if (a.causal && k0 + kl > q) s[r] += -1e9f;
            mx = fmaxf(mx, s[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = expf(m_run - m_new);
        float ls = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = expf(s[r] - m_new); ls += s[r]; }
        l_run = l_run * alpha + ls;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = (r & 3) + 8 * (r >> 2) + 4 * half;
            const float a0 = Vs[key * 64 + l31], a1 = Vs[key * 64 + 32 + l31];
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, s[r], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, s[r], o1, 0, 0, 0);
        }
        __syncthreads();
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int d = (r & 3) + 8 * (r >> 2) + 4 * half;
        Os[wave][l31][d] = o0[r] / l_tot;
        Os[wave][l31][d + 32] = o1[r] / l_tot;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = i * 64 + lane, row = idx >> 4, c = idx & 15;
        if (q0w + row < t) {
            f32x4 v;
            v[0] = Os[wave][row][c * 4 + 0]; v[1] = Os[wave][row][c * 4 + 1];
            v[2] = Os[wave][row][c * 4 + 2]; v[3] = Os[wave][row][c * 4 + 3];
            *reinterpret_cast<f32x4*>(ctx + ((size_t)u * t + q0w + row) * hidden + h * 64 + c * 4) = v;
        }
    }
}

// exp(x) for the soft-max weights (x <= 0 up to rounding) in 6 instructions: x log2(e) is formed in two pieces so that
// the argument of v_exp_f32 carries no rounding error of its own (the library expf spends twice as many on range checks
// these arguments never need).  Error ~1.5 ulp; exp(-inf) = 0.
__device__ __forceinline__ float exp_neg(float x)
{
    const float t = x * 1.44269502e+00f;
    float e = fmaf(x, 1.44269502e+00f, -t);
    e = fmaf(x, 1.92596303e-08f, e);
    const float p = __builtin_amdgcn_exp2f(t);
    return (x == -INFINITY) ? 0.0f : fmaf(p, e * 6.93147182e-01f, p);
}


// ---- K/V storage of the decode-time attention kernels ---------------------------------------------------------
// KV16 = false: float32 rows (every float32 precision mode).  KV16 = true (`precision: bf16` only): the caches hold bf16
// (round to nearest even when a row is written), half the bytes of the HBM-bound key loop; the rows go to the bf16 matrix
// instruction as loaded, q and P as two bf16 terms each (16 significant bits), accumulation, scores and soft-max in
// float32 (kv_frags_bf16 / split_bf16x2 below).  A lane's 16 K values of one key are, in both
// modes, 16-byte loads: float32 4 x (4 consecutive d at 16 q + 4 g), bf16 2 x (8 consecutive d at 32 (q / 2) + 8 g);
// the MFMA's k index is only a summation label, so q fragments simply follow the same order (kv_d_of).
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));

template <bool KV16> __device__ __forceinline__ int kv_d_of(int q, int g)      // first of the 4 consecutive d of fragment q (0..3)
{
    return KV16 ? 32 * (q >> 1) + 8 * g + 4 * (q & 1) : 16 * q + 4 * g;
}
__device__ __forceinline__ f32x4 bf16_lo_hi(uint32_t w0, uint32_t w1)           // 4 bf16 in two words -> 4 floats
{
    f32x4 r;
    r[0] = __builtin_bit_cast(float, w0 << 16); r[1] = __builtin_bit_cast(float, w0 & 0xffff0000u);
    r[2] = __builtin_bit_cast(float, w1 << 16); r[3] = __builtin_bit_cast(float, w1 & 0xffff0000u);
    return r;
}
__device__ __forceinline__ u32x2_t f32x4_to_bf16(f32x4 x)                       // round to nearest even (v_cvt_pk_bf16_f32)
{
    return __builtin_bit_cast(u32x2_t, __builtin_convertvector(x, bf16x4_t));
}
__device__ __forceinline__ f32x4 bf16_round(f32x4 x)
{
    const u32x2_t b = f32x4_to_bf16(x);
    return bf16_lo_hi(b[0], b[1]);
}
// registers a key block waits in between its loads and its MFMAs
template <bool KV16> struct KvRegs;
template <> struct KvRegs<false> { f32x4 k[4]; f32x4 v[4]; };
template <> struct KvRegs<true> { u32x4_t k[2]; u32x2_t v[4]; };
template <bool KV16> __device__ __forceinline__ void kv_load(KvRegs<KV16>& r, const void* kbase, size_t krow_elem, const void* vbase,
                                                             const size_t (&vrow_elem)[4], int g, int c)
{
    // krow_elem / vrow_elem: element index of the row's d = 0 inside the cache
    if constexpr (KV16) {
        const uint16_t* kp = static_cast<const uint16_t*>(kbase) + krow_elem + 8 * g;
        r.k[0] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(kp));
        r.k[1] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(kp + 32));
#pragma unroll
        for (int i = 0; i < 4; ++i)
            r.v[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(static_cast<const uint16_t*>(vbase) + vrow_elem[i] + 4 * c));
    } else {
        const float* kp = static_cast<const float*>(kbase) + krow_elem + 4 * g;
#pragma unroll
        for (int db = 0; db < 4; ++db) r.k[db] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(kp + db * 16));   // streamed once per step
#pragma unroll
        for (int i = 0; i < 4; ++i)
            r.v[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(static_cast<const float*>(vbase) + vrow_elem[i] + 4 * c));
    }
}
template <bool KV16> __device__ __forceinline__ void kv_unpack(const KvRegs<KV16>& r, f32x4 (&kc4)[4], f32x4 (&vc4)[4])
{
    if constexpr (KV16) {
        kc4[0] = bf16_lo_hi(r.k[0][0], r.k[0][1]); kc4[1] = bf16_lo_hi(r.k[0][2], r.k[0][3]);
        kc4[2] = bf16_lo_hi(r.k[1][0], r.k[1][1]); kc4[3] = bf16_lo_hi(r.k[1][2], r.k[1][3]);
#pragma unroll
        for (int i = 0; i < 4; ++i) vc4[i] = bf16_lo_hi(r.v[i][0], r.v[i][1]);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) { kc4[i] = r.k[i]; vc4[i] = r.v[i]; }
    }
}

// ---- float32 products on the fp16 matrix pipe, as gemm_split.hip does them (x = x1 + 2^-11 x2, x1 = fp16(x),
// x2 = fp16(2^11 (x - x1)); a.b = a1 b1 + 2^-11 (a1 b2 + a2 b1) + O(2^-22)): the decode-time attention kernels' dot
// products took a third of their time on the exact-f32 matrix instruction (1/16 of the fp16 rate) - matrix-pipe time the
// GEMMs of the other steps in flight could not use.  Operands must fit fp16's range: the engine checks K / V rows where
// they are written and q where it is assembled (range flag -> the call is repeated with the exact-f32 products).
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
constexpr float ATT_F16_SCALE = 2048.0f;
__device__ __forceinline__ void split_f16x2(const f32x4& x, f16x4_t& hi, f16x4_t& lo)
{
    hi = __builtin_convertvector(x, f16x4_t);
    lo = __builtin_convertvector((x - __builtin_convertvector(hi, f32x4)) * ATT_F16_SCALE, f16x4_t);
}
__device__ __forceinline__ f32x4 mfma16(const f16x4_t& a, const f16x4_t& b, const f32x4& c)
{
    return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bool fits_f16(const f32x4& x)
{
    const float m = fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fmaxf(fabsf(x[2]), fabsf(x[3])));
    return m < 65504.0f;                                   // false for NaN as well
}

// ---- `precision: bf16`: K / V rows are bf16 already, so they go to the bf16 matrix instruction as they are loaded (no
// widening, 1/16 of the exact-f32 instruction's passes); the float32 side of each product (q, P) goes in as two bf16 terms
// (x = x1 + x2 to 16 significant bits, both in float32's exponent range: no scaling and no range guard).
__device__ __forceinline__ void split_bf16x2(const f32x4& x, bf16x4_t& hi, bf16x4_t& lo)
{
    hi = __builtin_convertvector(x, bf16x4_t);
    lo = __builtin_convertvector(x - __builtin_convertvector(hi, f32x4), bf16x4_t);
}
__device__ __forceinline__ f32x4 mfma16(const bf16x4_t& a, const bf16x4_t& b, const f32x4& c)
{
    typedef short s16x4_t __attribute__((ext_vector_type(4)));
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4_t, a), __builtin_bit_cast(s16x4_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ bf16x4_t bf16x4_of(uint32_t w0, uint32_t w1)
{
    return __builtin_bit_cast(bf16x4_t, u32x2_t{w0, w1});
}
// K fragments in kv_d_of order; V tile j = V[key 4g + r][d = 4c + j], r = 0..3: the 16-bit halves of four rows
__device__ __forceinline__ void kv_frags_bf16(const KvRegs<true>& r, bf16x4_t (&kb)[4], bf16x4_t (&vb)[4])
{
    kb[0] = bf16x4_of(r.k[0][0], r.k[0][1]); kb[1] = bf16x4_of(r.k[0][2], r.k[0][3]);
    kb[2] = bf16x4_of(r.k[1][0], r.k[1][1]); kb[3] = bf16x4_of(r.k[1][2], r.k[1][3]);
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        vb[2 * w] = bf16x4_of((r.v[0][w] & 0xffffu) | (r.v[1][w] << 16), (r.v[2][w] & 0xffffu) | (r.v[3][w] << 16));
        vb[2 * w + 1] = bf16x4_of((r.v[0][w] >> 16) | (r.v[1][w] & 0xffff0000u), (r.v[2][w] >> 16) | (r.v[3][w] & 0xffff0000u));
    }
}

// ------------------------------------------------------------------------------------------------
// Decoder self-attention at position `pos` (model.py:588-600 with reserve_last; no mask).
// One workgroup per (utterance, head) serves all beams.  The beams' own k/v of position `pos` are
// appended to the cache (slot = beam index; layout [utterance][head][position][slot][64], so the rows one
// workgroup gathers sit a few KB apart); history position j < pos of beam b lives in slot anc[b][j].  Beams of one utterance share most of their history, so beam_step_kernel keeps, per
// utterance, the list of LIVE (position, slot) rows - rows referenced by at least one beam - and this
// kernel streams exactly those rows once, flash-style on the exact-f32 16x16x4 MFMA (same operand
// mapping as the cross-attention kernel below): S^T[key][beam] for all beams at once, entries whose
// key is not on that beam's ancestry are masked to -inf.  HBM traffic = live rows, not beams x pos.
// The new position's K keys (one per beam, valid for that beam only) are taken from LDS.
// ------------------------------------------------------------------------------------------------
// F16X3 (`f32_f16x3`, float32 caches): the dot products as three fp16 matrix products per multiply, exactly as the cross-attention
// kernel below does them; q, k and v of the new position are checked against fp16's range where they are assembled (every
// history row was checked the same way by the step that wrote it), range flag -> the engine repeats the call with F16X3 off.
template <int NB, bool KV16, bool F16X3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NB == 1 ? 3 : (F16X3 ? 1 : 2)))) void decoder_self_attention_kernel(SelfAttnArgs a)
{
    static_assert(!(KV16 && F16X3), "bf16 caches go to the bf16 matrix instruction");
    if (*a.done) return;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int BEAMS = NB * 16;
    const int h = blockIdx.x, u = blockIdx.y;
    const int K = a.k, hidden = a.heads * 64, width = a.width, ls = a.lds_stride;
    const int pos = *a.pos;
    const int n_live = min(a.live_cnt[u], K * pos);
    float* qs = reinterpret_cast<float*>(smem_raw);        // [BEAMS][64]
    float* knew = qs + BEAMS * 64;                         // [BEAMS][64]
    float* vnew = knew + BEAMS * 64;                       // [BEAMS][64]
    int32_t* ancs = reinterpret_cast<int32_t*>(vnew + BEAMS * 64);   // [K][ls]
    int32_t* live = ancs + K * ls;                         // [K * ls]  (position << 5 | slot)
    // the merge buffers reuse the ancs/live area once the key loop is over
    float (*wo)[NB][64][17] = reinterpret_cast<float (*)[NB][64][17]>(ancs);
    float (*wm)[NB][16] = reinterpret_cast<float (*)[NB][16]>(reinterpret_cast<float*>(ancs) + 4 * NB * 64 * 17);
    float (*wl)[NB][16] = wm + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c = lane & 15;

    // ---- q, k, v of the new position: reduce the split-K slabs, add the bias, scale q (utils.py:628, Q3)
    for (int i = tid; i < BEAMS * 16; i += 256) {
        const int kb = i >> 4, cc = i & 15;
        f32x4 q4 = {0.f, 0.f, 0.f, 0.f}, k4 = q4, v4 = q4;
        if (kb < K) {
            const size_t rowoff = (size_t)(u * K + kb) * 3 * hidden + h * 64 + cc * 4;
            for (int z = 0; z < a.nsplit; ++z) {
                const float* p = a.qkv + z * a.split_stride + rowoff;
                q4 += *reinterpret_cast<const f32x4*>(p);
                k4 += *reinterpret_cast<const f32x4*>(p + hidden);
                v4 += *reinterpret_cast<const f32x4*>(p + 2 * hidden);
            }
            const float* bp = a.bias + h * 64 + cc * 4;
            q4 += *reinterpret_cast<const f32x4*>(bp);
            k4 += *reinterpret_cast<const f32x4*>(bp + hidden);
            v4 += *reinterpret_cast<const f32x4*>(bp + 2 * hidden);
            q4 = q4 * 0.125f;
            if constexpr (F16X3) {
                if (!(fits_f16(q4) && fits_f16(k4) && fits_f16(v4)) && a.range_flag) atomicOr(a.range_flag, 1);
            }
            const size_t off = ((((size_t)u * a.heads + h) * width + pos) * K + kb) * 64 + cc * 4;   // head-major cache
            if constexpr (KV16) {
                // the row is used as the cache will return it from the next step on: rounded once, here
                const u32x2_t kb16 = f32x4_to_bf16(k4), vb16 = f32x4_to_bf16(v4);
                *reinterpret_cast<u32x2_t*>(static_cast<uint16_t*>(a.kcache) + off) = kb16;
                *reinterpret_cast<u32x2_t*>(static_cast<uint16_t*>(a.vcache) + off) = vb16;
                k4 = bf16_lo_hi(kb16[0], kb16[1]);
                v4 = bf16_lo_hi(vb16[0], vb16[1]);
            } else {
                *reinterpret_cast<f32x4*>(static_cast<float*>(a.kcache) + off) = k4;
                *reinterpret_cast<f32x4*>(static_cast<float*>(a.vcache) + off) = v4;
            }
        }
        reinterpret_cast<f32x4*>(qs)[kb * 16 + cc] = q4;
        reinterpret_cast<f32x4*>(knew)[kb * 16 + cc] = k4;
        reinterpret_cast<f32x4*>(vnew)[kb * 16 + cc] = v4;
    }
    {
        const int32_t* anc = ((pos & 1) ? a.anc1 : a.anc0) + (size_t)u * K * width;
        for (int i = tid; i < K * pos; i += 256) {
            const int kb = i / pos, j = i - kb * pos;
            ancs[kb * ls + j] = anc[(size_t)kb * width + j];
        }
        const int32_t* lsrc = a.live + (size_t)u * K * width;
        for (int i = tid; i < n_live; i += 256) live[i] = lsrc[i];
    }
    __syncthreads();

    f32x4 qf[NB][4];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int db = 0; db < 4; ++db) qf[nb][db] = *reinterpret_cast<const f32x4*>(qs + (nb * 16 + c) * 64 + kv_d_of<KV16>(db, g));
    bf16x4_t qbh[KV16 ? NB : 1][4], qbl[KV16 ? NB : 1][4];
    if constexpr (KV16) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int db = 0; db < 4; ++db) split_bf16x2(qf[nb][db], qbh[nb][db], qbl[nb][db]);
    }
    f16x4_t qh[F16X3 ? NB : 1][4], ql[F16X3 ? NB : 1][4];
    if constexpr (F16X3) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int db = 0; db < 4; ++db) split_f16x2(qf[nb][db], qh[nb][db], ql[nb][db]);
    }
    float m_run[NB], l_run[NB];
    f32x4 o[NB][4];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        m_run[nb] = -INFINITY; l_run[nb] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[nb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // one block of 16 keys: kc4 = K rows (lane: key c, d = db*16 + 4g ..), vc4 = V rows (lane: key 4g+r, d = 4c ..),
    // cj/cs = position (-1: the new position) and slot (-2: no such key) of key 4g+r
    // KV16: kop / vop are bf16 fragments (K in kv_d_of order; V tile j = V[key 4g + r][d = 4c + j], r = 0..3) for the bf16
    // matrix instruction, q and P go in as two bf16 terms; otherwise float32 rows for the exact-f32 instruction
    auto compute = [&](const auto (&kop)[4], const auto (&vop)[4], const int (&cj)[4], const int (&cs)[4]) {
        f16x4_t kh[4], kl[4], vh[4], vl[4];
        if constexpr (F16X3) {
#pragma unroll
            for (int db = 0; db < 4; ++db) split_f16x2(kop[db], kh[db], kl[db]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {                       // tile j: V[key 4g + r][d = 4c + j], r = 0..3 (the MFMA's k = key)
                const f32x4 tv = {vop[0][j], vop[1][j], vop[2][j], vop[3][j]};
                split_f16x2(tv, vh[j], vl[j]);
            }
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            if constexpr (F16X3) {
                f32x4 sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    sc = mfma16(kh[db], ql[nb][db], sc);
                    sc = mfma16(kl[db], qh[nb][db], sc);
                    s = mfma16(kh[db], qh[nb][db], s);
                }
                s = s + sc * (1.0f / ATT_F16_SCALE);
            } else if constexpr (KV16) {
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    s = mfma16(kop[db], qbl[nb][db], s);
                    s = mfma16(kop[db], qbh[nb][db], s);
                }
            } else {
#pragma unroll
                for (int db = 0; db < 4; ++db)
#pragma unroll
                    for (int j = 0; j < 4; ++j) s = __builtin_amdgcn_mfma_f32_16x16x4f32(kop[db][j], qf[nb][db][j], s, 0, 0, 0);
            }
            const int beam = nb * 16 + c, bl = min(beam, K - 1);
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int want = (cj[r] < 0) ? beam : ancs[bl * ls + max(cj[r], 0)];
                const bool ok = cs[r] >= 0 && beam < K && want == cs[r];
                s[r] = ok ? s[r] : -INFINITY;
                mx = fmaxf(mx, s[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run[nb], mx);
            const float m_use = (m_new == -INFINITY) ? 0.0f : m_new;     // beam without a key so far: all weights 0
            const float alpha = exp_neg(m_run[nb] - m_use);
            float lsum = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) { s[r] = exp_neg(s[r] - m_use); lsum += s[r]; }
            l_run[nb] = l_run[nb] * alpha + lsum;
            m_run[nb] = m_new;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[nb][j] = o[nb][j] * alpha;
            if constexpr (F16X3) {
                f16x4_t ph, pl;
                split_f16x2(s, ph, pl);                         // P^T[key 4g + r][beam c]: the B operand, k = key
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 tc = {0.f, 0.f, 0.f, 0.f};
                    tc = mfma16(vh[j], pl, tc);
                    tc = mfma16(vl[j], ph, tc);
                    o[nb][j] = mfma16(vh[j], ph, o[nb][j]);
                    o[nb][j] = o[nb][j] + tc * (1.0f / ATT_F16_SCALE);
                }
            } else if constexpr (KV16) {
                bf16x4_t pbh, pbl;
                split_bf16x2(s, pbh, pbl);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o[nb][j] = mfma16(vop[j], pbl, o[nb][j]);
                    o[nb][j] = mfma16(vop[j], pbh, o[nb][j]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[nb][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(vop[r][j], s[r], o[nb][j], 0, 0, 0);
            }
        }
    };

    // ---- live history rows, 16 per block, blocks dealt round-robin to the 4 waves; loads are branch-free
    // (clamped list index) and run one block ahead of the MFMAs
    const int nblk = (n_live + 15) >> 4;
    const size_t head_base = ((size_t)u * a.heads + h) * width;   // rows of this (utterance, head): [position][slot][64]
    // two register sets, used alternately (no copies): the next block's loads fly under this block's MFMAs
    KvRegs<KV16> kv[2];
    int ej[2][4], es[2][4];
    auto load_block = [&](auto setc, int blk) {
        constexpr int S = decltype(setc)::value;
        const int key0 = blk * 16;
        const int e = live[min(key0 + c, n_live - 1)];
        size_t vrow[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int idx = key0 + 4 * g + r;
            const int er = live[min(idx, n_live - 1)];
            vrow[r] = ((head_base + (er >> 5)) * K + (er & 31)) * 64;
            ej[S][r] = er >> 5;
            es[S][r] = idx < n_live ? (er & 31) : -2;
        }
        kv_load<KV16>(kv[S], a.kcache, ((head_base + (e >> 5)) * K + (e & 31)) * 64, a.vcache, vrow, g, c);
    };
    auto compute_set = [&](auto setc) {
        constexpr int S = decltype(setc)::value;
        if constexpr (KV16) {
            bf16x4_t kb[4], vb[4];
            kv_frags_bf16(kv[S], kb, vb);
            compute(kb, vb, ej[S], es[S]);
        } else {
            f32x4 kc4[4], vc4[4];
            kv_unpack<KV16>(kv[S], kc4, vc4);
            compute(kc4, vc4, ej[S], es[S]);
        }
    };
    using sc0 = std::integral_constant<int, 0>;
    using sc1 = std::integral_constant<int, 1>;
    if (wave < nblk) load_block(sc0{}, wave);
    for (int blk = wave; blk < nblk; blk += 8) {
        if (blk + 4 < nblk) load_block(sc1{}, blk + 4);
        compute_set(sc0{});
        if (blk + 4 >= nblk) break;
        if (blk + 8 < nblk) load_block(sc0{}, blk + 8);
        compute_set(sc1{});
    }
    // ---- the new position: key = beam, from LDS
#pragma unroll
    for (int sb = 0; sb < NB; ++sb) {
        if (wave == ((nblk + sb) & 3)) {
            f32x4 kc4[4], vc4[4];
            int cj[4], cs[4];
#pragma unroll
            for (int db = 0; db < 4; ++db) kc4[db] = *reinterpret_cast<const f32x4*>(knew + (sb * 16 + c) * 64 + kv_d_of<KV16>(db, g));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kbm = sb * 16 + 4 * g + r;
                vc4[r] = reinterpret_cast<const f32x4*>(vnew)[kbm * 16 + c];
                cj[r] = -1;
                cs[r] = kbm < K ? kbm : -2;
            }
            if constexpr (KV16) {                              // rows in LDS hold the rounded values: the conversion is exact
                bf16x4_t kb[4], vb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    kb[j] = __builtin_convertvector(kc4[j], bf16x4_t);
                    vb[j] = __builtin_convertvector(f32x4{vc4[0][j], vc4[1][j], vc4[2][j], vc4[3][j]}, bf16x4_t);
                }
                compute(kb, vb, cj, cs);
            } else {
                compute(kc4, vc4, cj, cs);
            }
        }
    }
    __syncthreads();                                       // every wave is done with ancs / live: reuse as merge buffers

    // o[nb][j][r'] = O[beam nb*16+c][d = 16g + 4r' + j] of this wave's keys (unnormalised)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        float l = l_run[nb];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        if (g == 0) { wm[wave][nb][c] = m_run[nb]; wl[wave][nb][c] = l; }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) wo[wave][nb][16 * g + 4 * r + j][c] = o[nb][j][r];
    }
    __syncthreads();
    for (int i = tid; i < NB * 16 * 16; i += 256) {
        const int beam = i >> 4, d4 = i & 15, nb = beam >> 4, bc = beam & 15;
        if (beam < K) {
            const float M = fmaxf(fmaxf(wm[0][nb][bc], wm[1][nb][bc]), fmaxf(wm[2][nb][bc], wm[3][nb][bc]));
            float sc[4], L = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { sc[w] = exp_neg(wm[w][nb][bc] - M); L += sc[w] * wl[w][nb][bc]; }
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float acc = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) acc += sc[w] * wo[w][nb][4 * d4 + e][bc];
                v[e] = acc / L;
            }
            *reinterpret_cast<f32x4*>(a.ctx + (size_t)(u * K + beam) * hidden + h * 64 + 4 * d4) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Decoder cross-attention (model.py:604-616): one workgroup per (utterance, head) serves all beams in
// ONE pass over that utterance's K and V (flash-style, exact-f32 16x16x4 MFMA), so the kernel is bound
// by the HBM stream of K/V and not by per-beam dot products.
//   S^T[key][beam] = K.Q^T : A = K block (16 keys x 4 d), B = Q^T (4 d x 16 beams)
//   O^T[d][beam]  += V^T.P^T: A = V^T (16 d x 4 keys),    B = P^T = the S^T accumulator itself
// The MFMA k index is a summation label, so a lane feeds 4 consecutive d (resp. one key row) from one
// 16-byte load.  The 4 waves take key blocks round-robin with an online softmax each and are merged
// through LDS at the end.  NB = beam groups of 16 (beam_size <= 16*NB).
// Mask: -1e9 on padded encoder positions exactly as the reference (common_attention.py:202); keys
// beyond t (block padding) get -inf.
// ------------------------------------------------------------------------------------------------
template <int NB, bool KV16, bool F16X3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NB == 1 && !F16X3 ? 3 : 2)))
void decoder_cross_attention_kernel(CrossAttnArgs a)
{
    if (*a.done) return;
    __shared__ float wo[4][NB][64][17];
    __shared__ float wm[4][NB][16], wl[4][NB][16];
    const int h = blockIdx.x, u = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int hidden = a.heads * 64, t = a.t_arr ? a.t_arr[u] : a.t, K = a.k;

    // K and V of this (utterance, head) are two contiguous [t][64] arrays, so a 16-key block is 4 KB of each and every load
    // of a block is (per-lane base + block offset) + an immediate: no per-load address arithmetic.  The last block may run
    // past row t - 1: the engine allocates ASR_XATT_SLACK_ROWS spare rows behind K, V and the padding flags, and those keys
    // get -inf below.
    const size_t key_row0 = a.t_arr ? (size_t)a.row_off[u] : (size_t)u * t;               // utterance's first key among all keys
    const size_t kv_base = key_row0 * hidden + (size_t)h * t * 64;                        // element index of key 0, d 0
    const int32_t* pptr = a.pad2 + key_row0 + 4 * g;                                      // key 4g (+ r)
    float m_run[NB], l_run[NB];
    f32x4 o[NB][4];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        m_run[nb] = -INFINITY; l_run[nb] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[nb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    f32x4 qf[NB][4], qb[4];                                     // filled below, after the first block's loads are issued
    f16x4_t qh[F16X3 ? NB : 1][4], ql[F16X3 ? NB : 1][4];       // F16X3: q as two fp16 terms
    bf16x4_t qbh[KV16 ? NB : 1][4], qbl[KV16 ? NB : 1][4];      // KV16: q as two bf16 terms
    const int nblk = (t + 15) >> 4;
    // two register sets, used alternately (no copies): the next block's loads fly under this block's MFMAs
    KvRegs<KV16> kv[2];
    int32_t padv[2][4];
    auto load_block = [&](auto setc, int blk) {
        constexpr int S = decltype(setc)::value;
        const size_t row0 = kv_base + (size_t)blk * 1024;                                  // key blk * 16
        const size_t vrow[4] = {row0 + (size_t)(4 * g) * 64, row0 + (size_t)(4 * g + 1) * 64, row0 + (size_t)(4 * g + 2) * 64,
                                row0 + (size_t)(4 * g + 3) * 64};
        kv_load<KV16>(kv[S], a.kc, row0 + (size_t)c * 64, a.v, vrow, g, c);                // streamed once: kept out of the caches
        const int32_t* pr = pptr + blk * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r) padv[S][r] = pr[r];
    };
    auto compute_block = [&](auto setc, int blk) {
        constexpr int S = decltype(setc)::value;
        f32x4 kc4[4], vc4[4];
        KvRegs<KV16> kr = kv[S];
        float bc[4];
        const int keyq = blk * 16 + 4 * g;
#pragma unroll
        for (int i = 0; i < 4; ++i) bc[i] = (keyq + i >= t) ? -INFINITY : (padv[S][i] ? -1e9f : 0.0f);
        if (blk == nblk - 1) {
            // spare rows past t hold whatever was there (possibly NaN): an -inf score alone would leave 0 x NaN in P.V
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (keyq + i >= t) kr.v[i] = {};
            if (blk * 16 + c >= t) {
#pragma unroll
                for (int db = 0; db < (KV16 ? 2 : 4); ++db) kr.k[db] = {};
            }
        }
        bf16x4_t kb[4], vb[4];
        if constexpr (KV16) {
            kv_frags_bf16(kr, kb, vb);
        } else {
            kv_unpack<KV16>(kr, kc4, vc4);
        }
        f16x4_t kh[4], kl[4], vh[4], vl[4];
        if constexpr (F16X3) {
#pragma unroll
            for (int db = 0; db < 4; ++db) split_f16x2(kc4[db], kh[db], kl[db]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {                       // tile j: V[key 4g + r][d = 4c + j], r = 0..3 (the MFMA's k = key)
                const f32x4 tv = {vc4[0][j], vc4[1][j], vc4[2][j], vc4[3][j]};
                split_f16x2(tv, vh[j], vl[j]);
            }
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            if constexpr (F16X3) {
                f32x4 sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    sc = mfma16(kh[db], ql[nb][db], sc);
                    sc = mfma16(kl[db], qh[nb][db], sc);
                    s = mfma16(kh[db], qh[nb][db], s);
                }
                s = s + sc * (1.0f / ATT_F16_SCALE);
            } else if constexpr (KV16) {
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    s = mfma16(kb[db], qbl[nb][db], s);         // small terms first
                    s = mfma16(kb[db], qbh[nb][db], s);
                }
            } else {
#pragma unroll
                for (int db = 0; db < 4; ++db)
#pragma unroll
                    for (int j = 0; j < 4; ++j) s = __builtin_amdgcn_mfma_f32_16x16x4f32(kc4[db][j], qf[nb][db][j], s, 0, 0, 0);
            }
            // s[r] = S[beam nb*16+c][key key0 + 4g + r]
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) { s[r] += bc[r]; mx = fmaxf(mx, s[r]); }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run[nb], mx);
            const float alpha = exp_neg(m_run[nb] - m_new);
            float ls = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) { s[r] = exp_neg(s[r] - m_new); ls += s[r]; }
            l_run[nb] = l_run[nb] * alpha + ls;
            m_run[nb] = m_new;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[nb][j] = o[nb][j] * alpha;
            if constexpr (F16X3) {
                f16x4_t ph, pl;
                split_f16x2(s, ph, pl);                         // P^T[key 4g + r][beam c]: the B operand, k = key
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 tc = {0.f, 0.f, 0.f, 0.f};
                    tc = mfma16(vh[j], pl, tc);
                    tc = mfma16(vl[j], ph, tc);
                    o[nb][j] = mfma16(vh[j], ph, o[nb][j]);
                    o[nb][j] = o[nb][j] + tc * (1.0f / ATT_F16_SCALE);
                }
            } else if constexpr (KV16) {
                bf16x4_t pbh, pbl;
                split_bf16x2(s, pbh, pbl);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o[nb][j] = mfma16(vb[j], pbl, o[nb][j]);
                    o[nb][j] = mfma16(vb[j], pbh, o[nb][j]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[nb][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(vc4[r][j], s[r], o[nb][j], 0, 0, 0);
            }
        }
    };
    using xc0 = std::integral_constant<int, 0>;
    using xc1 = std::integral_constant<int, 1>;
    if (wave < nblk) load_block(xc0{}, wave);                  // the first K/V block flies while q is assembled

    // q fragments: split-K slabs fetched two at a time (independent loads in flight), summed in slab order
#pragma unroll
    for (int db = 0; db < 4; ++db) qb[db] = *reinterpret_cast<const f32x4*>(a.qbias + h * 64 + kv_d_of<KV16>(db, g));
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int beam = min(nb * 16 + c, K - 1);
        const size_t off = (size_t)(u * K + beam) * hidden + h * 64;
        f32x4 acc4[4];
#pragma unroll
        for (int db = 0; db < 4; ++db) acc4[db] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int z0 = 0; z0 < a.nsplit; z0 += 2) {
            f32x4 t4[2][4];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float* src = a.qpart + (size_t)min(z0 + j, a.nsplit - 1) * a.split_stride + off;
#pragma unroll
                for (int db = 0; db < 4; ++db) t4[j][db] = *reinterpret_cast<const f32x4*>(src + kv_d_of<KV16>(db, g));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (z0 + j < a.nsplit) {
#pragma unroll
                    for (int db = 0; db < 4; ++db) acc4[db] += t4[j][db];
                }
        }
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            f32x4 q4 = acc4[db] + qb[db];
            q4 = q4 * 0.125f;                                  // utils.py:628, after the bias (Q3)
            if (nb * 16 + c >= K) q4 = f32x4{0.f, 0.f, 0.f, 0.f};
            qf[nb][db] = q4;
            if constexpr (F16X3) {
                split_f16x2(q4, qh[nb][db], ql[nb][db]);
                if (!fits_f16(q4) && a.range_flag) atomicOr(a.range_flag, 1);
            }
            if constexpr (KV16) split_bf16x2(q4, qbh[nb][db], qbl[nb][db]);
        }
    }

    for (int blk = wave; blk < nblk; blk += 8) {
        if (blk + 4 < nblk) load_block(xc1{}, blk + 4);
        compute_block(xc0{}, blk);
        if (blk + 4 >= nblk) break;
        if (blk + 8 < nblk) load_block(xc0{}, blk + 8);
        compute_block(xc1{}, blk + 4);
    }
    // o[nb][j][r'] = O[beam nb*16+c][d = 16g + 4r' + j] of this wave's keys (unnormalised)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        float l = l_run[nb];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        if (g == 0) { wm[wave][nb][c] = m_run[nb]; wl[wave][nb][c] = l; }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) wo[wave][nb][16 * g + 4 * r + j][c] = o[nb][j][r];
    }
    __syncthreads();
    for (int i = tid; i < NB * 16 * 16; i += 256) {
        const int beam = i >> 4, d4 = i & 15, nb = beam >> 4, bc = beam & 15;
        if (beam < K) {
            const float M = fmaxf(fmaxf(wm[0][nb][bc], wm[1][nb][bc]), fmaxf(wm[2][nb][bc], wm[3][nb][bc]));
            float sc[4], L = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { sc[w] = exp_neg(wm[w][nb][bc] - M); L += sc[w] * wl[w][nb][bc]; }
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float acc = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) acc += sc[w] * wo[w][nb][4 * d4 + e][bc];
                v[e] = acc / L;
            }
            *reinterpret_cast<f32x4*>(a.ctx + (size_t)(u * K + beam) * hidden + h * 64 + 4 * d4) = v;
        }
    }
}


// ------------------------------------------------------------------------------------------------
// The many-query attention on the 16-bit matrix instruction (v_mfma_f32_32x32x16_*: 8 passes for 16 k-values where the exact-f32
// instruction above takes 16 passes for 2), same grid, same operand roles (S^T = K.Q^T, O^T += V^T.P^T, a lane owns a query).
//   MODE 13 (`f32_f16x3`): every operand as x1 + 2^-11 x2 (fp16 pair, as gemm_split.hip and the decode-time cross-attention do):
//           three products per multiply, the 2^-11 group in accumulators of its own; operands must fit fp16 (range_flag -> the
//           engine repeats the call on the exact-f32 kernel);
//   MODE 1  (`bf16`): K and V rounded to bf16 (what the decode-time kernels' bf16 caches hold), q and P as two bf16 terms.
// K / V tiles are split ONCE per workgroup on their way into LDS: K planes [key][8 x 16 B] (chunk ^ (key & 7)), V planes
// TRANSPOSED [d][8 x 8 B] (four keys a chunk, chunk ^ ((d >> 1) & 7)) so that a lane's A operand of O^T += V^T.P^T - eight keys
// of one d - is two 8-byte reads.  The k index of the second product is a summation label only: key(s, half, j) =
// 16 s + 8 (j >> 2) + 4 half + (j & 3) is the order in which a lane's S^T accumulator already holds its 16 keys, so P goes from
// the accumulator into the B operand without leaving the lane.
// ------------------------------------------------------------------------------------------------
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE> __device__ __forceinline__ f32x16 mma32x16(const u32x4_t& a, const u32x4_t& b, const f32x16& c)
{
    if constexpr (MODE == 13) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// x -> (first term, second term) of 8 values; MODE 13: fp16 pair with the 2^11 scale, MODE 1: bf16 pair
template <int MODE> __device__ __forceinline__ void split8(const f32x8_t& x, u32x4_t& hi, u32x4_t& lo)
{
    if constexpr (MODE == 13) {
        const f16x8_t h = __builtin_convertvector(x, f16x8_t);
        hi = __builtin_bit_cast(u32x4_t, h);
        lo = __builtin_bit_cast(u32x4_t, __builtin_convertvector((x - __builtin_convertvector(h, f32x8_t)) * ATT_F16_SCALE, f16x8_t));
    } else {
        const bf16x8_t h = __builtin_convertvector(x, bf16x8_t);
        hi = __builtin_bit_cast(u32x4_t, h);
        lo = __builtin_bit_cast(u32x4_t, __builtin_convertvector(x - __builtin_convertvector(h, f32x8_t), bf16x8_t));
    }
}
__device__ __forceinline__ float absmax8(const f32x8_t& x)
{
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) m = fmaxf(m, fabsf(x[i]));
    return m;
}

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void encoder_attention16_kernel(MhaArgs a, int32_t* range_flag)
{
    constexpr bool F16 = MODE == 13;
    constexpr int NPL = F16 ? 2 : 1;               // planes of K and V in LDS (bf16: one term)
    __shared__ u32x4_t Ks[NPL][32 * 8];            // [plane][key][16-B chunk of 8 d ^ (key & 7)]
    __shared__ u32x2_t Vt[NPL][64 * 8];            // [plane][d][8-B chunk of 4 keys ^ ((d >> 1) & 7)]
    __shared__ float Bias[32];
    __shared__ float Os[4][32][65];

    const int qt = blockIdx.x, h = blockIdx.y, u = blockIdx.z;
    const int hidden = a.heads * 64;
    const int t = a.tq, tk = a.tk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int q0w = qt * 128 + wave * 32;
    const int q = q0w + l31;
    const float* qbase = a.q + (size_t)u * t * a.ldq;
    const float* kbase = a.k + (size_t)u * tk * a.ldk;
    const float* vbase = a.v + (size_t)u * tk * a.ldv;
    const int32_t* pad = a.pad;
    float* ctx = a.ctx;
    float amax = 0.f;                              // MODE 13: the largest magnitude this thread has split

    // q fragments: B operand of S^T = K.Q^T, lane (query l31, half) holds d = 16 s + 8 half .. + 8 of k-step s
    u32x4_t qh[4], ql[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        f32x8_t v = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (q < t) {
            const float* p = qbase + (size_t)q * a.ldq + h * 64 + 16 * s4 + 8 * half;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(p), v1 = *reinterpret_cast<const f32x4*>(p + 4);
            v = f32x8_t{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]} * 0.125f;   // q *= key_depth_per_head ** -0.5 (utils.py:628)
        }
        if constexpr (F16) amax = fmaxf(amax, absmax8(v));
        split8<MODE>(v, qh[s4], ql[s4]);
    }

    f32x16 o[2], o2[F16 ? 2 : 1];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; o2[0][r] = 0.f; if constexpr (F16) o2[1][r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    // staging roles: K - thread -> (key = tid / 8, 8 d at 8 (tid % 8)); V - thread -> (4 keys at 4 (tid / 32), 2 d at 2 (tid % 32))
    const int skey = tid >> 3, sc = tid & 7;
    const int vkg = tid >> 5, vdp = tid & 31;

    int nkt = (tk + 31) / 32;
    if (a.causal) nkt = min(nkt, (min(qt * 128 + 127, t - 1) >> 5) + 1);
    for (int kt = 0; kt < nkt; ++kt) {
        const int k0 = kt * 32;
        {
            f32x8_t kv = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (k0 + skey < tk) {
                const float* p = kbase + (size_t)(k0 + skey) * a.ldk + h * 64 + 8 * sc;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(p), v1 = *reinterpret_cast<const f32x4*>(p + 4);
                kv = f32x8_t{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            }
            f32x2 vv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                vv[r] = f32x2{0.f, 0.f};
                if (k0 + 4 * vkg + r < tk) vv[r] = *reinterpret_cast<const f32x2*>(vbase + (size_t)(k0 + 4 * vkg + r) * a.ldv + h * 64 + 2 * vdp);
            }
            const f32x8_t vt = {vv[0][0], vv[1][0], vv[2][0], vv[3][0], vv[0][1], vv[1][1], vv[2][1], vv[3][1]};   // d = 2 vdp: 4 keys, d + 1: 4 keys
            if constexpr (F16) amax = fmaxf(amax, fmaxf(absmax8(kv), absmax8(vt)));
            u32x4_t k_hi, k_lo, v_hi, v_lo;
            split8<MODE>(kv, k_hi, k_lo);
            split8<MODE>(vt, v_hi, v_lo);
            Ks[0][skey * 8 + (sc ^ (skey & 7))] = k_hi;
            if constexpr (F16) Ks[1][skey * 8 + (sc ^ (skey & 7))] = k_lo;
            // rows d = 2 vdp and 2 vdp + 1 share (d >> 1) & 7 = vdp & 7
            const int vch = vkg ^ (vdp & 7);
            Vt[0][(2 * vdp) * 8 + vch] = u32x2_t{v_hi[0], v_hi[1]};
            Vt[0][(2 * vdp + 1) * 8 + vch] = u32x2_t{v_hi[2], v_hi[3]};
            if constexpr (F16) {
                Vt[1][(2 * vdp) * 8 + vch] = u32x2_t{v_lo[0], v_lo[1]};
                Vt[1][(2 * vdp + 1) * 8 + vch] = u32x2_t{v_lo[2], v_lo[3]};
            }
        }
        if (tid < 32) {
            const int kk = k0 + tid;
            Bias[tid] = (kk >= tk) ? -INFINITY : ((pad && pad[(size_t)u * tk + kk]) ? -1e9f : 0.0f);
        }
        __syncthreads();

        f32x16 s, s2;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; s2[r] = 0.f; }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int ch = (2 * s4 + half) ^ (l31 & 7);
            const u32x4_t kh = Ks[0][l31 * 8 + ch];
            if constexpr (F16) {
                const u32x4_t kl = Ks[1][l31 * 8 + ch];
                s2 = mma32x16<MODE>(kh, ql[s4], s2);
                s2 = mma32x16<MODE>(kl, qh[s4], s2);
                s = mma32x16<MODE>(kh, qh[s4], s);
            } else {
                s = mma32x16<MODE>(kh, ql[s4], s);
                s = mma32x16<MODE>(kh, qh[s4], s);
            }
        }
        // s[r] = S[query l31][key k0 + (r&3) + 8*(r>>2) + 4*half]
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kl = (r & 3) + 8 * (r >> 2) + 4 * half;
            if constexpr (F16) s[r] += s2[r] * (1.0f / ATT_F16_SCALE);
            s[r] = s[r] + Bias[kl];
            if (a.causal && k0 + kl > q) s[r] += -1e9f;
            mx = fmaxf(mx, s[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = expf(m_run - m_new);
        float ls = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = expf(s[r] - m_new); ls += s[r]; }
        l_run = l_run * alpha + ls;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; o2[0][r] *= alpha; if constexpr (F16) o2[1][r] *= alpha; }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const f32x8_t p8 = {s[8 * ks], s[8 * ks + 1], s[8 * ks + 2], s[8 * ks + 3], s[8 * ks + 4], s[8 * ks + 5], s[8 * ks + 6], s[8 * ks + 7]};
            u32x4_t ph, pl;
            split8<MODE>(p8, ph, pl);
#pragma unroll
            for (int dh = 0; dh < 2; ++dh) {
                const int d = l31 + 32 * dh, sw = (d >> 1) & 7;
                const u32x2_t c0 = Vt[0][d * 8 + ((4 * ks + half) ^ sw)], c1 = Vt[0][d * 8 + ((4 * ks + 2 + half) ^ sw)];
                const u32x4_t vh = {c0[0], c0[1], c1[0], c1[1]};
                if constexpr (F16) {
                    const u32x2_t e0 = Vt[1][d * 8 + ((4 * ks + half) ^ sw)], e1 = Vt[1][d * 8 + ((4 * ks + 2 + half) ^ sw)];
                    const u32x4_t vl = {e0[0], e0[1], e1[0], e1[1]};
                    o2[dh] = mma32x16<MODE>(vh, pl, o2[dh]);
                    o2[dh] = mma32x16<MODE>(vl, ph, o2[dh]);
                    o[dh] = mma32x16<MODE>(vh, ph, o[dh]);
                } else {
                    o[dh] = mma32x16<MODE>(vh, pl, o[dh]);
                    o[dh] = mma32x16<MODE>(vh, ph, o[dh]);
                }
            }
        }
        __syncthreads();
    }

    if constexpr (F16) {
        // NaN and Inf fail the comparison too
        if (!(amax < 65504.0f) && range_flag) atomicOr(range_flag, 1);
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int d = (r & 3) + 8 * (r >> 2) + 4 * half;
        float v0 = o[0][r], v1 = o[1][r];
        if constexpr (F16) { v0 += o2[0][r] * (1.0f / ATT_F16_SCALE); v1 += o2[1][r] * (1.0f / ATT_F16_SCALE); }
        Os[wave][l31][d] = v0 / l_tot;
        Os[wave][l31][d + 32] = v1 / l_tot;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = i * 64 + lane, row = idx >> 4, c = idx & 15;
        if (q0w + row < t) {
            f32x4 v;
            v[0] = Os[wave][row][c * 4 + 0]; v[1] = Os[wave][row][c * 4 + 1];
            v[2] = Os[wave][row][c * 4 + 2]; v[3] = Os[wave][row][c * 4 + 3];
            *reinterpret_cast<f32x4*>(ctx + ((size_t)u * t + q0w + row) * hidden + h * 64 + c * 4) = v;
        }
    }
}

}  // namespace

hipError_t launch_mha(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const int32_t* pad, int causal,
                      int n, int tq, int tk, int heads, float* ctx, int mode, int32_t* range_flag, hipStream_t s)
{
    if (n <= 0 || tq <= 0) return hipSuccess;
    if (tk <= 0 || !q || !k || !v || !ctx || ((ldq | ldk | ldv) & 3) || (causal && tq != tk)) return hipErrorInvalidValue;
    MhaArgs a{q, k, v, ldq, ldk, ldv, pad, tq, tk, heads, causal, ctx};
    const dim3 grid((tq + 127) / 128, heads, n);
    if (mode == 13) hipLaunchKernelGGL(encoder_attention16_kernel<13>, grid, dim3(256), 0, s, a, range_flag);
    else if (mode == 1) hipLaunchKernelGGL(encoder_attention16_kernel<1>, grid, dim3(256), 0, s, a, nullptr);
    else hipLaunchKernelGGL(encoder_attention_kernel, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_encoder_attention(const float* qkv, const int32_t* pad, int n, int t, int heads, float* ctx, int mode,
                                    int32_t* range_flag, hipStream_t s)
{
    const int hidden = heads * 64;
    return launch_mha(qkv, 3 * hidden, qkv + hidden, 3 * hidden, qkv + 2 * hidden, 3 * hidden, pad, 0, n, t, t, heads, ctx, mode,
                      range_flag, s);
}

size_t self_attention_lds_bytes(int k, int lds_stride)
{
    const int nb = k <= 16 ? 1 : 2;
    const size_t lists = (size_t)2 * k * lds_stride * sizeof(int32_t);
    const size_t merge = (size_t)nb * (4 * 64 * 17 + 2 * 4 * 16) * sizeof(float);
    return (size_t)3 * nb * 16 * 64 * sizeof(float) + (lists > merge ? lists : merge);
}

template <int NB, bool KV16, bool F16X3>
hipError_t launch_self(const SelfAttnArgs& a, size_t lds, hipStream_t s)
{
    auto kern = decoder_self_attention_kernel<NB, KV16, F16X3>;
    static std::atomic<bool> attr_done[64];      // zero-initialised; several handles' host threads may launch first at once (the call is idempotent)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_done[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           150 * 1024);
        if (e != hipSuccess) return e;
        attr_done[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3(a.heads, a.b), dim3(256), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_decoder_self_attention(const SelfAttnArgs& a, hipStream_t s)
{
    if (a.b <= 0) return hipSuccess;
    const size_t lds = self_attention_lds_bytes(a.k, a.lds_stride);
    if (lds > 150 * 1024 || a.k > ASR_MAX_BEAM || a.lds_stride > a.width || a.lds_stride < 2 || !a.live || !a.live_cnt)
        return hipErrorInvalidValue;
    if (a.kv_bf16) return a.k <= 16 ? launch_self<1, true, false>(a, lds, s) : launch_self<2, true, false>(a, lds, s);
    if (a.f16x3) return a.k <= 16 ? launch_self<1, false, true>(a, lds, s) : launch_self<2, false, true>(a, lds, s);
    return a.k <= 16 ? launch_self<1, false, false>(a, lds, s) : launch_self<2, false, false>(a, lds, s);
}

hipError_t launch_decoder_cross_attention(const CrossAttnArgs& a, hipStream_t s)
{
    if (a.b <= 0) return hipSuccess;
    if (a.k > ASR_MAX_BEAM || a.k < 1 || a.t < 1) return hipErrorInvalidValue;
    if (a.kv_bf16) {
        if (a.k <= 16) hipLaunchKernelGGL((decoder_cross_attention_kernel<1, true, false>), dim3(a.heads, a.b), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((decoder_cross_attention_kernel<2, true, false>), dim3(a.heads, a.b), dim3(256), 0, s, a);
    } else if (a.f16x3) {
        if (a.k <= 16) hipLaunchKernelGGL((decoder_cross_attention_kernel<1, false, true>), dim3(a.heads, a.b), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((decoder_cross_attention_kernel<2, false, true>), dim3(a.heads, a.b), dim3(256), 0, s, a);
    } else {
        if (a.k <= 16) hipLaunchKernelGGL((decoder_cross_attention_kernel<1, false, false>), dim3(a.heads, a.b), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((decoder_cross_attention_kernel<2, false, false>), dim3(a.heads, a.b), dim3(256), 0, s, a);
    }
    return hipGetLastError();
}
