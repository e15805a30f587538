// f32 "NT" GEMM on the gfx950 BF16 matrix cores by operand splitting:  C[M,N] = A[M,K] . W[N,K]^T.
//
// The f32-input MFMA (gemm_f32.hip) runs at 1/16 of the bf16 MFMA rate.  Every float is the exact sum of
// three bf16 values, x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (8 + 8 + 8
// significand bits), and a product of two bf16 values is exact in f32.  So
//     x.y = hh + (hm + mh) + (hl + lh + mm) + O(2^-24 |x.y|)
// and six v_mfma_f32_32x32x16_bf16 (f32 accumulate) per 16-deep k-step replace eight
// v_mfma_f32_32x32x2_f32: 192 instead of 512 matrix-pipe cycles at the accuracy of an f32 FMA chain
// (tests/test_gpu_kernels.py compares both against float64).  NPROD selects how many of the products
// are kept: 6 = "f32 via bf16x6" (default product path), 3 = hh+hm+mh (~2^-16), 1 = plain bf16.
//
// Replaces utils.dense (transformer/utils.py:546-557) and the GLU pairing of utils.py:559-561.
// Weights are split once at load time into bf16 planes [NPL][N][K]; activations are split while they
// are staged into LDS.  LDS image per plane: [rows][4 chunks of 16 B] (BK = 32 bf16), chunk XOR-swizzled
// by (row >> 2) & 3 so that both the 16-byte stores and the fragment reads are conflict-free.
#include "asr_common.h"
#include "gemm_epilogue.h"
#include <string.h>
#include <type_traits>

namespace {

constexpr int SBK = 32;   // k-depth of one LDS stage

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// 8 floats -> NPL bf16 planes of 8 values each (packed 2 per dword, element 0 in the low half)
template <int NPL>
__device__ __forceinline__ void split8(const f32x4& x0, const f32x4& x1, u32x4 (&out)[NPL])
{
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
        f32x2 v = pr < 2 ? f32x2{x0[2 * pr], x0[2 * pr + 1]} : f32x2{x1[2 * pr - 4], x1[2 * pr - 3]};
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            const bf16x2 b = __builtin_convertvector(v, bf16x2);
            out[pl][pr] = __builtin_bit_cast(uint32_t, b);
            if (pl + 1 < NPL) v = v - __builtin_convertvector(b, f32x2);      // exact
        }
    }
}

template <int BM, int BN, int WM, int WN, int EPI, int NPROD>
__global__ __launch_bounds__(256) void gemm_split_nt_kernel(GemmArgs p, int nt_m, int nt_n, int kchunk)
{
    constexpr int NPL = NPROD == 6 ? 3 : NPROD == 3 ? 2 : 1;
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
    static_assert(BM % 64 == 0 && BN % 64 == 0, "staging covers 64 rows per pass");
    constexpr int A_LD = BM / 64, B_LD = BN / 64;

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    u32x4* As = reinterpret_cast<u32x4*>(smem_raw);    // [NPL][BM][4]
    u32x4* Bs = As + NPL * BM * 4;                     // [NPL][BN][4]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // XCD-aware tile order (same as gemm_f32.hip)
    const int total = nt_m * nt_n;
    const int bid = blockIdx.x;
    const int q8 = total >> 3, r8 = total & 7, xcd = bid & 7;
    const int tlin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    constexpr int GM = 8;
    const int group = tlin / (GM * nt_n);
    const int first_m = group * GM;
    const int gsz = min(nt_m - first_m, GM);
    const int bm = first_m + (tlin % gsz);
    const int bn = (tlin % (GM * nt_n)) / gsz;
    const int m0 = bm * BM, n0 = bn * BN;

    const int z = blockIdx.y;
    const int kbeg = z * kchunk;
    const int kend = min(p.K, kbeg + kchunk);
    const int nk = (kend - kbeg) / SBK;                // K % 32 == 0 (checked by the launcher)

    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int n = 0; n < NI; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.0f;

    // staging: thread -> row (tid >> 2) + 64 i, k-chunk (tid & 3) of 8; rows past M / N are clamped (their
    // products land in C rows / columns that are never stored), so every load is unconditional
    const int srow = tid >> 2, sch = tid & 3;
    const float* arow[A_LD];
    const uint16_t* brow[B_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) arow[i] = p.A + (size_t)min(m0 + srow + 64 * i, p.M - 1) * p.lda + sch * 8;
#pragma unroll
    for (int i = 0; i < B_LD; ++i) brow[i] = p.Wp + (size_t)min(n0 + srow + 64 * i, p.N - 1) * p.K + sch * 8;
    const size_t wplane = (size_t)p.N * p.K;

    f32x4 ra[A_LD][2];
    u32x4 rb[B_LD][NPL];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            ra[i][0] = *reinterpret_cast<const f32x4*>(arow[i] + k0);
            ra[i][1] = *reinterpret_cast<const f32x4*>(arow[i] + k0 + 4);
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) rb[i][pl] = *reinterpret_cast<const u32x4*>(brow[i] + pl * wplane + k0);
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int row = srow + 64 * i;
            u32x4 pieces[NPL];
            split8<NPL>(ra[i][0], ra[i][1], pieces);
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) As[(pl * BM + row) * 4 + (sch ^ ((row >> 2) & 3))] = pieces[pl];
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const int row = srow + 64 * i;
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) Bs[(pl * BN + row) * 4 + (sch ^ ((row >> 2) & 3))] = rb[i][pl];
        }
    };

    const int l31 = lane & 31, lhalf = lane >> 5;
    if (nk > 0) gload(kbeg);
    for (int kt = 0; kt < nk; ++kt) {
        lstore();
        __syncthreads();
        gload(kt + 1 < nk ? kbeg + (kt + 1) * SBK : kbeg);        // next tile (or a harmless re-read) flies under the MFMAs
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ch = 2 * ks + lhalf;
            bf16x8 af[NPL][MI], bf[NPL][NI];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int row = wm0 + i * 32 + l31;
                    af[pl][i] = __builtin_bit_cast(bf16x8, As[(pl * BM + row) * 4 + (ch ^ ((row >> 2) & 3))]);
                }
#pragma unroll
                for (int n = 0; n < NI; ++n) {
                    const int row = wn0 + n * 32 + l31;
                    bf[pl][n] = __builtin_bit_cast(bf16x8, Bs[(pl * BN + row) * 4 + (ch ^ ((row >> 2) & 3))]);
                }
            }
            // smallest products first; (a-plane, w-plane) pairs: 6 -> lh hl mm mh hm hh, 3 -> mh hm hh, 1 -> hh
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int n = 0; n < NI; ++n) {
                    if constexpr (NPROD == 6) {
                        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][i], bf[0][n], acc[i][n], 0, 0, 0);
                        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[2][n], acc[i][n], 0, 0, 0);
                        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[1][n], acc[i][n], 0, 0, 0);
                    }
                    if constexpr (NPROD >= 3) {
                        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[0][n], acc[i][n], 0, 0, 0);
                        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[1][n], acc[i][n], 0, 0, 0);
                    }
                    acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][n], acc[i][n], 0, 0, 0);
                }
        }
        __syncthreads();
    }

    gemm_store_tile<EPI, MI, NI>(p, acc, m0, n0, wm0, wn0, z, lane);
}

template <int BM, int BN, int WM, int WN, int EPI, int NPROD>
hipError_t launch_split_tile(const GemmArgs& a, hipStream_t s)
{
    constexpr int NPL = NPROD == 6 ? 3 : NPROD == 3 ? 2 : 1;
    const int nt_m = (a.M + BM - 1) / BM, nt_n = (a.N + BN - 1) / BN;
    const int split = a.split_k < 1 ? 1 : a.split_k;
    int kchunk = (a.K + split - 1) / split;
    kchunk = (kchunk + SBK - 1) / SBK * SBK;
    const size_t lds = (size_t)NPL * (BM + BN) * 4 * sizeof(u32x4);
    auto kern = gemm_split_nt_kernel<BM, BN, WM, WN, EPI, NPROD>;
    static bool attr_done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3(nt_m * nt_n, split, 1), dim3(256), lds, s, a, nt_m, nt_n, kchunk);
    return hipGetLastError();
}

template <int EPI, int NPROD>
hipError_t launch_split_epi(const GemmArgs& a, int tile, hipStream_t s)
{
    switch (tile) {
        case 1: return launch_split_tile<128, 128, 64, 64, EPI, NPROD>(a, s);
        case 3: return launch_split_tile<64, 128, 32, 64, EPI, NPROD>(a, s);
        case 4: return launch_split_tile<128, 64, 64, 32, EPI, NPROD>(a, s);
        default: return launch_split_tile<64, 64, 32, 32, EPI, NPROD>(a, s);
    }
}

template <int NPROD>
hipError_t launch_split_nprod(const GemmArgs& a, int tile, hipStream_t s)
{
    switch (a.epilogue) {
        case EPI_BIAS: return launch_split_epi<EPI_BIAS, NPROD>(a, tile, s);
        case EPI_GLU: return launch_split_epi<EPI_GLU, NPROD>(a, tile, s);
        case EPI_PARTIAL: return launch_split_epi<EPI_PARTIAL, NPROD>(a, tile, s);
        case EPI_RELU: return launch_split_epi<EPI_RELU, NPROD>(a, tile, s);
        case EPI_SIGMOID: return launch_split_epi<EPI_SIGMOID, NPROD>(a, tile, s);
        case EPI_TANH: return launch_split_epi<EPI_TANH, NPROD>(a, tile, s);
        case EPI_SWISH: return launch_split_epi<EPI_SWISH, NPROD>(a, tile, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace

int gemm_split_planes(int nprod) { return nprod == 6 ? 3 : nprod == 3 ? 2 : 1; }

// host-side split of a weight matrix into bf16 planes (round to nearest even, like v_cvt_pk_bf16_f32)
static inline uint16_t f32_to_bf16_rne(float x, float* back)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) {            // inf / nan: truncate (keep a nan a nan)
        uint32_t hb = u >> 16;
        if ((u & 0x007fffffu) && !(hb & 0x7fu)) hb |= 1u;
        const uint32_t v = hb << 16;
        memcpy(back, &v, 4);
        return (uint16_t)hb;
    }
    const uint32_t r = u + 0x7fffu + ((u >> 16) & 1u);
    const uint32_t v = r & 0xffff0000u;
    memcpy(back, &v, 4);
    return (uint16_t)(r >> 16);
}

void gemm_split_weights_host(const float* w, size_t count, int nplanes, uint16_t* planes)
{
    for (size_t i = 0; i < count; ++i) {
        float v = w[i];
        for (int pl = 0; pl < nplanes; ++pl) {
            float back;
            planes[(size_t)pl * count + i] = f32_to_bf16_rne(v, &back);
            v = v - back;
        }
    }
}

hipError_t launch_gemm_split(const GemmArgs& a, int nprod, int tile_hint, hipStream_t s)
{
    if (a.M <= 0 || a.N <= 0) return hipSuccess;
    if ((a.K % SBK) || (a.lda & 3) || !a.Wp) return hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(a.A) | reinterpret_cast<uintptr_t>(a.Wp)) & 15) return hipErrorInvalidValue;
    if (a.split_k > 1 && a.epilogue != EPI_PARTIAL) return hipErrorInvalidValue;
    if (a.epilogue == EPI_GLU && (a.N & 31)) return hipErrorInvalidValue;
    int tile = tile_hint;
    if (tile == 0) {
        const long tiles128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
        tile = tiles128 >= 512 ? 1 : 2;
    }
    switch (nprod) {
        case 6: return launch_split_nprod<6>(a, tile, s);
        case 3: return launch_split_nprod<3>(a, tile, s);
        case 1: return launch_split_nprod<1>(a, tile, s);
    }
    return hipErrorInvalidValue;
}
