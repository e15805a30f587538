// f32 "NT" GEMM on the gfx950 BF16 matrix cores by operand splitting:  C[M,N] = A[M,K] . W[N,K]^T.
//
// The f32-input MFMA (gemm_f32.hip) runs at 1/16 of the bf16 MFMA rate.  Every float is the exact sum of
// three bf16 values, x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (8 + 8 + 8
// significand bits), and a product of two bf16 values is exact in f32.  So
//     x.y = hh + (hm + mh) + (hl + lh + mm) + O(2^-24 |x.y|)
// and six v_mfma_f32_32x32x16_bf16 (f32 accumulate) per 16-deep k-step replace eight
// v_mfma_f32_32x32x2_f32: 192 instead of 512 matrix-pipe cycles at the accuracy of an f32 FMA chain
// (tests/test_gpu_kernels.py compares both against float64).  NPROD selects how many of the products
// are kept: 6 = "f32 via bf16x6", 3 = hh+hm+mh (~2^-16), 1 = plain bf16.
//
// NPROD = 13 is "f32 via fp16x3": a float is split as x = x1 + 2^-11 x2 with x1 = fp16(x), x2 = fp16(2^11 (x - x1))
// (11 + 11 significand bits, the scale keeps x2 in fp16's normal range), and
//     x.y = x1 y1 + 2^-11 (x1 y2 + x2 y1) + O(2^-22 |x.y|)
// with the two groups in two f32 accumulators that are combined in the epilogue: THREE v_mfma_f32_32x32x16_f16 per k-step.
// Measured against float64 its rms error is below the exact-f32 kernel's (the small terms do not round against the
// large ones).  What it costs is range: an operand of magnitude >= 65504 does not fit x1, and below 2^-14 x1 and x2 sink
// into fp16's subnormals (absolute error <= 2^-36 per element: nothing beside ordinary elements, but a relative loss for
// a row that is that small throughout).  Weights are checked row by row when they are packed; activations are checked as
// they are split (a row's largest magnitude over the workgroup's K slice must lie in [2^-10, 65504) or be 0), and a kernel
// that meets one outside
// sets GemmArgs.range_flag - the caller must then discard the result (the engine repeats the call on the bf16x6 kernels).
//
// Replaces utils.dense (transformer/utils.py:546-557) and the GLU pairing of utils.py:559-561.
//
// Shape of the kernel: a workgroup is 4 waves side by side along N; wave w owns columns [32 NI w, 32 NI (w+1)) of
// the BM x BN tile (BM = 32 MI up to 160, BN = 128 NI) for ALL BM rows.  So
//   * W fragments are private to a wave and never touch LDS.  Weights are split ONCE at load time
//     (gemm_split_weights_host) and stored in MFMA fragment order: [N/32 row panels][K/32 k-tiles][plane][k-step]
//     [lane] x 16 bytes, lane (l31, half) holding row l31, k = 16 kstep + 8 half .. + 8.  A wave's load of one
//     fragment is one contiguous 1 KB run and its registers ARE the MFMA operand;
//   * A (activations, f32 [M][K], small and L2-resident) is shared by the 4 waves: staged through LDS, split into
//     planes on the way in (LDS image per plane [rows][4 chunks of 16 B], chunk XOR (row >> 2) & 3: stores and
//     fragment reads conflict-free), double-buffered with ONE barrier per k-tile;
//   * two register sets keep the global loads of the next two k-tiles in flight, and the side work of a k-tile is
//     dealt out between its MFMAs by hand (see "the k-tile" below).
// M = 640 beam rows x N = 8192 (the GLU layer) is 4 x 64 tiles of 160 x 128: one workgroup per CU.
#include "asr_common.h"
#include "gemm_epilogue.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <thread>
#include <type_traits>
#include <vector>

namespace {

constexpr int SBK = 32;   // k-depth of one LDS stage

// Weight fragment loads.  ASR_GEMM_W_NT (development build): non-temporal, meant to keep the weight stream from displacing the
// activations in the XCD's L2.  Measured, same-box A/B of the whole bench (round 3): 185.3 -> 229.0 ms per step - the row tiles of
// a column tile share their weight panel THROUGH the L2, and non-temporal lines do not stay long enough for that.  Not used.
#ifdef ASR_GEMM_W_NT
#define W_LOAD(p) __builtin_nontemporal_load(p)
#else
#define W_LOAD(p) (*(p))
#endif

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr float F16_MAX = 65504.0f;
constexpr float F16_TINY = 0.0009765625f;    // 2^-10
constexpr float F16X3_SCALE = 2048.0f;       // 2^11

template <int NPROD> struct SplitMode {
    static constexpr bool F16 = NPROD == 13;
    static constexpr int NPL = F16 ? 2 : NPROD == 6 ? 3 : NPROD == 3 ? 2 : 1;    // operand planes
    static constexpr int NMMA = F16 ? 3 : NPROD;                                 // MFMAs per fragment pair per k-step
};

template <bool F16>
__device__ __forceinline__ f32x16 mma32(const u32x4& a, const u32x4& b, const f32x16& c)
{
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Which tiles an XCD works on.  Workgroups are dealt round-robin over the 8 XCDs (workgroup b runs on XCD b % 8: observed,
// used for speed only), every XCD has an L2 of its own, and whatever two XCDs both read crosses the fabric twice.  The
// decode-row GEMMs are bound by exactly that traffic when several engine handles share the GPU (tools/corun_bench.hip: a
// plain streaming read beside the GLU GEMM slows BOTH - memory-side bandwidth, not CU residency), so for few row tiles the
// XCDs partition the problem as a pz x pm x pn grid (pz pm pn = 8): XCD (xz, xm, xn) owns the K-slices z = xz (mod pz), the
// xm-th of pm contiguous blocks of row tiles and the xn-th of pn contiguous blocks of column tiles, M-tile fastest.
// Memory-side reads per launch are then pn A + pm W (A = the activations, W = the weight planes) instead of up to 8 A + 2 W.
// pz = 0: the legacy order (each XCD a contiguous run of tiles, groups of 8 row tiles: the encoder's 334 row tiles).
struct TileMap { int pz, pm, pn; };

template <int MI, int NI, int NW, int EPI, int NPROD>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW == 4 ? 3 : 2))) void gemm_split_nt_kernel(GemmArgs p, int nt_m, int nt_n, int kchunk, TileMap map)
{
    static_assert(NW == 4 || NW == 8, "4 or 8 waves side by side");
    constexpr int FPT = 2048 / (NW * 64);              // floats of a 64 x 32 A pass per thread: 8 (4 waves) or 4 (8 waves)
    constexpr int LPT = FPT / 4, PAIRS = FPT / 2;      // 16-byte loads / float pairs per thread per pass
    constexpr bool F16 = SplitMode<NPROD>::F16;
    constexpr int NPL = SplitMode<NPROD>::NPL;
    constexpr int WN = NI * 32, BM = MI * 32, BN = NW * WN;
    constexpr int A_LD = (BM + 63) / 64;               // staging passes of 64 rows
    constexpr int AR = A_LD * 64;                      // LDS rows (tile padded to whole passes: every staging load and
                                                       // store is unconditional, so the compiler's wait counts are exact)
    constexpr int STAGE = NPL * AR * 4;                // u32x4 per LDS stage

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    u32x4* lds = reinterpret_cast<u32x4*>(smem_raw);   // [2][NPL][AR][4]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    int bm, bn, z;
    if (map.pz == 0) {
        // XCD-aware tile order (same as gemm_f32.hip): each XCD gets a contiguous run of tiles, M-tile fastest
        const int total = nt_m * nt_n;
        const int bid = blockIdx.x;
        const int q8 = total >> 3, r8 = total & 7, xcd = bid & 7;
        const int tlin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
        constexpr int GM = 8;
        const int group = tlin / (GM * nt_n);
        const int first_m = group * GM;
        const int gsz = min(nt_m - first_m, GM);
        bm = first_m + (tlin % gsz);
        bn = (tlin % (GM * nt_n)) / gsz;
        z = blockIdx.y;
    } else {
        const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;          // li-th workgroup dealt to this XCD
        const int xz = xcd % map.pz, xmn = xcd / map.pz, xm = xmn % map.pm, xn = xmn / map.pm;
        const int m_lo = xm * nt_m / map.pm, nm = (xm + 1) * nt_m / map.pm - m_lo;
        const int n_lo = xn * nt_n / map.pn, nn = (xn + 1) * nt_n / map.pn - n_lo;
        const int nz = max(p.split_k, 1) / map.pz;                      // K-slices per XCD (this launch is 1-D: every XCD pads to the largest share)
        if (li >= nm * nn * nz) return;                                 // (the grid is padded to the largest XCD's share)
        bm = m_lo + li % nm;
        const int rest = li / nm;
        bn = n_lo + rest % nn;
        z = xz + map.pz * (rest / nn);
    }
    const int m0 = bm * BM, n0 = bn * BN;

    const int kbeg = z * kchunk;
    const int kend = min(p.K, kbeg + kchunk);
    const int nk = (kend - kbeg) / SBK;                // K % 32 == 0 (checked by the launcher)
    const int kt0 = kbeg / SBK;                        // first k-tile of this slice
    const int nkt = p.K / SBK;                         // k-tiles per panel

    const int wn0 = wave * WN;

    f32x16 acc[MI][NI];
    f32x16 acc2[F16 ? MI : 1][F16 ? NI : 1];           // fp16x3: the 2^-11 group (x1 y2 + x2 y1)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int n = 0; n < NI; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][n][r] = 0.0f;
                if constexpr (F16) acc2[i][n][r] = 0.0f;
            }
    float amax = 0.0f;                                  // fp16x3: largest |activation| this thread has split

    // A staging: thread -> row tid / (32 / FPT) + 64 i, FPT floats at k = FPT (tid % (32 / FPT)).  Rows past M are clamped
    // (their products land in C rows that are never stored).
    const int srow = tid / (32 / FPT), sch = tid % (32 / FPT);
    const float* arow[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) arow[i] = p.A + (size_t)min(m0 + srow + 64 * i, p.M - 1) * p.lda + sch * FPT;
    // W fragments: panel (n0 + wn0) / 32 + n of this wave (clamped past the last panel: columns never stored)
    const int npanels = (p.N + 31) >> 5;
    const u32x4* wfrag[NI];
#pragma unroll
    for (int n = 0; n < NI; ++n)
        wfrag[n] = reinterpret_cast<const u32x4*>(p.Wp) + (size_t)min(((n0 + wn0) >> 5) + n, npanels - 1) * nkt * (NPL * 128) + lane;

    // register sets (set = k-tile parity): A as loaded, W as the MFMA consumes it
    f32x4 ra[2][A_LD][LPT];
    u32x4 rw[2][2][NPL][NI];                           // [set][k-step][plane][n]
    auto ktile_k0 = [&](int t) { return kbeg + (t < nk ? t : 0) * SBK; };                       // past the slice: a harmless re-read
    auto ktile_woff = [&](int t) { return (size_t)(kt0 + (t < nk ? t : 0)) * (NPL * 128); };
    auto gload_a = [&](auto setc, int t) {
        constexpr int S = decltype(setc)::value;
        const int k0 = ktile_k0(t);
#pragma unroll
        for (int i = 0; i < A_LD; ++i)
#pragma unroll
            for (int hh = 0; hh < LPT; ++hh) ra[S][i][hh] = *reinterpret_cast<const f32x4*>(arow[i] + k0 + 4 * hh);
    };
    auto gload_w = [&](auto setc, int t, int ks) {
        constexpr int S = decltype(setc)::value;
        const size_t off = ktile_woff(t) + ks * 64;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int n = 0; n < NI; ++n) {
                const u32x4 v = W_LOAD(&wfrag[n][off + pl * 128]);
                if (ks == 0) rw[S][0][pl][n] = v; else rw[S][1][pl][n] = v;
            }
    };
    uint32_t pieces[A_LD][NPL][PAIRS];
    f32x2 rest[A_LD];                                  // what is left of the pair being split
    // A staging, in small steps (so that they can be dealt out between MFMAs): per pass of 64 rows, for each of the
    // thread's pairs of floats: (pair, 0) = plane 0 and the remainder, (pair, 1) = the other planes; then one store group
    constexpr int APASS = PAIRS * 2 + 1;
    auto astage = [&](auto setc, int buf, int op) {
        constexpr int S = decltype(setc)::value;
        u32x4* As = lds + buf * STAGE;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
#pragma unroll
            for (int pr = 0; pr < PAIRS; ++pr) {
                if (op == i * APASS + 2 * pr) {
                    const f32x4& x = ra[S][i][pr >> 1];
                    f32x2 v = {x[2 * (pr & 1)], x[2 * (pr & 1) + 1]};
                    if constexpr (F16) {
#ifdef ASR_GEMM_NO_SPLIT       // measurement only (WRONG low plane, no range check): what the split's VALU work costs.  Same-box A/B of
                               // the whole bench (round 3): 186.8 -> 183.0 ms per step with a non-zero low plane - what activations
                               // split once by their producers could buy at most; with an all-zero low plane 183.7 -> 175.5 ms: the
                               // chip is power-bound, a matrix instruction on zeros is cheaper than one on data
                        const f16x2 b = __builtin_convertvector(v, f16x2);
                        pieces[i][0][pr] = __builtin_bit_cast(uint32_t, b);
                        pieces[i][1][pr] = __builtin_bit_cast(uint32_t, b) ^ 0x00550055u;
#else
                        amax = fmaxf(amax, fmaxf(fabsf(v[0]), fabsf(v[1])));
                        const f16x2 b = __builtin_convertvector(v, f16x2);
                        pieces[i][0][pr] = __builtin_bit_cast(uint32_t, b);
                        rest[i] = (v - __builtin_convertvector(b, f32x2)) * F16X3_SCALE;          // exact
#endif
                    } else {
                        const bf16x2 b = __builtin_convertvector(v, bf16x2);
                        pieces[i][0][pr] = __builtin_bit_cast(uint32_t, b);
                        if (NPL > 1) rest[i] = v - __builtin_convertvector(b, f32x2);              // exact
                    }
                }
                if (op == i * APASS + 2 * pr + 1) {
                    f32x2 v = rest[i];
                    if constexpr (F16) {
#ifndef ASR_GEMM_NO_SPLIT
                        pieces[i][1][pr] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
#endif
                    } else {
#pragma unroll
                        for (int pl = 1; pl < NPL; ++pl) {
                            const bf16x2 b = __builtin_convertvector(v, bf16x2);
                            pieces[i][pl][pr] = __builtin_bit_cast(uint32_t, b);
                            if (pl + 1 < NPL) v = v - __builtin_convertvector(b, f32x2);           // exact
                        }
                    }
                }
            }
            if (op == i * APASS + 2 * PAIRS) {
                const int row = srow + 64 * i;
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    if constexpr (PAIRS == 4) {
                        As[(pl * AR + row) * 4 + (sch ^ ((row >> 2) & 3))] =
                            u32x4{pieces[i][pl][0], pieces[i][pl][1], pieces[i][pl][2], pieces[i][pl][3]};
                    } else {                           // half a 16-byte chunk per thread
                        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                        u32x2* dst = reinterpret_cast<u32x2*>(&As[(pl * AR + row) * 4 + ((sch >> 1) ^ ((row >> 2) & 3))]) + (sch & 1);
                        *dst = u32x2{pieces[i][pl][0], pieces[i][pl][1]};
                    }
                }
            }
        }
    };

    const int l31 = lane & 31, lhalf = lane >> 5;
    u32x4 af[2][NPL][MI];                              // A fragments of the current / next k-step
    auto fload_one = [&](auto slotc, int buf, int ks, int op) {          // op = i * NPL + pl
        constexpr int S = decltype(slotc)::value;
        const u32x4* As = lds + buf * STAGE;
        const int ch = 2 * ks + lhalf;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
                if (op == i * NPL + pl) {
                    const int row = i * 32 + l31;
                    af[S][pl][i] = As[(pl * AR + row) * 4 + (ch ^ ((row >> 2) & 3))];
                }
    };

    // ---- the k-tile, scheduled by hand ------------------------------------------------------------------
    // Left to itself the compiler sinks every fragment read to just before the MFMA that uses it (it minimises live
    // ranges), which exposes one LDS latency per MFMA group: 3800 instead of 1920 cycles per k-tile at 160 x 128.
    // So the side work of a k-tile is dealt out evenly between its MFMAs and pinned with sched_barrier:
    //   step 0 (MFMAs of k-step 0): A fragment reads of k-step 1; A of tile kt+1 from registers into the other LDS
    //           buffer, split a pair of floats at a time; W loads (tile kt+1, k-step 1); barrier;
    //   step 1 (MFMAs of k-step 1): first A fragment reads of tile kt+1; global A loads of tile kt+3 (into the set
    //           just staged); W loads (tile kt+2, k-step 0) into the registers step 0 has just finished with.
    constexpr int NMMA = SplitMode<NPROD>::NMMA;
    constexpr int NM = MI * NI * NMMA;                   // MFMAs per 16-deep k-step
    constexpr int NF = NPL * MI;                         // A fragment reads per k-step
    constexpr int NA = A_LD * APASS;                     // A staging ops per k-tile
    constexpr int NWL = NPL * NI;                        // W loads per k-step
    constexpr int NAL = A_LD * LPT;                      // A loads per k-tile
    int g_k0 = 0;
    size_t g_woff = 0;
    auto side_gload_a = [&](auto setc, int op) {
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int i = 0; i < A_LD; ++i)
#pragma unroll
            for (int hh = 0; hh < LPT; ++hh)
                if (op == i * LPT + hh) ra[S][i][hh] = *reinterpret_cast<const f32x4*>(arow[i] + g_k0 + 4 * hh);
    };
    auto side_gload_w = [&](auto setc, auto ksc, int op) {             // op = pl * NI + n
        constexpr int S = decltype(setc)::value;
        constexpr int KS = decltype(ksc)::value;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int n = 0; n < NI; ++n)
                if (op == pl * NI + n) rw[S][KS][pl][n] = W_LOAD(&wfrag[n][g_woff + KS * 64 + pl * 128]);
    };
    // the MFMAs of k-step KS of the tile in W set WS, A fragments from slot FS; side ops [n NOPS / NM, (n + 1) NOPS / NM)
    // follow the n-th MFMA
    auto chunk = [&](auto fslotc, auto wsetc, auto ksc, auto nopsc, auto&& side) {
        constexpr int FS = decltype(fslotc)::value;
        constexpr int WS = decltype(wsetc)::value;
        constexpr int KS = decltype(ksc)::value;
        constexpr int NOPS = decltype(nopsc)::value;
        auto one = [&](int idx, const u32x4& a, const u32x4& b, f32x16& c) {
            c = mma32<F16>(a, b, c);
            const int lo = idx * NOPS / NM, hi = (idx + 1) * NOPS / NM;
#pragma unroll
            for (int e = 0; e < (NOPS + NM - 1) / NM; ++e)
                if (lo + e < hi) side(lo + e);
            __builtin_amdgcn_sched_barrier(0);
        };
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int n = 0; n < NI; ++n) {
                const int base = (i * NI + n) * NMMA;
                int q = 0;
                if constexpr (F16) {
                    one(base + q++, af[FS][0][i], rw[WS][KS][1][n], acc2[i][n]);        // x1 y2
                    one(base + q++, af[FS][1][i], rw[WS][KS][0][n], acc2[i][n]);        // x2 y1
                    one(base + q++, af[FS][0][i], rw[WS][KS][0][n], acc[i][n]);         // x1 y1
                } else {
                    // smallest products first; (a-plane, w-plane) pairs: 6 -> lh hl mm mh hm hh, 3 -> mh hm hh, 1 -> hh
                    if constexpr (NPROD == 6) {
                        one(base + q++, af[FS][2][i], rw[WS][KS][0][n], acc[i][n]);
                        one(base + q++, af[FS][0][i], rw[WS][KS][2][n], acc[i][n]);
                        one(base + q++, af[FS][1][i], rw[WS][KS][1][n], acc[i][n]);
                    }
                    if constexpr (NPROD >= 3) {
                        one(base + q++, af[FS][1][i], rw[WS][KS][0][n], acc[i][n]);
                        one(base + q++, af[FS][0][i], rw[WS][KS][1][n], acc[i][n]);
                    }
                    one(base + q++, af[FS][0][i], rw[WS][KS][0][n], acc[i][n]);
                }
            }
    };
    using c0 = std::integral_constant<int, 0>;
    using c1 = std::integral_constant<int, 1>;
    // one k-tile kt whose parity is P: W set P holds tile kt, W set 1-P and A set 1-P hold tile kt+1
    auto ktile = [&](auto parc, int kt) {
        constexpr int P = decltype(parc)::value;
        using cP = std::integral_constant<int, P>;
        using cQ = std::integral_constant<int, 1 - P>;
        const int buf = P;
        g_woff = ktile_woff(kt + 1);
        chunk(c0{}, cP{}, c0{}, std::integral_constant<int, NF + NA + NWL>{}, [&](int op) {
            if (op < NF) fload_one(c1{}, buf, 1, op);
            else if (op < NF + NA) astage(cQ{}, buf ^ 1, op - NF);
            else side_gload_w(cQ{}, c1{}, op - NF - NA);
        });
        __syncthreads();
        g_k0 = ktile_k0(kt + 3);
        g_woff = ktile_woff(kt + 2);
        chunk(c1{}, cP{}, c1{}, std::integral_constant<int, NF + NAL + NWL>{}, [&](int op) {
            if (op < NF) fload_one(c0{}, buf ^ 1, 0, op);
            else if (op < NF + NAL) side_gload_a(cQ{}, op - NF);
            else side_gload_w(cP{}, c0{}, op - NF - NAL);
        });
    };
    if (nk > 0) {
        gload_a(c0{}, 0);
        gload_w(c0{}, 0, 0);
        gload_w(c0{}, 0, 1);
        gload_a(c1{}, 1);
        gload_w(c1{}, 1, 0);
#pragma unroll
        for (int op = 0; op < NA; ++op) astage(c0{}, 0, op);
        gload_a(c0{}, 2);
        __syncthreads();
#pragma unroll
        for (int op = 0; op < NF; ++op) fload_one(c0{}, 0, 0, op);
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            ktile(c0{}, kt);
            ktile(c1{}, kt + 1);
        }
        if (kt < nk) ktile(c0{}, kt);                     // odd tile count
    }

    if constexpr (F16) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int n = 0; n < NI; ++n) acc[i][n] += acc2[i][n] * (1.0f / F16X3_SCALE);
        // The 32 / FPT neighbouring lanes that stage one A row hold, together, that row's largest magnitude over the
        // workgroup's whole K slice (>= 128 values).  Outside fp16's range, or so small throughout (below 2^-10, not 0) that
        // x1 / x2 fall into fp16's subnormals and lose relative accuracy: the caller repeats the call on the bf16x6 kernels.
        // (Judging single threads' 16-value pieces instead would trip on sparse rows, e.g. ReLU outputs.)
#pragma unroll
        for (int o = 1; o < 32 / FPT; o <<= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
        if ((!(amax < F16_MAX) || (amax > 0.0f && amax < F16_TINY)) && p.range_flag) atomicOr(p.range_flag, 1);
    }
    gemm_store_tile<EPI, MI, NI>(p, acc, m0, n0, 0, wn0, z, lane);
}

template <int MI, int NI, int NW, int EPI, int NPROD>
hipError_t launch_split_tile(const GemmArgs& a, hipStream_t s)
{
    constexpr int NPL = SplitMode<NPROD>::NPL;
    constexpr int BM = MI * 32, BN = NW * NI * 32;
    constexpr int AR = (BM + 63) / 64 * 64;
    const int nt_m = (a.M + BM - 1) / BM, nt_n = (a.N + BN - 1) / BN;
    const int split = a.split_k < 1 ? 1 : a.split_k;
    int kchunk = (a.K + split - 1) / split;
    kchunk = (kchunk + SBK - 1) / SBK * SBK;
    const size_t lds = (size_t)2 * NPL * AR * 4 * sizeof(u32x4);
    auto kern = gemm_split_nt_kernel<MI, NI, NW, EPI, NPROD>;
    static std::atomic<bool> attr_done[64];      // zero-initialised; several handles' host threads may launch first at once (the call is idempotent)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_done[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done[dev].store(true, std::memory_order_release);
    }
    // Few row tiles (the decode rows): partition the tiles over the XCDs so that as little as possible is read by more
    // than one of them.  Memory-side reads = pn A + pm W; the partition must leave no XCD more than a third above its
    // even share of workgroups.
    TileMap map{0, 0, 0};
    int per_xcd = 0;
    static const int forced = getenv("ASR_AMD_XCD_MAP") ? atoi(getenv("ASR_AMD_XCD_MAP")) : -1;     // development: 0 = legacy order everywhere
    if (nt_m <= 32 && forced != 0) {
        const double A = 4.0 * a.M * a.K, W = 2.0 * NPL * (double)a.N * a.K;
        const int total = nt_m * nt_n * split;
        double best = 0.0;
        for (int pz = 1; pz <= 8; pz *= 2) {
            if (split % pz) continue;
            for (int pm = 1; pm * pz <= 8; pm *= 2) {
                const int pn = 8 / (pm * pz);
                if (pm > nt_m || pn > nt_n) continue;
                const int share = ((nt_m + pm - 1) / pm) * ((nt_n + pn - 1) / pn) * (split / pz);
                if (share * 8 > total + total / 3 + 8) continue;
                // (What the model leaves out: an XCD that runs its share in several rounds of 32 workgroups re-fetches its slice
                // of A once per round or so - the weight panels streaming through the 4 MiB L2 evict it.  Measured, 1280 x 8192 x
                // 1024 (80 workgroups per XCD): 149 MB fetched with either of the two best partitions, 75 / 88 MB by this formula.)
                const double traffic = pn * A + pm * W;
                if (!per_xcd || traffic < best) { best = traffic; map = TileMap{pz, pm, pn}; per_xcd = share; }
            }
        }
    }
    if (per_xcd) hipLaunchKernelGGL(kern, dim3(8 * per_xcd, 1, 1), dim3(NW * 64), lds, s, a, nt_m, nt_n, kchunk, map);
    else hipLaunchKernelGGL(kern, dim3(nt_m * nt_n, split, 1), dim3(NW * 64), lds, s, a, nt_m, nt_n, kchunk, map);
    return hipGetLastError();
}

template <int EPI, int NPROD>
hipError_t launch_split_epi(const GemmArgs& a, int tile, hipStream_t s)
{
    switch (tile) {
#ifdef ASR_GEMM_ALL_TILES                                                    // development (tools/gemm_split_bench.hip): the tiles that lost
        case 1: return launch_split_tile<4, 1, 4, EPI, NPROD>(a, s);     // 128 x 128
        case 5: return launch_split_tile<5, 1, 4, EPI, NPROD>(a, s);     // 160 x 128
        case 7: return launch_split_tile<2, 2, 4, EPI, NPROD>(a, s);     // 64 x 256, 4 waves of 64 columns
#endif
#ifdef ASR_GEMM_ALL_TILES
        case 3: return launch_split_tile<1, 1, 4, EPI, NPROD>(a, s);     // 32 x 128
        case 4: return launch_split_tile<1, 1, 8, EPI, NPROD>(a, s);     // 32 x 256, 8 waves of 32 columns
        case 11: return launch_split_tile<5, 1, 8, EPI, NPROD>(a, s);    // 160 x 256
        case 12: return launch_split_tile<4, 2, 4, EPI, NPROD>(a, s);    // 128 x 256, 4 waves of 64 columns
#endif
        case 6: return launch_split_tile<3, 1, 8, EPI, NPROD>(a, s);     // 96 x 256, 8 waves of 32 columns
        case 10: return launch_split_tile<4, 1, 8, EPI, NPROD>(a, s);    // 128 x 256, 8 waves of 32 columns
        case 9: return launch_split_tile<2, 1, 8, EPI, NPROD>(a, s);     // 64 x 256, 8 waves of 32 columns
        default: return launch_split_tile<2, 1, 4, EPI, NPROD>(a, s);    // 64 x 128
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

int gemm_split_planes(int nprod) { return nprod == 6 ? 3 : nprod == 3 || nprod == 13 ? 2 : 1; }

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

size_t gemm_split_weight_elems(int n, int k, int nplanes) { return (size_t)((n + 31) / 32) * 32 * (size_t)k * nplanes; }

// [n][k] f32 -> [n/32 panels][k/32 k-tiles][plane][k-step (2)][lane (64)] x 8 values of 16 bits: lane (l31, half) of
// k-step s holds row 32 panel + l31, k = 32 kt + 16 s + 8 half .. + 8 (the B operand of v_mfma_f32_32x32x16_*); rows past n
// are zero.  nprod 6 / 3 / 1: bf16 planes h, m, l; nprod 13: fp16 planes y1 = fp16(y), y2 = fp16(2^11 (y - y1)).
// Returns false (nprod 13 only) when the matrix does not suit fp16x3: a weight with |w| >= 65504 or not finite, or a row
// whose largest magnitude is below 2^-10 (and not 0).
bool gemm_split_weights_host(const float* w, int n, int k, int nprod, uint16_t* out)
{
    const int nplanes = gemm_split_planes(nprod);
    const int nkt = k / 32;
    std::atomic<bool> ok{true};
    const int npanels = (n + 31) / 32;
    if (n & 31) memset(out + (size_t)(npanels - 1) * nkt * nplanes * 1024, 0, (size_t)nkt * nplanes * 1024 * sizeof(uint16_t));
    auto pack_rows = [&](int r0, int r1) {
        for (int r = r0; r < r1; ++r) {
            const int panel = r >> 5, l31 = r & 31;
            if (nprod == 13) {
                float rmax = 0.f;
                for (int c = 0; c < k; ++c) rmax = fmaxf(rmax, fabsf(w[(size_t)r * k + c]));
                if (rmax > 0.f && rmax < 0.0009765625f) ok = false;
            }
            for (int c = 0; c < k; ++c) {
                const int kt = c >> 5, ks = (c >> 4) & 1, half = (c >> 3) & 1, e = c & 7;
                float v = w[(size_t)r * k + c];
                uint16_t* blk = out + ((size_t)panel * nkt + kt) * nplanes * 1024 + (size_t)(ks * 64 + half * 32 + l31) * 8 + e;
                if (nprod == 13) {
                    if (!(fabsf(v) < 65504.0f)) ok = false;
                    const _Float16 y1 = (_Float16)v;
                    const _Float16 y2 = (_Float16)((v - (float)y1) * 2048.0f);
                    memcpy(&blk[0], &y1, 2);
                    memcpy(&blk[1024], &y2, 2);
                    continue;
                }
                for (int pl = 0; pl < nplanes; ++pl) {
                    float back;
                    blk[(size_t)pl * 1024] = f32_to_bf16_rne(v, &back);
                    v = v - back;
                }
            }
        }
    };
    // rows are independent (a row writes its own 16-byte pieces): panels dealt to a few host threads
    const size_t work = (size_t)n * k;
    const int nthreads = work < ((size_t)1 << 20) ? 1 : (int)std::min<size_t>(8, std::max(1u, std::thread::hardware_concurrency()));
    if (nthreads <= 1) {
        pack_rows(0, n);
    } else {
        std::vector<std::thread> th;
        const int per = ((npanels + nthreads - 1) / nthreads) * 32;
        for (int t = 0; t < nthreads; ++t) {
            const int r0 = std::min(n, t * per), r1 = std::min(n, (t + 1) * per);
            if (r0 < r1) th.emplace_back(pack_rows, r0, r1);
        }
        for (auto& x : th) x.join();
    }
    return ok.load();
}

// 1 = 128x128, 2 = 64x128, 5 = 160x128, 7 = 64x256 as 4 waves of 64 columns, 9 = 64x256 as 8 waves of 32 columns (M x N).
// Measured with tools/gemm_split_bench.hip on MI355X: 64-row tiles win at every shape of the model, the 640 decode rows as
// well as the 21376 encoder rows (several workgroups per CU hide each other's barrier and load latencies; under a dense
// MFMA load the chip lowers its clock, so the taller tiles' better MFMA-per-byte ratio does not show as wall time).  The
// 8-wave workgroup splits an A tile once for 256 columns instead of 128 (the split is ~15 % of the fp16x3 kernel's time) at
// the same number of waves per CU: 5-10 % faster wherever there are enough 256-column tiles to go round, i.e. everywhere
// but the 1024 x 1024 projections of the decode rows.
static int legacy_tile(int M, int N, int K)
{
    const long t256 = (long)((M + 63) / 64) * ((N + 255) / 256);
    return t256 * std::max(1, K / 1024) >= 64 ? 9 : 2;
}

// The 8-wave kernels hold 158-222 registers per lane: ONE workgroup per CU, so a launch of more than 256 workgroups runs
// in rounds and the last round leaves most of the chip idle (640 x 8192 as 64 x 256 tiles = 320 workgroups = two rounds:
// 55 us; as 96 x 256 tiles = 224 workgroups = one round: 42 us).  For the decode rows of a whole batch (M = 640: seven
// 96-row tiles waste 5 % of the rows) the 96 x 256 tile is taken when tiles x K-slices lands in (205, 256]: measured with
// tools/gemm_split_bench.hip (cold weights), GLU 55 -> 42 us, FFN2 27 -> 23 us (8 slices), logits 26 -> 23 us (2 slices);
// QKV (84 tiles x 2 = 168 workgroups) stays on 64 x 256 x 2 slices (240 workgroups: 17 us against 20 us).
// Returns the number of K-slices to run 96 x 256 tiles with, 0 = do not.
static int tall_slices(int M, int N, int K, bool may_split)
{
    if (M <= 576 || M > 4096 || legacy_tile(M, N, K) != 9) return 0;
    const int rows = (M + 95) / 96 * 96;
    if (rows * 100 > M * 106) return 0;                            // more than 6 % of padded rows
    const long tiles = (long)(rows / 96) * ((N + 255) / 256);
    int split = 1;
    if (may_split)
        while (split * 2 <= ASR_MAX_SPLITK && tiles * split * 2 <= 256 && K / (split * 2) >= 128) split *= 2;
    const long wgs = tiles * split;
    return (wgs > 205 && wgs <= 256) ? split : 0;
}

int gemm_split_pick_tile(int M, int N, int K, int split_k, int epilogue)
{
    const int ts = tall_slices(M, N, K, epilogue == EPI_PARTIAL);
    if (ts && ts == std::max(1, split_k)) return 6;
    // Many rows (the encoder's 21376 .. 100000, the teacher-forced decoder): 128 x 256 tiles.  Every weight fragment a wave loads
    // serves four 32-row MFMA blocks instead of two, i.e. half the L2 -> register weight traffic per product.  Round 3, alone on the
    // GPU (tools/gemm_split_bench, 21376 rows, fp16x3): QKV 541 -> 480 us, GLU 1274 -> 1174, FFN2 656 -> 620, projections 161 -> 145
    // (bf16: FFN2 317 -> 265, QKV 225 -> 210); 160 x 256 spills.  With four handles sharing the GPU, whole bench: 186.9 -> 185.9 ms
    // per step, config 5 (50 x 30 s) +0.8 % / bf16 +1.8 %.  (Round 1 measured 64-row tiles ahead for these rows too - with the
    // 4-wave kernel of that time.)  ASR_AMD_ENC_TILE=6|9|10: development.
    static const int enc_tile = getenv("ASR_AMD_ENC_TILE") ? atoi(getenv("ASR_AMD_ENC_TILE")) : 10;
    if (M > 4096 && std::max(1, split_k) == 1 && (enc_tile == 6 || enc_tile == 9 || enc_tile == 10)) return enc_tile;
    return legacy_tile(M, N, K);
}

// K-slices for the decode-time projections: a k-tile costs a workgroup about 0.3 us while it waits on its own loads and
// barrier, about 0.45 us (six products) of matrix pipe when workgroups queue on a CU; every slab costs one more write +
// read of M x N floats by the consumer.  Constants fitted to tools/gemm_split_bench.hip and to an end-to-end sweep of the
// bench (tools/splits_sweep.sh).
int gemm_split_pick_split_k(int M, int N, int K, int nprod)
{
    if (const int ts = tall_slices(M, N, K, true)) return ts;
    const int bn = legacy_tile(M, N, K) == 9 ? 256 : 128;
    const double tiles = (double)((M + 63) / 64) * ((N + bn - 1) / bn);
    if (tiles * (bn / 128) >= 1024) return 1;
    const double pipe_us = 0.45 * (nprod == 6 ? 1.0 : nprod == 3 || nprod == 13 ? 0.6 : 0.4), lat_us = 0.3;
    int best = 1;
    double best_t = 1e30;
    for (int split = 1; split <= ASR_MAX_SPLITK; split *= 2) {
        if (split > 1 && K / split < 128) break;
        const double kt = ceil((double)K / split / 32.0);
        const double rounds = ceil(tiles * split / 256.0 - 1e-9);
        const double t = fmax(rounds * kt * pipe_us, kt * lat_us) + split * (double)M * N * 8.0 / 4.0e6;
        if (t < best_t - 1e-9) { best_t = t; best = split; }
    }
    return best;
}

hipError_t launch_gemm_split(const GemmArgs& a, int nprod, int tile_hint, hipStream_t s)
{
    if (a.M <= 0 || a.N <= 0) return hipSuccess;
    if ((a.K % SBK) || (a.lda & 3) || !a.Wp) return hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(a.A) | reinterpret_cast<uintptr_t>(a.Wp)) & 15) return hipErrorInvalidValue;
    if (a.split_k > 1 && a.epilogue != EPI_PARTIAL) return hipErrorInvalidValue;
    if (a.epilogue == EPI_GLU && (a.N & 31)) return hipErrorInvalidValue;
    const int tile = tile_hint ? tile_hint : gemm_split_pick_tile(a.M, a.N, a.K, a.split_k, a.epilogue);
    switch (nprod) {
        case 13: return launch_split_nprod<13>(a, tile, s);
        case 6: return launch_split_nprod<6>(a, tile, s);
        case 3: return launch_split_nprod<3>(a, tile, s);
        case 1: return launch_split_nprod<1>(a, tile, s);
    }
    return hipErrorInvalidValue;
}
