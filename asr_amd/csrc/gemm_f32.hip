// fp32 "NT" GEMM on the gfx950 matrix cores:  C[M,N] = A[M,K] . W[N,K]^T  (+ epilogue).
//
// Replaces utils.dense (transformer/utils.py:546-557: reshape(x,[-1,in]) @ kernel^T + bias with
// kernel stored [out, in]) and the GLU pairing of utils.py:559-561 / model.py:441.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 -- exact f32 products accumulated in f32 (a k-ordered fmaf
// chain, no reduced-precision path), which is what the 1e-3 logits parity bar needs.  Both
// operands are K-contiguous in HBM, so tiles are staged with 16-byte loads of 4 consecutive k and
// each lane feeds four MFMAs from one ds_read_b128 (lanes 0-31 take k-chunk 2q, lanes 32-63 chunk
// 2q+1: the MFMA k index is only a summation label, so any A/B-consistent assignment is valid).
// LDS image: [rows][8 chunks of 16 B] with the chunk XOR-swizzled by (row & 7).
#include "asr_common.h"
#include "gemm_epilogue.h"
#include <math.h>
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int BK = 32;   // k-depth of one LDS stage (floats)
bool g_force_reg_stage = false;   // development / test switch: always the general (MODE 0) main loop
#ifndef ASR_GEMM_DEBUG
#define ASR_GEMM_DEBUG 0   // development only: 1 = no global loads in the loop, 2 = no LDS stores, 4 = no barrier, 8 = loads re-read tile 0
#endif


// MODE 0: register-staged tiles, any K % 4 == 0 (a k-chunk past K is zeroed on its way into LDS).
// MODE 2: the same for K % 64 == 0 (every K-slice is a whole number of k-tile PAIRS), without those selects.
// (A variant that staged tiles with global_load_lds_dwordx4, swizzle applied on the source address, measured
// within +-3 % of this register-staged loop at every tile size and was dropped.)
template <int BM, int BN, int WM, int WN, int EPI, int MODE>
__global__ __launch_bounds__(256) void gemm_f32_nt_kernel(GemmArgs p, int nt_m, int nt_n, int kchunk)
{
    constexpr bool KTAIL = MODE == 0;
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
    constexpr int A_LD = BM / 32, B_LD = BN / 32;      // 16-B loads per thread per stage

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    f32x4* As = reinterpret_cast<f32x4*>(smem_raw);    // [2][BM][8]
    f32x4* Bs = As + 2 * BM * 8;                       // [2][BN][8]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // XCD-aware tile order: blocks b and b+8 share an XCD (its L2); give each XCD a contiguous run
    // of tiles, M-tile fastest inside groups of 8 M-tiles, so the W panel of a column tile is read
    // by one L2 and the A panels of a group stay resident in it.
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
    const int nk = (kend - kbeg + BK - 1) / BK;

    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int n = 0; n < NI; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.0f;

    const int lrow = tid >> 3, lch = tid & 7;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // Software pipeline: two register sets hold the global loads of the next two k-tiles (set = tile
    // parity), two LDS buffers hold the current and the next tile, and the A/B fragments of k-chunk q+1
    // are read from LDS before the MFMAs of chunk q are issued.  The LDS stores of tile kt+1 are issued in
    // front of the MFMAs of chunk 2, the barrier and the first fragment read of tile kt+1 in front of the
    // MFMAs of chunk 3: no LDS latency sits between two MFMA groups (two waves of a SIMD run in phase, so
    // a bubble in one is a bubble in both - measured with tools/mfma_probe.hip).
    const float* arow[A_LD];
    const float* brow[B_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) arow[i] = p.A + (size_t)min(m0 + lrow + 32 * i, p.M - 1) * p.lda;
#pragma unroll
    for (int i = 0; i < B_LD; ++i) brow[i] = p.W + (size_t)min(n0 + lrow + 32 * i, p.N - 1) * p.ldw;

    f32x4 ra[2][A_LD], rb[2][B_LD];
    bool kvalid[2] = {false, false};
    // Branch-free loads (so that the compiler can count outstanding loads per register set): rows past
    // M / N are clamped to the last row (their products land in C rows / columns that are never stored)
    // and a k-chunk past kend is loaded from k = kbeg and zeroed on its way into LDS.
    auto gload = [&](auto setc, int k0) {
        constexpr int S = decltype(setc)::value;
        const int kk = k0 + lch * 4;
        const bool kok = kk < kend;
        const int kc = (ASR_GEMM_DEBUG & 8) ? kbeg + lch * 4 : (kok ? kk : kbeg);   // debug 8: always the first tile (cache-resident)
        kvalid[S] = kok;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) ra[S][i] = *reinterpret_cast<const f32x4*>(arow[i] + kc);
#pragma unroll
        for (int i = 0; i < B_LD; ++i) rb[S][i] = *reinterpret_cast<const f32x4*>(brow[i] + kc);
    };
    auto lstore = [&](auto setc, int buf) {
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int row = lrow + 32 * i;
            As[(buf * BM + row) * 8 + (lch ^ (row & 7))] = (!KTAIL || kvalid[S]) ? ra[S][i] : zero4;
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const int row = lrow + 32 * i;
            Bs[(buf * BN + row) * 8 + (lch ^ (row & 7))] = (!KTAIL || kvalid[S]) ? rb[S][i] : zero4;
        }
    };

    const int l31 = lane & 31, lhalf = lane >> 5;
    f32x4 af[2][MI], bf[2][NI];
    auto fload = [&](auto slotc, int buf, int q) {
        constexpr int S = decltype(slotc)::value;
        const int ch = 2 * q + lhalf;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row = wm0 + i * 32 + l31;
            af[S][i] = As[(buf * BM + row) * 8 + (ch ^ (row & 7))];
        }
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            const int row = wn0 + n * 32 + l31;
            bf[S][n] = Bs[(buf * BN + row) * 8 + (ch ^ (row & 7))];
        }
    };
    auto mma = [&](auto slotc) {
        constexpr int S = decltype(slotc)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int n = 0; n < NI; ++n)
                    acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[S][i][j], bf[S][n][j], acc[i][n], 0, 0, 0);
    };
    using c0 = std::integral_constant<int, 0>;
    using c1 = std::integral_constant<int, 1>;

    {
        // One k-tile; SET = register set that holds tile kt+1 (the other set holds tile kt+2, in flight).
        // The side work of a tile - fragment reads of the next chunk, the selects + LDS stores of tile kt+1, the
        // global loads of tile kt+3 - is dealt out ONE operation after each MFMA (an f32 MFMA occupies the pipe
        // for 64 cycles, a side operation issues in a few), instead of in blocks between MFMA groups where it
        // would leave the matrix pipe idle; sched_barrier pins that order.
        constexpr int NM = 4 * MI * NI;                          // MFMAs per chunk
        auto side_fload = [&](auto slotc, int buf, int q, int op) {   // op < MI + NI: one fragment read
            constexpr int S = decltype(slotc)::value;
            const int ch = 2 * q + lhalf;
#pragma unroll
            for (int i = 0; i < MI; ++i)
                if (op == i) { const int row = wm0 + i * 32 + l31; af[S][i] = As[(buf * BM + row) * 8 + (ch ^ (row & 7))]; }
#pragma unroll
            for (int n = 0; n < NI; ++n)
                if (op == MI + n) { const int row = wn0 + n * 32 + l31; bf[S][n] = Bs[(buf * BN + row) * 8 + (ch ^ (row & 7))]; }
        };
        auto side_lstore = [&](auto setc, int buf, int op) {     // op < A_LD + B_LD: one 16-byte LDS store
            constexpr int S = decltype(setc)::value;
            if (ASR_GEMM_DEBUG & 2) return;
#pragma unroll
            for (int i = 0; i < A_LD; ++i)
                if (op == i) { const int row = lrow + 32 * i; As[(buf * BM + row) * 8 + (lch ^ (row & 7))] = (!KTAIL || kvalid[S]) ? ra[S][i] : zero4; }
#pragma unroll
            for (int i = 0; i < B_LD; ++i)
                if (op == A_LD + i) { const int row = lrow + 32 * i; Bs[(buf * BN + row) * 8 + (lch ^ (row & 7))] = (!KTAIL || kvalid[S]) ? rb[S][i] : zero4; }
        };
        int g_kc = 0;
        auto side_gload = [&](auto setc, int op) {               // op < A_LD + B_LD: one 16-byte global load
            constexpr int S = decltype(setc)::value;
            if (ASR_GEMM_DEBUG & 1) return;
#pragma unroll
            for (int i = 0; i < A_LD; ++i)
                if (op == i) ra[S][i] = *reinterpret_cast<const f32x4*>(arow[i] + g_kc);
#pragma unroll
            for (int i = 0; i < B_LD; ++i)
                if (op == A_LD + i) rb[S][i] = *reinterpret_cast<const f32x4*>(brow[i] + g_kc);
        };
        // chunk q of tile (buf): MFMAs from fragment slot q & 1, `side(n)` after the n-th MFMA
        auto chunk = [&](auto slotc, auto&& side) {
            constexpr int S = decltype(slotc)::value;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int n = 0; n < NI; ++n) {
                        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[S][i][j], bf[S][n][j], acc[i][n], 0, 0, 0);
                        side((j * MI + i) * NI + n);
                        __builtin_amdgcn_sched_barrier(0);
                    }
        };
        constexpr int NF = MI + NI, NL = A_LD + B_LD;
        constexpr int LS_PER = (NL + 2 * (NM - NF) - 1) / (2 * (NM - NF));    // LDS stores per slot over chunks 1 and 2
        auto ktile = [&](auto setc, int kt) {
            constexpr int S = decltype(setc)::value;
            using cS = std::integral_constant<int, S>;
            const int buf = kt & 1;
            // chunk 0: fragments of chunk 1
            chunk(c0{}, [&](int n) { if (n < NF) side_fload(c1{}, buf, 1, n); });
            // chunk 1: fragments of chunk 2, then the first LDS stores of tile kt+1
            chunk(c1{}, [&](int n) {
                if (n < NF) side_fload(c0{}, buf, 2, n);
                else
#pragma unroll
                    for (int e = 0; e < LS_PER; ++e) side_lstore(cS{}, buf ^ 1, (n - NF) * LS_PER + e);
            });
            // chunk 2: fragments of chunk 3, then the remaining LDS stores
            chunk(c0{}, [&](int n) {
                if (n < NF) side_fload(c1{}, buf, 3, n);
                else
#pragma unroll
                    for (int e = 0; e < LS_PER; ++e) side_lstore(cS{}, buf ^ 1, ((NM - NF) + (n - NF)) * LS_PER + e);
            });
            if (!(ASR_GEMM_DEBUG & 4)) __syncthreads();
            // chunk 3: first fragments of tile kt+1, then the global loads of tile kt+3 into the set just stored
            {
                const int k0 = kbeg + (kt + 3) * BK, kk = k0 + lch * 4;
                const bool kok = kk < kend;
                g_kc = kok ? kk : kbeg;
                fload(c0{}, buf ^ 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                constexpr int GL_PER = (NL + NM - 1) / NM;
                chunk(c1{}, [&](int n) {
#pragma unroll
                    for (int e = 0; e < GL_PER; ++e) side_gload(cS{}, n * GL_PER + e);
                });
                kvalid[S] = kok;
            }
        };

        // Every step is unconditional (tiles past the K range load a clamped address and are stored to LDS as
        // zeros), so the number of outstanding loads is static and the waits are exact; an odd tile count
        // costs one all-zero tile.
        if (nk > 0) {
            gload(c0{}, kbeg);
            gload(c1{}, kbeg + BK);
            lstore(c0{}, 0);
            __syncthreads();
            gload(c0{}, kbeg + 2 * BK);
            fload(c0{}, 0, 0);
            for (int kt = 0; kt < nk; kt += 2) {
                ktile(c1{}, kt);
                ktile(c0{}, kt + 1);
            }
        }
    }

    gemm_store_tile<EPI, MI, NI>(p, acc, m0, n0, wm0, wn0, z, lane);
}

template <int BM, int BN, int WM, int WN, int EPI, int MODE>
hipError_t launch_tile_impl(const GemmArgs& a, hipStream_t s)
{
    const int nt_m = (a.M + BM - 1) / BM, nt_n = (a.N + BN - 1) / BN;
    const int split = a.split_k < 1 ? 1 : a.split_k;
    int kchunk = (a.K + split - 1) / split;
    constexpr int KG = MODE == 2 ? 2 * BK : BK;       // MODE 2 runs k-tiles in pairs and has no zero tile to pad with
    kchunk = (kchunk + KG - 1) / KG * KG;
    const size_t lds = (size_t)2 * (BM + BN) * 8 * sizeof(f32x4);
    auto kern = gemm_f32_nt_kernel<BM, BN, WM, WN, EPI, MODE>;
    static std::atomic<bool> attr_done[64];      // zero-initialised; several handles' host threads may launch first at once (the call is idempotent)   // per instantiation, per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_done[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done[dev].store(true, std::memory_order_release);
    }
    dim3 grid(nt_m * nt_n, split, 1);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a, nt_m, nt_n, kchunk);
    return hipGetLastError();
}

template <int BM, int BN, int WM, int WN, int EPI>
hipError_t launch_tile(const GemmArgs& a, hipStream_t s)
{
    if ((a.K % (2 * BK)) == 0 && !g_force_reg_stage) return launch_tile_impl<BM, BN, WM, WN, EPI, 2>(a, s);   // every production shape
    return launch_tile_impl<BM, BN, WM, WN, EPI, 0>(a, s);
}

template <int EPI>
hipError_t launch_epi(const GemmArgs& a, int tile, hipStream_t s)
{
    switch (tile) {
        case 1: return launch_tile<128, 128, 64, 64, EPI>(a, s);
        case 3: return launch_tile<64, 128, 32, 64, EPI>(a, s);
        case 4: return launch_tile<128, 64, 64, 32, EPI>(a, s);
        default: return launch_tile<64, 64, 32, 32, EPI>(a, s);
    }
}

}  // namespace

void gemm_force_reg_stage(bool on) { g_force_reg_stage = on; }

int gemm_pick_split_k(int M, int N, int K, int nprod)
{
    if (nprod) return gemm_split_pick_split_k(M, N, K, nprod);
    // Decode-time projections (M = batch*beam rows) run 64x64 tiles, 4 waves each.  The f32 MFMA pipe is
    // the bound, so what matters is how evenly the waves fill the 1024 SIMDs: w waves per SIMD cost
    // ceil(w) turns of (K / split) / 32 k-tiles (0.45 us each); every extra slab costs one more write +
    // read of M x N floats.  Constants fitted to tools/gemm_bench.hip on MI355X.
    // bf16-plane modes: a k-tile costs nprod 32-cycle MFMAs per 16-deep step instead of eight 64-cycle ones
    const double ktile_us = nprod == 0 ? 0.45 : 0.45 * (nprod * 32.0) / 512.0 + 0.05;
    const double tiles = (double)((M + 63) / 64) * ((N + 63) / 64);
    const double w1 = tiles * 4.0 / 1024.0;
    int best = 1;
    double best_t = 1e30;
    for (int split = 1; split <= ASR_MAX_SPLITK; split *= 2) {
        if (split > 1 && K / split < 128) break;
        const double ktiles = ceil((double)K / split / 32.0);
        const double t = ceil(w1 * split - 1e-9) * ktiles * ktile_us + split * (double)M * N * 4.0 / 4.0e6;
        if (t < best_t - 1e-9) { best_t = t; best = split; }
    }
    return best;
}

// Tile choice of the exact-f32 GEMM (1 = 128x128, 2 = 64x64, 4 = 128x64; 3 = 64x128 only by hint).
int gemm_f32_pick_tile(int M, int N, int split_k, int epilogue)
{
    const long tiles128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    if (tiles128 >= 512) return 1;                              // encoder-sized: 128x128
    (void)split_k; (void)epilogue;    // (128x64 for the unsplit 640-row QKV measured the same as 64x64 in the pipeline)
    return 2;                                                   // decode rows: 64x64
}

hipError_t launch_gemm_f32(const GemmArgs& a, int tile_hint, hipStream_t s)
{
    if (a.M <= 0 || a.N <= 0) return hipSuccess;
    if ((a.K & 3) || (a.lda & 3) || (a.ldw & 3)) return hipErrorInvalidValue;
    if (a.split_k > 1 && a.epilogue != EPI_PARTIAL) return hipErrorInvalidValue;
    if (a.epilogue == EPI_GLU && (a.N & 31)) return hipErrorInvalidValue;
    const int tile = tile_hint ? tile_hint : gemm_f32_pick_tile(a.M, a.N, a.split_k, a.epilogue);
    switch (a.epilogue) {
        case EPI_BIAS: return launch_epi<EPI_BIAS>(a, tile, s);
        case EPI_GLU: return launch_epi<EPI_GLU>(a, tile, s);
        case EPI_PARTIAL: return launch_epi<EPI_PARTIAL>(a, tile, s);
        case EPI_RELU: return launch_epi<EPI_RELU>(a, tile, s);
        case EPI_SIGMOID: return launch_epi<EPI_SIGMOID>(a, tile, s);
        case EPI_TANH: return launch_epi<EPI_TANH>(a, tile, s);
        case EPI_SWISH: return launch_epi<EPI_SWISH>(a, tile, s);
    }
    return hipErrorInvalidValue;
}
