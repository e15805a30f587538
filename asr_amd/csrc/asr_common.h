// Internal declarations shared by the HIP translation units of libasr_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <atomic>

#define ASR_D_HEAD 64          // hidden_units / num_heads of every reference config
#define ASR_MAX_BEAM 32
#define ASR_MAX_SPLITK 16
#define ASR_XATT_SLACK_ROWS 16  // spare rows behind the cross-attention K / V arrays and padding flags (attention.hip reads whole 16-key blocks)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- GEMM (gemm_f32.hip) -------------------------------------------------------------------
enum GemmEpilogue {
    EPI_BIAS = 0,      // C = acc + bias (bias may be null)
    EPI_GLU = 1,       // W rows packed per 32: 16 value rows | 16 gate rows; C[m, n/2] = (a+b1)*sigmoid(g+b2)
    EPI_PARTIAL = 2,   // C[z][m][n] = acc of K-slice z (split-K slabs, reduced by the consumer)
    EPI_RELU = 3, EPI_SIGMOID = 4, EPI_TANH = 5, EPI_SWISH = 6   // 1-argument ff activations
};

struct GemmArgs {
    const float* A;    // [M, K] row-major, leading dimension lda
    const float* W;    // [N, K] row-major ("kernel [out, in]", utils.py:551), leading dimension ldw
    const uint16_t* Wp; // gemm_split.hip only: the same matrix as bf16 planes in panel order (gemm_split_weights_host; W unused)
    const float* bias; // [N] or null
    float* C;          // [M, ldc] (EPI_PARTIAL: [split_k][M][ldc])
    int M, N, K;
    int lda, ldw, ldc;
    int split_k;       // >= 1; > 1 only with EPI_PARTIAL
    int epilogue;
    int32_t* range_flag; // gemm_split.hip, nprod 13 only: set to 1 when an activation does not fit fp16 (may be null)
};
// tile_hint: 0 = choose from the shape, 1 = 128x128, 2 = 64x64, 3 = 64x128(M x N), 4 = 128x64
int gemm_f32_pick_tile(int M, int N, int split_k, int epilogue);
hipError_t launch_gemm_f32(const GemmArgs& a, int tile_hint, hipStream_t s);
// number of K-slices the decode-time projections should use for this shape (fills the chip)
int gemm_pick_split_k(int M, int N, int K, int nprod);
// true: always use the register-staged main loop (the DMA-staged loop needs K % 32 == 0 and is the default when it applies)
void gemm_force_reg_stage(bool on);

// f32 GEMM on the 16-bit matrix cores by operand splitting (gemm_split.hip); nprod = 13 (f32 accuracy by three fp16
// products; operands must fit fp16's range, see GemmArgs.range_flag), 6 (f32 accuracy by six bf16 products), 3 or 1 (bf16).
// Needs K % 32 == 0 and a.Wp = the weight matrix as made by gemm_split_weights_host (gemm_split_planes(nprod) bf16 planes in
// the kernel's panel order; gemm_split_weight_elems() uint16 elements).
// tile_hint: 0 = choose from the shape, 2 = 64x128, 9 = 64x256 (8 waves), 6 = 96x256, 10 = 128x256 (M x N); with -DASR_GEMM_ALL_TILES also 1 = 128x128,
// 5 = 160x128, 7 = 64x256 (4 waves); an unknown hint runs 64x128
hipError_t launch_gemm_split(const GemmArgs& a, int nprod, int tile_hint, hipStream_t s);
int gemm_split_pick_tile(int M, int N, int K, int split_k, int epilogue);
int gemm_split_pick_split_k(int M, int N, int K, int nprod);
int gemm_split_planes(int nprod);
size_t gemm_split_weight_elems(int n, int k, int nplanes);
bool gemm_split_weights_host(const float* w, int n, int k, int nprod, uint16_t* out);

// ---- row kernels (rowops.hip) ----------------------------------------------------------------
hipError_t launch_frame_stack_lens(const float* raw, const int32_t* lens, int n, int raw_frames, int raw_dim, int left,
                                   int right, int sub, float* out, hipStream_t s);
hipError_t launch_frame_stack(const float* raw, int n, int raw_frames, int raw_dim, int left, int right,
                              int sub, float* out, hipStream_t s);
// feature front end (frontend.hip; third_party/feat_convert/io/fea.py)
hipError_t launch_fea_add_delt(const float* x, int n, int dim, float* out, hipStream_t s);
hipError_t launch_fea_cmvn_global(const float* x, const float* mean, const float* var, int n, int dim, float* out, hipStream_t s);
hipError_t launch_fea_cmvn_utt(const float* x, int n, int dim, float* out, hipStream_t s);
hipError_t launch_fea_splice(const float* x, int n, int dim, int left, int right, int sub, int offset, int edge, float* out,
                             hipStream_t s);
hipError_t launch_pad_mask(const float* X, int rows, int d, int32_t* pad, hipStream_t s);
// contrib layer_norm over (t, hidden) per utterance + timing signal; scratch >= n*(2*chunks+2) doubles
hipError_t launch_input_norm_timing(const float* x, const float* gamma, const float* beta, const float* pe,
                                    int n, int t, int hidden, double* scratch, float* y, hipStream_t s);
size_t input_norm_scratch_doubles(int n, int t, int hidden);
// y = LN(resid + dbias + sum_z part[z]) * scale + bias; rows with mask[row] != 0 are zeroed;
// pad2_out (optional) receives (sum |y| == 0)
hipError_t launch_add_layer_norm(const float* resid, const float* part, int nsplit, size_t split_stride,
                                 const float* dbias, const float* scale, const float* bias,
                                 const int32_t* mask, int32_t* pad2_out, int rows, int hidden, float* y,
                                 hipStream_t s);
hipError_t launch_add_layer_norm_done(const float* resid, const float* part, int nsplit, size_t split_stride,
                                      const float* dbias, const float* scale, const float* bias,
                                      const int32_t* mask, int32_t* pad2_out, int rows, int hidden, float* y,
                                      const int32_t* done, hipStream_t s);
hipError_t launch_embed_pe(const float* E, const int32_t* preds0, const int32_t* preds1, int preds_stride,
                           const int32_t* pos_dev, const float* pe, float mult, int rows, int hidden, float* y,
                           const int32_t* done, hipStream_t s);
// teacher-forced decoder input: y[r] = E[shift_right(Y)[r]] * mult + pe[r % ty]; Y int32 [rows] = [n][ty]
hipError_t launch_embed_shifted(const float* E, const int32_t* Y, int rows, int ty, int vocab, const float* pe, float mult,
                                int hidden, float* y, hipStream_t s);
// row_loss[r] = (Y[r] != 0) * (logsumexp(logits[r]) - logits[r][Y[r]])
hipError_t launch_xent_rows(const float* logits, const int32_t* Y, int rows, int vocab, float* row_loss, hipStream_t s);
// kv [n*t, 2*hidden] (k | v) -> k, v [n][heads][t][64], float32 or (kv_bf16) bf16 rounded to nearest even
// range_flag (may be null): set when a value does not fit fp16 (what the F16X3 attention kernels need)
hipError_t launch_pack_cross_kv(const float* kv, int n, int t, int heads, void* k, void* v, int kv_bf16, int32_t* range_flag, hipStream_t s);
// logits [nsplit][rows][vocab] split-K slabs (summed in slab order; nsplit = 1: plain logits)
hipError_t launch_log_softmax_topk(const float* logits, int nsplit, size_t split_stride, int rows, int vocab, int k,
                                   float* values, int32_t* indices, const int32_t* done, hipStream_t s);

struct BeamState {       // device pointers; double-buffered arrays are indexed by (*pos & 1)
    float* scores[2];          // [R]
    int32_t* finished[2];      // [R]
    int32_t* preds[2];         // [R, width]
    int32_t* anc[2];           // [R, width] slot (0..K-1) holding position j of row r's history
    int32_t* live;             // [B][K*width] (position << 5 | slot) of every cache row some beam of the utterance still
                               // descends from, positions 0 .. *pos-1, ordered by position then slot
    int32_t* live_cnt;         // [B] entries in live
    int32_t* pos;              // [4] pos = decode steps done so far (= position of the token fed next), done, unfinished, arrivals
    int32_t* done;             // [1] loop has ended
    int32_t* unfinished;       // [1] scratch counter
    int32_t* tokens;           // [R] last tokens (trace)
    int32_t* parents;          // [R] last parents, global row index (trace)
    int width;                 // max_target_length + 2
    int limit;                 // min(T + 50, max_target_length)
    const int32_t* limit_u;    // null, or [B]: utterance u's own limit (a joint decode of groups with different T): once
                               // limit_u[u] + 1 steps are done the utterance's beams count as finished - they keep their
                               // scores and are padded with 3, exactly what further steps do to finished beams
};
hipError_t launch_beam_init(const BeamState& st, int b, int k, hipStream_t s);
// one pruning step + the next step's decoder input (y_next[r] = E[token r] * mult + pe[pos + 1]) + the loop condition
hipError_t launch_beam_step(const BeamState& st, const float* cand_scores, const int32_t* cand_tokens,
                            int b, int k, const float* E, const float* pe, float mult, int hidden, float* y_next,
                            hipStream_t s);
// stand-alone pruning step for asr_op_beam_step
hipError_t launch_beam_prune(const float* scores, const int32_t* finished, const float* cand_scores,
                             const int32_t* cand_tokens, int b, int k, float* new_scores, int32_t* tokens,
                             int32_t* parents, int32_t* finished_out, hipStream_t s);

// ---- attention (attention.hip) ---------------------------------------------------------------
// mode: 13 = products as three fp16 MFMAs (f32_f16x3; operands outside fp16's range set *range_flag), 1 = bf16 K / V with
// two-term q and P (`precision: bf16`), anything else = the exact-f32 matrix instruction
hipError_t launch_encoder_attention(const float* qkv, const int32_t* pad, int n, int t, int heads,
                                    float* ctx, int mode, int32_t* range_flag, hipStream_t s);
// general many-query attention: q rows [n*tq] (stride ldq), k / v rows [n*tk]; pad [n][tk] or null; causal needs tq == tk
hipError_t launch_mha(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const int32_t* pad, int causal,
                      int n, int tq, int tk, int heads, float* ctx, int mode, int32_t* range_flag, hipStream_t s);
struct SelfAttnArgs {
    const float* qkv;      // [nsplit][R][3*hidden] partial qkv projections (q | k | v), bias not yet added
    int nsplit; size_t split_stride;
    const float* bias;     // [3*hidden]
    void* kcache;          // [b][heads][width][K][64] for this block: head-major, then position, beam slot inside;
    void* vcache;          // float32, or bf16 when kv_bf16
    int kv_bf16;
    const int32_t* anc0;   // BeamState.anc[0], anc[1]
    const int32_t* anc1;
    const int32_t* live;   // BeamState.live, live_cnt
    const int32_t* live_cnt;
    const int32_t* pos;
    const int32_t* done;
    float* ctx;            // [R, hidden]
    int b, k, heads, width;
    int lds_stride;        // >= (max position this decode can reach) + 1, <= width: row pitch of the LDS score / ancestry tables
    int f16x3;             // float32 caches only: products as three fp16 MFMAs per multiply (operands must fit fp16's range)
    int32_t* range_flag;   // f16x3: set when q, k or v of the new position does not fit (may be null)
};
size_t self_attention_lds_bytes(int k, int lds_stride);
hipError_t launch_decoder_self_attention(const SelfAttnArgs& a, hipStream_t s);
struct CrossAttnArgs {
    const float* qpart;    // [nsplit][R][hidden] partial q projections
    int nsplit; size_t split_stride;
    const float* qbias;    // [hidden]
    const void* kc;        // [b][heads][t][64] + ASR_XATT_SLACK_ROWS rows of 64 values that may be read (not used)
    const void* v;         // [b][heads][t][64] + the same slack; float32, or bf16 when kv_bf16
    int kv_bf16;
    const int32_t* pad2;   // [b][t] + ASR_XATT_SLACK_ROWS ints of slack
    const int32_t* done;
    float* ctx;            // [R, hidden]
    int b, k, heads, t;
    const int32_t* t_arr;  // null, or [b]: utterance u has t_arr[u] keys (a joint decode of groups with different t; `t` is then unused)
    const int32_t* row_off;// with t_arr: [b] row index of utterance u's first key among all utterances' keys: its K / V are
                           // [heads][t_arr[u]][64] at element row_off[u] * heads * 64, its padding flags at pad2 + row_off[u]
    int f16x3;             // float32 K / V only: products as three fp16 MFMAs per multiply (operands must fit fp16's range)
    int32_t* range_flag;   // f16x3: set when a q value does not fit (may be null); K / V are checked where they are written
};
hipError_t launch_decoder_cross_attention(const CrossAttnArgs& a, hipStream_t s);
