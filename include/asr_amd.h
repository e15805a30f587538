/*
 * asr_amd.h -- C ABI of libasr_amd.so: the MI355X (gfx950) implementation of the batched
 * inference path of shiyuzh2007/ASR (fbank frame stacking -> encoder -> beam-search decoder).
 *
 * The reference has no FFI for this path: it sits behind `Evaluator.beam_search(X)`
 * (transformer/evaluate.py:46-47), i.e. one `sess.run(model.prediction, feed_dict)` per batch.
 * Each entry point below names the reference interface it stands in for (file:line relative to
 * /root/reference/transformer/).  The Python host (asr_amd/) binds these with ctypes.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success and a
 * non-zero asr_status otherwise; asr_last_error() returns the message of the last failure on
 * that handle (or of the last failed asr_create when handle == NULL).  A handle owns one GPU's
 * weights, K/V caches and workspaces; it is not thread-safe; calls block until outputs are
 * valid.  Host-pointer arguments are ordinary pageable memory unless a name ends in `_dev`.
 */
#ifndef ASR_AMD_H
#define ASR_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct asr_handle asr_handle;

enum asr_status {
    ASR_OK = 0,
    ASR_ERR_ARG = 1,        /* bad argument / shape                     */
    ASR_ERR_HIP = 2,        /* a HIP runtime call or kernel failed      */
    ASR_ERR_WEIGHTS = 3,    /* unknown / missing / mis-shaped variable  */
    ASR_ERR_STATE = 4,      /* call order (e.g. decode before finalize) */
    ASR_ERR_CAPACITY = 5    /* output buffer too small                  */
};

enum asr_ff_activation { ASR_FF_RELU = 0, ASR_FF_GLU = 1, ASR_FF_SIGMOID = 2, ASR_FF_TANH = 3, ASR_FF_SWISH = 4 };

/* Arithmetic of the dense layers (utils.dense).  The reference computes in float32 (model.py:35); every mode
 * accumulates in float32 and keeps activations, attention, layer norms and the beam search in float32.
 *   F32         exact-f32 matrix instructions (v_mfma_f32_32x32x2_f32): products and sums in f32.
 *   F32_BF16X6  float32 operands split exactly into three bf16 terms each, six bf16 matrix products per k-step
 *               (terms below 2^-24 of the product dropped): f32-chain accuracy at 2.7x the matrix rate.  A layer
 *               whose input width is not a multiple of 32 runs on the F32 kernel in this and the next two modes.
 *   BF16X3      two terms each, three products (~2^-16 relative product error).
 *   BF16        operands rounded to bf16 (north_star's "bf16 MFMA" mode; tolerance stated in the tests).  This mode also
 *               stores the decoder's K/V caches (self-attention history and the per-utterance cross-attention K/V) as
 *               bf16: half the bytes of the HBM-bound attention loops; scores, soft-max and P.V stay float32.
 *   F32_F16X3   float32 operands split as x = x1 + 2^-11 x2 with x1 = fp16(x), x2 = fp16(2^11 (x - x1)) (11 + 11 significand
 *               bits); three fp16 matrix products per k-step, the 2^-11 group in an accumulator of its own: measured
 *               against float64 its error is below the exact-f32 kernel's, at 5.3x the f32 matrix rate.  fp16 holds
 *               magnitudes below 65504: weights are checked at asr_finalize_weights (a checkpoint that does not fit runs
 *               as F32_BF16X6), activations while they are split - a call that meets one outside the range is repeated
 *               on the F32_BF16X6 kernels before it returns, so results never depend on the range.
 */
enum asr_precision { ASR_PRECISION_F32 = 0, ASR_PRECISION_F32_BF16X6 = 1, ASR_PRECISION_BF16X3 = 2, ASR_PRECISION_BF16 = 3,
                     ASR_PRECISION_F32_F16X3 = 4 };

/* Model + search configuration: the YAML keys read by the inference graph
 * (model.py:19-45,434-442; config_template_*.yaml). */
typedef struct asr_config {
    int32_t input_dim;                   /* train.input_dim (320 = 80 x (3+1))            */
    int32_t hidden_units;                /* hidden_units                                  */
    int32_t num_heads;                   /* num_heads (hidden_units / num_heads must be 64) */
    int32_t num_blocks;                  /* num_blocks (encoder and decoder)              */
    int32_t dst_vocab_size;              /* dst_vocab_size                                */
    int32_t ff_activation;               /* enum asr_ff_activation (ff_activation)        */
    int32_t scale_embedding;             /* scale_embedding                               */
    int32_t tie_embedding_and_softmax;   /* tie_embedding_and_softmax                     */
    int32_t beam_size;                   /* test.beam_size (1..32)                        */
    int32_t max_target_length;           /* test.max_target_length                        */
    float   lp_alpha;                    /* test.lp_alpha                                 */
    int32_t device;                      /* HIP device ordinal this handle lives on       */
    int32_t precision;                   /* enum asr_precision (build extension; YAML key test.precision; the Python host's default
                                            is F32_F16X3, a zeroed struct means F32) */
    int32_t sub_batches;                 /* 1 = beam-search the batch as one run of kernels, 2..4 = as that many independent
                                            parts on their own streams (same results: utterances are independent), 0 = env
                                            ASR_AMD_SUBBATCH or automatic (two halves from 320 beam rows);
                                            build extension, YAML key test.sub_batches */
    int32_t shared_gpu;                  /* 0 = this handle has the GPU to itself: K-slices and tiles are chosen so that every
                                            kernel fills the chip alone (lowest latency of one batch); 1 = other handles decode on
                                            the same GPU at the same time (asr_amd.model.EnginePool): fewer K-slices, i.e. fewer
                                            bytes, because the other handles' kernels fill the chip.  Same results up to the
                                            summation order of the K-slices.  Build extension, YAML key test.shared_gpu */
    int32_t reserved[1];                 /* must be zero                                  */
} asr_config;

/* Number of HIP devices visible to this process (0 when there is none or on error).  Replaces the
 * `/gpu:%d` device list of Model.__init__ (model.py:27). */
int asr_device_count(void);

/* Replaces: `eval(config.model)(config, num_gpus)` + `build_test_model()` for ONE device
 * (evaluate.py:26-27; model.py:162-204). */
int asr_create(const asr_config* cfg, asr_handle** out);
void asr_destroy(asr_handle* h);
const char* asr_last_error(const asr_handle* h);

/* Replaces: `model.saver.restore(sess, latest_checkpoint)` (evaluate.py:36).  `tf_var_name` is the
 * TensorFlow variable name, `data` float32 host memory of `shape[0..ndim)`.
 * Unknown names are rejected with ASR_ERR_WEIGHTS unless they are optimizer slots / counters
 * (`.../Adam`, `.../Adam_1`, `beta1_power`, `beta2_power`, `global_step`), which are ignored. */
int asr_load_tensor(asr_handle* h, const char* tf_var_name, const float* data,
                    const int64_t* shape, int32_t ndim);
/* Fails with ASR_ERR_WEIGHTS (message lists the names) while any graph variable is missing;
 * on success packs the weights into their device layouts. */
int asr_finalize_weights(asr_handle* h);
/* Instead of asr_load_tensor + asr_finalize_weights: `h` uses the packed device weights `from` holds (one copy per GPU for
 * all handles of an asr_amd.model.EnginePool; the weights are read-only once finalized).  The reference's counterpart is
 * its single graph: every sess.run of evaluate.py:46-47 reads the one set of variables restored at evaluate.py:36.  Both
 * handles must live on the same GPU with the same model shape and precision mode (else ASR_ERR_ARG); `from` must be
 * finalized, `h` must not be (ASR_ERR_STATE).  The store is reference-counted: either handle may be destroyed first.
 * asr_device_bytes counts the store on the handle that filled it only. */
int asr_share_weights(asr_handle* h, asr_handle* from);
/* asr_config.shared_gpu / sub_batches for the calls that follow (same value ranges; sub_batches 0 = automatic).  A pool
 * of handles switches to the shared-GPU regime while it decodes batches side by side and back when one of its handles is
 * used alone (Model.predict, Evaluator.loss), so that a lone call gets the K-slices, tiles and two-stream halves that
 * fill the chip by themselves.  Not while another thread is inside a call on `h`.  No counterpart in the reference. */
int asr_set_sharing(asr_handle* h, int32_t shared_gpu, int32_t sub_batches);

/* Replaces: `Evaluator.beam_search(X)` for one device shard (evaluate.py:46-47; model.py:176-178):
 * X [n, t, d] float32 row-major, all-zero rows are padding (model.py:450).  Y receives int32
 * [n, *len_out] row-major (row stride *len_out), hypotheses right-padded with 3;
 * y_capacity = number of int32 elements Y can hold (n * (max_target_length + 1) always suffices).
 * n == 0 returns *len_out = 0 (model.py:184-187). */
int asr_decode(asr_handle* h, const float* X, int32_t n, int32_t t, int32_t d,
               int32_t* Y, int64_t y_capacity, int32_t* len_out);

/* Several batches of the reference's bucketed reader (utils.py:285-323), each padded to its OWN T, decoded as ONE beam
 * search.  No counterpart in the reference, whose translate loop (evaluate.py:52-66) runs one sess.run per batch: ten-frame
 * buckets of a few utterances leave every decode position streaming all weights for a handful of rows.  X = group after
 * group, group g being [n_g[g], t_g[g], d] row-major; Y [sum n_g, *len_out] in the same order.  Each group goes through the
 * encoder with its own T (the input layer norm runs over (T, H), model.py:453) and its utterances attend over their own T
 * keys; the decode steps run over all utterances' beams together.  An utterance's own loop limit min(T + 50,
 * max_target_length) (model.py:310) is kept: once it is reached its beams keep their scores and are padded with 3, which is
 * what the reference's loop does to finished beams (model.py:238-253), so hypotheses read up to the first 3 - what
 * indices_to_words does (utils.py) - equal those of group-by-group asr_decode calls up to float32 summation order (K-slice
 * counts follow the row count); the trailing columns of Y differ: *len_out is the joint loop's length. */
int asr_decode_groups(asr_handle* h, const float* X, int32_t n_groups, const int32_t* n_g, const int32_t* t_g, int32_t d,
                      int32_t* Y, int64_t y_capacity, int32_t* len_out);
/* Same, with X already resident in this device's memory (what bench.py times).  If
 * raw_frames > 0, X_dev is instead RAW fbank [n, raw_frames, raw_dim] and the frame-stacking
 * front end (left, right, sub) runs on the device first (third_party/feat_convert/io/fea.py:142-149),
 * giving t = ceil(raw_frames / sub), d = raw_dim * (left + right + 1). */
int asr_decode_dev(asr_handle* h, const float* X_dev, int32_t n, int32_t t, int32_t d,
                   int32_t raw_frames, int32_t raw_dim, int32_t left, int32_t right, int32_t sub,
                   int32_t* Y, int64_t y_capacity, int32_t* len_out);

/* Replaces: `Evaluator.loss(X, Y)` for one device shard (evaluate.py:49-50 -> model.loss_sum, model.py:171-190):
 * encoder, the teacher-forced decoder on shift_right(Y) (model.py:492-557; causal self-attention, cross-attention
 * over the encoder output) and test_loss (model.py:350-360) = sum over positions with Y != 0 of the softmax cross
 * entropy against Y.  Y int32 [n, ty], 0 = padding.  n == 0 gives 0.0 (model.py:184-187). */
int asr_loss(asr_handle* h, const float* X, int32_t n, int32_t t, int32_t d,
             const int32_t* Y, int32_t ty, double* loss_sum);

/* Front end alone.  Replaces tf_fea_splice (+ the `sub` subsampling implied by the
 * `feats.scp.left3_sub3` inputs): feats [n_frames, dim] -> out [ceil(n_frames/sub), dim*(left+right+1)]. */
int asr_stack_frames(asr_handle* h, const float* feats, int32_t n_frames, int32_t dim,
                     int32_t left, int32_t right, int32_t sub, float* out);

/* Feature front end of third_party/feat_convert/io/fea.py, one utterance per call (feats [n_frames, dim] row-major):
 *  asr_fea_add_delt     np_fea_add_delt / tf_fea_add_delt (fea.py:61-72,117-128): out [n_frames, 3*dim] =
 *                       feats | delta | delta-delta with the Kaldi windows of fea.py:20-52 (edge frames repeated);
 *                       bit-identical to the float32 numpy evaluation.
 *  asr_fea_cmvn_global  tf_fea_cmvn_global (fea.py:130-132): (feats - mean) / var, mean / var [dim] (the reference
 *                       divides by `var` as given); bit-identical.
 *  asr_fea_cmvn_utt     tf_fea_cmvn_utt (fea.py:134-140): per-column (x - mean) * rsqrt(E[x^2] - mean^2 + 1e-12).
 *  asr_fea_splice       tf_fea_splice (fea.py:142-149) generalised: keeps frames offset, offset+sub, ...;
 *                       edge = 0 pads the context with zeros (the reference), edge = 1 repeats the first / last
 *                       frame (Kaldi splice-feats); out [ceil((n_frames-offset)/sub), dim*(left+right+1)].
 *                       asr_stack_frames == offset 0, edge 0. */
int asr_fea_add_delt(asr_handle* h, const float* feats, int32_t n_frames, int32_t dim, float* out);
int asr_fea_cmvn_global(asr_handle* h, const float* feats, int32_t n_frames, int32_t dim,
                        const float* mean, const float* var, float* out);
int asr_fea_cmvn_utt(asr_handle* h, const float* feats, int32_t n_frames, int32_t dim, float* out);
int asr_fea_splice(asr_handle* h, const float* feats, int32_t n_frames, int32_t dim, int32_t left, int32_t right,
                   int32_t sub, int32_t offset, int32_t edge, float* out);

/* Parity probe.  Replaces `Model.encoder(X)` (model.py:206-209,444-490): X [n,t,d] ->
 * enc_out [n,t,hidden_units], pad_out (optional) [n,t] int32 (1 = padded frame). */
int asr_encode(asr_handle* h, const float* X, int32_t n, int32_t t, int32_t d,
               float* enc_out, int32_t* pad_out);

/* Parity probe: asr_decode that also returns, for the first `trace_steps` decode steps,
 * the pre-softmax logits [trace_steps, n*beam, vocab] and the pruned beam scores / tokens /
 * parents [trace_steps, n*beam] (model.py:277-301).  Any trace pointer may be NULL. */
int asr_decode_trace(asr_handle* h, const float* X, int32_t n, int32_t t, int32_t d,
                     int32_t* Y, int64_t y_capacity, int32_t* len_out, int32_t trace_steps,
                     float* logits, float* scores, int32_t* tokens, int32_t* parents);

/* ---- single-kernel probes (host pointers; for the per-kernel parity tests) -------------- */

/* utils.dense (utils.py:546-557): y[m,n] = x[m,k] . w[n,k]^T + bias[n] (bias may be NULL).
 * split_k >= 1 exercises the split-K path (partials reduced in fixed order). */
int asr_op_dense(asr_handle* h, const float* x, const float* w, const float* bias,
                 int32_t m, int32_t n, int32_t k, int32_t split_k, float* y);
/* ff_hidden's first dense with GLU (utils.py:559-561; model.py:441):
 * y[m,n] = (x.w1^T + b1) * sigmoid(x.w2^T + b2). */
int asr_op_dense_glu(asr_handle* h, const float* x, const float* w1, const float* b1,
                     const float* w2, const float* b2, int32_t m, int32_t n, int32_t k, float* y);
/* utils.residual (utils.py:494-507; common_layers.py:333-338):
 * y = layer_norm(x + delta + dbias, eps 1e-6) * scale + bias; dbias may be NULL. */
int asr_op_add_layer_norm(asr_handle* h, const float* x, const float* delta, const float* dbias,
                          const float* scale, const float* bias, int32_t rows, int32_t hidden,
                          float* y);
/* `embedding(...)` of the tokens fed at decode step `position` + that position's timing signal
 * (model.py:569-576 with utils.embedding utils.py:509-526 and add_timing_signal_1d): y[r] = emb[tokens[r]] *
 * (scale ? sqrt(hidden) : 1) + signal[position].  emb [vocab, hidden], hidden = the handle's hidden_units; runs the
 * decode step's own kernel. */
int asr_op_embedding(asr_handle* h, const float* emb, const int32_t* tokens, int32_t rows, int32_t vocab,
                     int32_t hidden, int32_t position, int32_t scale, float* y);
/* The layout step between the encoder-side K/V projection and the decode-time cross attention (no counterpart of its own
 * in the reference: split_heads of utils.py:618-619 done once per utterance instead of once per step): kv [n*t, 2*heads*64]
 * (k | v) -> k, v [n][heads][t][64], float32 or (as_bf16) bf16 rounded to nearest even (2-byte elements). */
int asr_op_pack_cross_kv(asr_handle* h, const float* kv, int32_t n, int32_t t, int32_t heads, int32_t as_bf16,
                         void* k, void* v);
/* tf.contrib.layers.layer_norm over (t,hidden) per utterance + add_timing_signal_1d
 * (model.py:453,456): x [n,t,hidden] -> y. */
int asr_op_input_norm_timing(asr_handle* h, const float* x, const float* gamma, const float* beta,
                             int32_t n, int32_t t, int32_t hidden, float* y);
/* Allocation-free variant of asr_op_input_norm_timing.  NO COUNTERPART IN THE REFERENCE: it exists for callers that
 * cannot allocate device memory (tight serving loops, harnesses that forbid hipMalloc).  The call itself launches on the
 * handle's private stream and waits for it, so it cannot be recorded into a HIP graph by a caller's stream capture.
 * x_dev / gamma_dev / beta_dev / y_dev are DEVICE pointers on the handle's GPU, 16-byte aligned (x, y [n,t,hidden];
 * gamma, beta [hidden]); the per-utterance partial sums that the other entry point keeps in a hipMalloc'ed buffer live
 * in a static device array of the library (1 MiB per GPU, held until the process ends).  Calls no hipMalloc / hipFree
 * and no allocating routine.  Limits: n <= 1024 utterances per call and t <= the positions of the handle's timing
 * table (max(512, max_target_length + 2), or more once another entry point has grown it); beyond either the call
 * fails with ASR_ERR_CAPACITY and touches nothing.  All calls on one GPU share that GPU's array (the last call's
 * contents are overwritten, nothing needs clearing): calls on one GPU from several host threads - handles of one device in
 * an EnginePool - take turns on a per-GPU mutex of the library, held from the launch to the end of the wait; a caller that
 * launched its own kernels on that array would race.  Blocks until y_dev is valid.  Results are bit-identical to
 * asr_op_input_norm_timing. */
int asr_op_input_norm_timing_noalloc(asr_handle* h, const float* x_dev, const float* gamma_dev, const float* beta_dev,
                                     int32_t n, int32_t t, int32_t hidden, float* y_dev);
/* Decode-time encoder-decoder attention core of one decoder block at one step (model.py:604-616 -> utils.py:612-631,
 * common_attention.py:202 for the padding bias): q [n*beams, heads*64] = the query projection BEFORE its bias (qbias
 * [heads*64] is added, then the 1/sqrt(64) scale of utils.py:628), kv [n*t, 2*heads*64] = the encoder-side (k | v)
 * projection WITH its bias, pad [n*t] int32 (1 = padded position: -1e9 on its score) -> ctx [n*beams, heads*64]
 * (combine_heads applied).  Runs the decode step's own kernels (pack_cross_kv + the cross-attention kernel) in the variant
 * named: 0 = exact-float32 products, 1 = three fp16 products per multiply (the default mode's), 2 = K / V stored as
 * bf16 (`precision: bf16`).  *range_flag (may be NULL) receives 1 when variant 1 met an operand outside fp16's range
 * (the engine then repeats the call on variant 0; ctx is unspecified). */
int asr_op_decoder_cross_attention(asr_handle* h, const float* q, const float* qbias, const float* kv, const int32_t* pad,
                                   int32_t n, int32_t t, int32_t heads, int32_t beams, int32_t variant, float* ctx,
                                   int32_t* range_flag);
/* Decode-time self-attention core of one decoder block at step `pos` (model.py:588-600 -> utils.py:612-631 with
 * reserve_last, no bias): every beam attends over ITS OWN history and the new position.  qkv [n*beams, 3*heads*64] = the
 * (q | k | v) projection of the new position BEFORE its bias (bias [3*heads*64] is added; q then scaled by 1/sqrt(64));
 * hist_k / hist_v [n][pos][beams][heads*64] = the K / V rows written at earlier steps, by (position, beam slot);
 * anc [n][beams][pos] = the slot of beam b's ancestor at position j (what the reference gets by gathering its cache with
 * the surviving beams' indices, model.py:296-301; here rows never move).  -> ctx [n*beams, heads*64].  Builds the cache,
 * the ancestry table and the list of live (position, slot) rows as the engine holds them and runs the decode step's own
 * kernel; kv_bf16 = 1: the `precision: bf16` variant (cache rows and the new rows rounded to bf16), 2: float32 caches with the
 * products as three fp16 matrix products per multiply (what `f32_f16x3` runs; operands must fit fp16's range), 0: float32 caches
 * on the exact-f32 matrix instruction. pos may be 0. */
int asr_op_decoder_self_attention(asr_handle* h, const float* qkv, const float* bias, const float* hist_k, const float* hist_v,
                                  const int32_t* anc, int32_t n, int32_t pos, int32_t heads, int32_t beams, int32_t kv_bf16,
                                  float* ctx);
/* Encoder self-attention core (utils.py:624-631; common_attention.py:295-334): qkv [n,t,3*hidden]
 * (q | k | v, q NOT yet scaled), pad [n,t] int32 -> ctx [n,t,hidden]. */
int asr_op_encoder_attention(asr_handle* h, const float* qkv, const int32_t* pad,
                             int32_t n, int32_t t, float* ctx);
/* model.test_output (model.py:345-347): logits [rows,vocab] -> top-k of log_softmax,
 * ordered by (value desc, index asc): values [rows,k], indices [rows,k]. */
int asr_op_log_softmax_topk(asr_handle* h, const float* logits, int32_t rows, int32_t vocab,
                            int32_t k, float* values, int32_t* indices);
/* One pruning step (model.py:280-301): scores [b*k], finished [b*k] (0/1), cand_scores /
 * cand_tokens [b*k, k] -> new_scores, tokens, parents (global row index), finished_out. */
int asr_op_beam_step(asr_handle* h, const float* scores, const int32_t* finished,
                     const float* cand_scores, const int32_t* cand_tokens, int32_t b, int32_t k,
                     float* new_scores, int32_t* tokens, int32_t* parents, int32_t* finished_out);

/* ---- instrumentation ---------------------------------------------------------------------- */

/* mode 0: normal.  mode 1: bracket every kernel launch of subsequent calls with HIP events on the
 * stream it is launched on and accumulate per-kernel-class time; the host waits for every bracket, so each
 * kernel runs alone (its solo duration).  mode 2: the same brackets left in the queues and read back
 * afterwards: the durations kernels have in the running pipeline, with the other sub-batches' kernels beside
 * them (what rocprofv3 --kernel-trace sees; bench.py's roofline leg).  For measurement and nothing else. */
int asr_set_profile(asr_handle* h, int32_t mode);
/* Returns up to `cap` classes accumulated since the last asr_set_profile(h, 1 or 2):
 * names[i] (static strings), total milliseconds, launch counts, algorithmic FLOPs and bytes.
 * asr_set_profile times an EMPTY event pair (median of 33, issued the way the mode issues its brackets) and that figure is subtracted from every
 * bracket; the total subtracted is returned as the last class, named "(event_pair_overhead)". */
int asr_get_profile(asr_handle* h, int32_t cap, const char** names, double* ms, int64_t* launches,
                    double* flops, double* bytes, int32_t* n_out);
/* Device-memory footprint of the handle in bytes (caches + workspaces + the weight store if this handle filled it). */
int64_t asr_device_bytes(const asr_handle* h);

/* ASR_PRECISION_F32_F16X3 only: how many calls on this handle were repeated on the F32_BF16X6 kernels because an operand
 * did not fit fp16 (diagnostics; the results of such calls are those of F32_BF16X6).  No counterpart in the reference. */
int64_t asr_range_retries(const asr_handle* h);

/* Host utility, no GPU involved: crc32c (Castagnoli) continued from `crc` (0 to start).  The checksum TensorFlow
 * stores per tensor and per index block in the V2 checkpoints that `saver.restore` (evaluate.py:36) reads and
 * avg_checkpoints.py:107 writes; asr_amd/tf_bundle.py uses it to verify / write such files. */
uint32_t asr_crc32c(const void* data, int64_t n, uint32_t crc);

#ifdef __cplusplus
}
#endif
#endif /* ASR_AMD_H */
