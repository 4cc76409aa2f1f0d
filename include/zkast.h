/* zkast.h — C ABI of libzkast.so: MI355X-native (gfx950) sliding-window log-mel + AST two-stage inference.
 *
 * This is the drop-in boundary for ONE path of daostler-tum/zenker-audio-detection: what
 * `forward_probs` (src/test_long_audio_windows_2stage.py:104-113) does per batch —
 *     fx(batch, sampling_rate=16000, return_tensors="pt")     -> zk_logmel / zk_features_expand
 *     model(feats).logits                                      -> zk_ast_forward
 *     torch.softmax(logits, dim=1)                             -> zk_softmax
 * — and the two-stage cascade around it (:301-340)              -> zk_two_stage.
 * The reference has no FFI of its own (pure Python); the binding a maintainer adds is the ctypes layer in
 * zenker-audio-detection_amd/zkast/lib.py.
 *
 * Conventions
 *  - plain C, no torch types.  Every pointer argument documented "host|device" may be either: the library asks the
 *    HIP runtime (hipPointerGetAttributes) and stages host memory itself.
 *  - the caller allocates every output; the library returns no owned memory except zk_last_error()'s string.
 *  - return value 0 = ok, negative = error (ZK_E_*); no exceptions or aborts cross the boundary.
 *  - a context is bound to one GPU and one HIP stream and is NOT thread-safe; calls are synchronous on return
 *    unless the context was switched to async mode with zk_set_async(ctx, 1) (then call zk_synchronize()).
 *    The context's own stream is NON-BLOCKING: it does not wait for work queued on the NULL stream or on another
 *    library's stream.  Device buffers handed in must be complete (synchronize their producer), or the producer's
 *    stream must be given to the context with zk_set_stream().
 */
#ifndef ZKAST_H
#define ZKAST_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zk_ctx zk_ctx;

enum { ZK_OK = 0, ZK_E_ARG = -1, ZK_E_HIP = -2, ZK_E_STATE = -3, ZK_E_SHAPE = -4, ZK_E_NOMEM = -5 };

/* compute modes of the transformer GEMMs (all accumulate in fp32; LayerNorm/softmax/residual stream are fp32) */
enum {
  ZK_F16 = 1,   /* one fp16 MFMA pass.  Fastest; ~3e-3 max-abs logit error on the synthetic weight sets.        */
  ZK_F16C8 = 2, /* fp16 pass + ONE fp8 (e4m3) pass that carries both split-correction products (they are 2^-11 of
                   the result, so 4 significant bits suffice): fp32-grade products at 2 matrix-pipe passes.  1e-4 on
                   ordinary weight sets; on an input-sensitive model it reaches 8.6e-4 ... 1.07e-3 over the 3 599
                   windows of a 30-min recording, i.e. it sits AT the tolerance there: use ZK_F16MIX              */
  ZK_F16X3 = 3, /* (hi,lo) fp16 operand pairs, 3 MFMA passes: fp32-equivalent products; meets the 1e-3 tolerance */
  ZK_F16MIX = 4 /* ZK_F16X3 or ZK_F16C8 per encoder layer and kernel group (layers only exchange the fp32 residual stream,
                   producers write the plane format their consumer reads).  The default assignment is the cheapest one
                   measured to keep >= 20 % of the 1e-3 tolerance on a 3 599-window recording of the input-sensitive
                   weight set; zk_model_set_layer_modes sets another                                    */
};

/* tensor dtypes accepted by zk_model_load */
enum { ZK_DT_F32 = 0, ZK_DT_F16 = 1, ZK_DT_BF16 = 2 };

typedef struct {
  const char* name;    /* state-dict key, transformers 4.x or 5.x scheme                                          */
  const void* data;    /* HOST pointer, C-contiguous; the library copies, the caller keeps ownership             */
  int32_t ndim;
  int64_t shape[4];
  int32_t dtype;       /* ZK_DT_*                                                                                 */
} zk_tensor_desc;

/* mirrors config.json of the model directory (ASTConfig, $TF/.../configuration_audio_spectrogram_transformer.py:50-64) */
typedef struct {
  int32_t hidden_size, num_hidden_layers, num_attention_heads, intermediate_size;
  int32_t patch_size, frequency_stride, time_stride, max_length, num_mel_bins, num_labels;
  float layer_norm_eps;
} zk_ast_config;

/* ---- context ------------------------------------------------------------------------------------------------ */
/* replaces the module-level DEVICE of the reference (src/test_long_audio_windows_2stage.py:48) */
int zk_create(int device_id, zk_ctx** out);
void zk_destroy(zk_ctx* ctx);
const char* zk_last_error(zk_ctx* ctx);           /* ctx may be NULL: last error of zk_create                     */
int zk_set_stream(zk_ctx* ctx, void* hip_stream); /* hipStream_t; NULL = the context's own stream                  */
int zk_set_async(zk_ctx* ctx, int enable);
int zk_synchronize(zk_ctx* ctx);
/* exact last-layer pruning: only tokens 0/1 feed the head (ASTModel.forward:304), so the last layer's attention
 * queries, O projection and MLP run on those two rows per window only.  Default on; results are unchanged.       */
int zk_set_prune_last_layer(zk_ctx* ctx, int enable);
/* exact layer-0 constant-row reuse: a 1 s window fills 98 of the extractor's 1024 frames (feature_extraction_...py:143-151),
 * so 1094 of its 1214 tokens (cls, distillation, the padding-only patches) enter layer 0 with values that do not depend
 * on the window; their embedding rows and layer-0 LayerNorm + q|k|v rows are computed once per model and copied.  Applies
 * to forwards from the feature slot (zk_logmel / zk_features_set / zk_two_stage), never to caller-provided
 * input_values.  Default on; results are bit-identical.                                                            */
int zk_set_layer0_reuse(zk_ctx* ctx, int enable);
/* ... and their ATTENTION state: the scores of the constant queries against the constant keys do not depend on the window
 * either, so the running softmax state (max, sum, unnormalised output) of every constant query over the first 1024 constant
 * keys is tabulated once per model; per window the constant queries only add the remaining 190 keys, the 120 real queries
 * see all 1214 keys in the order [constant | real] (0.27 M instead of 1.47 M query-key pairs per window and head in layer 0,
 * eager_attention_forward, modeling_audio_spectrogram_transformer.py:102-127).  The algorithm is exact; the keys are summed in
 * another order than the plain kernel's, so logits agree with it to rounding (~1e-6), not bit for bit.  Only together with
 * zk_set_layer0_reuse, for 51..100-frame windows, in the split compute modes.  Default on.                            */
int zk_set_layer0_attention(zk_ctx* ctx, int enable);
int zk_set_micro_batch(zk_ctx* ctx, int32_t windows); /* forward is chunked into micro-batches; 0 = auto (default)    */
const char* zk_version(void);

/* ---- model -------------------------------------------------------------------------------------------------- */
/* replaces load_stage_model (src/test_long_audio_windows_2stage.py:86-98): weights of stage 0 (Idle/Swallow) or
 * stage 1 (Healthy/Zenker) plus that stage's extractor statistics (preprocessor_config.json: mean, std).          */
int zk_model_load(zk_ctx* ctx, int stage, const zk_tensor_desc* tensors, int32_t n_tensors, const zk_ast_config* cfg,
                  float fx_mean, float fx_std, int32_t compute_mode);
int zk_model_set_compute_mode(zk_ctx* ctx, int stage, int32_t compute_mode);
/* one compute mode (ZK_F16 / ZK_F16C8 / ZK_F16X3) per encoder layer (n = num_hidden_layers entries), or per kernel group of
 * each layer (n = 4 * num_hidden_layers: [layer][QKV GEMM, QK^T of attention, O projection, MLP = FC1 + FC2]); the model's
 * mode becomes ZK_F16MIX.  ZK_F16C8 for QK^T needs the ZK_F16C8 QKV GEMM of that layer (its epilogue writes k's c8 plane).
 * The reference has one dtype per model (model.to(DEVICE), src/test_long_audio_windows_2stage.py:96); this is the knob
 * that trades matrix-pipe passes against the logit tolerance where the error is made.                              */
int zk_model_set_layer_modes(zk_ctx* ctx, int stage, const int32_t* modes, int32_t n);
/* the extractor statistics used when a forward runs from the feature slot (fx.mean / fx.std of that stage) */
int zk_model_set_fx(zk_ctx* ctx, int stage, float fx_mean, float fx_std);

/* ---- feature extraction ------------------------------------------------------------------------------------- */
/* replaces window_audio (:62-75) + ASTFeatureExtractor._extract_fbank_features.  Windows are
 * audio[first_start + i*hop : ... + win], i < n_windows; samples past n_samples read as 0.  The un-normalised
 * log-mel rows stay on the device in the context (the "feature slot"); n_frames = 1 + (win-400)/160, capped at 1024. */
int zk_logmel(zk_ctx* ctx, const float* audio /*host|device; NULL = the audio slot*/, int64_t n_samples, int64_t first_start, int64_t hop,
              int32_t win, int32_t n_windows);
/* replaces ASTFeatureExtractor.__call__(...)["input_values"]: zero-pad the slot to 1024 rows and apply
 * (x - mean) / (2*std) when do_normalize != 0.  out: (n_windows, 1024, 128) fp32, host|device.                     */
int zk_features_expand(zk_ctx* ctx, float mean, float std, int32_t do_normalize, float* out /*host|device*/);
/* copy the compact slot out: (n_windows, n_frames, 128) fp32 */
int zk_features_get(zk_ctx* ctx, float* out /*host|device*/, int32_t* n_windows, int32_t* n_frames);

/* the inverse: put compact un-normalised log-mel features (n_windows, n_frames, 128) fp32 host|device into the slot, as
 * if zk_logmel had just produced them.  Replaces re-running the extractor when a feature cache exists
 * (src/test_long_audio_windows_2stage_cache.py:163-168 loads a (N,1024,128) bundle; the compact store keeps only the
 * n_frames real rows, and each stage applies its own mean/std when it reads the slot).                              */
int zk_features_set(zk_ctx* ctx, const float* feats /*host|device*/, int32_t n_windows, int32_t n_frames);

/* ---- transformer -------------------------------------------------------------------------------------------- */
/* replaces ASTForAudioClassification.forward: input_values (B,1024,128) fp32 host|device -> logits (B,labels).
 * input_values == NULL: run on the feature slot (normalised with the stage's mean/std); win_idx (host|device,
 * may be NULL) then selects/gathers B windows of the slot.                                                        */
int zk_ast_forward(zk_ctx* ctx, int stage, const float* input_values, const int32_t* win_idx, int32_t B,
                   float* logits /*host|device*/);
/* torch.softmax(logits, dim=1) */
int zk_softmax(zk_ctx* ctx, const float* logits /*host|device*/, int32_t n, int32_t num_labels,
               float* probs /*host|device*/);

/* ---- cascade ------------------------------------------------------------------------------------------------ */
/* replaces main():301-340 for one recording: log-mel -> stage-1 -> gate -> stage-2 on the gated windows.
 * Gate: argmax == 1 and p_swallow >= thr1 (and p_swallow >= fwd_min_prob when fwd_min_prob >= 0,
 * ..._cache.py:471-478).  Outputs (host|device, caller-sized): s1_logits (N,2), swallow_idx (N), n_swallow (1),
 * s2_logits (N,2) of which the first *n_swallow rows are valid.                                                   */
/* audio == NULL: run on the audio slot (zk_audio_load); n_samples is then ignored.                                  */
int zk_two_stage(zk_ctx* ctx, const float* audio, int64_t n_samples, int64_t first_start, int64_t hop, int32_t win,
                 int32_t n_windows, float thr1, float fwd_min_prob, float* s1_logits, int32_t* swallow_idx,
                 int32_t* n_swallow, float* s2_logits);
/* the gate alone, on device: logits (N,2) -> probs (N,2, may be NULL), ascending indices, count */
int zk_gate(zk_ctx* ctx, const float* logits, int32_t n, float thr1, float fwd_min_prob, float* probs,
            int32_t* swallow_idx, int32_t* n_swallow);

/* ---- multi-GPU: one process per GPU, windows / patients are sharded by the host code (zkast/dist.py,
 * zkast/batch.py); the ONLY device exchange is the all-gather of per-window logits after each stage.  The reference is
 * single-process (src/run_batch_simple_2stage.py:258-292 walks the patient list in one interpreter, the per-patient
 * script owns one DEVICE, src/test_long_audio_windows_2stage.py:48), so these entry points replace nothing — they are
 * what lets that loop and forward_probs (:104-113) run on 8 GPUs.  RCCL (librccl.so.1) is dlopen()ed on first use.   */
/* rank 0: create the 128-byte RCCL unique id; the caller ships it to the other ranks (any host channel)            */
int zk_comm_unique_id(void* out_128_bytes);
/* collective over all ranks: bind an RCCL communicator (over xGMI inside a node) to the context.  world == 1 with
 * unique_id == NULL needs no RCCL and makes every gather below a copy.                                              */
int zk_comm_init(zk_ctx* ctx, int32_t rank, int32_t world, const void* unique_id_128_bytes);
int zk_comm_destroy(zk_ctx* ctx);
int zk_comm_info(zk_ctx* ctx, int32_t* rank, int32_t* world);
/* all-gather of equally sized logit shards on the context's stream: local (rows_per_rank, cols) fp32 host|device ->
 * all (world, rows_per_rank, cols) host|device, rank-major.  Ragged shards are padded by the caller (dist.py).       */
int zk_allgather_logits(zk_ctx* ctx, const float* local, int32_t rows_per_rank, int32_t cols, float* all);
/* the same for opaque bytes (per-patient summary records of the batch driver, barrier / max-over-ranks of a timing)   */
int zk_comm_allgather_bytes(zk_ctx* ctx, const void* send, int64_t bytes, void* recv);

/* ---- load_audio (next row) ---------------------------------------------------------------------- */
/* torchaudio.functional.resample(wav, orig, new) defaults (sinc_interp_hann, width 6, rolloff 0.99);
 * out has ceil(new*n_in/orig) samples.  PARITY UNPINNED (torchaudio source absent).                               */
int zk_resample(zk_ctx* ctx, const float* in /*host|device*/, int64_t n_in, int32_t orig_sr, int32_t new_sr,
                float* out /*host|device*/, int64_t n_out);

/* torchaudio.load + wav.mean(dim=0) (src/test_long_audio_windows_2stage.py:54-56) for the sample data of a RIFF/WAVE
 * "data" chunk (the header is parsed by the caller): interleaved little-endian samples -> mono float32, n_frames =
 * n_bytes / (channels * bits/8).  format_tag 1 = integer PCM (bits 8 / 16 / 24 / 32), 3 = IEEE float (bits 32 / 64);
 * scaling x / 2^(bits-1) (8-bit: (x - 128) / 128).  out holds n_frames floats.                                     */
int zk_wav_decode(zk_ctx* ctx, const void* data /*host|device*/, int64_t n_bytes, int32_t format_tag, int32_t bits,
                  int32_t channels, float* out /*host|device*/);

/* load_audio as ONE call that stays on the device (src/test_long_audio_windows_2stage.py:53-59): one upload of the
 * data chunk's bytes, zk_wav_decode, zk_resample to target_sr when sr differs; the mono float32 recording is kept in
 * the context ("audio slot") for zk_logmel / zk_two_stage(audio = NULL).  n_samples_out may be NULL.                */
int zk_audio_load(zk_ctx* ctx, const void* data /*host|device*/, int64_t n_bytes, int32_t format_tag, int32_t bits,
                  int32_t channels, int32_t sr, int32_t target_sr, int64_t* n_samples_out);
/* length of the audio slot and, when out != NULL, a copy of it (the np.float32 (T,) that load_audio returns)         */
int zk_audio_get(zk_ctx* ctx, float* out /*host|device*/, int64_t* n_samples);

/* ---- introspection / measurement ---------------------------------------------------------------------------- */
/* per-kernel-class HIP-event timing over the calls made since zk_prof_begin (on the context's stream).
 * zk_prof_get: name in {"gemm_qkv","gemm_o","gemm_fc1","gemm_fc2","gemm_patch","attention","layernorm","logmel",
 * "embed","head","wav_decode","resample","allgather"} -> accumulated milliseconds and launch count
 * ("allgather": the RCCL collective of zk_allgather_logits alone, without its staging copies — it includes the wait for the
 * slowest peer; the byte gathers are not counted;
 * inside a ZK_F16MIX model the layers that run ZK_F16X3 are other kernels and count apart: "gemm_qkv_x3","gemm_o_x3",
 * "gemm_fc1_x3","gemm_fc2_x3","attention_x3"). */
int zk_prof_begin(zk_ctx* ctx);
int zk_prof_end(zk_ctx* ctx);
int zk_prof_get(zk_ctx* ctx, const char* name, double* ms, int64_t* launches);
/* FLOPs the launches of that class EXECUTED algorithmically (2*M*N*K per GEMM launch; 4*Sq*S*64 per head for attention),
 * i.e. with the exact last-layer pruning taken into account and WITHOUT counting the 3 MFMA passes of ZK_F16X3 */
int zk_prof_get_flops(zk_ctx* ctx, const char* name, double* flops);
/* debug tap used by the parity tests: keep the fp32 residual stream of the FIRST micro-batch after encoder layer
 * `layer` (-1 = embeddings output, -2 = off) of the next forward; get copies (n_windows, 1214, 768) to host.        */
int zk_debug_set_tap(zk_ctx* ctx, int32_t layer);
int zk_debug_get_tap(zk_ctx* ctx, float* out /*host*/, int32_t n_windows);

/* ---- test hooks: run ONE kernel on caller-provided fp32 HOST data (tests/test_kernels_gpu.py) ------------------ */
/* Flags ORed into `epi` (zk_test_gemm) or `nsplit` (zk_test_layernorm, zk_test_attention), ZK_F16C8 only: the kernel reads
 * its x planes (TILED_IN, GEMM) / writes its output planes (TILED_OUT: LayerNorm, attention, the GEMM's GELU epilogue)
 * in the k-slice-major tile layout the forward uses between its GEMM-side kernels; results must equal the row-major
 * ones bit for bit.                                                                                                 */
#define ZK_TEST_TILED_IN 0x100
#define ZK_TEST_TILED_OUT 0x200
/* zk_test_gemm: the rows M .. ceil(M/256)*256 of the hook's x planes hold NaN patterns instead of zeros (the ZK_F16C8
 * kernel reads its last row block whole; what lies behind row M must never reach a result)                           */
#define ZK_TEST_POISON_PAD 0x400
/* zk_test_gemm: tell the ZK_F16C8 launcher that the x planes are allocated for exactly M rows; it must refuse the launch
 * (ZK_E_SHAPE) unless M is a multiple of 256                                                                         */
#define ZK_TEST_SHORT_X 0x800
/* LayerNorm(768): x (rows,768) -> out (rows,768) = hi (+ lo when nsplit is 3 or 2) of the output planes            */
int zk_test_layernorm(zk_ctx* ctx, const float* x, const float* gamma, const float* beta, int32_t rows, float eps,
                      int32_t nsplit, float* out);
/* out = x[M,K] . w[N,K]^T + bias with epilogue epi (0 store, 1 gelu, 2 residual += (out is in/out), 3 patch-embed:
 * M % 1212 == 0, N == 768, pos (1214,768), out ((M/1212)*1214, 768) in/out); N % 256 == 0, K % 64 == 0          */
int zk_test_gemm(zk_ctx* ctx, const float* x, const float* w, const float* bias, int32_t M, int32_t N, int32_t K,
                 int32_t epi, int32_t nsplit, const float* pos, float* out);
/* qkv (W*1214, 2304) -> out (W*1214, 768): softmax(q k^T / 8) v per head                                            */
int zk_test_attention(zk_ctx* ctx, const float* qkv, int32_t W, int32_t nsplit, float* out);
/* the c8 plane of ZK_F16C8 as raw 16-bit entries: activations (is_weight 0): byte 0 = e4m3((x - fp16(x)) * 2^11),
 * byte 1 = e4m3(x); weights: byte 0 = e4m3(w * 2^w_exp), byte 1 = e4m3((w - fp16(w)) * 2^(w_exp + 11)); OCP e4m3,
 * round to nearest even, operands clamped to +-448                                                                  */
int zk_test_split_c8(zk_ctx* ctx, const float* x, int64_t n, int32_t w_exp, int32_t is_weight, uint16_t* out);

#ifdef __cplusplus
}
#endif
#endif
