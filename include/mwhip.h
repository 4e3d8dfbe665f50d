/* mwhip.h — C ABI of the MI355X (gfx950) Whisper transcription engine.
 *
 * The reference has no FFI: its boundary is three Python calls into the
 * un-vendored `whisperx` package (reference transcribe.py:107-113 load_model,
 * :117 load_audio, :123 model.transcribe).  This header is what a binding for
 * that boundary talks to; each entry point names the call it stands behind.
 * Audio decoding (load_audio) stays on the host and has no entry point here.
 *
 * Conventions
 *  - every function returns 0 on success or a negative mw_status; the message
 *    of the last failure is available from mw_last_error().  Nothing throws or
 *    aborts across this boundary.
 *  - the caller owns every buffer it passes; the context owns all device
 *    memory (weights, caches, workspaces, captured graph).
 *  - a data pointer may be a host pointer or a device pointer of the
 *    context's GPU; the library asks the HIP runtime which one it is.
 *    Device buffers are read and written on the context's own (non-blocking)
 *    stream, which is not ordered against any stream of the caller: an input
 *    buffer must be complete before the call, and every call returns only
 *    after its outputs are complete.
 *  - a context is bound to one GPU and one HIP stream and is not thread-safe;
 *    separate contexts are independent.  Multi-GPU = one process per GPU.
 *  - audio is float32 mono 16 kHz in [-1, 1); a chunk is 480000 samples (30 s).
 */
#ifndef MWHIP_H
#define MWHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mw_ctx mw_ctx;

typedef enum {
    MW_OK = 0,
    MW_ERR_ARG = -1,        /* bad argument (null, out of range, wrong shape)   */
    MW_ERR_STATE = -2,      /* call order violated (e.g. decode before encode) */
    MW_ERR_HIP = -3,        /* a HIP runtime call failed                        */
    MW_ERR_NAME = -4,       /* unknown or duplicate weight name                 */
    MW_ERR_MISSING = -5,    /* mw_finalize: a required weight was never set     */
    MW_ERR_NOMEM = -6
} mw_status;

typedef enum { MW_F32 = 0, MW_F16 = 1, MW_BF16 = 2 } mw_dtype;

/* How the engine keeps the Linear weights of the encoder and decoder layers in HBM (what the
 * compute_type argument of whisperx.load_model, reference transcribe.py:32,109, selects there).
 *  MW_WEIGHTS_F16: fp16, fp32 accumulation everywhere.
 *  MW_WEIGHTS_FP8: OCP e4m3 codes with one fp32 scale per output row (scale = max|w| / 448,
 *    round to nearest even), quantised by mw_set_weight from whatever dtype arrives.  Encoder
 *    products fed by a LayerNorm (QKV, MLP in) and the cross K/V projection run on the gfx950 fp8
 *    MFMA with their input rows quantised the same way (one scale per row); every other product
 *    multiplies fp16 activations with the exact fp16 image of the codes.  Convolutions,
 *    embeddings, LayerNorm parameters and biases stay fp16/fp32. */
typedef enum { MW_WEIGHTS_F16 = 0, MW_WEIGHTS_FP8 = 1 } mw_weight_format;

/* How the per-layer cross-attention K/V of the encoder output is cached for the decoder (the largest stream of
 * a decode step: 245.8 MB per sequence per step at large-v3 in fp16).  No counterpart in the reference.
 *  MW_KV_F16: fp16.
 *  MW_KV_FP8: OCP e4m3 codes with one fp32 scale per (sequence, head, frame) row of 64 values (scale =
 *    max|.| / 448, round to nearest even), written by the projection's epilogue; 130.6 MB per sequence per
 *    step.  Only together with MW_WEIGHTS_FP8. */
typedef enum { MW_KV_F16 = 0, MW_KV_FP8 = 1 } mw_kv_format;

/* Architecture of the checkpoint (what `MODEL_SIZE`, reference transcribe.py:33,
 * resolves to inside whisperx.load_model). head_dim must be 64. */
typedef struct {
    int32_t n_mels;        /* 80 or 128 */
    int32_t d_model;
    int32_t n_heads;
    int32_t enc_layers;
    int32_t dec_layers;
    int32_t ffn_dim;
    int32_t vocab;
    int32_t n_audio_ctx;   /* 1500 */
    int32_t n_text_ctx;    /* 448  */
    int32_t weight_format; /* mw_weight_format; fp8 needs d_model and ffn_dim to be multiples of 128 */
    int32_t kv_format;     /* mw_kv_format */
} mw_model_desc;

/* Decoding rules; the asr_options / language arguments of whisperx.load_model
 * (reference transcribe.py:107-113) resolve to these. Arrays are copied. */
typedef struct {
    const int32_t* prompt;          int32_t n_prompt;          /* forced ids, >= 1            */
    const int32_t* suppress;        int32_t n_suppress;        /* masked at every sampled step */
    const int32_t* begin_suppress;  int32_t n_begin_suppress;  /* also masked at the first one */
    int32_t eot;                                               /* end-of-text id               */
    int32_t max_new;                                           /* sampled tokens per chunk     */
} mw_decode_opts;

/* ---- engine life cycle: whisperx.load_model (reference transcribe.py:107-113) ---- */

/* Create an engine on GPU `device` able to process up to `max_batch` (1..512) chunks per call.  Beyond 32
 * sequences the decode projections take all sequences in one pass over the weights (in blocks of 128 inside a
 * launch): every weight is streamed from HBM once per step however many sequences there are. */
int mw_create(const mw_model_desc* desc, int device, int max_batch, mw_ctx** out);

/* Hand over one tensor of a Hugging Face-layout Whisper state dict
 * ("model.encoder.conv1.weight", ...; "proj_out.weight" is accepted and ignored,
 * it is tied to the token embedding).  The data is copied/converted at once. */
int mw_set_weight(mw_ctx* ctx, const char* name, const void* data,
                  const int64_t* shape, int ndim, int dtype);

/* Check that every tensor arrived; builds derived tables. Must precede any compute. */
int mw_finalize(mw_ctx* ctx);

int mw_set_decode_opts(mw_ctx* ctx, const mw_decode_opts* opts);

/* A second engine on the same GPU that shares the (finalized) parent's weights and has its own
 * stream, workspaces, K/V caches and decode options (copied from the parent at this moment).
 * Parent and forks may be driven from different host threads at the same time: at batch 16 most
 * decode launches are latency-bound, so two or three engines in flight raise throughput.
 * Destroy every fork before its parent.  No counterpart in the reference (single engine). */
int mw_fork(mw_ctx* parent, mw_ctx** out);

/* mw_fork with a max_batch of its own (1..512): workspaces and caches are sized per context, so a set of
 * contexts can split roles, e.g. forks of 16 that run the encoder half of passes and forks of 64 that decode
 * what they encoded (mw_encode_into). */
int mw_fork_sized(mw_ctx* parent, int max_batch, mw_ctx** out);

/* mw_fork_sized with a cross-attention K/V cache for `cache_sequences` <= max_batch sequences only.  The hypotheses of
 * a recording (mw_decode_beam, mw_decode_sample) share the recording's cache, so an engine that decodes n recordings
 * with b hypotheses each needs max_batch = n * b but a cache of n: at large-v3 a cached sequence is 0.25 GB.  Encoder
 * passes, greedy and teacher-forced decoding on such a context take at most cache_sequences chunks.  Behind whisperx's
 * default decoding (beam 5; reference transcribe.py:107-113,123). */
int mw_fork_cached(mw_ctx* parent, int max_batch, int cache_sequences, mw_ctx** out);

/* Free everything the context owns (reference transcribe.py:153-154 `del model`). */
void mw_destroy(mw_ctx* ctx);

/* Message of the most recent failure on this context (or of mw_create when ctx is NULL). */
const char* mw_last_error(const mw_ctx* ctx);

/* ---- the hot path: model.transcribe (reference transcribe.py:123) ---- */

/* Stage 1: log-mel features of `batch` chunks. pcm: float32 [batch][480000].
 * mel_out (optional): float32 [batch][n_mels][3000]. The features also stay in the
 * context for mw_encode. */
int mw_logmel(mw_ctx* ctx, const float* pcm, int batch, float* mel_out);

/* mw_logmel for callers that cannot allocate device memory (tight serving loops, harnesses that forbid
 * hipMalloc): no counterpart in the reference.  It synchronises its own stream and reads its events before it
 * returns, so it cannot itself be captured into a HIP graph.  pcm float32 [batch][480000] and
 * mel_out float32 [batch][n_mels][3000] must both be device memory of the context's GPU (a host buffer would
 * need staging, which allocates) and both are required; the features also stay in the context for mw_encode.
 * The call allocates and frees nothing, on the device or through any library: the only memory its kernels use
 * beyond pcm, mel_out and the context's tables is a file-scope __device__ array in the library (one copy per
 * GPU, 128 chunks x 94 block maxima x 4 B = 47 KiB, held until the process ends).  batch beyond 128 is
 * refused (MW_ERR_ARG).  All contexts of one GPU share that array: calls of this entry point on one GPU take
 * turns (a per-GPU lock inside the library is held for the duration of the call, which blocks until its two
 * launches are done); from several processes on one GPU every process has its own copy.  Results are bit-identical
 * to mw_logmel. */
int mw_logmel_noalloc(mw_ctx* ctx, const float* pcm, int batch, float* mel_out);

/* Stage 2: encoder + per-layer cross-attention K/V for the decoder.
 * mel (optional): float32 [batch][n_mels][3000] to use instead of the features left by
 * mw_logmel. enc_out (optional): float32 [batch][n_audio_ctx][d_model]. */
int mw_encode(mw_ctx* ctx, const float* mel, int batch, float* enc_out);

/* Stages 1 and 2 of `batch` chunks on `ctx` (its stream and workspaces; pcm float32 [batch][480000]), with the
 * decoder's cross-attention K/V written into the cache of `dst` at sequences slot0 .. slot0 + batch - 1.
 * `dst` is `ctx` itself, its parent or another fork of that parent; slot0 + batch <= dst's max_batch.  The
 * call returns when the K/V are complete.  Several contexts may fill disjoint slots of one destination at the
 * same time (each from its own thread); the destination must not be decoding meanwhile.  With this the chunks
 * of model.transcribe (reference transcribe.py:123, batch_size chunks per encoder pass) can be decoded in
 * larger groups than they were encoded in: every decoder weight is streamed once per step for the whole group.
 * No counterpart in the reference. */
int mw_encode_into(mw_ctx* ctx, const float* pcm, int batch, mw_ctx* dst, int slot0);

/* Declare sequences 0 .. batch - 1 of the context's cross-attention cache encoded (by mw_encode_into calls that
 * have returned): mw_decode_greedy / mw_decode_beam / mw_decode_forced of `batch` sequences may follow. */
int mw_set_encoded(mw_ctx* ctx, int batch);

/* Test hook: what the cross-attention K/V projection left in the cache for one sequence of one decoder layer, as
 * float32 [n_heads][n_audio_ctx][64] each (the decoder reads exactly these rows at every step; an e4m3 cache is
 * returned as code * row scale).  `sequence` is among those encoded last (mw_encode, or mw_encode_into +
 * mw_set_encoded).  No counterpart in the reference's call surface: the tensors are those of the cross-attention
 * branch of Hugging Face's WhisperAttention (modeling_whisper.py:321-335: K = Wk enc, no bias; V = Wv enc + bv; head
 * split), which whisperx's backend keeps internal behind model.transcribe (reference transcribe.py:123). */
int mw_read_cross_kv(mw_ctx* ctx, int layer, int sequence, float* k_out, float* v_out);

/* Stage 3: greedy decoding of the chunks encoded last.
 * out_ids: int32 [batch][max_new] (padded with eot); out_len: int32 [batch] = tokens
 * before eot.  logits_out (optional): float32 [max_new][batch][vocab], the raw
 * (unmasked) logits every sampled token was chosen from; rows after every chunk has
 * finished are left untouched. */
int mw_decode_greedy(mw_ctx* ctx, int batch, int32_t* out_ids, int32_t* out_len, float* logits_out);

/* Teacher-forced decoder pass over given ids (parity checks): tokens int32 [batch][n];
 * logits_out float32 [batch][n][vocab]. */
int mw_decode_forced(mw_ctx* ctx, int batch, const int32_t* tokens, int n, float* logits_out);

/* Beam search at temperature 0 over the encoder state of the last mw_encode(n_audio): the decoding
 * whisperx runs by default behind model.transcribe (the reference overrides only initial_prompt,
 * transcribe.py:111,123).  n_audio * beam sequences must fit max_batch; beam in 1..8; patience in
 * [1, 4] (finished hypotheses kept per recording = round(beam * patience)).  The winner maximises
 * cumulative log-probability per token.
 * out_ids int32 [n_audio][max_new] padded with eot; out_len int32 [n_audio]; out_avg_logprob
 * float32 [n_audio] or NULL (sum / (len + 1)).  Optional (NULL to skip): every ranked hypothesis,
 * hyp_ids int32 [n_audio][beam][max_new], hyp_len int32 [n_audio][beam] (-1 = none),
 * hyp_sum_logprob float32 [n_audio][beam]. */
int mw_decode_beam(mw_ctx* ctx, int n_audio, int beam, float patience, int32_t* out_ids, int32_t* out_len,
                   float* out_avg_logprob, int32_t* hyp_ids, int32_t* hyp_len, float* hyp_sum_logprob);

/* Sampling at a temperature over the encoder state of the last mw_encode(n_audio): the decodes of the temperature
 * fallback (asr_options temperatures / best_of / compression_ratio_threshold / log_prob_threshold of
 * whisperx.load_model, reference transcribe.py:107-113; the loop that walks the temperatures is host code, see
 * fallback.py).  best_of (1..8) independent samples per recording, n_audio * best_of <= max_batch; every sampled
 * token is drawn from softmax(logits / temperature) under the suppress rules of mw_set_decode_opts, as a Gumbel
 * arg-max over counter-based uniforms of (seed, sequence, position, token id): the same arguments give the same
 * ids.  temperature 0: the arg-max (greedy decoding with its log-probabilities).  A sample's score is the sum
 * of log-softmax(logits) of its tokens, end-of-text included; the recording's result is the sample with the highest
 * score per token.  active (optional, uint8 [n_audio], host memory): recordings with 0 are not decoded (length 0, no
 * cross-attention reads: a fallback pass spares the K/V stream of the windows that passed; their rows still ride
 * through the other launches of every step).
 * out_ids int32 [n_audio][max_new] padded with eot; out_len int32 [n_audio]; out_avg_logprob float32 [n_audio] or
 * NULL (score / (len + 1)).  Optional (NULL to skip): every sample, smp_ids int32 [n_audio][best_of][max_new],
 * smp_len int32 [n_audio][best_of], smp_sum_logprob float32 [n_audio][best_of]; logits_out float32
 * [max_new][n_audio*best_of][vocab] as in mw_decode_greedy (ended sequences are then not skipped). */
int mw_decode_sample(mw_ctx* ctx, int n_audio, int best_of, float temperature, uint32_t seed, const uint8_t* active,
                     int32_t* out_ids, int32_t* out_len, float* out_avg_logprob, int32_t* smp_ids, int32_t* smp_len,
                     float* smp_sum_logprob, float* logits_out);

/* The uniform in (0, 1) behind the draw of mw_decode_sample for sequence `sequence` (= recording * best_of + sample),
 * sampled position `position` (0 = the first token after the prompt) and token id `token`: host code, no GPU needed.
 * The token drawn at a position is argmax over the unsuppressed ids of logits[id] / temperature - log(-log(u(id))). */
float mw_sample_uniform(uint32_t seed, uint32_t sequence, uint32_t position, uint32_t token);

/* Stages 1-3 in one call: pcm float32 [batch][480000] -> ids. */
int mw_transcribe_batch(mw_ctx* ctx, const float* pcm, int batch, int32_t* out_ids, int32_t* out_len);

/* ---- the step before the path: speech-activity windows ---- */

/* Frame log-energies of mono 16 kHz PCM, the input of the window search that stands where whisperx
 * runs its VAD before cutting <= 30 s windows (the reference passes vad_options to it:
 * transcribe.py:43-46,112; the thresholds and the window merge live on the host side).
 *   out_db[f] = 10*log10( mean of x[f*hop + i]^2 over i < frame that exist + 1e-10 ),
 *   f < n_frames, and n_frames must equal ceil(n_samples / hop).
 * pcm: float32 [n_samples], host or device memory.  out_db: float32 [n_frames], host or device. */
int mw_frame_energy_db(mw_ctx* ctx, const float* pcm, int64_t n_samples, int frame, int hop, float* out_db, int64_t n_frames);

/* ---- measurement ---- */

/* GPU time in ms of the last call's stages, from HIP events on the context's stream:
 * [0] log-mel  [1] encoder  [2] cross K/V  [3] decode loop  [4] number of decode steps run.
 * (The reference prints time.time() deltas per stage: transcribe.py:114,124-125.) */
int mw_stage_ms(mw_ctx* ctx, float out[5]);

/* Re-run only the decode-time cross-attention kernel `iters` times back to back on the current
 * caches (cycling through the layers) and report its mean duration (ms, HIP events on the
 * context's stream) and the K/V bytes one launch must read. */
int mw_profile_cross_attention(mw_ctx* ctx, int batch, int iters, float* mean_ms, double* bytes_per_launch);

/* Re-run only the six weight-streaming projections of a decoder layer (QKV, attention out, cross-attention query
 * and out, MLP in and out; the launches of a real step without the two attention launches between them) for
 * `iters` layers back to back, cycling through the model's layers, and report the mean time per layer (ms, HIP
 * events on the context's stream, launch boundaries included) and the weight bytes of a layer's six matrices.
 * Leaves the decode state at position 0 of a fresh decode. */
int mw_profile_decode_gemvs(mw_ctx* ctx, int batch, int iters, float* mean_ms_per_layer, double* weight_bytes_per_layer);

/* Engine options: key "graph" (1 = capture the decode step into a hipGraph, default 1),
 * "check_every" (decode steps between checks for all-finished, default 4; a check is read one check later, so
 * that the stream never runs dry: a pass ends at most 2 * check_every - 1 steps after its last sequence),
 * "prefill" (1 = greedy decoding and beam search send all forced prompt positions through the decoder
 * in one pass instead of one step per position, default 1; the reference's initial_prompt, transcribe.py:40,111,
 * makes the prompt several dozen ids long),
 * "skip_done" (1 = the decode cross-attention does not read the K/V cache of sequences that have ended, or of
 * recordings whose beam search is complete: their next token is end-of-text whatever the step computes;
 * default 1),
 * "xattn_stream_from" (number of (sequence, head) pairs from which the decode cross-attention runs as one
 * workgroup per pair that walks all 1500 keys, instead of four key ranges per pair merged by the last to finish:
 * large decode groups fill the GPU without the ranges and their hand-off costs bandwidth; default 1024,
 * 0 = never),
 * "xattn_stream_per_cu" (0..8: resident workgroups per compute unit that walking form is held to; 0 = what its
 * registers allow, the default, and the fastest for an engine that has the GPU to itself.  With several engines on
 * one GPU set 1: the stream keeps 97 % of its rate alone, and the short memory queues it leaves let the
 * latency-bound launches of the other engines through: 149 -> 142 ms per step of the benchmark's arrangement),
 * "dec_gemm_fat" (0/1: the projections of a decode group of more than 64 sequences as fewer workgroups with twice
 * the weight rows each, which halves the times the activations pass through the compute units; default 0, the
 * faster form for an engine that has the GPU to itself by 4 %.  With several engines on one GPU set 1: 144 -> 140 ms
 * per step of the benchmark's arrangement). */
int mw_set_option(mw_ctx* ctx, const char* key, int value);

int mw_synchronize(mw_ctx* ctx);

/* ------------------------------------------------------------------------------------------------
 * Forced alignment: the acoustic model behind whisperx.load_align_model / whisperx.align
 * (reference transcribe.py:127-135; for language "zh" whisperx loads a Hugging Face Wav2Vec2ForCTC checkpoint of
 * the wav2vec2-large-xlsr-53 family: layer-norm feature extractor with convolution biases, stable-layer-norm
 * transformer).  The library computes the frame-wise log-probabilities over the character vocabulary; the
 * trellis search over them is host code above this boundary.
 * ------------------------------------------------------------------------------------------------ */
typedef struct mw_aligner mw_aligner;

typedef struct {
    int32_t n_conv;             /* layers of the feature extractor (7)                              */
    int32_t conv_dim;           /* channels of every layer (512); a multiple of 64, at most 512     */
    int32_t conv_kernel[8];     /* (10, 3, 3, 3, 3, 2, 2)                                           */
    int32_t conv_stride[8];     /* (5, 2, 2, 2, 2, 2, 2): one frame per 320 samples                 */
    int32_t hidden;             /* 1024; n_heads * 64                                               */
    int32_t n_heads;            /* 16                                                               */
    int32_t n_layers;           /* 24                                                               */
    int32_t ffn_dim;            /* 4096                                                             */
    int32_t pos_kernel;         /* 128: taps of the positional convolution                          */
    int32_t pos_groups;         /* 16                                                               */
    int32_t vocab;              /* rows of lm_head                                                  */
    int32_t max_samples;        /* samples one call takes through the model: the longest waveform of
                                 * mw_align_emissions / mw_align_segment, or n_segments x the longest
                                 * segment of a mw_align_segments pack (<= 9600000)                   */
} mw_align_desc;

/* Frames the feature extractor makes of n_samples samples (valid convolutions). */
int64_t mw_align_frames(const mw_align_desc* desc, int64_t n_samples);

/* Create / destroy an aligner on `device`.  mw_align_last_error(NULL) describes a failed mw_align_create on this thread. */
int mw_align_create(const mw_align_desc* desc, int device, mw_aligner** out);
void mw_align_destroy(mw_aligner* a);
const char* mw_align_last_error(const mw_aligner* a);

/* Hand over one tensor of the checkpoint under its Hugging Face state-dict name ("wav2vec2.feature_extractor.
 * conv_layers.0.conv.weight", ..., "lm_head.bias"), host or device memory, shape as stored.  The positional
 * convolution is taken either as "....pos_conv_embed.conv.weight" or as its weight-norm pair ("weight_g"/"weight_v"
 * or "parametrizations.weight.original0"/"original1").  mw_align_finalize checks that every tensor arrived with the
 * shape the description implies, converts them and frees the staging copies. */
int mw_align_set_weight(mw_aligner* a, const char* name, const void* data, const int64_t* shape, int ndim, int dtype);
int mw_align_finalize(mw_aligner* a);

/* A second aligner on the same GPU that shares the (finalized) parent's weights and has its own stream and
 * workspaces: segments are independent, and one segment alone (about 1500 frames) fills a fraction of the chip, so
 * several in flight from different host threads raise throughput.  Destroy every fork before its parent. */
int mw_align_fork(mw_aligner* parent, mw_aligner** out);

/* pcm: n_samples float32 mono 16 kHz samples (host or device; the caller normalises them if the checkpoint's
 * feature extractor asks for it) -> out: float32 [n_frames][vocab] log-softmax of the CTC logits (host or device,
 * room for mw_align_frames(desc, n_samples) rows), *n_frames rows written.  Blocks until out is complete. */
int mw_align_emissions(mw_aligner* a, const float* pcm, int64_t n_samples, float* out, int32_t* n_frames);

/* The search inside whisperx.align (get_trellis + backtrack of its alignment.py, reference transcribe.py:131), on the
 * host: emission float32 [n_frames][vocab] log-probabilities (host memory), tokens[n_tokens] the ids of the
 * segment's characters (-1: a character outside the vocabulary, aligned as a wildcard that scores the best
 * non-blank entry of a frame).  Fills, for every frame t, token_index[t] (which token the best path stands on) and
 * score[t] (probability of what the path emits there).  MW_ERR_STATE: no path (fewer frames than tokens). */
int mw_align_path(const float* emission, int32_t n_frames, int32_t vocab, const int32_t* tokens, int32_t n_tokens,
                  int32_t blank_id, int32_t* token_index, float* score);

/* mw_align_emissions followed by mw_align_path for one segment, without bringing the emissions to the host: only the
 * columns the search reads (the blank and the segment's 1..1024 tokens) are gathered on the GPU and copied.  pcm as
 * for mw_align_emissions; token_index and score need room for mw_align_frames(desc, n_samples) entries. */
int mw_align_segment(mw_aligner* a, const float* pcm, int64_t n_samples, const int32_t* tokens, int32_t n_tokens,
                     int32_t blank_id, int32_t* token_index, float* score, int32_t* n_frames);

/* mw_align_segment for a pack of 1..32 segments in ONE pass of the acoustic model (whisperx.align walks the segments
 * one at a time, reference transcribe.py:131; segments are independent, and one segment is a fraction of what fills
 * the GPU).  Every segment takes a slice as long as the longest one (zero-padded behind), so n_segments x the longest
 * n_samples must fit max_samples; the convolutions, LayerNorms and matrix products run once over all slices, the
 * positional convolution and the attention use each segment's own frames, so a segment's result does not depend on
 * what shares its pack.  pcm[b]: segment b's samples (host or device); tokens: the segments' token lists back to
 * back, n_tokens[b] of them for segment b; token_index, score: the segments' frames back to back (room for the sum of
 * mw_align_frames(desc, n_samples[b])); n_frames[b] frames of segment b; status[b]: MW_OK, or MW_ERR_STATE when
 * segment b has no path (fewer frames than tokens; its stretch of token_index / score is then undefined). */
int mw_align_segments(mw_aligner* a, int32_t n_segments, const float* const* pcm, const int64_t* n_samples,
                      const int32_t* tokens, const int32_t* n_tokens, int32_t blank_id, int32_t* token_index, float* score,
                      int32_t* n_frames, int32_t* status);

#ifdef __cplusplus
}
#endif
#endif /* MWHIP_H */
