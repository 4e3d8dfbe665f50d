// Internal launch interface between the engine and its gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

namespace mw {

typedef _Float16 f16;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int kChunkSamples = 480000;
constexpr int kFrames = 3000;
constexpr int kNfft = 400;
constexpr int kHop = 160;
constexpr int kFreqPad = 256;   // 201 DFT bins padded to 8 tiles of 32
constexpr int kHeadDim = 64;

// e4m3 (OCP, max 448) conversion of four floats, round to nearest even in hardware; the inputs are clamped to
// the finite range first so that no overflow encoding can appear.  Byte i of the result is value i.
__device__ __forceinline__ unsigned pack4_e4m3(float a, float b, float c, float d) {
    a = fminf(fmaxf(a, -448.f), 448.f); b = fminf(fmaxf(b, -448.f), 448.f);
    c = fminf(fmaxf(c, -448.f), 448.f); d = fminf(fmaxf(d, -448.f), 448.f);
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}

// ---------------------------------------------------------------- log-mel
struct MelTables {
    const float* basis_cos;   // [400][256]  hann[n]*cos(2*pi*f*n/400), zero for f >= 201
    const float* basis_sin;   // [400][256] -hann[n]*sin(...)
    const float* filt;        // [256][n_mels_pad] mel filter bank, zero padded
    int n_mels, n_mels_pad;
};
// pcm [B][480000] -> raw log10 mel [B][n_mels][3000] + per-block maxima [B][94]
void launch_mel_dft(const float* pcm, int B, const MelTables& t, float* logmel, float* blockmax, hipStream_t s);
// clamp/scale in place, and write the time-major fp16 copy melT [B][3002][n_mels] (rows 0 and 3001 stay zero)
void launch_mel_norm(float* logmel, const float* blockmax, int B, int n_mels, f16* melT, hipStream_t s);
// float32 features supplied by the caller -> melT
void launch_mel_import(const float* mel, int B, int n_mels, f16* melT, hipStream_t s);

// frame log-energies (dB) of mono PCM: out[f] over samples [f*hop, f*hop+frame) that exist, f < n_frames = ceil(n/hop)
void launch_frame_energy(const float* pcm, long n, int frame, int hop, float* out, long n_frames, hipStream_t s);

// ---------------------------------------------------------------- big GEMM (encoder, cross K/V)
enum GemmOut { OUT_F16 = 0, OUT_RESID_F32 = 1, OUT_QKV_SPLIT = 2, OUT_KV_SPLIT = 3, OUT_KV_SPLIT8 = 4 };
struct GemmParams {
    // C[m][n] = epilogue( sum_k A[m][k] * W[n][k] ), fp16 operands, fp32 accumulate
    const f16* A; long lda; int a_rows_per_batch; long a_batch_stride;
    const f16* W; long ldw;
    int M, N, K;
    const float* bias;        // [N] or null
    float scale; int scale_ncols;   // columns n < scale_ncols are multiplied by scale after the bias
    int gelu;
    int out_mode;
    void* C; long ldc; int c_rows_per_batch; long c_batch_stride;   // OUT_F16 / OUT_RESID_F32
    f16* q; f16* k; f16* v;   // split modes: q,k [B][H][T][64]; v is [B][H][64][Tpad] (QKV) or [B][H][T][64] (KV)
    int T, Tpad, H, D;
    // f8 != 0: A and W hold OCP e4m3 bytes (lda, ldw, K still count elements; K % 128 == 0) and the
    // product is scaled by a_scale[m] * w_scale[n] before the bias.  Output modes: OUT_F16 with GELU,
    // OUT_QKV_SPLIT, OUT_KV_SPLIT.  f8 == 0 with w_scale != null: fp16 operands, columns scaled by
    // w_scale[n] before the bias (W holds unscaled fp8 codes expanded to fp16); OUT_RESID_F32 without GELU, or OUT_F16.
    int f8; const float* a_scale; const float* w_scale;
    // OUT_KV_SPLIT8 (f8 products only): k, v as e4m3 codes [B][H][T][64] with one scale per (b, h, t) row of 64
    uint8_t* k8; uint8_t* v8; float* ks; float* vs;
    // groups > 1 (fp16 operands, OUT_F16 / OUT_RESID_F32 only): that many independent products of one shape in ONE
    // launch; group g reads A + g * a_group_stride and W + g * w_group_stride (elements), bias + g * N, and writes
    // C + g * c_group_stride (elements of C's type).  A grouped convolution as GEMMs over a group-major buffer.
    int groups; long a_group_stride, w_group_stride, c_group_stride;
};
void launch_gemm(const GemmParams& p, hipStream_t s);

// ---------------------------------------------------------------- encoder helpers
// `advance` (optional): an int in device memory that the launch increments by one when it is done
void launch_layernorm(const float* x, const float* g, const float* b, f16* y, float* y32, int rows, int D, hipStream_t s,
                      int* advance = nullptr);
// LayerNorm with an fp8 (OCP e4m3) result: y8 [rows][D] bytes and one scale per row (y = y8 * yscale[row]);
// y32 (optional) receives the unquantised fp32 rows
void launch_layernorm_fp8(const float* x, const float* g, const float* b, uint8_t* y8, float* yscale, float* y32, int rows, int D,
                          hipStream_t s);
void launch_bcast_rows(const float* table, float* x, int B, int rows, int D, hipStream_t s);   // x[b][r][:] = table[r][:]
// q,k [B][H][T][64], vT [B][H][64][Tpad] -> out [B*T][H*64] fp16
// keys_of (optional, device, int [B]): batch element b attends over its first keys_of[b] <= T positions only and only its
// first keys_of[b] output rows are written (a packed batch of recordings of different lengths; the rest must hold finite values)
void launch_enc_attention(const f16* q, const f16* k, const f16* vT, f16* out, int B, int H, int T, int Tpad, hipStream_t s,
                          const int* keys_of = nullptr);

// ---------------------------------------------------------------- weight import
void launch_convert(const void* src, int src_dtype, void* dst, int dst_is_f16, long n, hipStream_t s);
// fp8 weight import: [rows][cols] of any dtype -> e4m3 bytes + one scale per row (cols % 4 == 0)
void launch_quant_rows(const void* src, int src_dtype, uint8_t* dst, float* scale, int rows, int cols, hipStream_t s);
// e4m3 bytes -> the same values as fp16, unscaled (n % 16 == 0)
void launch_fp8_to_f16(const uint8_t* src, f16* dst, long n, hipStream_t s);
// conv weight [D][C][3] (any dtype) -> fp16 [D][3][C]
void launch_conv_permute(const void* src, int src_dtype, f16* dst, int D, int C, hipStream_t s);

// ---------------------------------------------------------------- decode step
struct DecState {          // lives in device memory; one per context
    int pos;               // position of the token whose embedding is in x
    int prompt_len;
    int max_new;
    int eot;
    int n_finished;
    int batch;
    int pad[2];
};
enum GemvOut { GV_F16 = 0, GV_RESID = 1, GV_QKV = 2, GV_LOGITS = 3 };
struct GemvParams {
    int B, N, K;              // B: any number of sequences; more than 32 run as groups of 32 (or 16) inside the launch
    const f16* W;             // [N][K] fp16 ...
    const uint8_t* W8;        // ... or, when non-null, [N][K] OCP e4m3 codes with w_scale [N] (K % 64 == 0); not for GV_LOGITS
    const float* w_scale;
    const float* bias;        // [N] or null
    // input: LayerNorm(x32; ln_g, ln_b) when ln_g != null, else xin16
    const float* x32; const float* ln_g; const float* ln_b;
    f16* ln_scratch;          // [B][K] fp16, used when K has no fused-LayerNorm instantiation
    const f16* xin16;
    int nv;                   // weight rows per workgroup (8 or 16); chosen by launch_gemv
    float scale; int scale_ncols; int gelu;
    int out_mode;
    f16* out16;               // GV_F16: [B][N];  GV_QKV: q [B][D]
    float* out32;             // GV_RESID: x [B][N] += ;  GV_LOGITS: logits [B][N]
    f16* kcache; f16* vcache; // GV_QKV: [B][H][Tctx][64] of this layer
    int H, D, Tctx;
    const DecState* st;
    // GV_LOGITS
    const uint8_t* mask_first; const uint8_t* mask_rest;
    float* part_val; int* part_idx; int n_parts;   // [B][n_parts]
    // GV_RESID, optional: scratch of the split-K form (partial tiles + one ticket per row group; the
    // tickets must be zero before the first launch, every launch leaves them zero again); sk_partial_floats and
    // sk_groups are the extents of ONE group of 16 (or 32) sequences, the buffers hold one such block per group
    float* sk_partial; long sk_partial_floats; int* sk_tickets; int sk_groups;
    // projections of 33..128 sequences (gemm_dec_kernel), optional: partial tiles of its K ranges (dg_partial_floats in
    // all) and the number of row groups the ticket array covers; it shares sk_tickets
    float* dg_partial; long dg_partial_floats; int dg_groups;
    int dg_fat;               // gemm_dec_kernel: twice the weight rows per workgroup (several engines share the GPU: bytes through the CUs count, not workgroups)
};
// floats of partial-tile scratch that cover every projection of a model (d_model D, MLP width F) at up to B sequences
inline long decgemm_partial_floats(int D, int F, int B = 128) { return (long)(3 * D > F ? 3 * D : F) * 128 * 8 * ((B + 127) / 128); }
// floats / ints of split-K scratch that cover every residual projection of a model of width D
inline long splitk_partial_floats(int D) { return (long)(D / 16) * 8 * 2 * 256; }
inline int splitk_groups(int D) { return D / 16; }
void launch_gemv(const GemvParams& p, hipStream_t s);
int gemv_logits_parts(int N);
// raise the dynamic-LDS limit of every decode kernel instantiation once (not legal inside stream capture)
void dec_kernels_prepare();

// src_rows (beam search only, else null): int [Tctx][B], physical cache row of each cached position of each sequence
// max_keys: upper bound of position + 1 over the steps this launch (or its captured graph) serves; launches
// with the same dec_self_attn_bucket(max_keys) are identical
void launch_dec_self_attn(const f16* q, const f16* kc, const f16* vc, f16* out, int B, int H, int Tctx,
                          const DecState* st, hipStream_t s, const int* src_rows = nullptr, int max_keys = 1 << 30);
int dec_self_attn_bucket(int n_keys);
constexpr int kSelfAttnBuckets = 4;
// Cross-attention splits each (sequence, head) over kXattnSplit workgroups; `partial` holds
// B*H*kXattnSplit*68 floats and `tickets` B*H ints that must be zero before the first launch
// (every launch leaves them zero again).
constexpr int kXattnSplit = 4;
// kv_div > 1: query row r attends over the encoder cache of recording (row0 + r) / kv_div (the beams of one
// recording; the prompt positions of one sequence).  row0: index of this launch's first row in the whole list.
// done != null: rows r with done[(row0 + r) / done_div] != 0 are skipped (`out` keeps what it held): sequences
// that have ended, recordings whose beam search is complete.
// stream_from: from this many (sequence, head) pairs on (and kv_div == 1) one workgroup per pair walks all the keys
// (cross_attn_stream_kernel) instead of four key ranges per pair merged through tickets; <= 0: never.
// stream_per_cu: resident workgroups per CU that form is held to (by an LDS request nothing uses): 0 = what its
// registers allow (three).  One per CU streams within 2 - 3 % of three (its loads in flight still cover the HBM round
// trip) and leaves the memory queues short for the latency-bound launches of other streams.
void launch_dec_cross_attn(const f16* q, const f16* kc, const f16* vc, f16* out, int B, int H, int T, float* partial,
                           int* tickets, hipStream_t s, int kv_div = 1, int row0 = 0, const int* done = nullptr, int done_div = 1,
                           int stream_from = 1024, int stream_per_cu = 0);
// The same over an e4m3 cache: k8, v8 [B][H][T][64] codes, ks, vs [B][H][T] row scales
void launch_dec_cross_attn8(const f16* q, const uint8_t* k8, const uint8_t* v8, const float* ks, const float* vs, f16* out, int B,
                            int H, int T, float* partial, int* tickets, hipStream_t s, int kv_div = 1, int row0 = 0,
                            const int* done = nullptr, int done_div = 1, int stream_from = 1024, int stream_per_cu = 0);

struct FinalizeParams {
    DecState* st;
    const float* part_val; const int* part_idx; int n_parts;
    const int* forced;          // [B][n_forced] ids fed before sampling starts (prompt, or teacher-forced ids)
    int n_forced;
    const f16* tok_emb; const f16* pos_emb; int D; int n_text_ctx; int vocab;
    float* x;                   // [B][D]
    int* out_ids;               // [B][max_new]
    int* out_len;               // [B]
    int* finished;              // [B]
};
void launch_dec_init(const FinalizeParams& p, int B, int prompt_len, int max_new, int eot, hipStream_t s);

// ---------------------------------------------------------------- prompt prefill (all forced positions in one pass)
// rows m = sequence * P + position.  forced: int [B][n_forced] (the first P of each row are used)
void launch_prefill_embed(const int* forced, int n_forced, int B, int P, const f16* tok_emb, const f16* pos_emb, float* x, int D,
                          hipStream_t s);
// qkv [B*P][3D] fp16 (scaled q | k | v) -> k, v rows into the layer's cache [B][H][Tctx][64] and causal attention out [B*P][D]; P <= 448
// cache_seq_stride: sequence b files its rows under cache sequence b * cache_seq_stride
void launch_prefill_self_attn(const f16* qkv, f16* kcache, f16* vcache, f16* out, int B, int H, int P, int Tctx, int D, hipStream_t s,
                              int cache_seq_stride = 1);
// Cross-attention of every prompt row: q, out [B*P][H*64] (rows m = sequence * P + position, q scaled); sequence b
// attends over the cached encoder K/V ([recordings][H][T][64]) of recording b / kv_rep.  One pass of the keys per
// (sequence, head, 128 prompt rows), on the matrix cores.  Also the decode step of beam search and of best-of-n
// sampling: B recordings with P hypotheses each are B "sequences" of P rows that share one K/V cache.
// done != null: a workgroup all of whose rows r have done[r / done_div] != 0 does nothing (`out` keeps what it held).
void launch_prefill_cross_attn(const f16* q, const f16* kc, const f16* vc, f16* out, int B, int P, int H, int T, int kv_rep,
                               hipStream_t s, const int* done = nullptr, int done_div = 1);
// The same over an e4m3 cache (codes k8, v8 and row scales ks, vs as for launch_dec_cross_attn8)
void launch_prefill_cross_attn8(const f16* q, const uint8_t* k8, const uint8_t* v8, const float* ks, const float* vs, f16* out, int B,
                                int P, int H, int T, int kv_rep, hipStream_t s, const int* done = nullptr, int done_div = 1);
// dx [B*rep][D] = x rows of the last position, `rep` copies each; st->pos = P - 1; src_rows (beam search, else null):
// int [Tctx][B*rep], its first P rows are pointed at the cache rows b * rep
void launch_prefill_last_rows(const float* x, float* dx, int B, int P, int D, DecState* st, hipStream_t s, int rep = 1,
                              int* src_rows = nullptr);

// ---------------------------------------------------------------- beam search bookkeeping
constexpr int kMaxBeam = 8;
struct BeamParams {
    DecState* st;
    int n_audio, beam, max_cand;            // sequences = n_audio * beam; max_cand finished hypotheses per recording
    const float* logits;                    // [n_audio*beam][vocab] of this step
    const uint8_t* mask_first; const uint8_t* mask_rest;
    const int* forced; int n_forced;        // [n_audio*beam][n_forced]
    const f16* tok_emb; const f16* pos_emb; int D; int n_text_ctx; int vocab;
    float* x;                               // [n_audio*beam][D] next input embeddings
    float* sum_lp;                          // [n_audio*beam] cumulative log-probability of each live beam
    int* hist;                              // [n_audio*beam][max_new] sampled tokens of each live beam
    int* src_rows;                          // [n_text_ctx][n_audio*beam] physical self-attention cache rows
    int* fin_tokens;                        // [n_audio][max_cand][max_new]
    float* fin_lp; int* fin_len;            // [n_audio][max_cand]
    int* fin_count; int* completed;         // [n_audio]
    // per (sequence, vocabulary part) of this step, written by the scan launch and merged by the step launch:
    float* part_m; float* part_s;           // [n_audio*beam][kBeamParts] running maximum and sum of exp(v - maximum)
    float* part_v; int* part_i;             // [n_audio*beam][kBeamParts][kMaxBeam + 1] best logits, descending, and their ids
};
constexpr int kBeamParts = 16;
void launch_beam_init(const BeamParams& p, int prompt_len, int max_new, int eot, hipStream_t s);
void launch_beam_step(const BeamParams& p, hipStream_t s);
void launch_dec_finalize(const FinalizeParams& p, int B, hipStream_t s);

// ---------------------------------------------------------------- sampling at a temperature (the fallback decodes)
// Set per call by the init launch and read by the step launches (captured steps must not carry them as arguments).
struct SampleState {
    float inv_temp;         // 1 / temperature; 0 = arg-max (temperature 0)
    uint32_t seed;
    int pad[2];
};
// best_of independent samples per recording: sequence r reads the cross K/V of recording r / best_of and keeps a
// self-attention cache of its own.
struct SampleParams {
    DecState* st; SampleState* ss;
    int n_audio, best_of;
    const float* logits;                    // [n_audio*best_of][vocab] of this step
    const uint8_t* mask_first; const uint8_t* mask_rest;
    const int* forced; int n_forced;        // [n_audio*best_of][n_forced]
    const f16* tok_emb; const f16* pos_emb; int D; int n_text_ctx; int vocab;
    float* x;                               // [n_audio*best_of][D] next input embeddings
    float* sum_lp;                          // [n_audio*best_of] sum of log-probabilities of the sampled tokens (end-of-text included)
    int* out_ids; int* out_len; int* finished;      // [n_audio*best_of][max_new], [n_audio*best_of], [n_audio*best_of]
    const uint8_t* active;                  // [n_audio] or null: recordings with 0 start ended (length 0, nothing sampled)
    int skip_ended;                         // 1: the cross-attention launches skip sequences whose `finished` flag is set
    // per (sequence, vocabulary part) of this step, written by the scan launch and merged by the step launch:
    float* part_m; float* part_s;           // running maximum and sum of exp(v - maximum) over the unmasked logits
    float* part_v; float* part_l; int* part_i;      // best perturbed score, the raw logit behind it, its id
};
// Uniform in (0, 1) for (seed, sequence, sampled position, token id): three rounds of the 32-bit murmur3 finaliser;
// the top 23 bits plus one half, so that the value is exact in float32 on every implementation.
__host__ __device__ inline uint32_t sample_mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
__host__ __device__ inline float sample_uniform(uint32_t seed, uint32_t seq, uint32_t pos, uint32_t tok) {
    uint32_t h = sample_mix32(seed ^ (seq * 0x9E3779B1u));
    h = sample_mix32(h ^ (pos * 0x85EBCA77u));
    h = sample_mix32(h ^ (tok * 0xC2B2AE3Du));
    return ((float)(h >> 9) + 0.5f) * (1.0f / 8388608.0f);
}
void launch_sample_init(const SampleParams& p, int prompt_len, int max_new, int eot, float inv_temp, uint32_t seed, int n_ended_at_start,
                        hipStream_t s);
void launch_sample_step(const SampleParams& p, hipStream_t s);

// ---------------------------------------------------------------- wav2vec2 front end (forced alignment)
// First convolution of the feature extractor on the raw waveform, with its LayerNorm over channels and GELU:
// x float32 [n] -> out fp16 [T0][C], T0 = (n - K0) / S0 + 1.  w [C][K0], b, g, beta [C] float32.  C % 64 == 0, C <= 512, K0 <= 16.
void launch_w2v_conv0(const float* x, long n, const float* w, const float* b, const float* g, const float* beta, f16* out, int T0, int C,
                      int K0, int S0, hipStream_t s);
// LayerNorm over the D channels of fp16 rows, in place, optionally followed by GELU (D % 8 == 0, D <= 4096)
void launch_ln16(f16* x, const float* g, const float* b, int rows, int D, float eps, int gelu, hipStream_t s);
// x float32 [T][D] -> pack fp16 [G][T + K][D / G]: group-major copy with K/2 zero rows in front and K - K/2 behind
void launch_w2v_pos_pack(const float* x, f16* pack, int T, int D, int G, int K, hipStream_t s);
// rows of z [rows][ld]: z[r][0..V) -= logsumexp(z[r][0..V))
void launch_log_softmax_rows(float* z, int rows, int ld, int V, hipStream_t s);
// out [rows][1 + J]: column 0 = z[r][blank], column 1 + j = z[r][tokens[j]], or for tokens[j] == -1 the largest z[r][v], v != blank
void launch_ctc_gather(const float* z, int rows, int ld, int V, const int32_t* tokens, int J, int blank, float* out, hipStream_t s);

}  // namespace mw
