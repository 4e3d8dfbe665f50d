// Engine behind the C ABI of include/mwhip.h: owns the weights, workspaces and K/V caches of one
// GPU, sequences the kernels of the hot path (log-mel -> encoder -> cross K/V -> greedy decode) on
// one HIP stream, and replays a captured hipGraph of one decode step per token.
#include "../../include/mwhip.h"
#include "mw_kernels.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

using namespace mw;

static thread_local std::string g_create_error;

// Scratch of mw_logmel_noalloc, the one entry point that allocates nothing: the per-block maxima of the
// log-mel kernel (94 floats per chunk, kNoAllocMaxBatch chunks), which mel_norm_kernel reduces to each chunk's
// maximum.  Every GPU the library runs on holds its own copy of the array at its own address until the process
// ends; in host code the array's NAME is only a stand-in for it, so the address on each device is asked for
// once (mw_create, that device current) and only that pointer is ever handed to a kernel.
constexpr int kNoAllocMaxBatch = 128;
constexpr int kMaxBatch = 512;                 // sequences one context holds (a decode group of 32 batches of 16)
constexpr int kMelBlocksPerChunk = (kFrames + 31) / 32;        // 94: the grid of launch_mel_dft
static __device__ __align__(256) float scratch[kNoAllocMaxBatch * kMelBlocksPerChunk];
constexpr int kMaxDevices = 64;
constexpr int kGraphSlots = 4;                 // captured decode steps kept per self-attention bucket
static float* g_scratch_ptr[kMaxDevices] = {};
static std::mutex g_scratch_mutex;
// one per GPU: every context and fork on a GPU shares that GPU's array, so calls on one GPU take turns
static std::mutex g_scratch_in_use[kMaxDevices];

// `device` must be current.  Returns hipSuccess once the device's address is known.
static hipError_t resolve_scratch(int device) {
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[device]) return hipSuccess;
    float* ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    g_scratch_ptr[device] = ptr;
    return hipSuccess;
}

// fp16 weights in `w`; or, in an fp8 engine, OCP e4m3 codes in `w8` with one scale per output row in `ws`
struct LinearW { f16* w = nullptr; float* b = nullptr; uint8_t* w8 = nullptr; float* ws = nullptr; };
struct NormW { float* g = nullptr; float* b = nullptr; };

struct EncLayerW { NormW ln1, ln2; LinearW qkv, out, fc1, fc2; };
struct DecLayerW { NormW ln1, ln2, ln3; LinearW qkv, out, cq, co, fc1, fc2; };

struct mw_ctx {
    mw_model_desc d{};
    int device = 0, max_batch = 0;
    int cache_seqs = 0;                 // sequences the cross-attention K/V cache holds (<= max_batch; hypotheses of a recording share one)
    hipStream_t stream = nullptr;
    std::string err;
    std::vector<void*> allocs;
    char* arena = nullptr; size_t arena_size = 0, arena_used = 0; int arena_mode = 0;     // dalloc: 1 = add up sizes, 2 = carve `arena`
    bool finalized = false;
    int use_graph = 1, check_every = 4, use_prefill = 1, skip_done = 1;
    int xattn_stream_from = 1024;       // (sequence, head) pairs from which the decode cross-attention walks all keys per workgroup
    int xattn_stream_per_cu = 0;        // resident workgroups per CU that form is held to (0: what its registers allow)
    int dec_gemm_fat = 0;               // projections of decode groups: 1 = fewer, larger workgroups (engines that share the GPU)
    bool can_end = true;                    // false when the decode options suppress end-of-text everywhere
    const mw_ctx* parent = nullptr;                       // non-null: weights and tables belong to `parent`

    // weights
    f16* conv1_w = nullptr; float* conv1_b = nullptr;     // [D][3][n_mels]
    f16* conv2_w = nullptr; float* conv2_b = nullptr;     // [D][3][D]
    float* enc_pos = nullptr;                             // [1500][D] fp32
    std::vector<EncLayerW> enc;
    NormW enc_ln;
    f16* tok_emb = nullptr; f16* dec_pos = nullptr;
    std::vector<DecLayerW> dec;
    NormW dec_ln;
    f16* ckv_w = nullptr; float* ckv_b = nullptr;         // [L][2D][D], [L][2D] (k part of the bias = 0)
    uint8_t* ckv_w8 = nullptr; float* ckv_ws = nullptr;   // fp8 engine: [L][2D][D] codes, [L][2D] scales
    bool fp8 = false;                                     // desc.weight_format == MW_WEIGHTS_FP8
    std::set<std::string> seen;
    bool enc_pos_set = false;

    // mel tables
    float *basis_cos = nullptr, *basis_sin = nullptr, *filt = nullptr;
    int n_mels_pad = 0;

    // staging for host -> device weight import
    void* stage = nullptr; size_t stage_bytes = 0;

    // encoder workspaces (sized for max_batch)
    float* pcm_dev = nullptr;
    float* logmel = nullptr; float* blockmax = nullptr;
    f16* melT = nullptr; f16* h1 = nullptr;
    float* x = nullptr; f16* xn = nullptr;
    uint8_t* xn8 = nullptr; float* xs = nullptr;          // fp8 engine: quantised LayerNorm rows [M][D] + row scales [M]
    f16* wexp = nullptr;                                  // fp8 engine: one weight matrix expanded to fp16 (max(D*D, D*F))
    f16 *q = nullptr, *k = nullptr, *vT = nullptr, *att = nullptr, *hid = nullptr, *enc_out = nullptr;
    float* enc_out32 = nullptr;
    f16 *crossK = nullptr, *crossV = nullptr;             // [L][B][H][1500][64]
    uint8_t *crossK8 = nullptr, *crossV8 = nullptr;       // MW_KV_FP8: e4m3 codes in the same layout ...
    float *crossKs = nullptr, *crossVs = nullptr;         // ... and one scale per row: [L][B][H][1500]
    bool kv8 = false;
    int Tpad = 0;
    int mel_batch = 0, enc_batch = 0;

    // decoder workspaces
    DecState* st = nullptr;
    float* dx = nullptr; f16 *dq = nullptr, *datt = nullptr, *dhid = nullptr, *dxn = nullptr;
    f16 *selfK = nullptr, *selfV = nullptr;               // [L][B][H][n_text_ctx][64]
    float* logits = nullptr; float* part_val = nullptr; int* part_idx = nullptr; int n_parts = 0;
    uint8_t *mask_first = nullptr, *mask_rest = nullptr;
    int *forced = nullptr, *out_ids = nullptr, *out_len = nullptr, *finished = nullptr;
    int forced_cap = 0;
    float* xpartial = nullptr; int* xtickets = nullptr;    // cross-attention key-range partials + ticket counters
    float* dgpartial = nullptr;                            // partial tiles of the projections of 33..128 sequences
    float* skpartial = nullptr; int* sktickets = nullptr;  // split-K partial tiles + ticket counters of the residual projections
    // beam search (allocated on first use)
    float *bm_sum_lp = nullptr, *bm_fin_lp = nullptr, *bm_part_m = nullptr, *bm_part_s = nullptr, *bm_part_v = nullptr;
    int* bm_part_i = nullptr;
    int *bm_hist = nullptr, *bm_src_rows = nullptr, *bm_fin_tokens = nullptr, *bm_fin_len = nullptr, *bm_fin_count = nullptr,
        *bm_completed = nullptr;
    // sampling at a temperature (allocated on first use)
    SampleState* ss = nullptr;
    float *sp_sum_lp = nullptr, *sp_part_m = nullptr, *sp_part_s = nullptr, *sp_part_v = nullptr, *sp_part_l = nullptr;
    int* sp_part_i = nullptr;
    uint8_t* sp_active = nullptr;

    // decode options
    std::vector<int32_t> prompt;
    int eot = 0, max_new = 0;
    bool opts_set = false;

    // captured decode step
    // one captured step per self-attention context bucket (the launch geometry of that kernel follows the position)
    // and, per bucket, up to kGraphSlots (batch size / forced length / kind of decode / logits flag) combinations kept
    // side by side: a ragged last batch, or a fallback pass between greedy ones, does not throw the other's graphs
    // away (a capture + instantiate of the ~260-launch step costs tens of ms per bucket)
    hipGraph_t graph[kSelfAttnBuckets][kGraphSlots] = {}; hipGraphExec_t graph_exec[kSelfAttnBuckets][kGraphSlots] = {};
    long graph_batch[kSelfAttnBuckets][kGraphSlots] = {}; int graph_logits[kSelfAttnBuckets][kGraphSlots] = {};
    int graph_next[kSelfAttnBuckets] = {};                // slot the next new combination of a bucket takes (round robin)

    // timing
    hipEvent_t ev[6]{};
    // all-finished polls: the counter is copied into pinned slot (poll & 1) and read one poll later, so the stream never drains
    int* fin_host = nullptr; hipEvent_t poll_ev[2]{};
    float stage_ms[5]{};
};

#define HIPCHK(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            char buf_[512];                                                                       \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            ctx->err = buf_;                                                                      \
            return MW_ERR_HIP;                                                                    \
        }                                                                                         \
    } while (0)

static int fail(mw_ctx* ctx, int code, const std::string& msg) {
    ctx->err = msg;
    return code;
}

template <typename T>
static int dalloc(mw_ctx* ctx, T** out, size_t count, bool zero = false) {
    void* p = nullptr;
    size_t bytes = count * sizeof(T) + 256;           // slack: vector loads may run a few bytes past a tail
    if (ctx->arena_mode) {                             // one allocation for many tensors: see alloc_weights
        bytes = (bytes + 255) / 256 * 256;
        if (ctx->arena_mode == 1) { ctx->arena_size += bytes; *out = nullptr; return MW_OK; }
        p = ctx->arena + ctx->arena_used;
        ctx->arena_used += bytes;
        if (zero && hipMemsetAsync(p, 0, bytes, ctx->stream) != hipSuccess) { ctx->err = "hipMemsetAsync failed"; return MW_ERR_HIP; }
        *out = reinterpret_cast<T*>(p);
        return MW_OK;
    }
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        ctx->err = std::string("hipMalloc of ") + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e);
        return MW_ERR_NOMEM;
    }
    ctx->allocs.push_back(p);
    if (zero) {
        e = hipMemsetAsync(p, 0, bytes, ctx->stream);
        if (e != hipSuccess) { ctx->err = "hipMemsetAsync failed"; return MW_ERR_HIP; }
    }
    *out = reinterpret_cast<T*>(p);
    return MW_OK;
}
#define ALLOC(ptr, count, ...)                                          \
    do { int r_ = dalloc(ctx, &(ptr), (count), ##__VA_ARGS__); if (r_) return r_; } while (0)

static bool is_device_ptr(const void* p) {
    hipPointerAttribute_t a;
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

// ---------------------------------------------------------------------------------------------
// creation
// ---------------------------------------------------------------------------------------------
static int build_mel_tables(mw_ctx* ctx) {
    const int n_mels = ctx->d.n_mels;
    ctx->n_mels_pad = ((n_mels + 31) / 32) * 32;
    std::vector<float> bc((size_t)kNfft * kFreqPad, 0.f), bs((size_t)kNfft * kFreqPad, 0.f);
    const double two_pi = 6.283185307179586476925286766559;
    for (int n = 0; n < kNfft; ++n) {
        const double w = 0.5 - 0.5 * cos(two_pi * n / kNfft);          // periodic Hann (torch.hann_window)
        for (int f = 0; f <= kNfft / 2; ++f) {
            const int r = (int)(((long)f * n) % kNfft);                 // exact angle reduction
            const double a = two_pi * r / kNfft;
            bc[(size_t)n * kFreqPad + f] = (float)(w * cos(a));
            bs[(size_t)n * kFreqPad + f] = (float)(-w * sin(a));
        }
    }
    // slaney mel scale + slaney normalisation, 0..8000 Hz (transformers audio_utils.py:638-729)
    auto hz_to_mel = [](double f) { return f >= 1000.0 ? 15.0 + log(f / 1000.0) * (27.0 / log(6.4)) : 3.0 * f / 200.0; };
    auto mel_to_hz = [](double m) { return m >= 15.0 ? 1000.0 * exp((log(6.4) / 27.0) * (m - 15.0)) : 200.0 * m / 3.0; };
    std::vector<double> pts(n_mels + 2);
    const double m_lo = hz_to_mel(0.0), m_hi = hz_to_mel(8000.0);
    for (int i = 0; i < n_mels + 2; ++i) pts[i] = mel_to_hz(m_lo + (m_hi - m_lo) * i / (n_mels + 1));
    std::vector<float> fl((size_t)kFreqPad * ctx->n_mels_pad, 0.f);
    for (int f = 0; f <= kNfft / 2; ++f) {
        const double hz = 8000.0 * f / (kNfft / 2);
        for (int m = 0; m < n_mels; ++m) {
            const double down = (hz - pts[m]) / (pts[m + 1] - pts[m]);
            const double up = (pts[m + 2] - hz) / (pts[m + 2] - pts[m + 1]);
            double v = fmin(down, up);
            if (v < 0) v = 0;
            v *= 2.0 / (pts[m + 2] - pts[m]);
            fl[(size_t)f * ctx->n_mels_pad + m] = (float)v;
        }
    }
    ALLOC(ctx->basis_cos, bc.size());
    ALLOC(ctx->basis_sin, bs.size());
    ALLOC(ctx->filt, fl.size());
    HIPCHK(hipMemcpy(ctx->basis_cos, bc.data(), bc.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->basis_sin, bs.data(), bs.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->filt, fl.data(), fl.size() * 4, hipMemcpyHostToDevice));
    return MW_OK;
}

static int alloc_linear(mw_ctx* ctx, LinearW& l, size_t out, size_t in, bool zero_bias = false) {
    if (ctx->fp8) { ALLOC(l.w8, out * in); ALLOC(l.ws, out); }
    else ALLOC(l.w, out * in);
    ALLOC(l.b, out, zero_bias);
    return MW_OK;
}
#define ALLOC_LINEAR(l, out, in, ...)                                   \
    do { int r_ = alloc_linear(ctx, (l), (out), (in), ##__VA_ARGS__); if (r_) return r_; } while (0)

static int alloc_weights_into(mw_ctx* ctx) {
    const mw_model_desc& d = ctx->d;
    const size_t D = d.d_model, F = d.ffn_dim, T = d.n_audio_ctx;
    ALLOC(ctx->conv1_w, D * 3 * d.n_mels); ALLOC(ctx->conv1_b, D);
    ALLOC(ctx->conv2_w, D * 3 * D); ALLOC(ctx->conv2_b, D);
    ALLOC(ctx->enc_pos, T * D);
    ctx->enc.resize(d.enc_layers);
    for (auto& l : ctx->enc) {
        ALLOC(l.ln1.g, D); ALLOC(l.ln1.b, D); ALLOC(l.ln2.g, D); ALLOC(l.ln2.b, D);
        ALLOC_LINEAR(l.qkv, 3 * D, D, true);
        ALLOC_LINEAR(l.out, D, D);
        ALLOC_LINEAR(l.fc1, F, D);
        ALLOC_LINEAR(l.fc2, D, F);
    }
    ALLOC(ctx->enc_ln.g, D); ALLOC(ctx->enc_ln.b, D);
    ALLOC(ctx->tok_emb, (size_t)d.vocab * D);
    ALLOC(ctx->dec_pos, (size_t)d.n_text_ctx * D);
    ctx->dec.resize(d.dec_layers);
    for (auto& l : ctx->dec) {
        ALLOC(l.ln1.g, D); ALLOC(l.ln1.b, D); ALLOC(l.ln2.g, D); ALLOC(l.ln2.b, D); ALLOC(l.ln3.g, D); ALLOC(l.ln3.b, D);
        ALLOC_LINEAR(l.qkv, 3 * D, D, true);
        ALLOC_LINEAR(l.out, D, D);
        ALLOC_LINEAR(l.cq, D, D);
        ALLOC_LINEAR(l.co, D, D);
        ALLOC_LINEAR(l.fc1, F, D);
        ALLOC_LINEAR(l.fc2, D, F);
    }
    ALLOC(ctx->dec_ln.g, D); ALLOC(ctx->dec_ln.b, D);
    if (ctx->fp8) { ALLOC(ctx->ckv_w8, (size_t)d.dec_layers * 2 * D * D); ALLOC(ctx->ckv_ws, (size_t)d.dec_layers * 2 * D); }
    else ALLOC(ctx->ckv_w, (size_t)d.dec_layers * 2 * D * D);
    ALLOC(ctx->ckv_b, (size_t)d.dec_layers * 2 * D, true);
    return MW_OK;
}

// The weights are some 1500 tensors, most of them small.  They live in ONE allocation: a first walk over the list adds
// up the sizes, a second one carves the block.  As allocations of their own they end up wherever the driver finds room,
// and in a process that has already freed large blocks (an earlier pool of engines) that placement made every engine
// that reads them side by side slower: a one-hour recording took 3.1 s instead of 2.4 s with nothing else changed,
// the decode loops of three engines 2066 ms each instead of 1711 (each kernel alone ran at the same speed).
static int alloc_weights(mw_ctx* ctx) {
    ctx->arena_mode = 1; ctx->arena_size = 0;
    int r = alloc_weights_into(ctx);
    ctx->arena_mode = 0;
    if (r) return r;
    void* block = nullptr;
    const hipError_t e = hipMalloc(&block, ctx->arena_size);
    if (e != hipSuccess) {
        ctx->err = std::string("hipMalloc of ") + std::to_string(ctx->arena_size) + " bytes (weights) failed: " + hipGetErrorString(e);
        return MW_ERR_NOMEM;
    }
    ctx->allocs.push_back(block);
    ctx->arena = static_cast<char*>(block); ctx->arena_used = 0;
    ctx->arena_mode = 2;
    r = alloc_weights_into(ctx);
    ctx->arena_mode = 0;
    return r;
}

// Everything one stream of work needs besides the weights: workspaces, K/V caches, decode state.
static int alloc_work(mw_ctx* ctx) {
    const mw_model_desc& d = ctx->d;
    const size_t D = d.d_model, F = d.ffn_dim, H = d.n_heads, T = d.n_audio_ctx, B = ctx->max_batch;
    const size_t M = B * T, MC = (size_t)ctx->cache_seqs * T;
    ctx->Tpad = (int)((T + 7) / 8 * 8);
    if (hipHostMalloc(reinterpret_cast<void**>(&ctx->fin_host), 2 * sizeof(int), hipHostMallocDefault) != hipSuccess) {
        ctx->err = "hipHostMalloc failed"; return MW_ERR_HIP;
    }
    for (auto& e : ctx->poll_ev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { ctx->err = "hipEventCreate failed"; return MW_ERR_HIP; }
    // encoder workspaces
    ALLOC(ctx->pcm_dev, B * (size_t)kChunkSamples);
    ALLOC(ctx->logmel, B * (size_t)d.n_mels * kFrames);
    ALLOC(ctx->blockmax, B * 128);
    ALLOC(ctx->melT, B * (size_t)(kFrames + 2) * d.n_mels, true);
    ALLOC(ctx->h1, B * (size_t)(kFrames + 2) * D, true);
    ALLOC(ctx->x, M * D); ALLOC(ctx->xn, M * D);
    // xn8: 256 rows of slack, the fp8 256x256 GEMM reads whole tiles of A without clamping the last one
    if (ctx->fp8) { ALLOC(ctx->xn8, (M + 256) * D, true); ALLOC(ctx->xs, M); ALLOC(ctx->wexp, D * std::max(D, F)); }
    ALLOC(ctx->q, M * D); ALLOC(ctx->k, M * D);
    ALLOC(ctx->vT, B * H * 64 * (size_t)ctx->Tpad, true);
    ALLOC(ctx->att, M * D); ALLOC(ctx->hid, M * F);
    ALLOC(ctx->enc_out, M * D); ALLOC(ctx->enc_out32, M * D);
    if (ctx->kv8) {
        ALLOC(ctx->crossK8, (size_t)d.dec_layers * MC * D); ALLOC(ctx->crossV8, (size_t)d.dec_layers * MC * D);
        ALLOC(ctx->crossKs, (size_t)d.dec_layers * MC * H); ALLOC(ctx->crossVs, (size_t)d.dec_layers * MC * H);
    } else {
        ALLOC(ctx->crossK, (size_t)d.dec_layers * MC * D);
        ALLOC(ctx->crossV, (size_t)d.dec_layers * MC * D);
    }
    // decoder workspaces
    ALLOC(ctx->st, 1, true);
    ALLOC(ctx->dx, B * D); ALLOC(ctx->dq, B * D); ALLOC(ctx->datt, B * D); ALLOC(ctx->dhid, B * F); ALLOC(ctx->dxn, B * D);
    ALLOC(ctx->selfK, (size_t)d.dec_layers * B * D * d.n_text_ctx, true);
    ALLOC(ctx->selfV, (size_t)d.dec_layers * B * D * d.n_text_ctx, true);
    ALLOC(ctx->logits, B * (size_t)d.vocab);
    ctx->n_parts = gemv_logits_parts(d.vocab);
    ALLOC(ctx->part_val, B * (size_t)ctx->n_parts); ALLOC(ctx->part_idx, B * (size_t)ctx->n_parts);
    ALLOC(ctx->mask_first, (size_t)d.vocab, true); ALLOC(ctx->mask_rest, (size_t)d.vocab, true);
    ALLOC(ctx->xpartial, B * H * kXattnSplit * 68); ALLOC(ctx->xtickets, B * H, true);
    const size_t seq_groups = (B + 15) / 16;                          // batches beyond 32 run as groups of 16 inside a launch
    ALLOC(ctx->skpartial, splitk_partial_floats(D) * seq_groups); ALLOC(ctx->sktickets, splitk_groups(D) * seq_groups, true);
    if (B > 32) ALLOC(ctx->dgpartial, decgemm_partial_floats((int)D, (int)F, (int)B));
    ctx->forced_cap = d.n_text_ctx;
    ALLOC(ctx->forced, B * (size_t)ctx->forced_cap);
    ALLOC(ctx->out_ids, B * (size_t)d.n_text_ctx); ALLOC(ctx->out_len, B); ALLOC(ctx->finished, B);
    return MW_OK;
}

extern "C" int mw_create(const mw_model_desc* desc, int device, int max_batch, mw_ctx** out) {
    if (!desc || !out) { g_create_error = "mw_create: null argument"; return MW_ERR_ARG; }
    *out = nullptr;
    const mw_model_desc& d = *desc;
    if (d.d_model <= 0 || d.n_heads <= 0 || d.d_model != d.n_heads * kHeadDim) { g_create_error = "mw_create: head_dim must be 64"; return MW_ERR_ARG; }
    if (d.d_model % 32 || d.ffn_dim % 32 || d.d_model > 2048 || d.ffn_dim > 5120) { g_create_error = "mw_create: d_model/ffn_dim must be multiples of 32, d_model <= 2048, ffn_dim <= 5120"; return MW_ERR_ARG; }
    if (d.n_mels % 8 || d.n_mels > 128 || d.n_mels <= 0) { g_create_error = "mw_create: n_mels must be a multiple of 8, <= 128"; return MW_ERR_ARG; }
    if (d.n_audio_ctx != kFrames / 2 || d.n_text_ctx <= 0 || d.n_text_ctx > 448) { g_create_error = "mw_create: n_audio_ctx must be 1500 and n_text_ctx <= 448"; return MW_ERR_ARG; }
    if (max_batch < 1 || max_batch > kMaxBatch) { g_create_error = "mw_create: max_batch must be in 1..512"; return MW_ERR_ARG; }
    if (d.vocab < 16 || d.enc_layers < 1 || d.dec_layers < 1) { g_create_error = "mw_create: bad vocab/layer count"; return MW_ERR_ARG; }
    if (d.weight_format != MW_WEIGHTS_F16 && d.weight_format != MW_WEIGHTS_FP8) { g_create_error = "mw_create: weight_format must be MW_WEIGHTS_F16 or MW_WEIGHTS_FP8"; return MW_ERR_ARG; }
    if (d.kv_format != MW_KV_F16 && d.kv_format != MW_KV_FP8) { g_create_error = "mw_create: kv_format must be MW_KV_F16 or MW_KV_FP8"; return MW_ERR_ARG; }
    if (d.kv_format == MW_KV_FP8 && d.weight_format != MW_WEIGHTS_FP8) { g_create_error = "mw_create: MW_KV_FP8 needs MW_WEIGHTS_FP8"; return MW_ERR_ARG; }
    if (d.weight_format == MW_WEIGHTS_FP8 && (d.d_model % 128 || d.ffn_dim % 128)) { g_create_error = "mw_create: fp8 weights need d_model and ffn_dim to be multiples of 128"; return MW_ERR_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        g_create_error = "mw_create: no such HIP device " + std::to_string(device) + " (" + std::to_string(ndev) + " visible)";
        return MW_ERR_HIP;
    }
    mw_ctx* ctx = new mw_ctx();
    ctx->d = d; ctx->device = device; ctx->max_batch = max_batch; ctx->cache_seqs = max_batch;
    ctx->fp8 = d.weight_format == MW_WEIGHTS_FP8;
    ctx->kv8 = d.kv_format == MW_KV_FP8;
    auto bail = [&](int code) { g_create_error = ctx->err; mw_destroy(ctx); return code; };
    if (hipSetDevice(device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return bail(MW_ERR_HIP); }
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { ctx->err = "hipStreamCreate failed"; return bail(MW_ERR_HIP); }
    for (auto& e : ctx->ev) if (hipEventCreate(&e) != hipSuccess) { ctx->err = "hipEventCreate failed"; return bail(MW_ERR_HIP); }
    if (hipError_t e = resolve_scratch(device); e != hipSuccess) {
        ctx->err = std::string("hipGetSymbolAddress(scratch) failed: ") + hipGetErrorString(e); return bail(MW_ERR_HIP);
    }
    int r = alloc_weights(ctx);
    if (r) return bail(r);
    r = alloc_work(ctx);
    if (r) return bail(r);
    r = build_mel_tables(ctx);
    if (r) return bail(r);
    dec_kernels_prepare();
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "stream sync failed in mw_create"; return bail(MW_ERR_HIP); }
    *out = ctx;
    return MW_OK;
}

extern "C" void mw_destroy(mw_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < kSelfAttnBuckets; ++i)
        for (int j = 0; j < kGraphSlots; ++j) {
            if (ctx->graph_exec[i][j]) (void)hipGraphExecDestroy(ctx->graph_exec[i][j]);
            if (ctx->graph[i][j]) (void)hipGraphDestroy(ctx->graph[i][j]);
        }
    for (void* p : ctx->allocs) (void)hipFree(p);
    if (ctx->stage) (void)hipFree(ctx->stage);
    for (auto& e : ctx->ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : ctx->poll_ev) if (e) (void)hipEventDestroy(e);
    if (ctx->fin_host) (void)hipHostFree(ctx->fin_host);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int mw_fork(mw_ctx* parent, mw_ctx** out) { return mw_fork_sized(parent, parent ? parent->max_batch : 0, out); }

extern "C" int mw_fork_sized(mw_ctx* parent, int max_batch, mw_ctx** out) { return mw_fork_cached(parent, max_batch, max_batch, out); }

extern "C" int mw_fork_cached(mw_ctx* parent, int max_batch, int cache_sequences, mw_ctx** out) {
    if (!parent || !out) return MW_ERR_ARG;
    *out = nullptr;
    if (!parent->finalized) return fail(parent, MW_ERR_STATE, "mw_fork: the parent must be finalized first");
    if (parent->parent) return fail(parent, MW_ERR_ARG, "mw_fork: fork from the original engine, not from a fork");
    if (max_batch < 1 || max_batch > kMaxBatch) return fail(parent, MW_ERR_ARG, "mw_fork_sized: max_batch must be in 1..512");
    if (cache_sequences < 1 || cache_sequences > max_batch) return fail(parent, MW_ERR_ARG, "mw_fork_cached: cache_sequences must be in 1..max_batch");
    mw_ctx* ctx = new mw_ctx();
    ctx->d = parent->d; ctx->device = parent->device; ctx->max_batch = max_batch; ctx->cache_seqs = cache_sequences;
    ctx->use_graph = parent->use_graph; ctx->check_every = parent->check_every; ctx->use_prefill = parent->use_prefill;
    ctx->skip_done = parent->skip_done; ctx->xattn_stream_from = parent->xattn_stream_from;
    ctx->xattn_stream_per_cu = parent->xattn_stream_per_cu;
    ctx->dec_gemm_fat = parent->dec_gemm_fat;
    ctx->parent = parent; ctx->finalized = true;
    // shared, read-only on the hot path
    ctx->conv1_w = parent->conv1_w; ctx->conv1_b = parent->conv1_b; ctx->conv2_w = parent->conv2_w; ctx->conv2_b = parent->conv2_b;
    ctx->enc_pos = parent->enc_pos; ctx->enc = parent->enc; ctx->enc_ln = parent->enc_ln;
    ctx->tok_emb = parent->tok_emb; ctx->dec_pos = parent->dec_pos; ctx->dec = parent->dec; ctx->dec_ln = parent->dec_ln;
    ctx->ckv_w = parent->ckv_w; ctx->ckv_b = parent->ckv_b; ctx->ckv_w8 = parent->ckv_w8; ctx->ckv_ws = parent->ckv_ws;
    ctx->fp8 = parent->fp8; ctx->kv8 = parent->kv8;
    ctx->basis_cos = parent->basis_cos; ctx->basis_sin = parent->basis_sin; ctx->filt = parent->filt; ctx->n_mels_pad = parent->n_mels_pad;
    ctx->prompt = parent->prompt; ctx->eot = parent->eot; ctx->max_new = parent->max_new; ctx->opts_set = parent->opts_set;
    ctx->can_end = parent->can_end;
    auto bail = [&](int code) { parent->err = "mw_fork: " + ctx->err; mw_destroy(ctx); return code; };
    if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return bail(MW_ERR_HIP); }
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { ctx->err = "hipStreamCreate failed"; return bail(MW_ERR_HIP); }
    for (auto& e : ctx->ev) if (hipEventCreate(&e) != hipSuccess) { ctx->err = "hipEventCreate failed"; return bail(MW_ERR_HIP); }
    int r = alloc_work(ctx);
    if (r) return bail(r);
    const size_t V = ctx->d.vocab;
    if (hipMemcpyAsync(ctx->mask_first, parent->mask_first, V, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(ctx->mask_rest, parent->mask_rest, V, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "copying the suppress masks failed"; return bail(MW_ERR_HIP); }
    *out = ctx;
    return MW_OK;
}

// Captured decode steps carry launch arguments that depend on options (whether the cross-attention reads the
// ended flags): drop them when such an option changes; the next decode captures again.
static int drop_graphs(mw_ctx* ctx) {
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < kSelfAttnBuckets; ++i)
        for (int j = 0; j < kGraphSlots; ++j) {
            if (ctx->graph_exec[i][j]) { (void)hipGraphExecDestroy(ctx->graph_exec[i][j]); ctx->graph_exec[i][j] = nullptr; }
            if (ctx->graph[i][j]) { (void)hipGraphDestroy(ctx->graph[i][j]); ctx->graph[i][j] = nullptr; }
        }
    return MW_OK;
}

extern "C" const char* mw_last_error(const mw_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

extern "C" int mw_set_option(mw_ctx* ctx, const char* key, int value) {
    if (!ctx || !key) return MW_ERR_ARG;
    if (!strcmp(key, "graph")) ctx->use_graph = value != 0;
    else if (!strcmp(key, "check_every")) ctx->check_every = value < 1 ? 1 : value;
    else if (!strcmp(key, "prefill")) ctx->use_prefill = value != 0;
    else if (!strcmp(key, "skip_done")) {
        if (ctx->skip_done != (value != 0)) { int r = drop_graphs(ctx); if (r) return r; }
        ctx->skip_done = value != 0;
    }
    else if (!strcmp(key, "xattn_stream_per_cu")) {
        if (value < 0 || value > 8) return fail(ctx, MW_ERR_ARG, "mw_set_option: xattn_stream_per_cu must be in 0..8");
        if (ctx->xattn_stream_per_cu != value) { int r = drop_graphs(ctx); if (r) return r; }
        ctx->xattn_stream_per_cu = value;
    }
    else if (!strcmp(key, "dec_gemm_fat")) {
        if (ctx->dec_gemm_fat != (value != 0)) { int r = drop_graphs(ctx); if (r) return r; }
        ctx->dec_gemm_fat = value != 0;
    }
    else if (!strcmp(key, "xattn_stream_from")) {
        if (ctx->xattn_stream_from != value) { int r = drop_graphs(ctx); if (r) return r; }
        ctx->xattn_stream_from = value;
    }
    else return fail(ctx, MW_ERR_ARG, std::string("mw_set_option: unknown key ") + key);
    return MW_OK;
}

extern "C" int mw_synchronize(mw_ctx* ctx) {
    if (!ctx) return MW_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return MW_OK;
}

// ---------------------------------------------------------------------------------------------
// weights
// ---------------------------------------------------------------------------------------------
// conv_c > 0: permute [D][C][3] -> [D][3][C].  dst8 != null: quantise [rows][cols] to e4m3 codes + row scales.
struct Slot { void* dst; long count; int to_f16; int conv_c; uint8_t* dst8 = nullptr; float* sdst = nullptr; int rows = 0, cols = 0; };

static bool resolve(mw_ctx* ctx, const std::string& name, Slot* s, std::vector<int64_t>* shape) {
    const mw_model_desc& d = ctx->d;
    const long D = d.d_model, F = d.ffn_dim;
    auto mat = [&](f16* p, long r, long c) { *s = {p, r * c, 1, 0}; *shape = {r, c}; return true; };
    // a Linear weight (or `sec`-th [r][c] block of a stacked one): fp16, or e4m3 codes + scales in an fp8 engine
    auto lin = [&](LinearW& l, long sec, long r, long c) {
        if (!ctx->fp8) return mat(l.w + sec * r * c, r, c);
        *s = {nullptr, r * c, 0, 0, l.w8 + sec * r * c, l.ws + sec * r, (int)r, (int)c};
        *shape = {r, c};
        return true;
    };
    auto vec = [&](float* p, long n) { *s = {p, n, 0, 0}; *shape = {n}; return true; };
    auto norm = [&](NormW& n, const std::string& leaf) {
        if (leaf == "weight") return vec(n.g, D);
        if (leaf == "bias") return vec(n.b, D);
        return false;
    };
    if (name == "model.encoder.conv1.weight") { *s = {ctx->conv1_w, D * d.n_mels * 3, 1, d.n_mels}; *shape = {D, d.n_mels, 3}; return true; }
    if (name == "model.encoder.conv1.bias") return vec(ctx->conv1_b, D);
    if (name == "model.encoder.conv2.weight") { *s = {ctx->conv2_w, D * D * 3, 1, (int)D}; *shape = {D, D, 3}; return true; }
    if (name == "model.encoder.conv2.bias") return vec(ctx->conv2_b, D);
    if (name == "model.encoder.embed_positions.weight") { *s = {ctx->enc_pos, (long)d.n_audio_ctx * D, 0, 0}; *shape = {d.n_audio_ctx, D}; return true; }
    if (name == "model.encoder.layer_norm.weight") return vec(ctx->enc_ln.g, D);
    if (name == "model.encoder.layer_norm.bias") return vec(ctx->enc_ln.b, D);
    if (name == "model.decoder.embed_tokens.weight") return mat(ctx->tok_emb, d.vocab, D);
    if (name == "model.decoder.embed_positions.weight") return mat(ctx->dec_pos, d.n_text_ctx, D);
    if (name == "model.decoder.layer_norm.weight") return vec(ctx->dec_ln.g, D);
    if (name == "model.decoder.layer_norm.bias") return vec(ctx->dec_ln.b, D);

    const bool is_enc = name.rfind("model.encoder.layers.", 0) == 0;
    const bool is_dec = name.rfind("model.decoder.layers.", 0) == 0;
    if (!is_enc && !is_dec) return false;
    const size_t p0 = strlen("model.encoder.layers.");
    const size_t p1 = name.find('.', p0);
    if (p1 == std::string::npos) return false;
    int li = 0;
    for (size_t i = p0; i < p1; ++i) { if (name[i] < '0' || name[i] > '9') return false; li = li * 10 + (name[i] - '0'); }
    if (li >= (is_enc ? d.enc_layers : d.dec_layers)) return false;
    const std::string rest = name.substr(p1 + 1);
    const size_t dot = rest.rfind('.');
    if (dot == std::string::npos) return false;
    const std::string mod = rest.substr(0, dot), leaf = rest.substr(dot + 1);
    const bool isw = leaf == "weight", isb = leaf == "bias";
    if (!isw && !isb) return false;

    auto proj = [&](LinearW& qkv, LinearW& out, const std::string& pj) {
        int sec = pj == "q_proj" ? 0 : pj == "k_proj" ? 1 : pj == "v_proj" ? 2 : pj == "out_proj" ? 3 : -1;
        if (sec < 0) return false;
        if (sec == 3) return isw ? lin(out, 0, D, D) : vec(out.b, D);
        if (sec == 1 && isb) return false;                             // k_proj has no bias (modeling_whisper.py:279)
        return isw ? lin(qkv, sec, D, D) : vec(qkv.b + sec * D, D);
    };
    auto fc = [&](LinearW& l, long o, long i) { return isw ? lin(l, 0, o, i) : vec(l.b, o); };

    if (is_enc) {
        EncLayerW& l = ctx->enc[li];
        if (mod.rfind("self_attn.", 0) == 0) return proj(l.qkv, l.out, mod.substr(10));
        if (mod == "self_attn_layer_norm") return norm(l.ln1, leaf);
        if (mod == "final_layer_norm") return norm(l.ln2, leaf);
        if (mod == "fc1") return fc(l.fc1, F, D);
        if (mod == "fc2") return fc(l.fc2, D, F);
        return false;
    }
    DecLayerW& l = ctx->dec[li];
    if (mod.rfind("self_attn.", 0) == 0) return proj(l.qkv, l.out, mod.substr(10));
    if (mod.rfind("encoder_attn.", 0) == 0) {
        const std::string pj = mod.substr(13);
        if (pj == "q_proj") return isw ? lin(l.cq, 0, D, D) : vec(l.cq.b, D);
        if (pj == "out_proj") return isw ? lin(l.co, 0, D, D) : vec(l.co.b, D);
        LinearW ckv{ctx->ckv_w, nullptr, ctx->ckv_w8, ctx->ckv_ws};            // all layers stacked: [L][2][D][D]
        if (pj == "k_proj") { if (isb) return false; return lin(ckv, (long)li * 2, D, D); }
        if (pj == "v_proj") return isw ? lin(ckv, (long)li * 2 + 1, D, D) : vec(ctx->ckv_b + (long)li * 2 * D + D, D);
        return false;
    }
    if (mod == "self_attn_layer_norm") return norm(l.ln1, leaf);
    if (mod == "encoder_attn_layer_norm") return norm(l.ln2, leaf);
    if (mod == "final_layer_norm") return norm(l.ln3, leaf);
    if (mod == "fc1") return fc(l.fc1, F, D);
    if (mod == "fc2") return fc(l.fc2, D, F);
    return false;
}

extern "C" int mw_set_weight(mw_ctx* ctx, const char* name_c, const void* data, const int64_t* shape, int ndim, int dtype) {
    if (!ctx) return MW_ERR_ARG;
    if (ctx->parent) return fail(ctx, MW_ERR_STATE, "mw_set_weight: a fork shares its parent's weights");
    if (!name_c || !data || !shape || ndim < 1 || ndim > 3) return fail(ctx, MW_ERR_ARG, "mw_set_weight: null or bad argument");
    if (dtype < MW_F32 || dtype > MW_BF16) return fail(ctx, MW_ERR_ARG, "mw_set_weight: dtype must be MW_F32, MW_F16 or MW_BF16");
    const std::string name(name_c);
    if (name == "proj_out.weight") return MW_OK;                     // tied to embed_tokens
    Slot s; std::vector<int64_t> want;
    if (!resolve(ctx, name, &s, &want)) return fail(ctx, MW_ERR_NAME, "mw_set_weight: unknown tensor name '" + name + "'");
    if ((int)want.size() != ndim) return fail(ctx, MW_ERR_ARG, "mw_set_weight: wrong rank for '" + name + "'");
    for (int i = 0; i < ndim; ++i)
        if (want[i] != shape[i]) return fail(ctx, MW_ERR_ARG, "mw_set_weight: wrong shape for '" + name + "' (dim " + std::to_string(i) + ": got " + std::to_string(shape[i]) + ", want " + std::to_string(want[i]) + ")");
    if (!ctx->seen.insert(name).second) return fail(ctx, MW_ERR_NAME, "mw_set_weight: '" + name + "' set twice");
    HIPCHK(hipSetDevice(ctx->device));
    const size_t esz = dtype == MW_F32 ? 4 : 2;
    const void* src = data;
    if (!is_device_ptr(data)) {
        const size_t bytes = (size_t)s.count * esz;
        if (bytes > ctx->stage_bytes) {
            HIPCHK(hipStreamSynchronize(ctx->stream));
            if (ctx->stage) (void)hipFree(ctx->stage);
            ctx->stage = nullptr; ctx->stage_bytes = 0;
            HIPCHK(hipMalloc(&ctx->stage, bytes));
            ctx->stage_bytes = bytes;
        }
        HIPCHK(hipStreamSynchronize(ctx->stream));                  // previous conversion may still read the stage
        HIPCHK(hipMemcpy(ctx->stage, data, bytes, hipMemcpyHostToDevice));
        src = ctx->stage;
    }
    if (s.dst8) launch_quant_rows(src, dtype, s.dst8, s.sdst, s.rows, s.cols, ctx->stream);
    else if (s.conv_c > 0) launch_conv_permute(src, dtype, reinterpret_cast<f16*>(s.dst), (int)want[0], s.conv_c, ctx->stream);
    else launch_convert(src, dtype, s.dst, s.to_f16, s.count, ctx->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));                      // the caller may free `data` on return
    if (name == "model.encoder.embed_positions.weight") ctx->enc_pos_set = true;
    return MW_OK;
}

extern "C" int mw_finalize(mw_ctx* ctx) {
    if (!ctx) return MW_ERR_ARG;
    if (ctx->parent) return MW_OK;
    const mw_model_desc& d = ctx->d;
    std::vector<std::string> need = {"model.encoder.conv1.weight", "model.encoder.conv1.bias", "model.encoder.conv2.weight",
                                     "model.encoder.conv2.bias", "model.encoder.layer_norm.weight", "model.encoder.layer_norm.bias",
                                     "model.decoder.embed_tokens.weight", "model.decoder.embed_positions.weight",
                                     "model.decoder.layer_norm.weight", "model.decoder.layer_norm.bias"};
    auto attn = [&](const std::string& p) {
        for (const char* pj : {"q_proj", "v_proj", "out_proj"}) { need.push_back(p + "." + pj + ".weight"); need.push_back(p + "." + pj + ".bias"); }
        need.push_back(p + ".k_proj.weight");
    };
    auto two = [&](const std::string& p) { need.push_back(p + ".weight"); need.push_back(p + ".bias"); };
    for (int i = 0; i < d.enc_layers; ++i) {
        const std::string p = "model.encoder.layers." + std::to_string(i);
        attn(p + ".self_attn"); two(p + ".self_attn_layer_norm"); two(p + ".final_layer_norm"); two(p + ".fc1"); two(p + ".fc2");
    }
    for (int i = 0; i < d.dec_layers; ++i) {
        const std::string p = "model.decoder.layers." + std::to_string(i);
        attn(p + ".self_attn"); attn(p + ".encoder_attn");
        two(p + ".self_attn_layer_norm"); two(p + ".encoder_attn_layer_norm"); two(p + ".final_layer_norm"); two(p + ".fc1"); two(p + ".fc2");
    }
    for (const auto& n : need)
        if (!ctx->seen.count(n)) return fail(ctx, MW_ERR_MISSING, "mw_finalize: weight '" + n + "' was never set");
    HIPCHK(hipSetDevice(ctx->device));
    if (!ctx->enc_pos_set) {     // sinusoid table (modeling_whisper.py:55-64): sin half then cos half
        const int T = d.n_audio_ctx, D = d.d_model, half = D / 2;
        std::vector<float> tab((size_t)T * D);
        const double inc = log(10000.0) / (half - 1);
        for (int t = 0; t < T; ++t)
            for (int c = 0; c < half; ++c) {
                const float inv = expf((float)(-inc * c));
                const float a = (float)t * inv;
                tab[(size_t)t * D + c] = sinf(a);
                tab[(size_t)t * D + half + c] = cosf(a);
            }
        HIPCHK(hipMemcpy(ctx->enc_pos, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    }
    if (ctx->stage) { (void)hipFree(ctx->stage); ctx->stage = nullptr; ctx->stage_bytes = 0; }
    ctx->finalized = true;
    return MW_OK;
}

extern "C" int mw_set_decode_opts(mw_ctx* ctx, const mw_decode_opts* o) {
    if (!ctx) return MW_ERR_ARG;
    if (!o || !o->prompt || o->n_prompt < 1) return fail(ctx, MW_ERR_ARG, "mw_set_decode_opts: a prompt of at least one id is required");
    const int V = ctx->d.vocab;
    if (o->max_new < 1 || o->n_prompt + o->max_new > ctx->d.n_text_ctx)
        return fail(ctx, MW_ERR_ARG, "mw_set_decode_opts: n_prompt + max_new must fit n_text_ctx");
    if (o->eot < 0 || o->eot >= V) return fail(ctx, MW_ERR_ARG, "mw_set_decode_opts: eot out of range");
    auto in_range = [&](const int32_t* a, int n) { for (int i = 0; i < n; ++i) if (a[i] < 0 || a[i] >= V) return false; return true; };
    if (!in_range(o->prompt, o->n_prompt) || (o->n_suppress && (!o->suppress || !in_range(o->suppress, o->n_suppress))) ||
        (o->n_begin_suppress && (!o->begin_suppress || !in_range(o->begin_suppress, o->n_begin_suppress))))
        return fail(ctx, MW_ERR_ARG, "mw_set_decode_opts: token id out of range");
    std::vector<uint8_t> rest(V, 0), first;
    for (int i = 0; i < o->n_suppress; ++i) rest[o->suppress[i]] = 1;
    first = rest;
    for (int i = 0; i < o->n_begin_suppress; ++i) first[o->begin_suppress[i]] = 1;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(ctx->mask_rest, rest.data(), V, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->mask_first, first.data(), V, hipMemcpyHostToDevice));
    ctx->prompt.assign(o->prompt, o->prompt + o->n_prompt);
    ctx->eot = o->eot; ctx->max_new = o->max_new; ctx->opts_set = true;
    const bool can_end = rest[o->eot] == 0;     // end-of-text suppressed at every position: no sequence ever ends early
    if (can_end != ctx->can_end) { int r = drop_graphs(ctx); if (r) return r; }
    ctx->can_end = can_end;
    return MW_OK;
}

// ---------------------------------------------------------------------------------------------
// hot path
// ---------------------------------------------------------------------------------------------
// cached: every sequence of the batch has a slot of its own in the cross-attention K/V cache (encoder passes, greedy decoding)
static int check_batch(mw_ctx* ctx, int batch, const char* who, bool cached = true) {
    if (!ctx->finalized) return fail(ctx, MW_ERR_STATE, std::string(who) + ": mw_finalize has not succeeded yet");
    const int limit = cached ? ctx->cache_seqs : ctx->max_batch;
    if (batch < 1 || batch > limit) return fail(ctx, MW_ERR_ARG, std::string(who) + ": batch must be in 1.." + std::to_string(limit));
    return MW_OK;
}

static int run_logmel(mw_ctx* ctx, const float* pcm_dev, int B) {
    MelTables t{ctx->basis_cos, ctx->basis_sin, ctx->filt, ctx->d.n_mels, ctx->n_mels_pad};
    launch_mel_dft(pcm_dev, B, t, ctx->logmel, ctx->blockmax, ctx->stream);
    launch_mel_norm(ctx->logmel, ctx->blockmax, B, ctx->d.n_mels, ctx->melT, ctx->stream);
    HIPCHK(hipGetLastError());
    ctx->mel_batch = B;
    return MW_OK;
}

extern "C" int mw_logmel(mw_ctx* ctx, const float* pcm, int batch, float* mel_out) {
    if (!ctx) return MW_ERR_ARG;
    int r = check_batch(ctx, batch, "mw_logmel", false);
    if (r) return r;
    if (!pcm) return fail(ctx, MW_ERR_ARG, "mw_logmel: pcm is null");
    HIPCHK(hipSetDevice(ctx->device));
    const float* src = pcm;
    if (!is_device_ptr(pcm)) {
        HIPCHK(hipMemcpyAsync(ctx->pcm_dev, pcm, (size_t)batch * kChunkSamples * 4, hipMemcpyHostToDevice, ctx->stream));
        src = ctx->pcm_dev;
    }
    HIPCHK(hipEventRecord(ctx->ev[0], ctx->stream));
    r = run_logmel(ctx, src, batch);
    if (r) return r;
    HIPCHK(hipEventRecord(ctx->ev[1], ctx->stream));
    if (mel_out)
        HIPCHK(hipMemcpyAsync(mel_out, ctx->logmel, (size_t)batch * ctx->d.n_mels * kFrames * 4, hipMemcpyDefault, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipEventElapsedTime(&ctx->stage_ms[0], ctx->ev[0], ctx->ev[1]));
    return MW_OK;
}

// mw_logmel for callers that cannot allocate: everything on the device, nothing allocated or freed.  The raw
// log-mel goes straight into the caller's output, which the second launch clamps and scales in place; the
// per-block maxima between the two launches live in this GPU's `scratch` array.
extern "C" int mw_logmel_noalloc(mw_ctx* ctx, const float* pcm, int batch, float* mel_out) {
    if (!ctx) return MW_ERR_ARG;
    int r = check_batch(ctx, batch, "mw_logmel_noalloc", false);
    if (r) return r;
    if (batch > kNoAllocMaxBatch)
        return fail(ctx, MW_ERR_ARG, "mw_logmel_noalloc: batch exceeds the static scratch (" + std::to_string(kNoAllocMaxBatch) + " chunks)");
    if (!pcm || !mel_out) return fail(ctx, MW_ERR_ARG, "mw_logmel_noalloc: pcm and mel_out are required");
    HIPCHK(hipSetDevice(ctx->device));
    if (!is_device_ptr(pcm) || !is_device_ptr(mel_out))
        return fail(ctx, MW_ERR_ARG, "mw_logmel_noalloc: pcm and mel_out must be device memory (staging host buffers would allocate)");
    float* blockmax = g_scratch_ptr[ctx->device];           // resolved by mw_create on this device
    if (!blockmax) return fail(ctx, MW_ERR_STATE, "mw_logmel_noalloc: the scratch address of this device was never resolved");
    // held until this call's two launches have finished (the call blocks on its stream anyway): two engine threads on
    // one GPU would otherwise overwrite each other's block maxima
    std::lock_guard<std::mutex> in_use(g_scratch_in_use[ctx->device]);
    MelTables t{ctx->basis_cos, ctx->basis_sin, ctx->filt, ctx->d.n_mels, ctx->n_mels_pad};
    HIPCHK(hipEventRecord(ctx->ev[0], ctx->stream));
    launch_mel_dft(pcm, batch, t, mel_out, blockmax, ctx->stream);           // writes every maximum it later reads: nothing to clear
    launch_mel_norm(mel_out, blockmax, batch, ctx->d.n_mels, ctx->melT, ctx->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ctx->ev[1], ctx->stream));
    ctx->mel_batch = batch;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipEventElapsedTime(&ctx->stage_ms[0], ctx->ev[0], ctx->ev[1]));
    return MW_OK;
}

static GemmParams gemm_base(const f16* A, long lda, const f16* W, long ldw, int M, int N, int K, const float* bias) {
    GemmParams p{};
    p.A = A; p.lda = lda; p.a_rows_per_batch = M > 0 ? M : 1; p.a_batch_stride = 0;
    p.W = W; p.ldw = ldw; p.M = M; p.N = N; p.K = K; p.bias = bias;
    p.scale = 1.f; p.scale_ncols = 0; p.gelu = 0; p.out_mode = OUT_F16;
    p.c_rows_per_batch = M > 0 ? M : 1; p.c_batch_stride = 0;
    return p;
}

static int run_encoder(mw_ctx* ctx, int B) {
    const mw_model_desc& d = ctx->d;
    const int D = d.d_model, F = d.ffn_dim, H = d.n_heads, T = d.n_audio_ctx, C = d.n_mels;
    const int M = B * T;
    hipStream_t s = ctx->stream;
    const float qscale = 1.0f / sqrtf((float)kHeadDim);
    // conv1 + GELU: rows are frames; A row (b,t) = melT[b][t .. t+2][:] (3*C contiguous halfs)
    {
        GemmParams p = gemm_base(ctx->melT, C, ctx->conv1_w, 3 * C, B * kFrames, D, 3 * C, ctx->conv1_b);
        p.a_rows_per_batch = kFrames; p.a_batch_stride = (long)(kFrames + 2) * C;
        p.gelu = 1; p.C = ctx->h1 + D; p.ldc = D; p.c_rows_per_batch = kFrames; p.c_batch_stride = (long)(kFrames + 2) * D;
        launch_gemm(p, s);
    }
    // x = positions; x += GELU(conv2(h1)) with stride 2: A row (b,t') = h1[b][2t' .. 2t'+2][:]
    launch_bcast_rows(ctx->enc_pos, ctx->x, B, T, D, s);
    {
        GemmParams p = gemm_base(ctx->h1, 2 * D, ctx->conv2_w, 3 * D, M, D, 3 * D, ctx->conv2_b);
        p.a_rows_per_batch = T; p.a_batch_stride = (long)(kFrames + 2) * D;
        p.gelu = 1; p.out_mode = OUT_RESID_F32; p.C = ctx->x; p.ldc = D;
        launch_gemm(p, s);
    }
    // fp8 engine: the two products fed by a LayerNorm (QKV, MLP in) run on the fp8 MFMA, their input rows
    // quantised by the LayerNorm launch itself; the two residual projections take fp16 activations as they
    // come (attention output, GELU output) and the weight codes expanded to fp16, scales in the epilogue.
    const bool f8 = ctx->fp8;
    auto resid = [&](const f16* A, int K, const LinearW& w) {
        GemmParams p = gemm_base(A, K, w.w, K, M, D, K, w.b);
        if (f8) { launch_fp8_to_f16(w.w8, ctx->wexp, (long)D * K, s); p.W = ctx->wexp; p.w_scale = w.ws; }
        p.out_mode = OUT_RESID_F32; p.C = ctx->x; p.ldc = D;
        launch_gemm(p, s);
    };
    auto norm_in = [&](const NormW& n) -> const f16* {
        if (f8) { launch_layernorm_fp8(ctx->x, n.g, n.b, ctx->xn8, ctx->xs, nullptr, M, D, s); return reinterpret_cast<const f16*>(ctx->xn8); }
        launch_layernorm(ctx->x, n.g, n.b, ctx->xn, nullptr, M, D, s);
        return ctx->xn;
    };
    auto fed_by_norm = [&](const f16* A, const LinearW& w, int N) {
        GemmParams p = gemm_base(A, D, f8 ? reinterpret_cast<const f16*>(w.w8) : w.w, D, M, N, D, w.b);
        if (f8) { p.f8 = 1; p.a_scale = ctx->xs; p.w_scale = w.ws; }
        return p;
    };
    for (int li = 0; li < d.enc_layers; ++li) {
        const EncLayerW& l = ctx->enc[li];
        {
            GemmParams p = fed_by_norm(norm_in(l.ln1), l.qkv, 3 * D);
            p.scale = qscale; p.scale_ncols = D; p.out_mode = OUT_QKV_SPLIT;
            p.q = ctx->q; p.k = ctx->k; p.v = ctx->vT; p.T = T; p.Tpad = ctx->Tpad; p.H = H; p.D = D;
            launch_gemm(p, s);
        }
        launch_enc_attention(ctx->q, ctx->k, ctx->vT, ctx->att, B, H, T, ctx->Tpad, s);
        resid(ctx->att, D, l.out);
        {
            GemmParams p = fed_by_norm(norm_in(l.ln2), l.fc1, F);
            p.gelu = 1; p.C = ctx->hid; p.ldc = F;
            launch_gemm(p, s);
        }
        resid(ctx->hid, F, l.fc2);
    }
    if (f8) launch_layernorm_fp8(ctx->x, ctx->enc_ln.g, ctx->enc_ln.b, ctx->xn8, ctx->xs, ctx->enc_out32, M, D, s);
    else launch_layernorm(ctx->x, ctx->enc_ln.g, ctx->enc_ln.b, ctx->enc_out, ctx->enc_out32, M, D, s);
    HIPCHK(hipGetLastError());
    return MW_OK;
}

// Cross-attention K/V of the B sequences encoded last, for every decoder layer, into the cache of `dst` (this
// context, or another context of the same model on the same GPU) at sequences slot0 .. slot0 + B - 1.
static int run_cross_kv(mw_ctx* ctx, int B, mw_ctx* dst = nullptr, int slot0 = 0) {
    if (!dst) dst = ctx;
    const mw_model_desc& d = ctx->d;
    const int D = d.d_model, H = d.n_heads, T = d.n_audio_ctx, M = B * T;
    const size_t per_layer = (size_t)dst->cache_seqs * T * D;
    const size_t at_slot = (size_t)slot0 * T * D;              // [L][cache_seqs][H][T][64]: a sequence is H*T*64 = T*D elements
    for (int li = 0; li < d.dec_layers; ++li) {
        GemmParams p = gemm_base(ctx->enc_out, D, ctx->ckv_w + (size_t)li * 2 * D * D, D, M, 2 * D, D, ctx->ckv_b + (size_t)li * 2 * D);
        if (ctx->fp8) {           // encoder output rows as the final LayerNorm quantised them (xn8, xs)
            p.A = reinterpret_cast<const f16*>(ctx->xn8); p.W = reinterpret_cast<const f16*>(ctx->ckv_w8 + (size_t)li * 2 * D * D);
            p.f8 = 1; p.a_scale = ctx->xs; p.w_scale = ctx->ckv_ws + (size_t)li * 2 * D;
        }
        p.out_mode = OUT_KV_SPLIT;
        if (ctx->kv8) {
            const size_t per_layer_rows = (size_t)dst->cache_seqs * T * H, rows_at_slot = (size_t)slot0 * T * H;
            p.out_mode = OUT_KV_SPLIT8; p.k8 = dst->crossK8 + li * per_layer + at_slot; p.v8 = dst->crossV8 + li * per_layer + at_slot;
            p.ks = dst->crossKs + li * per_layer_rows + rows_at_slot; p.vs = dst->crossVs + li * per_layer_rows + rows_at_slot;
        } else {
            p.k = dst->crossK + li * per_layer + at_slot; p.v = dst->crossV + li * per_layer + at_slot;
        }
        p.T = T; p.Tpad = T; p.H = H; p.D = D;
        launch_gemm(p, ctx->stream);
    }
    HIPCHK(hipGetLastError());
    if (dst == ctx) ctx->enc_batch = B;
    return MW_OK;
}

extern "C" int mw_encode(mw_ctx* ctx, const float* mel, int batch, float* enc_out) {
    if (!ctx) return MW_ERR_ARG;
    int r = check_batch(ctx, batch, "mw_encode");
    if (r) return r;
    HIPCHK(hipSetDevice(ctx->device));
    if (mel) {
        const float* src = mel;
        if (!is_device_ptr(mel)) {
            HIPCHK(hipMemcpyAsync(ctx->logmel, mel, (size_t)batch * ctx->d.n_mels * kFrames * 4, hipMemcpyHostToDevice, ctx->stream));
            src = ctx->logmel;
        }
        launch_mel_import(src, batch, ctx->d.n_mels, ctx->melT, ctx->stream);
        ctx->mel_batch = batch;
    } else if (ctx->mel_batch != batch) {
        return fail(ctx, MW_ERR_STATE, "mw_encode: no features for this batch; call mw_logmel first or pass mel");
    }
    HIPCHK(hipEventRecord(ctx->ev[1], ctx->stream));
    r = run_encoder(ctx, batch);
    if (r) return r;
    HIPCHK(hipEventRecord(ctx->ev[2], ctx->stream));
    r = run_cross_kv(ctx, batch);
    if (r) return r;
    HIPCHK(hipEventRecord(ctx->ev[3], ctx->stream));
    if (enc_out)
        HIPCHK(hipMemcpyAsync(enc_out, ctx->enc_out32, (size_t)batch * ctx->d.n_audio_ctx * ctx->d.d_model * 4, hipMemcpyDefault, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipEventElapsedTime(&ctx->stage_ms[1], ctx->ev[1], ctx->ev[2]));
    HIPCHK(hipEventElapsedTime(&ctx->stage_ms[2], ctx->ev[2], ctx->ev[3]));
    return MW_OK;
}

// Cross-attention of `rows` query rows over layer li's cached encoder K/V, in whichever format the cache has.
static void cross_attention(mw_ctx* ctx, int li, const f16* q, f16* out, int rows, int kv_div = 1, int row0 = 0,
                            const int* done = nullptr, int done_div = 1) {
    const mw_model_desc& d = ctx->d;
    const size_t per_layer = (size_t)ctx->cache_seqs * d.n_audio_ctx * d.d_model;
    // Several hypotheses per recording (beam search, best-of-n samples): the rows of a recording share its K/V, so one
    // workgroup per (recording, head) reads them ONCE for all of them (the prompt prefill's kernel; the single-query
    // kernel reads them once per hypothesis).  MW_SHARED_XATTN=0: the single-query kernel.
    static const bool shared_off = getenv("MW_SHARED_XATTN") && atoi(getenv("MW_SHARED_XATTN")) == 0;
    if (!shared_off && kv_div > 1 && row0 == 0 && rows % kv_div == 0) {
        const int n_rec = rows / kv_div;
        if (ctx->kv8) {
            const size_t per_layer_rows = (size_t)ctx->cache_seqs * d.n_audio_ctx * d.n_heads;
            launch_prefill_cross_attn8(q, ctx->crossK8 + li * per_layer, ctx->crossV8 + li * per_layer, ctx->crossKs + li * per_layer_rows,
                                       ctx->crossVs + li * per_layer_rows, out, n_rec, kv_div, d.n_heads, d.n_audio_ctx, 1, ctx->stream, done,
                                       done_div);
        } else {
            launch_prefill_cross_attn(q, ctx->crossK + li * per_layer, ctx->crossV + li * per_layer, out, n_rec, kv_div, d.n_heads,
                                      d.n_audio_ctx, 1, ctx->stream, done, done_div);
        }
        return;
    }
    if (ctx->kv8) {
        const size_t per_layer_rows = (size_t)ctx->cache_seqs * d.n_audio_ctx * d.n_heads;
        launch_dec_cross_attn8(q, ctx->crossK8 + li * per_layer, ctx->crossV8 + li * per_layer, ctx->crossKs + li * per_layer_rows,
                               ctx->crossVs + li * per_layer_rows, out, rows, d.n_heads, d.n_audio_ctx, ctx->xpartial, ctx->xtickets,
                               ctx->stream, kv_div, row0, done, done_div, ctx->xattn_stream_from, ctx->xattn_stream_per_cu);
    } else {
        launch_dec_cross_attn(q, ctx->crossK + li * per_layer, ctx->crossV + li * per_layer, out, rows, d.n_heads, d.n_audio_ctx,
                              ctx->xpartial, ctx->xtickets, ctx->stream, kv_div, row0, done, done_div, ctx->xattn_stream_from, ctx->xattn_stream_per_cu);
    }
}

// The end of a step: final LayerNorm (the same launch advances the position counter for the launches after
// it), logits with the partial arg-max, token bookkeeping.
static void enqueue_step_tail(mw_ctx* ctx, int B, bool want_logits, const FinalizeParams& fp, const BeamParams* beam = nullptr,
                              const SampleParams* samp = nullptr) {
    const mw_model_desc& d = ctx->d;
    const int D = d.d_model;
    hipStream_t s = ctx->stream;
    launch_layernorm(ctx->dx, ctx->dec_ln.g, ctx->dec_ln.b, ctx->dxn, nullptr, B, D, s, &ctx->st->pos);
    {
        GemvParams p{};
        p.B = B; p.N = d.vocab; p.K = D; p.W = ctx->tok_emb; p.scale = 1.f; p.H = d.n_heads; p.D = D; p.Tctx = d.n_text_ctx;
        p.st = ctx->st; p.nv = 16;
        p.xin16 = ctx->dxn; p.out_mode = GV_LOGITS; p.out32 = want_logits ? ctx->logits : nullptr;
        p.mask_first = ctx->mask_first; p.mask_rest = ctx->mask_rest;
        p.part_val = ctx->part_val; p.part_idx = ctx->part_idx; p.n_parts = ctx->n_parts;
        launch_gemv(p, s);
    }
    if (beam) launch_beam_step(*beam, s);
    else if (samp) launch_sample_step(*samp, s);
    else launch_dec_finalize(fp, B, s);
}

// The launches of decoder layer `li` for one position: six weight-streaming projections (GEMVs) and the two
// attention launches.  `parts`: kLayerGemv, kLayerAttn or both (the measurement entry points time them apart).
enum { kLayerGemv = 1, kLayerAttn = 2 };
static void enqueue_layer(mw_ctx* ctx, int li, int B, bool want_logits, const FinalizeParams& fp, int max_keys, const BeamParams* beam,
                          int parts = kLayerGemv | kLayerAttn, const SampleParams* samp = nullptr) {
    const mw_model_desc& d = ctx->d;
    const int D = d.d_model, F = d.ffn_dim, H = d.n_heads, Tc = d.n_text_ctx;
    hipStream_t s = ctx->stream;
    const float qscale = 1.0f / sqrtf((float)kHeadDim);
    const size_t self_layer = (size_t)ctx->max_batch * D * Tc;
    const bool gemv = parts & kLayerGemv, attn = parts & kLayerAttn;
    auto base = [&](int N, int K, const LinearW& w) {
        GemvParams p{};
        p.B = B; p.N = N; p.K = K; p.W = w.w; p.W8 = w.w8; p.w_scale = w.ws; p.bias = w.b; p.scale = 1.f; p.scale_ncols = 0; p.gelu = 0;
        p.H = H; p.D = D; p.Tctx = Tc; p.st = ctx->st; p.ln_scratch = ctx->dxn; p.nv = 16;
        p.sk_partial = ctx->skpartial; p.sk_partial_floats = splitk_partial_floats(D);
        p.sk_tickets = ctx->sktickets; p.sk_groups = splitk_groups(D);
        p.dg_partial = ctx->dgpartial; p.dg_partial_floats = decgemm_partial_floats(D, F, ctx->max_batch); p.dg_groups = splitk_groups(D) * ((ctx->max_batch + 15) / 16); p.dg_fat = ctx->dec_gemm_fat;
        return p;
    };
    const DecLayerW& l = ctx->dec[li];
    f16* sk = ctx->selfK + li * self_layer; f16* sv = ctx->selfV + li * self_layer;
    if (gemv) {   // LN1 + QKV; k/v go straight into the cache at the current position
        GemvParams p = base(3 * D, D, l.qkv);
        p.x32 = ctx->dx; p.ln_g = l.ln1.g; p.ln_b = l.ln1.b; p.scale = qscale; p.scale_ncols = D;
        p.out_mode = GV_QKV; p.out16 = ctx->dq; p.kcache = sk; p.vcache = sv;
        launch_gemv(p, s);
    }
    if (attn) launch_dec_self_attn(ctx->dq, sk, sv, ctx->datt, B, H, Tc, ctx->st, s, beam ? beam->src_rows : nullptr, max_keys);
    if (gemv) { GemvParams p = base(D, D, l.out); p.xin16 = ctx->datt; p.out_mode = GV_RESID; p.out32 = ctx->dx; launch_gemv(p, s); }
    if (gemv) {   // LN2 + cross-attention query
        GemvParams p = base(D, D, l.cq);
        p.x32 = ctx->dx; p.ln_g = l.ln2.g; p.ln_b = l.ln2.b; p.scale = qscale; p.scale_ncols = D;
        p.out_mode = GV_F16; p.out16 = ctx->dq;
        launch_gemv(p, s);
    }
    // sequences that have ended (recordings whose beam search is complete) are skipped: their next token is
    // end-of-text whatever the step computes
    // (not when the caller asked for the logits of every position: those are then the model's own throughout)
    if (attn) {
        if (beam) cross_attention(ctx, li, ctx->dq, ctx->datt, B, beam->beam, 0, (ctx->skip_done && ctx->can_end) ? beam->completed : nullptr, beam->beam);
        // samples: the cache of recording r / best_of, ended flags per sequence
        else if (samp) cross_attention(ctx, li, ctx->dq, ctx->datt, B, samp->best_of, 0, samp->skip_ended ? samp->finished : nullptr, 1);
        else cross_attention(ctx, li, ctx->dq, ctx->datt, B, 1, 0, (ctx->skip_done && ctx->can_end && !want_logits) ? fp.finished : nullptr, 1);
    }
    if (gemv) {
        {   GemvParams p = base(D, D, l.co); p.xin16 = ctx->datt; p.out_mode = GV_RESID; p.out32 = ctx->dx; launch_gemv(p, s); }
        {   // LN3 + fc1 + GELU
            GemvParams p = base(F, D, l.fc1);
            p.x32 = ctx->dx; p.ln_g = l.ln3.g; p.ln_b = l.ln3.b; p.gelu = 1; p.out_mode = GV_F16; p.out16 = ctx->dhid;
            launch_gemv(p, s);
        }
        {   GemvParams p = base(D, F, l.fc2); p.xin16 = ctx->dhid; p.out_mode = GV_RESID; p.out32 = ctx->dx; launch_gemv(p, s); }
    }
}

// One greedy step: 8 launches per layer + final norm + logits + token bookkeeping.
static void enqueue_step(mw_ctx* ctx, int B, bool want_logits, const FinalizeParams& fp, int max_keys, const BeamParams* beam = nullptr,
                         const SampleParams* samp = nullptr) {
    for (int li = 0; li < ctx->d.dec_layers; ++li) enqueue_layer(ctx, li, B, want_logits, fp, max_keys, beam, kLayerGemv | kLayerAttn, samp);
    enqueue_step_tail(ctx, B, want_logits, fp, beam, samp);
}

static FinalizeParams finalize_params(mw_ctx* ctx, int n_forced) {
    FinalizeParams fp{};
    fp.st = ctx->st; fp.part_val = ctx->part_val; fp.part_idx = ctx->part_idx; fp.n_parts = ctx->n_parts;
    fp.forced = ctx->forced; fp.n_forced = n_forced;
    fp.tok_emb = ctx->tok_emb; fp.pos_emb = ctx->dec_pos; fp.D = ctx->d.d_model; fp.n_text_ctx = ctx->d.n_text_ctx; fp.vocab = ctx->d.vocab;
    fp.x = ctx->dx; fp.out_ids = ctx->out_ids; fp.out_len = ctx->out_len; fp.finished = ctx->finished;
    return fp;
}

// Prompt prefill: positions 0 .. P-1 of all B sequences (ids in ctx->forced) through the decoder in one pass
// of M = B*P rows, leaving the self-attention caches filled, dx = the residual rows of position P-1 and
// st->pos = P-1, i.e. the state of a step-by-step run just before the final LayerNorm of step P-1.
// Borrows the encoder's workspaces (idle while decoding): x, xn, q (as the [M][3D] q|k|v rows), k (as the
// cross-attention queries), att, hid.  Same arithmetic as the steps (fp16 operands, fp32 accumulation, fp8
// weight codes expanded to fp16 with the row scale in the epilogue), on the GEMM kernels instead of the GEMVs.
// Beam search (beam > 1): B recordings, each prompt once; the k/v rows go to the cache rows of the recording's
// first beam, `src_rows` points every beam at them, and dx gets `beam` copies of each last row.
// kv_rep > 1 (samples): sequence b reads the cross-attention K/V of recording b / kv_rep.
static int run_prefill(mw_ctx* ctx, int B, int P, int beam = 1, int* src_rows = nullptr, int kv_rep = 1) {
    const mw_model_desc& d = ctx->d;
    const int D = d.d_model, F = d.ffn_dim, H = d.n_heads, Tc = d.n_text_ctx;
    const int M = B * P;
    hipStream_t s = ctx->stream;
    const float qscale = 1.0f / sqrtf((float)kHeadDim);
    const size_t self_layer = (size_t)ctx->max_batch * D * Tc;
    f16* qkv = ctx->q;          // [M][3D]: 3 * P <= 1500 always (P <= 448)
    f16* cq = ctx->k;           // [M][D]
    auto gemm = [&](const f16* A, int K, const LinearW& w, int N) {
        GemmParams p = gemm_base(A, K, w.w, K, M, N, K, w.b);
        if (ctx->fp8) { launch_fp8_to_f16(w.w8, ctx->wexp, (long)N * K, s); p.W = ctx->wexp; p.w_scale = w.ws; }
        return p;
    };
    launch_prefill_embed(ctx->forced, P * beam, B, P, ctx->tok_emb, ctx->dec_pos, ctx->x, D, s);     // forced: [B*beam][P], equal within a recording
    for (int li = 0; li < d.dec_layers; ++li) {
        const DecLayerW& l = ctx->dec[li];
        launch_layernorm(ctx->x, l.ln1.g, l.ln1.b, ctx->xn, nullptr, M, D, s);
        {   GemmParams p = gemm(ctx->xn, D, l.qkv, 3 * D); p.scale = qscale; p.scale_ncols = D; p.C = qkv; p.ldc = 3 * D; launch_gemm(p, s); }
        launch_prefill_self_attn(qkv, ctx->selfK + li * self_layer, ctx->selfV + li * self_layer, ctx->att, B, H, P, Tc, D, s, beam);
        {   GemmParams p = gemm(ctx->att, D, l.out, D); p.out_mode = OUT_RESID_F32; p.C = ctx->x; p.ldc = D; launch_gemm(p, s); }
        launch_layernorm(ctx->x, l.ln2.g, l.ln2.b, ctx->xn, nullptr, M, D, s);
        {   GemmParams p = gemm(ctx->xn, D, l.cq, D); p.scale = qscale; p.scale_ncols = D; p.C = cq; p.ldc = D; launch_gemm(p, s); }
        {   // every prompt row of a sequence against the cached encoder K/V of its recording, one pass of the keys
            const size_t per_layer = (size_t)ctx->cache_seqs * d.n_audio_ctx * D;
            if (ctx->kv8) {
                const size_t per_layer_rows = (size_t)ctx->cache_seqs * d.n_audio_ctx * H;
                launch_prefill_cross_attn8(cq, ctx->crossK8 + li * per_layer, ctx->crossV8 + li * per_layer, ctx->crossKs + li * per_layer_rows,
                                           ctx->crossVs + li * per_layer_rows, ctx->att, B, P, H, d.n_audio_ctx, kv_rep, s);
            } else {
                launch_prefill_cross_attn(cq, ctx->crossK + li * per_layer, ctx->crossV + li * per_layer, ctx->att, B, P, H, d.n_audio_ctx,
                                          kv_rep, s);
            }
        }
        {   GemmParams p = gemm(ctx->att, D, l.co, D); p.out_mode = OUT_RESID_F32; p.C = ctx->x; p.ldc = D; launch_gemm(p, s); }
        launch_layernorm(ctx->x, l.ln3.g, l.ln3.b, ctx->xn, nullptr, M, D, s);
        {   GemmParams p = gemm(ctx->xn, D, l.fc1, F); p.gelu = 1; p.C = ctx->hid; p.ldc = F; launch_gemm(p, s); }
        {   GemmParams p = gemm(ctx->hid, F, l.fc2, D); p.out_mode = OUT_RESID_F32; p.C = ctx->x; p.ldc = D; launch_gemm(p, s); }
    }
    launch_prefill_last_rows(ctx->x, ctx->dx, B, P, D, ctx->st, s, beam, src_rows);
    HIPCHK(hipGetLastError());
    return MW_OK;
}

// Poll number `n` of a decode loop for "all `need` sequences (recordings) have ended": the counter of the steps
// enqueued so far is copied to a pinned slot, and the slot of the PREVIOUS poll, which the GPU passed
// check_every steps ago, is what gets waited for and read.  The host therefore stays at most two polls ahead of
// the GPU and the stream never runs dry, at the price of up to check_every more steps after the last sequence
// has ended (their rows are padded with end-of-text and their cross-attention is skipped).
static int poll_finished(mw_ctx* ctx, int n, int need, bool* all_done) {
    *all_done = false;
    if (n > 0) {
        HIPCHK(hipEventSynchronize(ctx->poll_ev[(n - 1) & 1]));
        if (ctx->fin_host[(n - 1) & 1] >= need) { *all_done = true; return MW_OK; }
    }
    HIPCHK(hipMemcpyAsync(&ctx->fin_host[n & 1], &ctx->st->n_finished, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipEventRecord(ctx->poll_ev[n & 1], ctx->stream));
    return MW_OK;
}

// Capture (or re-use) the graph of one step for this batch size / forced length / logits flag and the
// self-attention bucket of `n_keys` = position + 1.  Returns the executable graph through `out`.
static int ensure_graph(mw_ctx* ctx, int B, bool want_logits, const FinalizeParams& fp, long key, int n_keys, hipGraphExec_t* out,
                        const BeamParams* beam = nullptr, const SampleParams* samp = nullptr) {
    const int b = dec_self_attn_bucket(n_keys);
    for (int j = 0; j < kGraphSlots; ++j)
        if (ctx->graph_exec[b][j] && ctx->graph_batch[b][j] == key && ctx->graph_logits[b][j] == (int)want_logits) {
            *out = ctx->graph_exec[b][j];
            return MW_OK;
        }
    const int j = ctx->graph_next[b];
    ctx->graph_next[b] = (j + 1) % kGraphSlots;
    if (ctx->graph_exec[b][j]) {           // the slot's old step may still be queued on the stream
        HIPCHK(hipStreamSynchronize(ctx->stream));
        (void)hipGraphExecDestroy(ctx->graph_exec[b][j]); ctx->graph_exec[b][j] = nullptr;
    }
    if (ctx->graph[b][j]) { (void)hipGraphDestroy(ctx->graph[b][j]); ctx->graph[b][j] = nullptr; }
    HIPCHK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    enqueue_step(ctx, B, want_logits, fp, n_keys, beam, samp);
    HIPCHK(hipStreamEndCapture(ctx->stream, &ctx->graph[b][j]));
    HIPCHK(hipGraphInstantiate(&ctx->graph_exec[b][j], ctx->graph[b][j], nullptr, nullptr, 0));
    ctx->graph_batch[b][j] = key; ctx->graph_logits[b][j] = (int)want_logits;
    *out = ctx->graph_exec[b][j];
    return MW_OK;
}

// Shared driver: feed `n_forced` given ids per sequence, then sample `max_new` more.
// `sc` (sampling at a temperature): the token bookkeeping of every step is the sampling launches', which read the
// step's full logits.
struct SampleCall { const SampleParams* p; float inv_temp; uint32_t seed; int n_ended_at_start; };
static int run_decode(mw_ctx* ctx, int B, int n_forced, int max_new, float* logits_out, bool logits_every_step, const SampleCall* sc = nullptr) {
    const mw_model_desc& d = ctx->d;
    hipStream_t s = ctx->stream;
    FinalizeParams fp = finalize_params(ctx, n_forced);
    const SampleParams* samp = sc ? sc->p : nullptr;
    if (samp) launch_sample_init(*samp, n_forced, max_new, ctx->eot, sc->inv_temp, sc->seed, sc->n_ended_at_start, s);
    else launch_dec_init(fp, B, n_forced, max_new, ctx->eot, s);
    HIPCHK(hipGetLastError());
    const bool copy_logits = logits_out != nullptr;
    const bool want_logits = copy_logits || samp != nullptr;
    const long graph_key = (samp ? ((1L << 48) | ((long)samp->best_of << 40) | ((long)samp->skip_ended << 39)) : 0L) | ((long)B * 1024 + n_forced);
    const int n_steps = max_new > 0 ? n_forced + max_new - 1 : n_forced;
    const bool graph = ctx->use_graph != 0;
    const size_t row = (size_t)B * d.vocab;
    int n_polls = 0;
    int steps_run = 0;
    int first_step = 0;
    // kv_div == 1 means "own cache" to the cross-attention launcher, so a single-row prompt keeps stepping
    if (ctx->use_prefill && !logits_every_step && max_new > 0 && n_forced >= 2) {
        int r = run_prefill(ctx, B, n_forced, 1, nullptr, samp ? samp->best_of : 1);    // positions 0 .. n_forced-1 in one pass
        if (r) return r;
        enqueue_step_tail(ctx, B, want_logits, fp, nullptr, samp);     // logits of the last prompt position choose the first sampled token
        HIPCHK(hipGetLastError());
        if (copy_logits) HIPCHK(hipMemcpyAsync(logits_out, ctx->logits, row * 4, hipMemcpyDefault, s));
        first_step = n_forced;
        steps_run = n_forced;
    }
    for (int step = first_step; step < n_steps; ++step) {       // step feeds position `step`: step + 1 cached keys
        if (graph) {
            hipGraphExec_t g;
            int r = ensure_graph(ctx, B, want_logits, fp, graph_key, step + 1, &g, nullptr, samp);
            if (r) return r;
            HIPCHK(hipGraphLaunch(g, s));
        } else { enqueue_step(ctx, B, want_logits, fp, step + 1, nullptr, samp); HIPCHK(hipGetLastError()); }
        ++steps_run;
        if (copy_logits) {
            // step feeds position `step`; its logits choose the token for position step + 1
            if (logits_every_step) {
                HIPCHK(hipMemcpyAsync(logits_out + (size_t)step * row, ctx->logits, row * 4, hipMemcpyDefault, s));
            } else if (step >= n_forced - 1) {
                HIPCHK(hipMemcpyAsync(logits_out + (size_t)(step - (n_forced - 1)) * row, ctx->logits, row * 4, hipMemcpyDefault, s));
            }
        }
        if (max_new > 0 && step >= n_forced - 1 && ((step - (n_forced - 1)) % ctx->check_every) == ctx->check_every - 1 && step + 1 < n_steps) {
            bool all_done = false;
            int r = poll_finished(ctx, n_polls++, B, &all_done);
            if (r) return r;
            if (all_done) break;
        }
    }
    ctx->stage_ms[4] = (float)steps_run;
    return MW_OK;
}

extern "C" int mw_decode_greedy(mw_ctx* ctx, int batch, int32_t* out_ids, int32_t* out_len, float* logits_out) {
    if (!ctx) return MW_ERR_ARG;
    int r = check_batch(ctx, batch, "mw_decode_greedy");
    if (r) return r;
    if (!ctx->opts_set) return fail(ctx, MW_ERR_STATE, "mw_decode_greedy: call mw_set_decode_opts first");
    if (ctx->enc_batch != batch) return fail(ctx, MW_ERR_STATE, "mw_decode_greedy: batch differs from the last mw_encode");
    if (!out_ids || !out_len) return fail(ctx, MW_ERR_ARG, "mw_decode_greedy: null output");
    HIPCHK(hipSetDevice(ctx->device));
    const int P = (int)ctx->prompt.size();
    std::vector<int32_t> f((size_t)batch * P);
    for (int b = 0; b < batch; ++b) memcpy(&f[(size_t)b * P], ctx->prompt.data(), P * 4);
    HIPCHK(hipMemcpyAsync(ctx->forced, f.data(), f.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev[3], ctx->stream));
    r = run_decode(ctx, batch, P, ctx->max_new, logits_out, false);
    if (r) return r;
    HIPCHK(hipEventRecord(ctx->ev[4], ctx->stream));
    HIPCHK(hipMemcpyAsync(out_ids, ctx->out_ids, (size_t)batch * ctx->max_new * 4, hipMemcpyDefault, ctx->stream));
    HIPCHK(hipMemcpyAsync(out_len, ctx->out_len, (size_t)batch * 4, hipMemcpyDefault, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipEventElapsedTime(&ctx->stage_ms[3], ctx->ev[3], ctx->ev[4]));
    return MW_OK;
}

// Beam search over the encoder state of the last mw_encode: n_audio recordings x beam hypotheses
// run as n_audio*beam sequences through the same step kernels (cross-attention reads the cache of
// recording r / beam, self-attention follows src_rows), bookkeeping in beam_step_kernel, the final
// top-up and ranking on the host.
extern "C" int mw_decode_beam(mw_ctx* ctx, int n_audio, int beam, float patience, int32_t* out_ids, int32_t* out_len,
                              float* out_avg_logprob, int32_t* hyp_ids, int32_t* hyp_len, float* hyp_sum_logprob) {
    if (!ctx) return MW_ERR_ARG;
    if (!ctx->opts_set) return fail(ctx, MW_ERR_STATE, "mw_decode_beam: call mw_set_decode_opts first");
    if (beam < 1 || beam > kMaxBeam) return fail(ctx, MW_ERR_ARG, "mw_decode_beam: beam must be in 1.." + std::to_string(kMaxBeam));
    if (!(patience >= 1.0f) || patience > 4.0f) return fail(ctx, MW_ERR_ARG, "mw_decode_beam: patience must be in [1, 4]");
    if (n_audio < 1 || n_audio * beam > ctx->max_batch)
        return fail(ctx, MW_ERR_ARG, "mw_decode_beam: n_audio * beam must be in 1.." + std::to_string(ctx->max_batch) + " (max_batch)");
    if (n_audio > ctx->cache_seqs) return fail(ctx, MW_ERR_ARG, "mw_decode_beam: n_audio exceeds the K/V cache (" + std::to_string(ctx->cache_seqs) + " recordings)");
    if (ctx->enc_batch != n_audio) return fail(ctx, MW_ERR_STATE, "mw_decode_beam: n_audio differs from the last mw_encode");
    if (!out_ids || !out_len) return fail(ctx, MW_ERR_ARG, "mw_decode_beam: null output");
    HIPCHK(hipSetDevice(ctx->device));
    const mw_model_desc& d = ctx->d;
    const int R = n_audio * beam, P = (int)ctx->prompt.size(), max_new = ctx->max_new;
    const int max_cand = (int)lroundf(beam * patience);
    const int cand_cap = (int)lroundf(kMaxBeam * 4.0f);
    if (!ctx->bm_sum_lp) {          // first use: sized for max_batch sequences and the largest candidate list
        const size_t MB = ctx->max_batch;
        ALLOC(ctx->bm_sum_lp, MB); ALLOC(ctx->bm_hist, MB * d.n_text_ctx); ALLOC(ctx->bm_src_rows, MB * d.n_text_ctx);
        ALLOC(ctx->bm_fin_tokens, MB * cand_cap * d.n_text_ctx); ALLOC(ctx->bm_fin_lp, MB * cand_cap);
        ALLOC(ctx->bm_fin_len, MB * cand_cap); ALLOC(ctx->bm_fin_count, MB); ALLOC(ctx->bm_completed, MB);
        ALLOC(ctx->bm_part_m, MB * kBeamParts); ALLOC(ctx->bm_part_s, MB * kBeamParts);
        ALLOC(ctx->bm_part_v, MB * kBeamParts * (kMaxBeam + 1)); ALLOC(ctx->bm_part_i, MB * kBeamParts * (kMaxBeam + 1));
    }
    std::vector<int32_t> f((size_t)R * P);
    for (int r = 0; r < R; ++r) memcpy(&f[(size_t)r * P], ctx->prompt.data(), P * 4);
    HIPCHK(hipMemcpyAsync(ctx->forced, f.data(), f.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev[3], ctx->stream));

    BeamParams bp{};
    bp.st = ctx->st; bp.n_audio = n_audio; bp.beam = beam; bp.max_cand = max_cand;
    bp.logits = ctx->logits; bp.mask_first = ctx->mask_first; bp.mask_rest = ctx->mask_rest;
    bp.forced = ctx->forced; bp.n_forced = P;
    bp.tok_emb = ctx->tok_emb; bp.pos_emb = ctx->dec_pos; bp.D = d.d_model; bp.n_text_ctx = d.n_text_ctx; bp.vocab = d.vocab;
    bp.x = ctx->dx; bp.sum_lp = ctx->bm_sum_lp; bp.hist = ctx->bm_hist; bp.src_rows = ctx->bm_src_rows;
    bp.fin_tokens = ctx->bm_fin_tokens; bp.fin_lp = ctx->bm_fin_lp; bp.fin_len = ctx->bm_fin_len;
    bp.fin_count = ctx->bm_fin_count; bp.completed = ctx->bm_completed;
    bp.part_m = ctx->bm_part_m; bp.part_s = ctx->bm_part_s; bp.part_v = ctx->bm_part_v; bp.part_i = ctx->bm_part_i;
    hipStream_t s = ctx->stream;
    launch_beam_init(bp, P, max_new, ctx->eot, s);
    HIPCHK(hipGetLastError());
    FinalizeParams fp = finalize_params(ctx, P);
    const int n_steps = P + max_new - 1;
    const bool graph = ctx->use_graph != 0;
    int n_polls = 0, steps_run = 0, first_step = 0;
    if (ctx->use_prefill && P >= 2) {                       // the prompt once per recording, in one pass
        int r = run_prefill(ctx, n_audio, P, beam, bp.src_rows);
        if (r) return r;
        enqueue_step_tail(ctx, R, true, fp, &bp);
        HIPCHK(hipGetLastError());
        first_step = steps_run = P;
    }
    for (int step = first_step; step < n_steps; ++step) {
        if (graph) {
            hipGraphExec_t g;
            // rows and prompt length in the low word (R <= 512, P <= 448), beam and candidate count in the high one
            int r = ensure_graph(ctx, R, true, fp, ((long)(beam * 64 + max_cand) << 32) | (long)(R * 1024 + P), step + 1, &g, &bp);
            if (r) return r;
            HIPCHK(hipGraphLaunch(g, s));
        } else { enqueue_step(ctx, R, true, fp, step + 1, &bp); HIPCHK(hipGetLastError()); }
        ++steps_run;
        if (step >= P - 1 && ((step - (P - 1)) % ctx->check_every) == ctx->check_every - 1 && step + 1 < n_steps) {
            bool all_done = false;
            int r = poll_finished(ctx, n_polls++, n_audio, &all_done);
            if (r) return r;
            if (all_done) break;
        }
    }
    ctx->stage_ms[4] = (float)steps_run;
    HIPCHK(hipEventRecord(ctx->ev[4], s));
    // ---- collect: finished lists, live beams for the top-up, then rank by log-probability per token
    std::vector<int32_t> hist((size_t)R * max_new), fin_tok((size_t)n_audio * max_cand * max_new), fin_len((size_t)n_audio * max_cand),
        fin_count(n_audio);
    std::vector<float> sum_lp(R), fin_lp((size_t)n_audio * max_cand);
    HIPCHK(hipMemcpyAsync(hist.data(), ctx->bm_hist, hist.size() * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(sum_lp.data(), ctx->bm_sum_lp, sum_lp.size() * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(fin_tok.data(), ctx->bm_fin_tokens, fin_tok.size() * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(fin_lp.data(), ctx->bm_fin_lp, fin_lp.size() * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(fin_len.data(), ctx->bm_fin_len, fin_len.size() * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(fin_count.data(), ctx->bm_fin_count, fin_count.size() * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipEventElapsedTime(&ctx->stage_ms[3], ctx->ev[3], ctx->ev[4]));
    // sampled tokens per live beam: every step run after the prompt appended one (the last token is chosen, not fed)
    const int live_len = std::min(max_new, steps_run - (P - 1));
    for (int a = 0; a < n_audio; ++a) {
        struct Hyp { const int32_t* tok; int len; float lp; };
        std::vector<Hyp> hyps;
        for (int c = 0; c < fin_count[a]; ++c)
            hyps.push_back({&fin_tok[((size_t)a * max_cand + c) * max_new], fin_len[(size_t)a * max_cand + c], fin_lp[(size_t)a * max_cand + c]});
        if ((int)hyps.size() < beam) {               // top up with the best live beams (stable in beam order)
            std::vector<int> order(beam);
            for (int j = 0; j < beam; ++j) order[j] = j;
            std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return sum_lp[a * beam + x] > sum_lp[a * beam + y]; });
            for (int j : order) {
                if ((int)hyps.size() >= beam) break;
                hyps.push_back({&hist[(size_t)(a * beam + j) * max_new], live_len, sum_lp[a * beam + j]});
            }
        }
        int best = 0;
        for (int i = 1; i < (int)hyps.size(); ++i)
            if (hyps[i].lp / std::max(hyps[i].len, 1) > hyps[best].lp / std::max(hyps[best].len, 1)) best = i;
        for (int i = 0; i < max_new; ++i) out_ids[(size_t)a * max_new + i] = i < hyps[best].len ? hyps[best].tok[i] : ctx->eot;
        out_len[a] = hyps[best].len;
        if (out_avg_logprob) out_avg_logprob[a] = hyps[best].lp / (hyps[best].len + 1);
        for (int c = 0; c < beam; ++c) {
            const bool have = c < (int)hyps.size();
            if (hyp_len) hyp_len[a * beam + c] = have ? hyps[c].len : -1;
            if (hyp_sum_logprob) hyp_sum_logprob[a * beam + c] = have ? hyps[c].lp : -INFINITY;
            if (hyp_ids)
                for (int i = 0; i < max_new; ++i)
                    hyp_ids[((size_t)a * beam + c) * max_new + i] = (have && i < hyps[c].len) ? hyps[c].tok[i] : ctx->eot;
        }
    }
    return MW_OK;
}

extern "C" float mw_sample_uniform(uint32_t seed, uint32_t sequence, uint32_t position, uint32_t token) {
    return sample_uniform(seed, sequence, position, token);
}

// Samples at a temperature over the encoder state of the last mw_encode: n_audio recordings x best_of samples run
// as n_audio*best_of sequences through the step kernels of greedy decoding (cross-attention reads the cache of
// recording r / best_of, every sequence keeps its own self-attention cache), token bookkeeping in the sampling
// launches, the choice among a recording's samples on the host.
extern "C" int mw_decode_sample(mw_ctx* ctx, int n_audio, int best_of, float temperature, uint32_t seed, const uint8_t* active,
                                int32_t* out_ids, int32_t* out_len, float* out_avg_logprob, int32_t* smp_ids, int32_t* smp_len,
                                float* smp_sum_logprob, float* logits_out) {
    if (!ctx) return MW_ERR_ARG;
    if (!ctx->opts_set) return fail(ctx, MW_ERR_STATE, "mw_decode_sample: call mw_set_decode_opts first");
    if (best_of < 1 || best_of > kMaxBeam) return fail(ctx, MW_ERR_ARG, "mw_decode_sample: best_of must be in 1.." + std::to_string(kMaxBeam));
    if (!(temperature >= 0.0f) || temperature > 100.0f) return fail(ctx, MW_ERR_ARG, "mw_decode_sample: temperature must be in [0, 100]");
    if (n_audio < 1 || n_audio * best_of > ctx->max_batch)
        return fail(ctx, MW_ERR_ARG, "mw_decode_sample: n_audio * best_of must be in 1.." + std::to_string(ctx->max_batch) + " (max_batch)");
    if (n_audio > ctx->cache_seqs) return fail(ctx, MW_ERR_ARG, "mw_decode_sample: n_audio exceeds the K/V cache (" + std::to_string(ctx->cache_seqs) + " recordings)");
    if (ctx->enc_batch != n_audio) return fail(ctx, MW_ERR_STATE, "mw_decode_sample: n_audio differs from the last mw_encode");
    if (!out_ids || !out_len) return fail(ctx, MW_ERR_ARG, "mw_decode_sample: null output");
    HIPCHK(hipSetDevice(ctx->device));
    const mw_model_desc& d = ctx->d;
    const int R = n_audio * best_of, P = (int)ctx->prompt.size(), max_new = ctx->max_new;
    if (!ctx->sp_part_i) {          // first use: sized for max_batch sequences.  Gated on the buffer allocated LAST, and
        const size_t MB = ctx->max_batch;           // every buffer only if it is still missing: an allocation that failed
        auto missing = [](const void* p) { return p == nullptr; };      // part-way is completed by the next call
        if (missing(ctx->ss)) ALLOC(ctx->ss, 1);
        if (missing(ctx->sp_sum_lp)) ALLOC(ctx->sp_sum_lp, MB);
        if (missing(ctx->sp_active)) ALLOC(ctx->sp_active, MB);
        if (missing(ctx->sp_part_m)) ALLOC(ctx->sp_part_m, MB * kBeamParts);
        if (missing(ctx->sp_part_s)) ALLOC(ctx->sp_part_s, MB * kBeamParts);
        if (missing(ctx->sp_part_v)) ALLOC(ctx->sp_part_v, MB * kBeamParts);
        if (missing(ctx->sp_part_l)) ALLOC(ctx->sp_part_l, MB * kBeamParts);
        ALLOC(ctx->sp_part_i, MB * kBeamParts);
    }
    int n_ended = 0;
    if (active) {
        for (int a = 0; a < n_audio; ++a) n_ended += active[a] ? 0 : best_of;
        HIPCHK(hipMemcpyAsync(ctx->sp_active, active, n_audio, hipMemcpyHostToDevice, ctx->stream));
    }
    std::vector<int32_t> f((size_t)R * P);
    for (int r = 0; r < R; ++r) memcpy(&f[(size_t)r * P], ctx->prompt.data(), P * 4);
    HIPCHK(hipMemcpyAsync(ctx->forced, f.data(), f.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev[3], ctx->stream));

    SampleParams sp{};
    sp.st = ctx->st; sp.ss = ctx->ss; sp.n_audio = n_audio; sp.best_of = best_of;
    sp.logits = ctx->logits; sp.mask_first = ctx->mask_first; sp.mask_rest = ctx->mask_rest;
    sp.forced = ctx->forced; sp.n_forced = P;
    sp.tok_emb = ctx->tok_emb; sp.pos_emb = ctx->dec_pos; sp.D = d.d_model; sp.n_text_ctx = d.n_text_ctx; sp.vocab = d.vocab;
    sp.x = ctx->dx; sp.sum_lp = ctx->sp_sum_lp; sp.out_ids = ctx->out_ids; sp.out_len = ctx->out_len; sp.finished = ctx->finished;
    sp.active = active ? ctx->sp_active : nullptr;
    // ended sequences cost no cross-attention reads (recordings switched off from the start included), unless the
    // caller wants the logits of every position
    sp.skip_ended = (ctx->skip_done && (ctx->can_end || n_ended > 0) && !logits_out) ? 1 : 0;
    sp.part_m = ctx->sp_part_m; sp.part_s = ctx->sp_part_s; sp.part_v = ctx->sp_part_v; sp.part_l = ctx->sp_part_l; sp.part_i = ctx->sp_part_i;
    SampleCall sc{&sp, temperature > 0.0f ? 1.0f / temperature : 0.0f, seed, n_ended};
    int r = run_decode(ctx, R, P, max_new, logits_out, false, &sc);
    if (r) return r;
    hipStream_t s = ctx->stream;
    HIPCHK(hipEventRecord(ctx->ev[4], s));
    std::vector<int32_t> ids((size_t)R * max_new), len(R);
    std::vector<float> lp(R);
    HIPCHK(hipMemcpyAsync(ids.data(), ctx->out_ids, ids.size() * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(len.data(), ctx->out_len, len.size() * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(lp.data(), ctx->sp_sum_lp, lp.size() * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipEventElapsedTime(&ctx->stage_ms[3], ctx->ev[3], ctx->ev[4]));
    // per recording: the sample with the highest log-probability per token (ties: the first)
    for (int a = 0; a < n_audio; ++a) {
        int best = a * best_of;
        for (int j = 1; j < best_of; ++j) {
            const int c = a * best_of + j;
            if (lp[c] / std::max(len[c], 1) > lp[best] / std::max(len[best], 1)) best = c;
        }
        memcpy(out_ids + (size_t)a * max_new, &ids[(size_t)best * max_new], (size_t)max_new * 4);
        out_len[a] = len[best];
        if (out_avg_logprob) out_avg_logprob[a] = lp[best] / (len[best] + 1);
    }
    if (smp_ids) memcpy(smp_ids, ids.data(), ids.size() * 4);
    if (smp_len) memcpy(smp_len, len.data(), len.size() * 4);
    if (smp_sum_logprob) memcpy(smp_sum_logprob, lp.data(), lp.size() * 4);
    return MW_OK;
}

extern "C" int mw_decode_forced(mw_ctx* ctx, int batch, const int32_t* tokens, int n, float* logits_out) {
    if (!ctx) return MW_ERR_ARG;
    int r = check_batch(ctx, batch, "mw_decode_forced");
    if (r) return r;
    if (ctx->enc_batch != batch) return fail(ctx, MW_ERR_STATE, "mw_decode_forced: batch differs from the last mw_encode");
    if (!tokens || !logits_out || n < 1 || n > ctx->d.n_text_ctx) return fail(ctx, MW_ERR_ARG, "mw_decode_forced: bad tokens / n / output");
    for (long i = 0; i < (long)batch * n; ++i)
        if (tokens[i] < 0 || tokens[i] >= ctx->d.vocab) return fail(ctx, MW_ERR_ARG, "mw_decode_forced: token id out of range");
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpyAsync(ctx->forced, tokens, (size_t)batch * n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    // logits come back step-major [n][batch][vocab]; transposed to [batch][n][vocab] below
    const size_t V = ctx->d.vocab;
    std::vector<float> tmp((size_t)n * batch * V);
    r = run_decode(ctx, batch, n, 0, tmp.data(), true);
    if (r) return r;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (int j = 0; j < n; ++j)
        for (int b = 0; b < batch; ++b)
            memcpy(logits_out + ((size_t)b * n + j) * V, &tmp[((size_t)j * batch + b) * V], V * 4);
    return MW_OK;
}

extern "C" int mw_transcribe_batch(mw_ctx* ctx, const float* pcm, int batch, int32_t* out_ids, int32_t* out_len) {
    if (!ctx) return MW_ERR_ARG;
    int r = check_batch(ctx, batch, "mw_transcribe_batch");
    if (r) return r;
    if (!ctx->opts_set) return fail(ctx, MW_ERR_STATE, "mw_transcribe_batch: call mw_set_decode_opts first");
    if (!pcm || !out_ids || !out_len) return fail(ctx, MW_ERR_ARG, "mw_transcribe_batch: null argument");
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const float* src = pcm;
    if (!is_device_ptr(pcm)) {
        HIPCHK(hipMemcpyAsync(ctx->pcm_dev, pcm, (size_t)batch * kChunkSamples * 4, hipMemcpyHostToDevice, s));
        src = ctx->pcm_dev;
    }
    const int P = (int)ctx->prompt.size();
    std::vector<int32_t> f((size_t)batch * P);
    for (int b = 0; b < batch; ++b) memcpy(&f[(size_t)b * P], ctx->prompt.data(), P * 4);
    HIPCHK(hipMemcpyAsync(ctx->forced, f.data(), f.size() * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));       // `f` is pageable host memory that dies with this scope
    HIPCHK(hipEventRecord(ctx->ev[0], s));
    r = run_logmel(ctx, src, batch);
    if (r) return r;
    HIPCHK(hipEventRecord(ctx->ev[1], s));
    r = run_encoder(ctx, batch);
    if (r) return r;
    HIPCHK(hipEventRecord(ctx->ev[2], s));
    r = run_cross_kv(ctx, batch);
    if (r) return r;
    HIPCHK(hipEventRecord(ctx->ev[3], s));
    r = run_decode(ctx, batch, P, ctx->max_new, nullptr, false);
    if (r) return r;
    HIPCHK(hipEventRecord(ctx->ev[4], s));
    HIPCHK(hipMemcpyAsync(out_ids, ctx->out_ids, (size_t)batch * ctx->max_new * 4, hipMemcpyDefault, s));
    HIPCHK(hipMemcpyAsync(out_len, ctx->out_len, (size_t)batch * 4, hipMemcpyDefault, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int i = 0; i < 4; ++i) HIPCHK(hipEventElapsedTime(&ctx->stage_ms[i], ctx->ev[i], ctx->ev[i + 1]));
    return MW_OK;
}

// Encoder half of a pass on one context, decoder half on another: log-mel, encoder and cross K/V of `batch`
// chunks run on `ctx` (its stream, its workspaces), the K/V land in the cache of `dst` at sequences
// slot0 .. slot0 + batch - 1.  Returns when they are complete.
extern "C" int mw_encode_into(mw_ctx* ctx, const float* pcm, int batch, mw_ctx* dst, int slot0) {
    if (!ctx) return MW_ERR_ARG;
    int r = check_batch(ctx, batch, "mw_encode_into");
    if (r) return r;
    if (!pcm || !dst) return fail(ctx, MW_ERR_ARG, "mw_encode_into: null argument");
    const mw_ctx* root_a = ctx->parent ? ctx->parent : ctx;
    const mw_ctx* root_b = dst->parent ? dst->parent : dst;
    if (root_a != root_b) return fail(ctx, MW_ERR_ARG, "mw_encode_into: both contexts must share weights (an engine and its forks)");
    if (slot0 < 0 || slot0 + batch > dst->cache_seqs)
        return fail(ctx, MW_ERR_ARG, "mw_encode_into: slot0 + batch must fit the destination's cache (" + std::to_string(dst->cache_seqs) + " sequences)");
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const float* src = pcm;
    if (!is_device_ptr(pcm)) {
        HIPCHK(hipMemcpyAsync(ctx->pcm_dev, pcm, (size_t)batch * kChunkSamples * 4, hipMemcpyHostToDevice, s));
        src = ctx->pcm_dev;
    }
    HIPCHK(hipEventRecord(ctx->ev[0], s));
    r = run_logmel(ctx, src, batch);
    if (r) return r;
    HIPCHK(hipEventRecord(ctx->ev[1], s));
    r = run_encoder(ctx, batch);
    if (r) return r;
    HIPCHK(hipEventRecord(ctx->ev[2], s));
    r = run_cross_kv(ctx, batch, dst, slot0);
    if (r) return r;
    HIPCHK(hipEventRecord(ctx->ev[3], s));
    HIPCHK(hipStreamSynchronize(s));
    for (int i = 0; i < 3; ++i) HIPCHK(hipEventElapsedTime(&ctx->stage_ms[i], ctx->ev[i], ctx->ev[i + 1]));
    return MW_OK;
}

extern "C" int mw_set_encoded(mw_ctx* ctx, int batch) {
    if (!ctx) return MW_ERR_ARG;
    int r = check_batch(ctx, batch, "mw_set_encoded");
    if (r) return r;
    ctx->enc_batch = batch;
    return MW_OK;
}

extern "C" int mw_read_cross_kv(mw_ctx* ctx, int layer, int sequence, float* k_out, float* v_out) {
    if (!ctx) return MW_ERR_ARG;
    const mw_model_desc& d = ctx->d;
    if (!k_out || !v_out) return fail(ctx, MW_ERR_ARG, "mw_read_cross_kv: null output");
    if (layer < 0 || layer >= d.dec_layers) return fail(ctx, MW_ERR_ARG, "mw_read_cross_kv: layer out of range");
    if (sequence < 0 || sequence >= ctx->enc_batch) return fail(ctx, MW_ERR_STATE, "mw_read_cross_kv: sequence is not among those encoded last");
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const size_t per_seq = (size_t)d.n_heads * d.n_audio_ctx * kHeadDim;                 // elements of one sequence in one layer
    const size_t at = ((size_t)layer * ctx->cache_seqs + sequence) * per_seq;
    if (ctx->kv8) {
        const size_t rows = per_seq / kHeadDim;
        std::vector<uint8_t> codes(per_seq);
        std::vector<float> scales(rows);
        const uint8_t* src8[2] = {ctx->crossK8, ctx->crossV8};
        const float* srcs[2] = {ctx->crossKs, ctx->crossVs};
        float* dst[2] = {k_out, v_out};
        for (int w = 0; w < 2; ++w) {
            HIPCHK(hipMemcpy(codes.data(), src8[w] + at, per_seq, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(scales.data(), srcs[w] + at / kHeadDim, rows * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < per_seq; ++i) {
                const uint8_t c = codes[i];
                const int e = (c >> 3) & 15, m = c & 7;
                const float v = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m * 0.125f, e - 7);
                dst[w][i] = ((c & 0x80) ? -v : v) * scales[i / kHeadDim];
            }
        }
        return MW_OK;
    }
    std::vector<f16> h(per_seq);
    HIPCHK(hipMemcpy(h.data(), ctx->crossK + at, per_seq * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < per_seq; ++i) k_out[i] = (float)h[i];
    HIPCHK(hipMemcpy(h.data(), ctx->crossV + at, per_seq * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < per_seq; ++i) v_out[i] = (float)h[i];
    return MW_OK;
}

extern "C" int mw_frame_energy_db(mw_ctx* ctx, const float* pcm, int64_t n_samples, int frame, int hop, float* out_db,
                                  int64_t n_frames) {
    if (!ctx) return MW_ERR_ARG;
    if (!pcm || !out_db) return fail(ctx, MW_ERR_ARG, "mw_frame_energy_db: null pointer");
    if (n_samples < 1 || hop < 1 || frame < hop || frame > (1 << 20))
        return fail(ctx, MW_ERR_ARG, "mw_frame_energy_db: need n_samples >= 1 and 1 <= hop <= frame <= 2^20");
    if (n_frames != (n_samples + hop - 1) / hop)
        return fail(ctx, MW_ERR_ARG, "mw_frame_energy_db: n_frames must be ceil(n_samples / hop)");
    HIPCHK(hipSetDevice(ctx->device));
    const bool pcm_dev = is_device_ptr(pcm), out_dev = is_device_ptr(out_db);
    float* d_out = out_db;
    if (!out_dev) HIPCHK(hipMalloc(&d_out, (size_t)n_frames * 4));
    int rc = MW_OK;
    if (pcm_dev) {
        launch_frame_energy(pcm, n_samples, frame, hop, d_out, n_frames, ctx->stream);
    } else {
        // host audio: staged in pieces of whole hops, each with the (frame - hop) samples its last frames reach into
        const int64_t piece_frames = (int64_t)(1 << 24) / hop;                      // ~16 M samples per piece
        const int64_t cap = piece_frames * hop + frame;
        float* d_pcm = nullptr;
        hipError_t e = hipMalloc(&d_pcm, (size_t)std::min<int64_t>(cap, n_samples) * 4);
        if (e != hipSuccess) { if (!out_dev) (void)hipFree(d_out); return fail(ctx, MW_ERR_HIP, "mw_frame_energy_db: out of device memory"); }
        for (int64_t f0 = 0; f0 < n_frames && rc == MW_OK; f0 += piece_frames) {
            const int64_t nf = std::min<int64_t>(piece_frames, n_frames - f0);
            const int64_t s0 = f0 * hop;
            const int64_t ns = std::min<int64_t>(n_samples - s0, (nf - 1) * hop + frame);
            if (hipMemcpyAsync(d_pcm, pcm + s0, (size_t)ns * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { rc = MW_ERR_HIP; break; }
            // the piece is its own signal of ns samples, except that only its first nf frames are wanted
            launch_frame_energy(d_pcm, ns, frame, hop, d_out + f0, nf, ctx->stream);
            if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = MW_ERR_HIP;   // d_pcm is reused by the next piece
        }
        (void)hipFree(d_pcm);
    }
    if (rc == MW_OK && hipGetLastError() != hipSuccess) rc = MW_ERR_HIP;
    if (rc == MW_OK && !out_dev && hipMemcpyAsync(out_db, d_out, (size_t)n_frames * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = MW_ERR_HIP;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = MW_ERR_HIP;
    if (!out_dev) (void)hipFree(d_out);
    if (rc != MW_OK) return fail(ctx, rc, "mw_frame_energy_db: HIP error");
    return MW_OK;
}

extern "C" int mw_stage_ms(mw_ctx* ctx, float out[5]) {
    if (!ctx || !out) return MW_ERR_ARG;
    for (int i = 0; i < 5; ++i) out[i] = ctx->stage_ms[i];
    return MW_OK;
}

extern "C" int mw_profile_cross_attention(mw_ctx* ctx, int batch, int iters, float* mean_ms, double* bytes_per_launch) {
    if (!ctx) return MW_ERR_ARG;
    int r = check_batch(ctx, batch, "mw_profile_cross_attention");
    if (r) return r;
    if (iters < 1 || !mean_ms || !bytes_per_launch) return fail(ctx, MW_ERR_ARG, "mw_profile_cross_attention: bad argument");
    if (ctx->enc_batch != batch) return fail(ctx, MW_ERR_STATE, "mw_profile_cross_attention: encode this batch first");
    HIPCHK(hipSetDevice(ctx->device));
    const mw_model_desc& d = ctx->d;
    // cycle over the layers so that successive launches read different cache regions, as in a real step
    cross_attention(ctx, 0, ctx->dq, ctx->datt, batch);
    HIPCHK(hipEventRecord(ctx->ev[5], ctx->stream));
    for (int i = 0; i < iters; ++i) cross_attention(ctx, i % d.dec_layers, ctx->dq, ctx->datt, batch);
    HIPCHK(hipEventRecord(ctx->ev[4], ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, ctx->ev[5], ctx->ev[4]));
    *mean_ms = ms / iters;
    *bytes_per_launch = ctx->kv8 ? 2.0 * batch * d.n_audio_ctx * ((double)d.d_model + 4.0 * d.n_heads)     // K and V: codes + row scales
                                 : 2.0 * batch * d.n_audio_ctx * (double)d.d_model * 2.0;                // K and V, fp16
    return MW_OK;
}

// The six weight-streaming projections of a decoder layer (QKV, attention out, cross-attention query and out,
// MLP in and out), back to back in the order and on the data of a real step, without the two attention launches
// between them; `iters` layers, cycling through the model's layers.  Reports the mean time per layer (HIP events
// on the context's stream, launch boundaries included) and the weight bytes a layer's six matrices hold.
extern "C" int mw_profile_decode_gemvs(mw_ctx* ctx, int batch, int iters, float* mean_ms_per_layer, double* weight_bytes_per_layer) {
    if (!ctx) return MW_ERR_ARG;
    int r = check_batch(ctx, batch, "mw_profile_decode_gemvs", false);
    if (r) return r;
    if (iters < 1 || !mean_ms_per_layer || !weight_bytes_per_layer) return fail(ctx, MW_ERR_ARG, "mw_profile_decode_gemvs: bad argument");
    if (!ctx->opts_set) return fail(ctx, MW_ERR_STATE, "mw_profile_decode_gemvs: call mw_set_decode_opts first");
    HIPCHK(hipSetDevice(ctx->device));
    const mw_model_desc& d = ctx->d;
    FinalizeParams fp = finalize_params(ctx, (int)ctx->prompt.size());
    // a defined state: position 0 of a fresh decode (the k/v rows the QKV launch files go to cache position 0)
    std::vector<int32_t> f((size_t)batch * ctx->prompt.size());
    for (int b = 0; b < batch; ++b) memcpy(&f[(size_t)b * ctx->prompt.size()], ctx->prompt.data(), ctx->prompt.size() * 4);
    HIPCHK(hipMemcpyAsync(ctx->forced, f.data(), f.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    launch_dec_init(fp, batch, (int)ctx->prompt.size(), ctx->max_new, ctx->eot, ctx->stream);
    for (int i = 0; i < d.dec_layers; ++i) enqueue_layer(ctx, i, batch, false, fp, 1, nullptr, kLayerGemv);       // warm-up pass
    HIPCHK(hipEventRecord(ctx->ev[5], ctx->stream));
    for (int i = 0; i < iters; ++i) enqueue_layer(ctx, i % d.dec_layers, batch, false, fp, 1, nullptr, kLayerGemv);
    HIPCHK(hipEventRecord(ctx->ev[4], ctx->stream));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, ctx->ev[5], ctx->ev[4]));
    *mean_ms_per_layer = ms / iters;
    const double D = d.d_model, F = d.ffn_dim;
    *weight_bytes_per_layer = (6.0 * D * D + 2.0 * D * F) * (ctx->fp8 ? 1.0 : 2.0);
    return MW_OK;
}
