// Forced-alignment acoustic model behind the mw_align_* entry points of include/mwhip.h: a wav2vec2 CTC
// network (Hugging Face Wav2Vec2ForCTC, the layer-norm / stable-layer-norm variant of wav2vec2-large-xlsr-53,
// which is what whisperx.load_align_model resolves Chinese to; reference transcribe.py:130-131) turned into
// frame-wise log-probabilities over its character vocabulary.  The trellis search over those emissions is host
// code (Python: manual-whisper_amd/alignment.py).
//
// Only the front end is new: the first convolution on the raw waveform, LayerNorm over fp16 rows, the group-major
// copy for the positional convolution and the log-softmax (kernels_w2v.hip).  The other six convolutions, the
// feature projection, the grouped positional convolution, the 24 transformer layers and the CTC head are the
// encoder's GEMM, LayerNorm and attention kernels: a strided convolution over a time-major [T][C] buffer is a
// GEMM whose A rows overlap (row t = k * C consecutive halfs starting at frame s * t: lda = s * C < K), and one
// group of the positional convolution is the same over a group-major copy of the hidden states (lda = D / G).
#include "../../include/mwhip.h"
#include "mw_kernels.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <thread>
#include <vector>

using namespace mw;

static thread_local std::string g_align_create_error;

namespace {

struct Lin { f16* w = nullptr; float* b = nullptr; };
struct Norm { float* g = nullptr; float* b = nullptr; };
struct ConvL { f16* w = nullptr; float* w32 = nullptr; float* b = nullptr; Norm ln; int k = 0, s = 0, cin = 0, cout = 0; };
struct Layer { Norm ln1, ln2; Lin qkv, out, fc1, fc2; };

}  // namespace

constexpr int kMaxSegmentTokens = 1024;                    // characters one mw_align_segment call places
constexpr int kMaxPackSegments = 32;                       // segments one mw_align_segments call takes through the model together
constexpr int kPackSlackRows = 64;                         // rows of slack per workspace: the frame counts of a pack's equal
                                                           // slices may add up to a few more than those of one long recording

struct mw_aligner {
    mw_align_desc d{};
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    std::vector<void*> allocs;
    bool finalized = false;
    mw_aligner* parent = nullptr;                          // a fork borrows its parent's weights
    std::map<std::string, std::vector<float>> host;       // tensors as handed over (float32), until mw_align_finalize
    std::map<std::string, std::vector<int64_t>> shapes;

    std::vector<ConvL> conv;
    Norm proj_ln; Lin proj;
    f16* pos_w = nullptr; float* pos_b = nullptr;          // [G][D/G][K][D/G], [D]
    std::vector<Layer> layers;
    Norm enc_ln;
    Lin head; int vpad = 0;

    // workspaces, sized for max_samples
    float* pcm = nullptr;
    f16* h[2] = {nullptr, nullptr};                        // ping-pong conv outputs [T_i][C]
    float* x = nullptr; f16* xn = nullptr;                 // residual stream [T][D], normalised rows
    f16* pack = nullptr;                                   // [G][T + K][D/G]
    f16 *q = nullptr, *k = nullptr, *vT = nullptr, *att = nullptr, *hid = nullptr;
    float* logits = nullptr;                               // [T][vpad]
    int32_t* tok_dev = nullptr;                            // [kMaxPackSegments * kMaxSegmentTokens]
    float* gath_dev = nullptr; float* gath_host = nullptr; // gathered log-probabilities of a pack's segments, back to back (device; pinned host)
    int32_t* keys_dev = nullptr;                           // [kMaxPackSegments] frames of each segment of the pack in flight
    size_t gath_floats = 0;
    int max_frames = 0, max_t0 = 0;
};

#define ACHK(call)                                                                                \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            char buf_[512];                                                                       \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            a->err = buf_;                                                                        \
            return MW_ERR_HIP;                                                                    \
        }                                                                                         \
    } while (0)

static int afail(mw_aligner* a, int code, const std::string& msg) { a->err = msg; return code; }

template <typename T>
static int aalloc(mw_aligner* a, T** out, size_t count, bool zero = false) {
    void* p = nullptr;
    const size_t bytes = count * sizeof(T) + 256;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) { a->err = std::string("hipMalloc of ") + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e); return MW_ERR_NOMEM; }
    a->allocs.push_back(p);
    if (zero && hipMemset(p, 0, bytes) != hipSuccess) { a->err = "hipMemset failed"; return MW_ERR_HIP; }
    *out = reinterpret_cast<T*>(p);
    return MW_OK;
}
#define AALLOC(ptr, count, ...) do { int r_ = aalloc(a, &(ptr), (count), ##__VA_ARGS__); if (r_) return r_; } while (0)

static long frames_after(const mw_align_desc& d, long n, int upto) {
    for (int i = 0; i < upto; ++i) n = n < d.conv_kernel[i] ? 0 : (n - d.conv_kernel[i]) / d.conv_stride[i] + 1;
    return n;
}

// Everything one stream of work needs besides the weights (a fork owns only this)
static int alloc_align_work(mw_aligner* a) {
    const mw_align_desc& d = a->d;
    a->max_t0 = (int)frames_after(d, d.max_samples, 1);
    a->max_frames = (int)frames_after(d, d.max_samples, d.n_conv);
    a->vpad = (d.vocab + 63) / 64 * 64;
    const size_t C = d.conv_dim, D = d.hidden, F = d.ffn_dim, T = a->max_frames + kPackSlackRows, H = d.n_heads;
    const size_t Tpad = (T + 7) / 8 * 8;
    AALLOC(a->pcm, (size_t)d.max_samples);
    AALLOC(a->h[0], ((size_t)a->max_t0 + kPackSlackRows) * C);
    AALLOC(a->h[1], ((size_t)frames_after(d, d.max_samples, 2) + kPackSlackRows) * C + C);
    AALLOC(a->x, T * D); AALLOC(a->xn, T * D);
    AALLOC(a->pack, (size_t)(T + (size_t)d.pos_kernel * kMaxPackSegments) * D);
    // att starts at zero: in a packed pass the rows past a segment's own frames are never written, and what they hold
    // is multiplied by a probability of zero in the attention of the layers that follow: it must be finite
    AALLOC(a->q, T * D); AALLOC(a->k, T * D); AALLOC(a->vT, H * 64 * (Tpad + 8 * kMaxPackSegments), true); AALLOC(a->att, T * D, true);
    AALLOC(a->hid, T * F);
    AALLOC(a->logits, T * (size_t)a->vpad);
    AALLOC(a->tok_dev, (size_t)kMaxPackSegments * kMaxSegmentTokens);
    AALLOC(a->keys_dev, (size_t)kMaxPackSegments);
    // gathered columns: one segment may have kMaxSegmentTokens characters over all the frames; a pack of shorter
    // segments shares the same room (checked per call)
    a->gath_floats = T * (size_t)(kMaxSegmentTokens + 1);
    AALLOC(a->gath_dev, a->gath_floats);
    if (hipHostMalloc(reinterpret_cast<void**>(&a->gath_host), a->gath_floats * 4, hipHostMallocDefault) != hipSuccess) {
        a->err = "hipHostMalloc failed"; return MW_ERR_NOMEM;
    }
    return MW_OK;
}

extern "C" int64_t mw_align_frames(const mw_align_desc* desc, int64_t n_samples) {
    if (!desc || n_samples < 0) return 0;
    return frames_after(*desc, n_samples, desc->n_conv);
}

extern "C" int mw_align_create(const mw_align_desc* desc, int device, mw_aligner** out) {
    if (!desc || !out) { g_align_create_error = "mw_align_create: null argument"; return MW_ERR_ARG; }
    *out = nullptr;
    const mw_align_desc& d = *desc;
    auto bad = [&](const char* m) { g_align_create_error = std::string("mw_align_create: ") + m; return MW_ERR_ARG; };
    if (d.n_conv < 2 || d.n_conv > 8) return bad("n_conv must be in 2..8");
    if (d.conv_dim < 64 || d.conv_dim % 64 || d.conv_dim > 512) return bad("conv_dim must be a multiple of 64, at most 512");
    if (d.conv_kernel[0] < 1 || d.conv_kernel[0] > 16) return bad("the first convolution's kernel must be in 1..16");
    for (int i = 0; i < d.n_conv; ++i) if (d.conv_kernel[i] < 1 || d.conv_stride[i] < 1) return bad("bad convolution kernel / stride");
    if (d.hidden <= 0 || d.n_heads <= 0 || d.hidden != d.n_heads * kHeadDim) return bad("head_dim must be 64");
    if (d.hidden % 64 || d.ffn_dim % 64 || d.hidden > 2048) return bad("hidden and ffn_dim must be multiples of 64, hidden <= 2048");
    if (d.pos_groups < 1 || d.hidden % d.pos_groups || (d.hidden / d.pos_groups) % 8 || d.pos_kernel < 2 || (d.pos_kernel * (d.hidden / d.pos_groups)) % 64)
        return bad("positional convolution: hidden / groups must be a multiple of 8 and kernel * hidden / groups a multiple of 64");
    if (d.n_layers < 1 || d.vocab < 2) return bad("bad layer count / vocabulary");
    if (d.max_samples < 400 || d.max_samples > 16000 * 600) return bad("max_samples must be in 400 .. 9600000");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { g_align_create_error = "mw_align_create: no such HIP device"; return MW_ERR_HIP; }
    mw_aligner* a = new mw_aligner();
    a->d = d; a->device = device;
    auto bail = [&](int code) { g_align_create_error = a->err; mw_align_destroy(a); return code; };
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking) != hipSuccess) { a->err = "stream creation failed"; return bail(MW_ERR_HIP); }
    int r = alloc_align_work(a);
    if (r) return bail(r);
    *out = a;
    return MW_OK;
}

extern "C" int mw_align_fork(mw_aligner* parent, mw_aligner** out) {
    if (!parent || !out) return MW_ERR_ARG;
    *out = nullptr;
    if (!parent->finalized) return afail(parent, MW_ERR_STATE, "mw_align_fork: the parent is not finalized");
    mw_aligner* a = new mw_aligner();
    a->d = parent->d; a->device = parent->device; a->parent = parent;
    a->conv = parent->conv; a->proj_ln = parent->proj_ln; a->proj = parent->proj; a->pos_w = parent->pos_w; a->pos_b = parent->pos_b;
    a->layers = parent->layers; a->enc_ln = parent->enc_ln; a->head = parent->head;
    a->finalized = true;
    auto bail = [&](int code) { parent->err = "mw_align_fork: " + a->err; mw_align_destroy(a); return code; };
    if (hipSetDevice(a->device) != hipSuccess || hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking) != hipSuccess) { a->err = "stream creation failed"; return bail(MW_ERR_HIP); }
    int r = alloc_align_work(a);
    if (r) return bail(r);
    *out = a;
    return MW_OK;
}

extern "C" void mw_align_destroy(mw_aligner* a) {
    if (!a) return;
    (void)hipSetDevice(a->device);
    if (a->stream) (void)hipStreamSynchronize(a->stream);
    for (void* p : a->allocs) (void)hipFree(p);
    if (a->gath_host) (void)hipHostFree(a->gath_host);
    if (a->stream) (void)hipStreamDestroy(a->stream);
    delete a;
}

extern "C" const char* mw_align_last_error(const mw_aligner* a) { return a ? a->err.c_str() : g_align_create_error.c_str(); }

extern "C" int mw_align_set_weight(mw_aligner* a, const char* name_c, const void* data, const int64_t* shape, int ndim, int dtype) {
    if (!a) return MW_ERR_ARG;
    if (a->finalized) return afail(a, MW_ERR_STATE, "mw_align_set_weight: already finalized");
    if (a->parent) return afail(a, MW_ERR_STATE, "mw_align_set_weight: a fork shares its parent's weights");
    if (!name_c || !data || !shape || ndim < 1 || ndim > 3) return afail(a, MW_ERR_ARG, "mw_align_set_weight: null or bad argument");
    if (dtype < MW_F32 || dtype > MW_BF16) return afail(a, MW_ERR_ARG, "mw_align_set_weight: dtype must be MW_F32, MW_F16 or MW_BF16");
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) { if (shape[i] < 1) return afail(a, MW_ERR_ARG, "mw_align_set_weight: bad shape"); n *= (size_t)shape[i]; }
    const size_t esz = dtype == MW_F32 ? 4 : 2;
    std::vector<uint8_t> raw(n * esz);
    ACHK(hipSetDevice(a->device));
    ACHK(hipMemcpy(raw.data(), data, raw.size(), hipMemcpyDefault));        // host or device source
    std::vector<float> f(n);
    if (dtype == MW_F32) memcpy(f.data(), raw.data(), n * 4);
    else if (dtype == MW_F16) { const _Float16* s = reinterpret_cast<const _Float16*>(raw.data()); for (size_t i = 0; i < n; ++i) f[i] = (float)s[i]; }
    else { const uint16_t* s = reinterpret_cast<const uint16_t*>(raw.data()); for (size_t i = 0; i < n; ++i) { uint32_t u = (uint32_t)s[i] << 16; memcpy(&f[i], &u, 4); } }
    const std::string name(name_c);
    a->host[name] = std::move(f);
    a->shapes[name] = std::vector<int64_t>(shape, shape + ndim);
    return MW_OK;
}

namespace {

struct Importer {
    mw_aligner* a;
    int rc = MW_OK;
    const std::vector<float>* get(const std::string& name, std::vector<int64_t> want) {
        if (rc) return nullptr;
        auto it = a->host.find(name);
        if (it == a->host.end()) { a->err = "mw_align_finalize: weight '" + name + "' was never set"; rc = MW_ERR_MISSING; return nullptr; }
        if (a->shapes[name] != want) { a->err = "mw_align_finalize: wrong shape for '" + name + "'"; rc = MW_ERR_ARG; return nullptr; }
        return &it->second;
    }
    float* up32(const std::vector<float>& v, size_t padded = 0) {
        float* p = nullptr;
        const size_t n = std::max(padded, v.size());
        if (aalloc(a, &p, n, true) != MW_OK) { rc = MW_ERR_NOMEM; return nullptr; }
        if (hipMemcpy(p, v.data(), v.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { a->err = "weight upload failed"; rc = MW_ERR_HIP; }
        return p;
    }
    f16* up16(const std::vector<_Float16>& v) {
        f16* p = nullptr;
        if (aalloc(a, &p, v.size(), true) != MW_OK) { rc = MW_ERR_NOMEM; return nullptr; }
        if (hipMemcpy(p, v.data(), v.size() * 2, hipMemcpyHostToDevice) != hipSuccess) { a->err = "weight upload failed"; rc = MW_ERR_HIP; }
        return p;
    }
    float* vec(const std::string& name, long n, size_t padded = 0) { auto* v = get(name, {n}); return v ? up32(*v, padded) : nullptr; }
    f16* mat(const std::string& name, long r, long c, long padded_rows = 0) {
        auto* v = get(name, {r, c});
        if (!v) return nullptr;
        std::vector<_Float16> h((size_t)std::max(r, padded_rows) * c, (_Float16)0.f);
        for (size_t i = 0; i < v->size(); ++i) h[i] = (_Float16)(*v)[i];
        return up16(h);
    }
    Norm norm(const std::string& prefix, long n) { Norm o; o.g = vec(prefix + ".weight", n); o.b = vec(prefix + ".bias", n); return o; }
};

}  // namespace

extern "C" int mw_align_finalize(mw_aligner* a) {
    if (!a) return MW_ERR_ARG;
    if (a->finalized) return MW_OK;
    ACHK(hipSetDevice(a->device));
    const mw_align_desc& d = a->d;
    const long C = d.conv_dim, D = d.hidden, F = d.ffn_dim, G = d.pos_groups, K = d.pos_kernel, W = D / G;
    Importer im{a};
    a->conv.resize(d.n_conv);
    for (int i = 0; i < d.n_conv; ++i) {
        ConvL& c = a->conv[i];
        c.k = d.conv_kernel[i]; c.s = d.conv_stride[i]; c.cin = i == 0 ? 1 : (int)C; c.cout = (int)C;
        const std::string p = "wav2vec2.feature_extractor.conv_layers." + std::to_string(i);
        auto* w = im.get(p + ".conv.weight", {C, c.cin, c.k});               // [out][in][k]
        if (!w) return im.rc;
        if (i == 0) c.w32 = im.up32(*w);                                     // [C][k] as stored (in = 1)
        else {                                                               // -> fp16 [out][k][in]: a GEMM row of k * C_in contiguous halfs
            std::vector<_Float16> h((size_t)C * c.k * C);
            for (long o = 0; o < C; ++o)
                for (long ci = 0; ci < C; ++ci)
                    for (long j = 0; j < c.k; ++j) h[((size_t)o * c.k + j) * C + ci] = (_Float16)(*w)[((size_t)o * C + ci) * c.k + j];
            c.w = im.up16(h);
        }
        c.b = im.vec(p + ".conv.bias", C);
        c.ln = im.norm(p + ".layer_norm", C);
        if (i > 0 && (c.k * C) % 64) return afail(a, MW_ERR_ARG, "mw_align_finalize: kernel * conv_dim must be a multiple of 64");
    }
    a->proj_ln = im.norm("wav2vec2.feature_projection.layer_norm", C);
    a->proj.w = im.mat("wav2vec2.feature_projection.projection.weight", D, C);
    a->proj.b = im.vec("wav2vec2.feature_projection.projection.bias", D);
    {   // positional convolution: weight norm over dims (0, 1) per tap folded in (modeling_wav2vec2.py:337-355), then
        // per group g an [W out][K * W in] matrix whose row is (tap, input channel)-major like the group-major copy's rows
        const std::string p = "wav2vec2.encoder.pos_conv_embed.conv.";
        std::vector<float> wfull;
        if (a->host.count(p + "weight")) { auto* w = im.get(p + "weight", {D, W, K}); if (!w) return im.rc; wfull = *w; }
        else {
            const bool par = a->host.count(p + "parametrizations.weight.original0") != 0;
            auto* g = im.get(p + (par ? "parametrizations.weight.original0" : "weight_g"), {1, 1, K});
            auto* v = im.get(p + (par ? "parametrizations.weight.original1" : "weight_v"), {D, W, K});
            if (!g || !v) return im.rc;
            std::vector<double> norm(K, 0.0);
            for (size_t i = 0; i < v->size(); ++i) norm[i % K] += (double)(*v)[i] * (*v)[i];
            wfull.resize(v->size());
            for (size_t i = 0; i < v->size(); ++i) wfull[i] = (float)((double)(*g)[i % K] * (*v)[i] / sqrt(norm[i % K]));
        }
        std::vector<_Float16> h((size_t)D * K * W);
        for (long o = 0; o < D; ++o)
            for (long ci = 0; ci < W; ++ci)
                for (long j = 0; j < K; ++j) h[((size_t)o * K + j) * W + ci] = (_Float16)wfull[((size_t)o * W + ci) * K + j];
        a->pos_w = im.up16(h);
        a->pos_b = im.vec(p + "bias", D);
    }
    a->layers.resize(d.n_layers);
    for (int i = 0; i < d.n_layers; ++i) {
        Layer& l = a->layers[i];
        const std::string p = "wav2vec2.encoder.layers." + std::to_string(i);
        l.ln1 = im.norm(p + ".layer_norm", D);
        l.ln2 = im.norm(p + ".final_layer_norm", D);
        std::vector<_Float16> w((size_t)3 * D * D);
        std::vector<float> b((size_t)3 * D);
        int sec = 0;
        for (const char* pj : {"q_proj", "k_proj", "v_proj"}) {
            auto* ww = im.get(p + ".attention." + pj + ".weight", {D, D});
            auto* bb = im.get(p + ".attention." + pj + ".bias", {D});
            if (!ww || !bb) return im.rc;
            for (size_t j = 0; j < ww->size(); ++j) w[(size_t)sec * D * D + j] = (_Float16)(*ww)[j];
            for (long j = 0; j < D; ++j) b[(size_t)sec * D + j] = (*bb)[j];
            ++sec;
        }
        l.qkv.w = im.up16(w); l.qkv.b = im.up32(b);
        l.out.w = im.mat(p + ".attention.out_proj.weight", D, D); l.out.b = im.vec(p + ".attention.out_proj.bias", D);
        l.fc1.w = im.mat(p + ".feed_forward.intermediate_dense.weight", F, D); l.fc1.b = im.vec(p + ".feed_forward.intermediate_dense.bias", F);
        l.fc2.w = im.mat(p + ".feed_forward.output_dense.weight", D, F); l.fc2.b = im.vec(p + ".feed_forward.output_dense.bias", D);
    }
    a->enc_ln = im.norm("wav2vec2.encoder.layer_norm", D);
    a->head.w = im.mat("lm_head.weight", d.vocab, D, a->vpad);             // rows past the vocabulary: zero
    a->head.b = im.vec("lm_head.bias", d.vocab, a->vpad);
    if (im.rc) return im.rc;
    a->host.clear(); a->shapes.clear();
    a->finalized = true;
    return MW_OK;
}

static GemmParams agemm(const f16* A, long lda, const f16* W, long ldw, int M, int N, int K, const float* bias) {
    GemmParams p{};
    p.A = A; p.lda = lda; p.a_rows_per_batch = M > 0 ? M : 1; p.a_batch_stride = 0;
    p.W = W; p.ldw = ldw; p.M = M; p.N = N; p.K = K; p.bias = bias;
    p.scale = 1.f; p.scale_ncols = 0; p.gelu = 0; p.out_mode = OUT_F16;
    p.c_rows_per_batch = M > 0 ? M : 1; p.c_batch_stride = 0;
    return p;
}

// Queue the acoustic model on the aligner's stream for a PACK of `n_seg` segments (no wait).  Every segment is laid
// out as a slice of `L` samples (the longest segment's length, the shorter ones zero-padded behind), so that the
// feature extractor's seven convolutions, the transformer layers' products and LayerNorms run once over all slices:
// a frame of the convolution stack only looks forward, so the first frames_after(n_b) frames of slice b are exactly
// what segment b gives alone and the frames behind them are finite and ignored.  Where frames mix across time the
// segment's own length is used: the positional convolution runs per segment over its own frames (zero padded as the
// model pads them), the attention masks the keys past them.  Log-probabilities land in a->logits, slice b at rows
// b * Tmax .. b * Tmax + frames[b] - 1; *t_stride = Tmax.
static int run_model(mw_aligner* a, int n_seg, const float* const* pcm, const int64_t* n_samples, int* frames, int* t_stride, const char* who) {
    if (!a->finalized) return afail(a, MW_ERR_STATE, std::string(who) + ": mw_align_finalize has not succeeded yet");
    const mw_align_desc& d = a->d;
    if (n_seg < 1 || n_seg > kMaxPackSegments) return afail(a, MW_ERR_ARG, std::string(who) + ": 1.." + std::to_string(kMaxPackSegments) + " segments per call");
    int64_t L = 0;
    for (int b = 0; b < n_seg; ++b) {
        if (!pcm[b]) return afail(a, MW_ERR_ARG, std::string(who) + ": null segment");
        L = std::max(L, n_samples[b]);
    }
    if (L * n_seg > d.max_samples)
        return afail(a, MW_ERR_ARG, std::string(who) + ": more than max_samples samples (" + std::to_string(d.max_samples) + ")" +
                                    (n_seg > 1 ? " in " + std::to_string(n_seg) + " slices of the longest segment's length" : ""));
    const int T = (int)frames_after(d, L, d.n_conv);               // frames of a slice
    for (int b = 0; b < n_seg; ++b) {
        frames[b] = (int)frames_after(d, n_samples[b], d.n_conv);
        if (frames[b] < 1) return afail(a, MW_ERR_ARG, std::string(who) + ": the input is shorter than one frame of the feature extractor");
    }
    if ((long)T * n_seg > a->max_frames + kPackSlackRows) return afail(a, MW_ERR_ARG, std::string(who) + ": the pack's frames exceed the workspaces");
    ACHK(hipSetDevice(a->device));
    hipStream_t s = a->stream;
    const int C = d.conv_dim, D = d.hidden, F = d.ffn_dim, H = d.n_heads, G = d.pos_groups, K = d.pos_kernel, W = D / G;
    const bool ragged = n_seg > 1;
    if (ragged) ACHK(hipMemsetAsync(a->pcm, 0, (size_t)L * n_seg * 4, s));
    for (int b = 0; b < n_seg; ++b) ACHK(hipMemcpyAsync(a->pcm + (size_t)b * L, pcm[b], (size_t)n_samples[b] * 4, hipMemcpyDefault, s));
    // ---- feature extractor: conv -> LayerNorm over channels -> GELU, seven times; slice b's frames lie Ti apart
    int Ti = (int)frames_after(d, L, 1);
    for (int b = 0; b < n_seg; ++b)
        launch_w2v_conv0(a->pcm + (size_t)b * L, L, a->conv[0].w32, a->conv[0].b, a->conv[0].ln.g, a->conv[0].ln.b, a->h[0] + (size_t)b * Ti * C, Ti, C,
                         d.conv_kernel[0], d.conv_stride[0], s);
    int cur = 0;
    for (int i = 1; i < d.n_conv; ++i) {
        const ConvL& c = a->conv[i];
        const int To = (Ti - c.k) / c.s + 1;
        GemmParams p = agemm(a->h[cur], (long)c.s * C, c.w, (long)c.k * C, To * n_seg, C, c.k * C, c.b);
        p.a_rows_per_batch = To; p.a_batch_stride = (long)Ti * C;
        p.C = a->h[cur ^ 1]; p.ldc = C; p.c_rows_per_batch = To; p.c_batch_stride = (long)To * C;
        launch_gemm(p, s);
        launch_ln16(a->h[cur ^ 1], c.ln.g, c.ln.b, To * n_seg, C, 1e-5f, 1, s);
        cur ^= 1; Ti = To;
    }
    const int M = T * n_seg;                                        // Ti == T now: rows of every product below
    // ---- feature projection: LayerNorm, Linear C -> D into the fp32 residual stream
    launch_ln16(a->h[cur], a->proj_ln.g, a->proj_ln.b, M, C, 1e-5f, 0, s);
    ACHK(hipMemsetAsync(a->x, 0, (size_t)M * D * 4, s));
    {
        GemmParams p = agemm(a->h[cur], C, a->proj.w, C, M, D, C, a->proj.b);
        p.out_mode = OUT_RESID_F32; p.C = a->x; p.ldc = D;
        launch_gemm(p, s);
    }
    // ---- x += GELU(grouped positional convolution of x): per segment over its own frames (the model pads a recording
    // with zeros at both ends), one GEMM per group over the group-major padded copy, all groups in one launch
    for (int b = 0; b < n_seg; ++b) {
        float* xb = a->x + (size_t)b * T * D;
        f16* packb = a->pack + (size_t)b * (T + K) * D;
        launch_w2v_pos_pack(xb, packb, frames[b], D, G, K, s);
        GemmParams p = agemm(packb, W, a->pos_w, (long)K * W, frames[b], W, K * W, a->pos_b);
        p.gelu = 1; p.out_mode = OUT_RESID_F32; p.C = xb; p.ldc = D;
        p.groups = G; p.a_group_stride = (long)(frames[b] + K) * W; p.w_group_stride = (long)W * K * W; p.c_group_stride = W;
        launch_gemm(p, s);
    }
    // ---- transformer layers (pre-LayerNorm), then the encoder's final LayerNorm
    const int Tpad = (T + 7) / 8 * 8;
    const float qscale = 1.0f / sqrtf((float)kHeadDim);
    ACHK(hipMemsetAsync(a->vT, 0, (size_t)n_seg * H * 64 * Tpad * 2, s));
    const int* keys_of = nullptr;
    if (ragged) {
        ACHK(hipMemcpyAsync(a->keys_dev, frames, (size_t)n_seg * 4, hipMemcpyHostToDevice, s));
        keys_of = a->keys_dev;
    }
    for (int li = 0; li < d.n_layers; ++li) {
        const Layer& l = a->layers[li];
        launch_layernorm(a->x, l.ln1.g, l.ln1.b, a->xn, nullptr, M, D, s);
        {
            GemmParams p = agemm(a->xn, D, l.qkv.w, D, M, 3 * D, D, l.qkv.b);
            p.scale = qscale; p.scale_ncols = D; p.out_mode = OUT_QKV_SPLIT;
            p.q = a->q; p.k = a->k; p.v = a->vT; p.T = T; p.Tpad = Tpad; p.H = H; p.D = D;
            launch_gemm(p, s);
        }
        launch_enc_attention(a->q, a->k, a->vT, a->att, n_seg, H, T, Tpad, s, keys_of);
        {   GemmParams p = agemm(a->att, D, l.out.w, D, M, D, D, l.out.b); p.out_mode = OUT_RESID_F32; p.C = a->x; p.ldc = D; launch_gemm(p, s); }
        launch_layernorm(a->x, l.ln2.g, l.ln2.b, a->xn, nullptr, M, D, s);
        {   GemmParams p = agemm(a->xn, D, l.fc1.w, D, M, F, D, l.fc1.b); p.gelu = 1; p.C = a->hid; p.ldc = F; launch_gemm(p, s); }
        {   GemmParams p = agemm(a->hid, F, l.fc2.w, F, M, D, F, l.fc2.b); p.out_mode = OUT_RESID_F32; p.C = a->x; p.ldc = D; launch_gemm(p, s); }
    }
    launch_layernorm(a->x, a->enc_ln.g, a->enc_ln.b, a->xn, nullptr, M, D, s);
    // ---- CTC head and log-softmax over the vocabulary
    ACHK(hipMemsetAsync(a->logits, 0, (size_t)M * a->vpad * 4, s));
    {
        GemmParams p = agemm(a->xn, D, a->head.w, D, M, a->vpad, D, a->head.b);
        p.out_mode = OUT_RESID_F32; p.C = a->logits; p.ldc = a->vpad;
        launch_gemm(p, s);
    }
    launch_log_softmax_rows(a->logits, M, a->vpad, d.vocab, s);
    ACHK(hipGetLastError());
    *t_stride = T;
    return MW_OK;
}

extern "C" int mw_align_emissions(mw_aligner* a, const float* pcm, int64_t n_samples, float* out, int32_t* n_frames) {
    if (!a) return MW_ERR_ARG;
    if (!pcm || !out || !n_frames) return afail(a, MW_ERR_ARG, "mw_align_emissions: null argument");
    int T = 0, stride = 0;
    if (int r = run_model(a, 1, &pcm, &n_samples, &T, &stride, "mw_align_emissions")) return r;
    const mw_align_desc& d = a->d;
    ACHK(hipMemcpy2DAsync(out, (size_t)d.vocab * 4, a->logits, (size_t)a->vpad * 4, (size_t)d.vocab * 4, T, hipMemcpyDefault, a->stream));
    ACHK(hipStreamSynchronize(a->stream));
    *n_frames = T;
    return MW_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// CTC forced alignment on the host: trellis over (frames, tokens) and the walk back, in float32 like the torch code
// of whisperx's alignment.py (get_trellis / backtrack, after torchaudio's forced-alignment tutorial).  A token of
// -1 is a wildcard (a character outside the vocabulary): it scores the best non-blank entry of the frame.
//   trellis[0][0] = 0, trellis[t][0] = sum of blank log-probabilities of frames 1..t, trellis[0][j > 0] = -inf,
//   trellis[t][0] = +inf for the last n_tokens - 1 frames (token 0 must have been left by then),
//   trellis[t + 1][j] = max(trellis[t][j] + blank(t), trellis[t][j - 1] + token_j(t)).
// The walk starts at the last frame on the last token and moves to token j - 1 wherever that was the larger term.
// `blank` [T] and `lp` [T][J] hold the only entries of the emissions the search reads: the blank's log-probability per
// frame, and per frame and token the token's (a wildcard's: the best non-blank entry of the frame).
static int path_search(const float* blank, const float* lp, int T, int J, long ld, int32_t* token_index, float* score) {
    const float inf = INFINITY;
    std::vector<float> tr((size_t)T * J);
    tr[0] = 0.f;
    for (int t = 1; t < T; ++t) tr[(size_t)t * J] = tr[(size_t)(t - 1) * J] + blank[t];
    for (int j = 1; j < J; ++j) tr[j] = -inf;
    if (J > 1) for (int t = std::max(0, T - (J - 1)); t < T; ++t) tr[(size_t)t * J] = inf;
    for (int t = 0; t + 1 < T; ++t) {
        const float* cur = &tr[(size_t)t * J];
        float* nxt = &tr[(size_t)(t + 1) * J];
        const float b = blank[t];
        const float* l = lp + (size_t)t * ld;
        for (int j = 1; j < J; ++j) {
            const float stay = cur[j] + b, move = cur[j - 1] + l[j];
            nxt[j] = stay > move ? stay : move;               // max; a NaN (inf - inf) cannot arise: -inf only meets finite terms
        }
    }
    int t = T - 1, j = J - 1, n = 0;
    auto put = [&](int jj, int tt, float p) { token_index[tt] = jj; score[tt] = expf(p); ++n; };
    put(j, t, blank[t]);
    while (j > 0) {
        if (t <= 0) return MW_ERR_STATE;                       // fewer frames than tokens: no path
        const float p_stay = blank[t - 1], p_change = lp[(size_t)(t - 1) * ld + j];
        const float stayed = tr[(size_t)(t - 1) * J + j] + p_stay, changed = tr[(size_t)(t - 1) * J + j - 1] + p_change;
        --t;
        const bool ch = changed > stayed;
        if (ch) --j;
        put(j, t, ch ? p_change : p_stay);
    }
    while (t > 0) { put(j, t - 1, blank[t - 1]); --t; }
    return n == T ? MW_OK : MW_ERR_STATE;
}

extern "C" int mw_align_path(const float* em, int32_t T, int32_t V, const int32_t* tokens, int32_t J, int32_t blank, int32_t* token_index,
                             float* score) {
    if (!em || !tokens || !token_index || !score || T < 1 || J < 1 || V < 2 || blank < 0 || blank >= V) return MW_ERR_ARG;
    bool wild = false;
    for (int j = 0; j < J; ++j) {
        if (tokens[j] < -1 || tokens[j] >= V) return MW_ERR_ARG;
        wild |= tokens[j] == -1;
    }
    std::vector<float> b(T), lp((size_t)T * J);
    for (int t = 0; t < T; ++t) {
        const float* row = em + (size_t)t * V;
        float m = -INFINITY;
        if (wild) for (int v = 0; v < V; ++v) if (v != blank && row[v] > m) m = row[v];
        b[t] = row[blank];
        for (int j = 0; j < J; ++j) lp[(size_t)t * J + j] = tokens[j] == -1 ? m : row[tokens[j]];
    }
    return path_search(b.data(), lp.data(), T, J, J, token_index, score);
}

// Segments end to end: the acoustic model over the whole pack, then each segment's search, with only the columns the
// search reads (the blank and the segment's tokens: [T][1 + n_tokens] instead of [T][vocab]) gathered on the GPU
// and brought to the host through a pinned buffer.  tokens: the segments' token lists back to back; token_index and
// score: the segments' frames back to back (mw_align_frames of each); status[b]: MW_OK, or MW_ERR_STATE when segment
// b has no path (fewer frames than tokens).
static int align_pack(mw_aligner* a, int n_seg, const float* const* pcm, const int64_t* n_samples, const int32_t* tokens, const int32_t* n_tokens,
                      int32_t blank_id, int32_t* token_index, float* score, int32_t* n_frames, int32_t* status, const char* who) {
    const mw_align_desc& d = a->d;
    if (blank_id < 0 || blank_id >= d.vocab) return afail(a, MW_ERR_ARG, std::string(who) + ": blank_id outside the vocabulary");
    if (n_seg < 1 || n_seg > kMaxPackSegments) return afail(a, MW_ERR_ARG, std::string(who) + ": 1.." + std::to_string(kMaxPackSegments) + " segments per call");
    long n_tok_all = 0;
    for (int b = 0; b < n_seg; ++b) {
        if (n_tokens[b] < 1 || n_tokens[b] > kMaxSegmentTokens) return afail(a, MW_ERR_ARG, std::string(who) + ": n_tokens must be in 1.." + std::to_string(kMaxSegmentTokens));
        for (int j = 0; j < n_tokens[b]; ++j)
            if (tokens[n_tok_all + j] < -1 || tokens[n_tok_all + j] >= d.vocab) return afail(a, MW_ERR_ARG, std::string(who) + ": token id outside the vocabulary");
        n_tok_all += n_tokens[b];
    }
    int frames[kMaxPackSegments], stride = 0;
    if (int r = run_model(a, n_seg, pcm, n_samples, frames, &stride, who)) return r;
    size_t need = 0;
    for (int b = 0; b < n_seg; ++b) need += (size_t)frames[b] * (n_tokens[b] + 1);
    if (need > a->gath_floats) return afail(a, MW_ERR_ARG, std::string(who) + ": frames x tokens of the pack exceed the gather buffer");
    hipStream_t s = a->stream;
    ACHK(hipMemcpyAsync(a->tok_dev, tokens, (size_t)n_tok_all * 4, hipMemcpyHostToDevice, s));
    size_t at = 0; long tok_at = 0;
    for (int b = 0; b < n_seg; ++b) {
        launch_ctc_gather(a->logits + (size_t)b * stride * a->vpad, frames[b], a->vpad, d.vocab, a->tok_dev + tok_at, n_tokens[b], blank_id,
                          a->gath_dev + at, s);
        at += (size_t)frames[b] * (n_tokens[b] + 1); tok_at += n_tokens[b];
    }
    ACHK(hipGetLastError());
    ACHK(hipMemcpyAsync(a->gath_host, a->gath_dev, need * 4, hipMemcpyDeviceToHost, s));
    ACHK(hipStreamSynchronize(s));
    // the searches of a pack's segments are independent (225 k trellis cells for a 30 s window of 150 characters: 1.5 ms
    // each): one host thread per segment beyond the first
    std::vector<size_t> g_at(n_seg), f_at(n_seg);
    std::vector<int> rc(n_seg, MW_OK);
    at = 0;
    size_t frame_at = 0;
    for (int b = 0; b < n_seg; ++b) {
        n_frames[b] = frames[b];
        g_at[b] = at; f_at[b] = frame_at;
        at += (size_t)frames[b] * (n_tokens[b] + 1); frame_at += frames[b];
    }
    auto search = [&](int b) {
        const int T = frames[b], ld = n_tokens[b] + 1;
        // column 0 of the gathered rows is the blank: copy it out so that the search reads two plain arrays
        std::vector<float> blank(T);
        for (int t = 0; t < T; ++t) blank[t] = a->gath_host[g_at[b] + (size_t)t * ld];
        rc[b] = path_search(blank.data(), a->gath_host + g_at[b] + 1, T, n_tokens[b], ld, token_index + f_at[b], score + f_at[b]);
    };
    {
        std::vector<std::thread> workers;
        for (int b = 1; b < n_seg; ++b) workers.emplace_back(search, b);
        search(0);
        for (auto& w : workers) w.join();
    }
    int worst = MW_OK;
    for (int b = 0; b < n_seg; ++b) {
        if (status) status[b] = rc[b];
        if (rc[b] != MW_OK) worst = rc[b];
    }
    if (worst == MW_ERR_STATE) a->err = std::string(who) + ": no path (fewer frames than tokens)";
    return worst;
}

extern "C" int mw_align_segment(mw_aligner* a, const float* pcm, int64_t n_samples, const int32_t* tokens, int32_t n_tokens, int32_t blank_id,
                                int32_t* token_index, float* score, int32_t* n_frames) {
    if (!a) return MW_ERR_ARG;
    if (!pcm || !tokens || !token_index || !score || !n_frames) return afail(a, MW_ERR_ARG, "mw_align_segment: null argument");
    return align_pack(a, 1, &pcm, &n_samples, tokens, &n_tokens, blank_id, token_index, score, n_frames, nullptr, "mw_align_segment");
}

extern "C" int mw_align_segments(mw_aligner* a, int32_t n_segments, const float* const* pcm, const int64_t* n_samples, const int32_t* tokens,
                                 const int32_t* n_tokens, int32_t blank_id, int32_t* token_index, float* score, int32_t* n_frames,
                                 int32_t* status) {
    if (!a) return MW_ERR_ARG;
    if (!pcm || !n_samples || !tokens || !n_tokens || !token_index || !score || !n_frames || !status)
        return afail(a, MW_ERR_ARG, "mw_align_segments: null argument");
    const int r = align_pack(a, n_segments, pcm, n_samples, tokens, n_tokens, blank_id, token_index, score, n_frames, status, "mw_align_segments");
    return r == MW_ERR_STATE ? MW_OK : r;          // segments without a path are reported through status[]
}
