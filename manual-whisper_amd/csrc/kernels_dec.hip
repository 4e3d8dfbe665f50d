// Decode-step kernels for gfx950: every launch of one greedy step, sized for a batch of at most
// 32 sequences.  All of them read the current position from device memory (DecState), so one
// captured hipGraph of a step is replayed for every token.
//
// Roofline: each kernel streams its weights (or the K/V cache) exactly once from HBM and does
// O(batch) work per byte, so all are HBM/latency-bound.  What decides their speed at batch 16 is
// the number of DEPENDENT memory round trips per workgroup, so every kernel here issues all of its
// global loads up front, branch-free (out-of-range pieces load a clamped, valid address and are
// zeroed by a select afterwards), and only then starts to compute.  The matrix products go through
// v_mfma_f32_16x16x32_f16 with the batch on the MFMA column index.
#include "mw_kernels.h"

#include <stdlib.h>

#include <mutex>
#include <type_traits>

namespace mw {

namespace {

// Streamed-once loads.  Decoder weights, the token embedding of the logits product and the cross-attention K/V
// cache are each read exactly once per step and a step moves far more than the 256 MiB Infinity Cache holds, so
// nothing is gained by keeping them in the caches: the non-temporal policy (`nt`) lets the lines pass through.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ uint4 load16(const void* p) {
    u32x4 v;
    if constexpr (NT) v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    else v = *reinterpret_cast<const u32x4*>(p);
    return make_uint4(v[0], v[1], v[2], v[3]);
}
template <bool NT>
__device__ __forceinline__ half8 load_half8(const void* p) {
    const uint4 u = load16<NT>(p);
    return *reinterpret_cast<const half8*>(&u);
}
template <bool NT>
__device__ __forceinline__ uint2 load8(const void* p) {
    u32x2 v;
    if constexpr (NT) v = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p));
    else v = *reinterpret_cast<const u32x2*>(p);
    return make_uint2(v[0], v[1]);
}
#ifdef MW_NT_GEMV
constexpr bool kNtGemv = MW_NT_GEMV;
#else
constexpr bool kNtGemv = false;
#endif
#ifdef MW_NT_XATTN
constexpr bool kNtXattn = MW_NT_XATTN;
#else
constexpr bool kNtXattn = true;      // measured: 22.9 -> 22.0 us per launch at 16 sequences, +3.4 % end to end with decode groups of 112
#endif
#ifdef MW_NT_LOGITS
constexpr bool kNtLogits = MW_NT_LOGITS;
#else
constexpr bool kNtLogits = false;
#endif

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float half_wave_sum(float v) {      // over the 32 lanes that share lane>>5
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// 16 e4m3 codes (one 16-byte weight fragment) -> two fp16 MFMA operands: codes 0..7 and 8..15.
// Every e4m3 value is exactly an fp16 value, so the product below is the product over the codes;
// the per-row weight scale is applied to the accumulated sum in the epilogue.
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void e4m3x16_to_f16(const uint4& w, half8& lo, half8& hi) {
    const unsigned u[4] = {w.x, w.y, w.z, w.w};
    half2v h[8];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        h[2 * d] = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(u[d], 1.0f, false);
        h[2 * d + 1] = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(u[d], 1.0f, true);
    }
    lo = half8{h[0][0], h[0][1], h[1][0], h[1][1], h[2][0], h[2][1], h[3][0], h[3][1]};
    hi = half8{h[4][0], h[4][1], h[5][0], h[5][1], h[6][0], h[6][1], h[7][0], h[7][1]};
}

// Batches beyond one MFMA row tile set (16*MT sequences; MT = 2 by default for such batches) run as GROUPS: the same launch carries one copy of
// its workgroups per group of 16*MT sequences (grid y, or z for the split-K form), all reading the same weight
// rows at the same time, so HBM sees the weights once and the other groups hit L2.  A group's workgroups see
// a parameter block whose row-indexed pointers start at the group's first sequence.
__device__ __forceinline__ void group_view(GemvParams& p, int g, int rows_per_group) {
    const long r0 = (long)g * rows_per_group;
    p.B = min(rows_per_group, p.B - (int)r0);
    if (p.x32) p.x32 += r0 * p.K;
    if (p.xin16) p.xin16 += r0 * p.K;
    if (p.out16) p.out16 += r0 * (p.out_mode == GV_QKV ? p.D : p.N);
    if (p.out32) p.out32 += r0 * p.N;
    if (p.kcache) { p.kcache += r0 * p.H * p.Tctx * 64; p.vcache += r0 * p.H * p.Tctx * 64; }
    if (p.part_val) { p.part_val += r0 * p.n_parts; p.part_idx += r0 * p.n_parts; }
    if (p.sk_partial) { p.sk_partial += (long)g * p.sk_partial_floats; p.sk_tickets += (long)g * p.sk_groups; }
}

// ------------------------------------------------------------------ skinny GEMM (weights streamed once)
// out[b][n] = epi( sum_k X[b][k] * W[n][k] ),  b < B <= 16*MT.
//
// Workgroup = NW waves that split K (wave w takes k-steps w, w+NW, ...; KPW of them, compile
// time); it owns `nv` weight rows in R tiles of 16: nv = 8 so that narrow outputs still spread over
// >= 160 workgroups (lanes 8..15 of each fragment re-read rows 0..7 from L1), 16 by default, and
// 17..32 (R = 2, the second tile partly duplicates) when 16 rows would need more than one
// workgroup per CU: past 256 workgroups a launch costs ~2.4 us more.  Weight
// fragments go straight from global memory to the MFMA A operand; the activation fragment is the
// B operand.  Plain input: fp16 activations, fetched with the weights.  LN input (XV > 0): the
// fp32 residual rows are loaded once (32 lanes per row, XV float4 per lane), normalised in
// registers (two-pass statistics, eps 1e-5) and staged in LDS as fp16 for all waves.
// Partial sums of the waves are combined through LDS in a fixed order: bit-for-bit reproducible.
//
// W8: the weights are e4m3 bytes with one fp32 scale per row.  A lane's 16-byte fragment then holds
// 16 codes, so a k-step covers 64 columns (same bytes and the same access pattern per step as the
// fp16 form, half as many steps): the fragment is expanded to two fp16 operands in registers and
// meets the activation columns 16*fg .. 16*fg+7 and 16*fg+8 .. 16*fg+15 of the step.
template <int MT, int KPW, int XV, int R, bool W8>
// The fused form always runs 512 threads.  With one row tile and up to 16 sequences it fits 128 VGPRs without
// spilling, which lets two workgroups share a CU (6.7 against 7.1 us for the QKV launch); the larger forms need more.
__global__ __launch_bounds__(XV > 0 ? 512 : 1024, (XV > 0 && R == 1 && MT == 1) ? 4 : 1) void gemv_kernel(GemvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (gridDim.y > 1) group_view(p, blockIdx.y, 16 * MT);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NW = blockDim.x >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int nv = p.nv;
    const int n0 = blockIdx.x * nv;
    constexpr int KSTEP = W8 ? 64 : 32;          // columns per k-step (64 bytes of a weight row either way)
    constexpr int XF = W8 ? 2 : 1;               // activation fragments per k-step
    const char* wp[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int local = nv == 8 ? (fr & 7) : min(r * 16 + fr, nv - 1);
        const long row = min(n0 + local, p.N - 1);
        wp[r] = (W8 ? reinterpret_cast<const char*>(p.W8) + row * p.K : reinterpret_cast<const char*>(p.W) + row * p.K * 2) + fg * 16;
    }
    const int nks = p.K / KSTEP;

    float4v* red = reinterpret_cast<float4v*>(smem);                       // [NW][R][MT][64]
    char* after_red = smem + (size_t)NW * R * MT * 64 * 16;

    // ---- every global load of this workgroup is issued here.  Loads return in issue order, so the
    // LayerNorm operands (cache hits) go first: the normalisation then runs while the weight
    // stream is still in flight instead of queueing behind it.
    // Fused form: 512 threads, K = 128 * XV (the launcher guarantees both); 32 lanes per row, 16 rows per pass.
    constexpr int XVN = XV > 0 ? XV : 1;
    float4 v[MT][XVN];
    float4 g4 = make_float4(0, 0, 0, 0), b4 = g4;
    const int l32 = tid & 31;
    const int gi = min(tid * 4, p.K - 4);       // 512 threads x 4 >= K; the surplus threads duplicate the last chunk
    if constexpr (XV > 0) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const float* xr = p.x32 + (long)min(t * 16 + (tid >> 5), p.B - 1) * p.K + l32 * 4;   // rows >= B: duplicates
#pragma unroll
            for (int j = 0; j < XV; ++j) v[t][j] = *reinterpret_cast<const float4*>(xr + 128 * j);
        }
        g4 = *reinterpret_cast<const float4*>(p.ln_g + gi);
        b4 = *reinterpret_cast<const float4*>(p.ln_b + gi);
    }
    uint4 wreg[R][KPW];
    bool kval[KPW];
#pragma unroll
    for (int c = 0; c < KPW; ++c) {
        const int ks = wave + NW * c;
        kval[c] = ks < nks;
#pragma unroll
        for (int r = 0; r < R; ++r) wreg[r][c] = load16<kNtGemv>(wp[r] + min(ks, nks - 1) * 64);
    }
    // epilogue operands: wave r finishes row tile r (the position too: read here, it costs no round trip of its own)
    const int pos = (p.out_mode == GV_QKV && p.st) ? p.st->pos : 0;
    const int nl = wave * 16 + 4 * fg;
    const int n = n0 + nl;
    const bool owner = wave < R && nl < nv && n < p.N;                     // N and nv are multiples of 4
    float4 bias4 = make_float4(0, 0, 0, 0), ws4 = make_float4(1, 1, 1, 1);
    float4 res4[MT];
    if (owner) {
        if (p.bias) bias4 = *reinterpret_cast<const float4*>(p.bias + n);
        if constexpr (W8) ws4 = *reinterpret_cast<const float4*>(p.w_scale + n);
        if (p.out_mode == GV_RESID) {
#pragma unroll
            for (int t = 0; t < MT; ++t)
                res4[t] = *reinterpret_cast<const float4*>(p.out32 + (long)min(t * 16 + fr, p.B - 1) * p.N + n);
        }
    }

    half8 xfrag[MT][KPW][XF];
    if constexpr (XV == 0) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const f16* xr = p.xin16 + (long)min(t * 16 + fr, p.B - 1) * p.K + fg * (8 * XF);   // rows >= B: duplicates, never stored
#pragma unroll
            for (int c = 0; c < KPW; ++c)
#pragma unroll
                for (int h = 0; h < XF; ++h)
                    xfrag[t][c][h] = *reinterpret_cast<const half8*>(xr + min(wave + NW * c, nks - 1) * KSTEP + 8 * h);
        }
    } else {
        // LayerNorm of the fp32 rows, two-pass statistics in registers
        float* gam = reinterpret_cast<float*>(after_red);                  // [K]
        float* bet = gam + p.K;                                            // [K]
        f16* xs = reinterpret_cast<f16*>(bet + p.K);                       // [16*MT][K+8]
        const int ldx = p.K + 8;
        *reinterpret_cast<float4*>(gam + gi) = g4;
        *reinterpret_cast<float4*>(bet + gi) = b4;
        __syncthreads();
        const float inv_k = 1.0f / (128 * XV);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int r = t * 16 + (tid >> 5);
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < XV; ++j) s += (v[t][j].x + v[t][j].y) + (v[t][j].z + v[t][j].w);
            const float mean = half_wave_sum(s) * inv_k;
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < XV; ++j) {
                v[t][j].x -= mean; v[t][j].y -= mean; v[t][j].z -= mean; v[t][j].w -= mean;
                q += (v[t][j].x * v[t][j].x + v[t][j].y * v[t][j].y) + (v[t][j].z * v[t][j].z + v[t][j].w * v[t][j].w);
            }
            const float rstd = rsqrtf(half_wave_sum(q) * inv_k + 1e-5f);
#pragma unroll
            for (int j = 0; j < XV; ++j) {
                const int c = l32 * 4 + 128 * j;
                const float4 g = *reinterpret_cast<const float4*>(gam + c);
                const float4 b = *reinterpret_cast<const float4*>(bet + c);
                *reinterpret_cast<half4*>(xs + r * ldx + c) =
                    half4{(f16)(v[t][j].x * rstd * g.x + b.x), (f16)(v[t][j].y * rstd * g.y + b.y),
                          (f16)(v[t][j].z * rstd * g.z + b.z), (f16)(v[t][j].w * rstd * g.w + b.w)};
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int c = 0; c < KPW; ++c)
#pragma unroll
                for (int h = 0; h < XF; ++h)
                    xfrag[t][c][h] = *reinterpret_cast<const half8*>(xs + (t * 16 + fr) * ldx + min(wave + NW * c, nks - 1) * KSTEP +
                                                                     fg * (8 * XF) + 8 * h);
    }

    float4v acc[R][MT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[r][t] = float4v{0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < KPW; ++c) {
        // a k-step past the end is cancelled on the ACTIVATION side: a select on the weight register would
        // make everything scheduled before it wait for the weight stream
        half8 wf[R][XF];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if constexpr (W8) e4m3x16_to_f16(wreg[r][c], wf[r][0], wf[r][XF - 1]);
            else wf[r][0] = *reinterpret_cast<half8*>(&wreg[r][c]);
        }
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int h = 0; h < XF; ++h) {
                half8 xf = xfrag[t][c][h];
                if (!kval[c]) xf = half8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[r][h], xf, acc[r][t], 0, 0, 0);
            }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int t = 0; t < MT; ++t) red[((wave * R + r) * MT + t) * 64 + lane] = acc[r][t];
    __syncthreads();
    if (!owner) return;

    // ---- epilogue: lane holds out[m][n..n+3]
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = t * 16 + fr;
        if (m >= p.B) continue;
        float4v v = {bias4.x, bias4.y, bias4.z, bias4.w};
        if constexpr (W8) {
            float4v a = red[(wave * MT + t) * 64 + lane];       // w = 0
            for (int w = 1; w < NW; ++w) a += red[((w * R + wave) * MT + t) * 64 + lane];
            v[0] += a[0] * ws4.x; v[1] += a[1] * ws4.y; v[2] += a[2] * ws4.z; v[3] += a[3] * ws4.w;
        } else {
            for (int w = 0; w < NW; ++w) v += red[((w * R + wave) * MT + t) * 64 + lane];
        }
        if (n < p.scale_ncols) v *= p.scale;
        if (p.gelu) { v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]); }
        if (p.out_mode == GV_RESID) {
            *reinterpret_cast<float4*>(p.out32 + (long)m * p.N + n) =
                make_float4(res4[t].x + v[0], res4[t].y + v[1], res4[t].z + v[2], res4[t].w + v[3]);
        } else {
            const half4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
            if (p.out_mode == GV_F16) {
                *reinterpret_cast<half4*>(p.out16 + (long)m * p.N + n) = h;
            } else {   // GV_QKV: q -> out16 [B][D]; k, v -> this layer's cache at the current position
                const int sec = n / p.D, nn = n - sec * p.D;
                if (sec == 0) {
                    *reinterpret_cast<half4*>(p.out16 + (long)m * p.D + nn) = h;
                } else {
                    f16* cache = sec == 1 ? p.kcache : p.vcache;
                    const int hh = nn >> 6, e = nn & 63;
                    *reinterpret_cast<half4*>(cache + (((long)m * p.H + hh) * p.Tctx + pos) * 64 + e) = h;
                }
            }
        }
    }
}

// ------------------------------------------------------------------ split-K residual GEMM
// x[b][n] += bias[n] + sum_k X[b][k] * W[n][k] for the three residual projections of a layer.
// In gemv_kernel every workgroup reads ALL of X (16 rows x K fp16), which for K = 5120 is more
// than its own weight slice: the launch is bound by activation re-reads, not by the weight stream.
// Here a workgroup owns 16*R weight rows and ONE of S ranges of K, so X is read N/(16R) times in
// total instead of N/8 times.  The S partial tiles of a row group are combined by whichever
// workgroup arrives last (same hand-off as the cross-attention merge above: one wave publishes with
// agent-scope stores, drains them, takes a ticket; the last ticket holder reads all S tiles back
// with agent-scope loads and sums them in index order, so the result is independent of arrival
// order), adds bias and residual, writes x and re-arms the ticket.
constexpr int kSplitKWaves = 4;
constexpr int kSplitKMax = 8;

template <int MT, int KPW, int R, bool W8>
__global__ __launch_bounds__(kSplitKWaves * 64) void gemv_splitk_kernel(GemvParams p, int ks_per) {
    constexpr int NW = kSplitKWaves, NT = R * MT;       // NT accumulator tiles per wave
    constexpr int KSTEP = W8 ? 64 : 32, XF = W8 ? 2 : 1;   // as in gemv_kernel
    __shared__ float4v red[NW][NT][64];
    if (gridDim.z > 1) group_view(p, blockIdx.z, 16 * MT);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int S = gridDim.y, split = blockIdx.y, rg = blockIdx.x;
    const int n0 = rg * 16 * R;
    const int nks = p.K / KSTEP;
    const int k0 = split * ks_per;
    const int kend = min(nks, k0 + ks_per);             // > k0: the launcher never creates an empty range

    uint4 wreg[R][KPW];
    half8 xfrag[MT][KPW][XF];
    bool kval[KPW];
    const char* wbase = W8 ? reinterpret_cast<const char*>(p.W8) : reinterpret_cast<const char*>(p.W);
    const long wrow_bytes = W8 ? p.K : 2L * p.K;
#pragma unroll
    for (int c = 0; c < KPW; ++c) {
        const int ks = k0 + wave + NW * c;
        kval[c] = ks < kend;
        const int kstep = min(ks, kend - 1);
#pragma unroll
        for (int r = 0; r < R; ++r)
            wreg[r][c] = load16<kNtGemv>(wbase + (long)(n0 + r * 16 + fr) * wrow_bytes + kstep * 64 + fg * 16);
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int h = 0; h < XF; ++h)
                xfrag[t][c][h] = *reinterpret_cast<const half8*>(p.xin16 + (long)min(t * 16 + fr, p.B - 1) * p.K + kstep * KSTEP +
                                                                 fg * (8 * XF) + 8 * h);
    }
    // epilogue operands of the merging wave (only the last arriver uses them)
    float4 bias4[R], ws4[R], res4[NT];
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int n = n0 + r * 16 + 4 * fg;
            bias4[r] = p.bias ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0, 0, 0, 0);
            ws4[r] = W8 ? *reinterpret_cast<const float4*>(p.w_scale + n) : make_float4(1, 1, 1, 1);
#pragma unroll
            for (int t = 0; t < MT; ++t)
                res4[r * MT + t] = *reinterpret_cast<const float4*>(p.out32 + (long)min(t * 16 + fr, p.B - 1) * p.N + n);
        }
    }

    float4v acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = float4v{0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < KPW; ++c) {
        half8 wf[R][XF];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if constexpr (W8) e4m3x16_to_f16(wreg[r][c], wf[r][0], wf[r][XF - 1]);
            else wf[r][0] = *reinterpret_cast<half8*>(&wreg[r][c]);
        }
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int h = 0; h < XF; ++h) {
                half8 xf = xfrag[t][c][h];
                if (!kval[c]) xf = half8{0, 0, 0, 0, 0, 0, 0, 0};       // cancel on the activation side (see gemv_kernel)
#pragma unroll
                for (int r = 0; r < R; ++r)
                    acc[r * MT + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[r][h], xf, acc[r * MT + t], 0, 0, 0);
            }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) red[wave][i][lane] = acc[i];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        acc[i] = red[0][i][lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) acc[i] += red[w][i][lane];
    }
    // publish: slab[rg][split][NT*4][64] floats, every store a contiguous 256 B per wave
    float* slab = p.sk_partial + (long)rg * S * (NT * 4 * 64);
    if (S > 1) {
        float* mine = slab + (long)split * (NT * 4 * 64) + lane;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __hip_atomic_store(mine + (i * 4 + j) * 64, acc[i][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's tile has left the CU
        int ticket = 0;
        if (lane == 0) ticket = __hip_atomic_fetch_add(p.sk_tickets + rg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = __shfl(ticket, 0);
        if (ticket != S - 1) return;
        // last arriver: all S tiles of this row group are complete; sum in index order
        float part[kSplitKMax][NT * 4];
#pragma unroll
        for (int i = 0; i < kSplitKMax; ++i) {
            const float* src = slab + (long)min(i, S - 1) * (NT * 4 * 64) + lane;
#pragma unroll
            for (int e = 0; e < NT * 4; ++e)
                part[i][e] = __hip_atomic_load(src + e * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int e = 0; e < NT * 4; ++e) {
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < kSplitKMax; ++i) v += i < S ? part[i][e] : 0.f;
            acc[e >> 2][e & 3] = v;
        }
        if (lane == 0) __hip_atomic_store(p.sk_tickets + rg, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int n = n0 + r * 16 + 4 * fg;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int m = t * 16 + fr, i = r * MT + t;
            if (m >= p.B) continue;
            *reinterpret_cast<float4*>(p.out32 + (long)m * p.N + n) =
                W8 ? make_float4(res4[i].x + (acc[i][0] * ws4[r].x + bias4[r].x), res4[i].y + (acc[i][1] * ws4[r].y + bias4[r].y),
                                 res4[i].z + (acc[i][2] * ws4[r].z + bias4[r].z), res4[i].w + (acc[i][3] * ws4[r].w + bias4[r].w))
                   : make_float4(res4[i].x + (acc[i][0] + bias4[r].x), res4[i].y + (acc[i][1] + bias4[r].y),
                                 res4[i].z + (acc[i][2] + bias4[r].z), res4[i].w + (acc[i][3] + bias4[r].w));
        }
    }
}

// ------------------------------------------------------------------ projections for 33..128 sequences
// out[b][n] = epi( sum_k X[b][k] * W[n][k] ) for all B <= 16*MT sequences of a decode group in ONE pass over the
// weights.  gemv_kernel serves such batches as groups of 32 inside a launch: every 16 weight rows fetch the whole
// activation matrix again, and at 128 sequences that is what a projection costs (N/16 * B * K * 2 bytes through
// the L1s: 105 MB for the MLP's first matrix, 14-20 us), not its 3-13 MB of weights.
// Here a workgroup owns 16*RT weight rows and ONE of S ranges of K for ALL sequences.  Its activation slice
// ([16*MT][range] fp16) is staged in LDS once, by LDS-DMA; wave (rt, mg) of the 8 multiplies row tile rt by the
// MT/MG sequence tiles of its group mg over the whole range: the weight fragments go from global memory straight
// to the MFMA A operand (all loads issued up front, behind the DMA), the activation fragments come from LDS,
// and no sum crosses waves.  The activation matrix is thus read N/(16*RT) times in total.  S > 1: the S partial
// tiles of a row group are merged by the workgroup that arrives last, in index order, with the ticket hand-off
// of gemv_splitk_kernel (every wave publishes its tiles with agent-scope stores and drains them, a workgroup
// barrier, one ticket per workgroup, the last arriver's waves read their own tiles back with agent-scope loads
// behind a second barrier).  Input rows are fp16 (the LayerNorm is its own launch at these batch sizes).
// Measured at 128 sequences, large-v3 shapes, per launch with its boundary (tools/gemv_bench): QKV 14.4 us
// (grouped gemv_kernel 16.8), MLP in 13.4 (20.4), MLP out 18.8 (20.6), the square projections 9.5-9.9 (9.3-9.6):
// what is left is the chain of round trips (load, publish, ticket, merge, store), not bytes.
// Tried and dropped: workgroups of 32 sequences over the whole of K (82 KB of activations in LDS, 160 weight
// registers per lane, no partial sums for K = 1280): a workgroup then takes in 250-410 KB, and one CU ingests
// about 25 GB/s however the loads are issued: 13.5 us for the square projections, 29.6 for the MLP's second matrix.
// Many workgroups with little data each beat few with much, merges included.
constexpr int kDecGemmWaves = 8;

// LDS image of the activation slice: 16-byte pieces, kDecPPS(KR) slots per row (the row's pieces at the longest
// range plus one pad slot, so that the 16 rows of an MFMA operand start in 16 different bank groups), filled by
// LDS-DMA in slot order (a wave instruction writes 64 consecutive slots).
template <int KR, bool W8> constexpr int dec_gemm_pps() { return KR * ((W8 ? 64 : 32) / 8) + 1; }
template <int MT, int KR, bool W8> constexpr int dec_gemm_lds_bytes() {
    return (16 * MT * dec_gemm_pps<KR, W8>() + kDecGemmWaves * 64 - 1) / (kDecGemmWaves * 64) * (kDecGemmWaves * 64) * 16;
}

template <int MT, int RT, int KR, bool W8>
__global__ __launch_bounds__(kDecGemmWaves * 64) void gemm_dec_kernel(GemvParams p, int ks_per) {
    constexpr int NW = kDecGemmWaves, MG = NW / RT, TPW = MT / MG;     // tiles of 16 sequences per wave
    constexpr int KSTEP = W8 ? 64 : 32, XF = W8 ? 2 : 1;
    constexpr int PPS = dec_gemm_pps<KR, W8>();
    static_assert(NW % RT == 0 && MT % MG == 0 && TPW >= 1, "8 waves = RT row tiles x MG sequence groups");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_last;
    // more than 16*MT sequences: blockIdx.z picks a block of 16*MT of them (its own rows, cache rows, partial slab
    // and tickets); the blocks of a row group read the same weights at about the same time, so HBM sends them once
    if (gridDim.z > 1) group_view(p, blockIdx.z, 16 * MT);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int rt = wave % RT, mg = wave / RT;
    const int S = gridDim.y, split = blockIdx.y, rg = blockIdx.x;
    const int n0 = rg * 16 * RT;
    const int nks = p.K / KSTEP;
    const int k0 = split * ks_per;
    const int nk = min(nks, k0 + ks_per) - k0;          // >= 1: the launcher never creates an empty range

    // ---- activation slice -> LDS by LDS-DMA: slot s = row * PPS + piece; slots past the range (and the pad slot)
    // repeat the row's last piece, rows >= B of the last tile that holds a sequence repeat the last sequence (never stored)
    {
        constexpr int NI = dec_gemm_lds_bytes<MT, KR, W8>() / 16 / (NW * 64);
        const int ppr = nk * (KSTEP / 8);
        const int live_rows = (p.B + 15) / 16 * 16;          // tiles past the last sequence are neither filled nor published
        const f16* xbase = p.xin16 + (long)k0 * KSTEP;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int slot = (j * NW + wave) * 64 + lane;
            if (((j * NW + wave) * 64) / PPS >= live_rows) continue;  // (wave-uniform)
            const int r = slot / PPS, c8 = slot - r * PPS;             // PPS is a compile-time constant
            const f16* src = xbase + (long)min(r, p.B - 1) * p.K + min(c8, ppr - 1) * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(smem + (j * NW + wave) * 1024), 16, 0, 0);
        }
    }
    // ---- weights of this wave's row tile over the whole range, straight into the MFMA operand registers
    const long wrow = min(n0 + rt * 16 + fr, p.N - 1);
    const char* wp = (W8 ? reinterpret_cast<const char*>(p.W8) + wrow * p.K : reinterpret_cast<const char*>(p.W) + wrow * p.K * 2) +
                     (long)k0 * 64 + fg * 16;
    uint4 wreg[KR];
#pragma unroll
    for (int c = 0; c < KR; ++c) wreg[c] = load16<kNtGemv>(wp + min(c, nk - 1) * 64);
    const int pos = (p.out_mode == GV_QKV && p.st) ? p.st->pos : 0;     // read here: no round trip of its own later
    // epilogue operands of this lane's four columns, requested behind the weights: bias, weight scales and (residual
    // outputs) the rows this tile is added to.  Only the workgroup that merges uses them, but it is the one whose
    // round trips are the launch's critical path, and nobody else writes these rows during the launch.
    const int n_ep = n0 + rt * 16 + 4 * fg;
    const bool n_ok = n_ep < p.N;
    float4 bias4 = make_float4(0, 0, 0, 0), ws4 = make_float4(1, 1, 1, 1);
    float4 res4[TPW];
    if (n_ok) {
        if (p.bias) bias4 = *reinterpret_cast<const float4*>(p.bias + n_ep);
        if constexpr (W8) ws4 = *reinterpret_cast<const float4*>(p.w_scale + n_ep);
        if (p.out_mode == GV_RESID) {
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int m = min((mg * TPW + t) * 16 + fr, p.B - 1);
                res4[t] = *reinterpret_cast<const float4*>(p.out32 + (long)m * p.N + n_ep);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the compiler does not count the LDS-DMA
    __syncthreads();
    const char* xs = smem;

    float4v acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = float4v{0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < KR; ++c) {
        if (c < nk) {                                   // the same for every lane of the workgroup
            half8 wf[XF];
            if constexpr (W8) e4m3x16_to_f16(wreg[c], wf[0], wf[XF - 1]);
            else wf[0] = *reinterpret_cast<half8*>(&wreg[c]);
#pragma unroll
            for (int t = 0; t < TPW; ++t)
#pragma unroll
                for (int h = 0; h < XF; ++h) {
                    const half8 xf = *reinterpret_cast<const half8*>(xs + (((mg * TPW + t) * 16 + fr) * PPS + c * (KSTEP / 8) + fg * XF + h) * 16);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[h], xf, acc[t], 0, 0, 0);
                }
        }
    }
    // lane holds out[(mg*TPW + t)*16 + fr][n .. n+3] of this K range, n = n0 + rt*16 + 4*fg
    // sequence tiles of this wave that hold a sequence (the products of the others were taken over unfilled LDS and go nowhere)
    const int n_live = __builtin_amdgcn_readfirstlane(min(TPW, max(0, (p.B + 15) / 16 - mg * TPW)));
    if (S > 1) {
        // slab[rg][split][wave][TPW][64 lanes] float4: every store and load 16 bytes per lane, 1 KiB contiguous per
        // wave, write-through / L1-bypassing (sc1: aux 16); narrower sc1 stores cost a fabric write each
        const long slab_floats = (long)S * (NW * TPW * 256);
        const __amdgpu_buffer_rsrc_t slab = __builtin_amdgcn_make_buffer_rsrc(p.sk_partial + (long)rg * slab_floats, 0,
                                                                             (int)(slab_floats * 4), 0x00020000);
        const int mine = (((split * NW + wave) * TPW) * 64 + lane) * 16;
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            if (t >= n_live) break;
            const u32x4 bits = {__float_as_uint(acc[t][0]), __float_as_uint(acc[t][1]), __float_as_uint(acc[t][2]), __float_as_uint(acc[t][3])};
            __builtin_amdgcn_raw_buffer_store_b128(bits, slab, mine + t * 1024, 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's tiles have left the CU
        __syncthreads();
        if (tid == 0) s_last = __hip_atomic_fetch_add(p.sk_tickets + rg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == S - 1;
        __syncthreads();
        if (!s_last) return;
        // last arriver: all S partials of this row group are complete; every wave sums its own tiles in index order
        // (this workgroup's own tiles are still in its registers: they take their place in the order without a read)
        float4v own[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) { own[t] = acc[t]; acc[t] = float4v{0, 0, 0, 0}; }
        for (int i = 0; i < S; ++i) {
            if (i == split) {
#pragma unroll
                for (int t = 0; t < TPW; ++t) acc[t] += own[t];
                continue;
            }
            const int theirs = (((i * NW + wave) * TPW) * 64 + lane) * 16;
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                if (t >= n_live) break;
                const u32x4 bits = __builtin_amdgcn_raw_buffer_load_b128(slab, theirs + t * 1024, 0, 16);
                acc[t][0] += __uint_as_float(bits[0]); acc[t][1] += __uint_as_float(bits[1]);
                acc[t][2] += __uint_as_float(bits[2]); acc[t][3] += __uint_as_float(bits[3]);
            }
        }
        if (tid == 0) __hip_atomic_store(p.sk_tickets + rg, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    const int n = n_ep;
    if (!n_ok) return;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int m = (mg * TPW + t) * 16 + fr;
        if (m >= p.B) continue;
        float4v o = {bias4.x, bias4.y, bias4.z, bias4.w};
        o[0] += acc[t][0] * ws4.x; o[1] += acc[t][1] * ws4.y; o[2] += acc[t][2] * ws4.z; o[3] += acc[t][3] * ws4.w;
        if (n < p.scale_ncols) o *= p.scale;
        if (p.gelu) { o[0] = gelu_erf(o[0]); o[1] = gelu_erf(o[1]); o[2] = gelu_erf(o[2]); o[3] = gelu_erf(o[3]); }
        if (p.out_mode == GV_RESID) {
            const float4 r4 = res4[t];
            *reinterpret_cast<float4*>(p.out32 + (long)m * p.N + n) = make_float4(r4.x + o[0], r4.y + o[1], r4.z + o[2], r4.w + o[3]);
        } else {
            const half4 h = {(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3]};
            if (p.out_mode == GV_F16) {
                *reinterpret_cast<half4*>(p.out16 + (long)m * p.N + n) = h;
            } else {   // GV_QKV: q -> out16 [B][D]; k, v -> this layer's cache at the current position
                const int sec = n / p.D, nn = n - sec * p.D;
                if (sec == 0) {
                    *reinterpret_cast<half4*>(p.out16 + (long)m * p.D + nn) = h;
                } else {
                    f16* cache = sec == 1 ? p.kcache : p.vcache;
                    const int hh = nn >> 6, e = nn & 63;
                    *reinterpret_cast<half4*>(cache + (((long)m * p.H + hh) * p.Tctx + pos) * 64 + e) = h;
                }
            }
        }
    }
}

// ------------------------------------------------------------------ logits + partial argmax
// logits[b][n] = xn[b] . E[n] over the tied token embedding (modeling_whisper.py:1080).
// Workgroup = 8 waves, each a tile of 16 vocabulary rows over the full K; the normalised
// activations (fp16) are staged once per workgroup in LDS.  Weights stream through two register
// banks of LCH k-steps so a bank's loads are in flight while the other feeds the MFMAs.  Each
// workgroup also reduces its 128 columns to one (max, lowest index) candidate per sequence under
// the suppress mask of this position; dec_finalize picks the winner.
constexpr int LCH = 8;      // (logits_wide_kernel's waits name this count)
template <int MT>
__global__ __launch_bounds__(512) void logits_kernel(GemvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // More than 16*MT sequences: every 128 vocabulary rows (a "part") are multiplied once per group of 16*MT sequences.
    // Workgroup ids are dealt to the 8 XCDs in turn and each XCD has its own L2, so the groups of one part take
    // CONSECUTIVE ids of ONE XCD (id = (slot * n_groups + group) * 8 + xcd, part = slot * 8 + xcd): the first of
    // them brings the part's 128 embedding rows into that L2 and the others find them there.  (With one grid row
    // per group the groups of a part sat 406 ids apart on different XCDs, and HBM sent the 133 MB embedding once
    // per group: 537 MB per launch at 112 sequences by the fetch counter.)
    int part = blockIdx.x;
    if (p.B > 16 * MT) {
        const int n_groups = (p.B + 16 * MT - 1) / (16 * MT);
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        part = (idx / n_groups) * 8 + xcd;
        if (part >= p.n_parts) return;                  // padding of the last slot (the whole workgroup leaves)
        group_view(p, idx % n_groups, 16 * MT);
    }
    f16* xs = reinterpret_cast<f16*>(smem);                                // [16*MT][K+8]
    const int ldx = p.K + 8;
    float* cand_v = reinterpret_cast<float*>(smem + (size_t)16 * MT * ldx * 2);   // [8][16*MT]
    int* cand_i = reinterpret_cast<int*>(cand_v + 8 * 16 * MT);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int tile = part * 8 + wave;
    const int n0 = tile * 16;
    const int wrow = min(n0 + fr, p.N - 1);
    const f16* wp = p.W + (long)wrow * p.K + fg * 8;
    const int nks = p.K >> 5;

    const int sample = p.st->pos - p.st->prompt_len;       // pos was advanced by the final LayerNorm launch
    uint4 wa[LCH], wb[LCH];
    auto wload = [&](uint4 (&dst)[LCH], int kb) {
#pragma unroll
        for (int c = 0; c < LCH; ++c) dst[c] = load16<kNtLogits>(wp + min(kb + c, nks - 1) * 32);
    };
    wload(wa, 0);
    wload(wb, LCH);
    for (int i = tid * 8; i < 16 * MT * p.K; i += blockDim.x * 8) {       // K is a multiple of 8
        const int r = i / p.K, c = i - r * p.K;
        *reinterpret_cast<uint4*>(xs + r * ldx + c) =
            r < p.B ? *reinterpret_cast<const uint4*>(p.xin16 + (long)r * p.K + c) : make_uint4(0, 0, 0, 0);
    }
    __syncthreads();

    float4v acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = float4v{0, 0, 0, 0};
    auto consume = [&](uint4 (&src)[LCH], int kb) {
#pragma unroll
        for (int c = 0; c < LCH; ++c) {
            uint4 w = src[c];
            if (kb + c >= nks) w = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const half8 xf = *reinterpret_cast<const half8*>(xs + (t * 16 + fr) * ldx + min(kb + c, nks - 1) * 32 + fg * 8);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<half8*>(&w), xf, acc[t], 0, 0, 0);
            }
        }
    };
    for (int kb = 0; kb < nks; kb += 2 * LCH) {
        consume(wa, kb);
        if (kb + 2 * LCH < nks) wload(wa, kb + 2 * LCH);
        consume(wb, kb + LCH);
        if (kb + 3 * LCH < nks) wload(wb, kb + 3 * LCH);
    }

    const uint8_t* mask = sample <= 0 ? p.mask_first : p.mask_rest;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = t * 16 + fr;
        float best = -INFINITY;
        int bidx = 0x7fffffff;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + 4 * fg + r;
            if (n < p.N && m < p.B) {
                const float v = acc[t][r];
                if (p.out32) p.out32[(long)m * p.N + n] = v;
                if (!mask[n] && (v > best || (v == best && n < bidx))) { best = v; bidx = n; }
            }
        }
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            const float ov = __shfl_xor(best, o);
            const int oi = __shfl_xor(bidx, o);
            if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
        }
        if (fg == 0) { cand_v[wave * 16 * MT + m] = best; cand_i[wave * 16 * MT + m] = bidx; }
    }
    __syncthreads();
    if (tid < 16 * MT && tid < p.B) {
        float best = cand_v[tid];
        int bidx = cand_i[tid];
        for (int w = 1; w < 8; ++w) {
            const float ov = cand_v[w * 16 * MT + tid];
            const int oi = cand_i[w * 16 * MT + tid];
            if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
        }
        p.part_val[(long)tid * p.n_parts + part] = best;
        p.part_idx[(long)tid * p.n_parts + part] = bidx;
    }
}

// The same product for decode groups (more than 32 sequences): ONE pass over a part's 128 embedding rows for up to 128
// sequences, where logits_kernel multiplies them once per 32 sequences and pulls them (and its slice of the
// activations) through the compute units again every time: 666 MB per token at 112 sequences against 250 MB here.
// All sequences' activations do not fit the LDS, so K goes by in chunks of 256 columns: two LDS images of
// [128][256] fp16 (rows 33 pieces of 16 bytes apart, as in gemm_dec_kernel) filled by LDS-DMA one chunk ahead, the
// weights in two register banks of one chunk each, also one chunk ahead.  More than 128 sequences: blocks of 128, the
// blocks of a part on consecutive workgroup ids of one XCD (see logits_kernel).
constexpr int LW_ROWS = 128, LW_PPS = 33, LW_ROUNDS = (LW_ROWS * LW_PPS + 511) / 512;      // 9 rounds of 512 slots
constexpr int LW_BUF = LW_ROUNDS * 512 * 16;                                                 // 73 728 bytes per image
constexpr size_t kLogitsWideLds = 2 * (size_t)LW_BUF + 8 * LW_ROWS * 8;

__global__ __launch_bounds__(512) void logits_wide_kernel(GemvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int part = blockIdx.x;
    if (p.B > LW_ROWS) {
        const int n_groups = (p.B + LW_ROWS - 1) / LW_ROWS;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        part = (idx / n_groups) * 8 + xcd;
        if (part >= p.n_parts) return;                  // padding of the last slot (the whole workgroup leaves)
        group_view(p, idx % n_groups, LW_ROWS);
    }
    float* cand_v = reinterpret_cast<float*>(smem + 2 * LW_BUF);      // [8][128]
    int* cand_i = reinterpret_cast<int*>(cand_v + 8 * LW_ROWS);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int n0 = (part * 8 + wave) * 16;
    const int wrow = min(n0 + fr, p.N - 1);
    const f16* wp = p.W + (long)wrow * p.K + fg * 8;
    const int nks = p.K >> 5, nchunks = (nks + LCH - 1) / LCH;
    const int n_live = __builtin_amdgcn_readfirstlane((p.B + 15) / 16);        // sequence tiles that hold a sequence
    const int sample = p.st->pos - p.st->prompt_len;       // pos was advanced by the final LayerNorm launch

    uint4 wa[LCH], wb[LCH];
    auto wload = [&](uint4 (&dst)[LCH], int chunk) {
#pragma unroll
        for (int c = 0; c < LCH; ++c) dst[c] = load16<kNtLogits>(wp + min(chunk * LCH + c, nks - 1) * 32);
    };
    // chunk -> LDS image (chunk & 1): slot = row * 33 + piece; pieces past K repeat the last one (their weights are
    // zeroed), rows past the last sequence of a live tile repeat the last sequence (never stored), dead tiles stay unfilled
    auto stage = [&](int chunk) {
        const int ppr = min(LCH * 4, (p.K - chunk * LCH * 32) / 8);
        const f16* xbase = p.xin16 + chunk * LCH * 32;
        char* dst = smem + (chunk & 1) * LW_BUF;
#pragma unroll
        for (int j = 0; j < LW_ROUNDS; ++j) {
            const int first = (j * 8 + wave) * 64;
            if (first / LW_PPS >= n_live * 16) continue;                  // (wave-uniform)
            const int slot = first + lane;
            const int r = slot / LW_PPS, c8 = slot - r * LW_PPS;
            const f16* src = xbase + (long)min(r, p.B - 1) * p.K + min(c8, ppr - 1) * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + (j * 8 + wave) * 1024), 16, 0, 0);
        }
    };
    float4v acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = float4v{0, 0, 0, 0};
    auto consume = [&](uint4 (&src)[LCH], int chunk) {
        const char* xs = smem + (chunk & 1) * LW_BUF;
#pragma unroll
        for (int c = 0; c < LCH; ++c) {
            uint4 w = src[c];
            if (chunk * LCH + c >= nks) w = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (t >= n_live) break;
                const half8 xf = *reinterpret_cast<const half8*>(xs + ((t * 16 + fr) * LW_PPS + c * 4 + fg) * 16);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<half8*>(&w), xf, acc[t], 0, 0, 0);
            }
        }
    };
    stage(0);
    wload(wa, 0);
    if (nchunks > 1) wload(wb, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the compiler does not count the LDS-DMA
    __syncthreads();
    for (int c = 0; c < nchunks; c += 2) {
        // even chunk from bank a, odd chunk from bank b; the image and the bank of chunk + 1 were requested a chunk ago
        // (the wait before the barrier is for the image; the LCH weight loads issued after it may stay in flight)
        if (c + 1 < nchunks) stage(c + 1);
        consume(wa, c);
        if (c + 2 < nchunks) { wload(wa, c + 2); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (c + 1 < nchunks) {
            if (c + 2 < nchunks) stage(c + 2);
            consume(wb, c + 1);
            if (c + 3 < nchunks) { wload(wb, c + 3); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }

    const uint8_t* mask = sample <= 0 ? p.mask_first : p.mask_rest;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        if (t >= n_live) break;
        const int m = t * 16 + fr;
        float best = -INFINITY;
        int bidx = 0x7fffffff;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + 4 * fg + r;
            if (n < p.N && m < p.B) {
                const float v = acc[t][r];
                if (p.out32) p.out32[(long)m * p.N + n] = v;
                if (!mask[n] && (v > best || (v == best && n < bidx))) { best = v; bidx = n; }
            }
        }
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            const float ov = __shfl_xor(best, o);
            const int oi = __shfl_xor(bidx, o);
            if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
        }
        if (fg == 0) { cand_v[wave * LW_ROWS + m] = best; cand_i[wave * LW_ROWS + m] = bidx; }
    }
    __syncthreads();
    if (tid < LW_ROWS && tid < p.B) {
        float best = cand_v[tid];
        int bidx = cand_i[tid];
        for (int w = 1; w < 8; ++w) {
            const float ov = cand_v[w * LW_ROWS + tid];
            const int oi = cand_i[w * LW_ROWS + tid];
            if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
        }
        p.part_val[(long)tid * p.n_parts + part] = best;
        p.part_idx[(long)tid * p.n_parts + part] = bidx;
    }
}

// ------------------------------------------------------------------ single-query attention over a cache
// One workgroup per (sequence, head, key range): scores over n keys ([n][64] fp16, contiguous),
// softmax, P.V.  Each 128-byte key/value row is fetched by 8 adjacent lanes (16 B each), 8 rows per
// wave instruction, so every wave instruction reads 1 KiB of contiguous memory.  A thread owns RPT
// rows (row = group + NG*i); all its K loads are issued together, and all its V loads are issued
// as soon as the K registers are free, so V is in flight during the softmax.
// Self-attention (modeling_whisper.py:473-478): n = position + 1, every cached key is visible to the
// newest query.  Cross-attention (:486-492): n = 1500 encoder frames, K/V cached once (:313-326).
//
// S > 1 splits the keys of one (sequence, head) over S workgroups.  A CU ingests a stream at a
// fixed rate, so a launch is only as fast as its most loaded CU: batch*heads = 320 equal workgroups
// on 256 CUs leave 3/4 of the chip idle for half of the launch, while 1280 quarter-size ones are 5
// per CU.  Each workgroup publishes (max, sum, unnormalised output) and takes a ticket; whoever
// draws the last ticket merges the S partials in index order (so the result does not depend on
// arrival order).  Hand-off between workgroups: the partial is written by ONE wave with
// agent-scope (write-through) stores, that wave drains them (s_waitcnt vmcnt(0)) before its lane 0
// adds to the ticket counter with an agent-scope atomic, and the last arriver's same wave reads
// every partial back with agent-scope loads only after its own add has returned.
constexpr int kPartStride = 68;     // floats per partial: max, sum, 64 outputs, 2 pad

//
// Beam search (several hypotheses per recording): `kv_div` > 1 makes sequence r read the cache of
// recording r / kv_div (cross-attention: one encoder pass per recording, shared by its beams).  IND:
// the self-attention cache is never reordered when beams change parents; instead `src_rows`
// [position][sequence] names the physical cache row that holds each position of a sequence's history.
template <int NT, int RPT, int S, bool IND = false, bool GRP = false, bool DONE = false>
__global__ __launch_bounds__(NT) void dec_attn_kernel(const f16* __restrict__ q, const f16* __restrict__ kc,
                                                      const f16* __restrict__ vc, f16* __restrict__ out, int H,
                                                      int tstride, int nfixed, const DecState* __restrict__ st,
                                                      float* __restrict__ partial, int* __restrict__ tickets,
                                                      int kv_div, const int* __restrict__ src_rows, int n_rows,
                                                      const int* __restrict__ done = nullptr, int done_div = 1) {
    constexpr int NG = NT / 8;            // row groups
    constexpr int NWV = NT / 64;
    __shared__ float part[NWV][64];
    __shared__ float red[2 * NWV];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Grid (sequences * H * S).  DONE (a form of its own, so that launches without ended flags keep the exact code
    // they had): grid (H * S, sequences), the same linear workgroup order, and the sequence comes without a division.
    const int bh = S == 1 ? blockIdx.x : DONE ? blockIdx.y * H + blockIdx.x / S : blockIdx.x / S;
    const int split = S == 1 ? 0 : blockIdx.x % S;
    if constexpr (DONE) {
        // cross-attention of a sequence that has ended (greedy) or of a recording that has all its hypotheses
        // (beam search): its token is fixed to end-of-text whatever this launch computes, so its K/V stay unread
        if (done[(blockIdx.y + ((GRP && !IND) ? n_rows : 0)) / done_div]) return;
    }
    const int ntot = st ? st->pos + 1 : nfixed;
    const int per = (ntot + S - 1) / S;
    const int t0 = split * per;
    const int n = min(ntot, t0 + per) - t0;              // >= 1: S > 1 is only used with ntot >= S
    const int rg = tid >> 3, dc = tid & 7;
    int seq = 0;
    long bh_kv = bh;
    if constexpr (IND || GRP) {         // compile-time: greedy decoding never reads kv_div (one more scalar-load round trip)
        seq = DONE ? blockIdx.y : bh / H;
        // GRP without IND: this launch serves query rows n_rows .. of a longer list (q, out, partial and tickets
        // are local to the launch; kc / vc are not)
        const int seq_all = (GRP && !IND) ? seq + n_rows : seq;
        bh_kv = (long)(seq_all / kv_div) * H + (bh - seq * H);
    }
    const f16* kb = kc + (bh_kv * tstride + t0) * 64 + dc * 8;
    const f16* vb = vc + (bh_kv * tstride + t0) * 64 + dc * 8;
    // element offset of row i of this thread from kb / vb (IND: through the cache-row table)
    long roff[IND ? RPT : 1];
    if constexpr (IND) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int t = min(rg + NG * i, n - 1);
            roff[i] = ((long)(src_rows[(long)(t0 + t) * n_rows + seq] - seq) * H * tstride + t) * 64;
        }
    }
    auto row_off = [&](int i) -> long {
        if constexpr (IND) return roff[i];
        else return (long)min(rg + NG * i, n - 1) * 64;
    };

    // Self-attention (S == 1: few rows per thread): V is fetched beside K, one round trip instead of two.
    // Cross-attention keeps one register set: its V loads are issued as soon as the scores have freed it.
    constexpr bool V_EARLY = S == 1;
    half8 kv[RPT];
    half8 vv[V_EARLY ? RPT : 1];
#pragma unroll
    for (int i = 0; i < RPT; ++i) kv[i] = load_half8<(S > 1) && kNtXattn>(kb + row_off(i));
    if constexpr (V_EARLY) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) vv[i] = *reinterpret_cast<const half8*>(vb + row_off(i));
    }
    float qv[8];
    {
        const half8 qh = *reinterpret_cast<const half8*>(q + (long)bh * 64 + dc * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) qv[j] = (float)qh[j];
    }
    // ---- scores (keys past n are duplicates of key n-1 and are masked out)
    float sv[RPT];
    float mx = -3.0e38f;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) d += qv[j] * (float)kv[i][j];
        d += __shfl_xor(d, 1);
        d += __shfl_xor(d, 2);
        d += __shfl_xor(d, 4);
        sv[i] = (rg + NG * i < n) ? d : -3.0e38f;
        mx = fmaxf(mx, sv[i]);
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        if constexpr (V_EARLY) kv[i] = vv[i];
        else kv[i] = load_half8<(S > 1) && kNtXattn>(vb + row_off(i));
    }
    // ---- softmax over all rows of the workgroup
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) mx = fmaxf(mx, red[w]);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        sv[i] = __expf(sv[i] - mx);                 // masked rows: exp(-huge) = 0
        sum += sv[i];
    }
    sum = wave_sum(sum) * 0.125f;                   // each row's score is held by its 8 lanes
    if (lane == 0) red[NWV + wave] = sum;
    // ---- P.V from registers
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < RPT; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += sv[i] * (float)kv[i][j];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float a = acc[j];
        a += __shfl_xor(a, 8);
        a += __shfl_xor(a, 16);
        a += __shfl_xor(a, 32);
        acc[j] = a;
    }
    if (lane < 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) part[wave][lane * 8 + j] = acc[j];
    }
    __syncthreads();
    if (wave != 0) return;
    float o = 0.f, l = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) { o += part[w][lane]; l += red[NWV + w]; }
    if constexpr (S == 1) {
        out[(long)bh * 64 + lane] = (f16)(o / l);
    } else {
        float* mine = partial + ((long)bh * S + split) * kPartStride;
        __hip_atomic_store(mine + 2 + lane, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0) {
            __hip_atomic_store(mine, mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(mine + 1, l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's partial has left the CU
        int ticket = 0;
        if (lane == 0) ticket = __hip_atomic_fetch_add(tickets + bh, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = __shfl(ticket, 0);
        if (ticket != S - 1) return;
        // last arriver: every partial of this (sequence, head) is complete; merge in index order
        float pm[S], pl[S], po[S];
        float gm = -3.0e38f;
#pragma unroll
        for (int i = 0; i < S; ++i) {
            const float* src = partial + ((long)bh * S + i) * kPartStride;
            pm[i] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pl[i] = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            po[i] = __hip_atomic_load(src + 2 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            gm = fmaxf(gm, pm[i]);
        }
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int i = 0; i < S; ++i) {
            const float w = __expf(pm[i] - gm);
            L += w * pl[i];
            O += w * po[i];
        }
        out[(long)bh * 64 + lane] = (f16)(O / L);
        if (lane == 0) __hip_atomic_store(tickets + bh, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
}

// ------------------------------------------------------------------ cross-attention of a large decode group
// One workgroup per (sequence, head) walks ALL the keys in chunks of 256 rows (8 per thread) with a
// running (max, sum, output) per thread: no key ranges, no published partials, no ticket.  With thousands of
// (sequence, head) pairs in a launch the ranges of dec_attn_kernel are not needed to fill the chip, and their
// hand-off is what that kernel pays for them: a read stream shaped like this launch (860 MB in 8960 items of 96 KB,
// tools/read_bw2.hip) runs at 6.85-6.9 TB/s with a barrier and an agent-scope store per item and at 6.35-6.45 TB/s
// once every item also takes a returning agent-scope atomic (the ticket) and every fourth merges: the wave that
// waits for its ticket holds its workgroup's slot.  Here the next chunk's K rows are requested as soon as the
// scores have freed their registers and its V rows as soon as the outputs have, so a workgroup always has loads in
// flight; the only barrier is the one before the final merge of the four waves.  Every thread keeps its own
// reference maximum (the 8 lanes of a row share it), merged at the end in a fixed order: lanes by butterfly, waves
// in index order.
// KV8: the cache holds e4m3 codes [B][H][T][64] with one fp32 scale per row (cross_attn8_kernel's format):
// score = (q . codes) * k_scale, output += p * v_scale * codes; eight rows a thread are 8 + 4 bytes each.
template <bool KV8> struct XRow { half8 d; };
template <> struct XRow<true> { uint2 d; float s; };
__device__ __forceinline__ void xrow_f32(const XRow<false>& r, float (&f)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (float)r.d[j];
}
__device__ __forceinline__ void e4m3x8_to_f32(const uint2& w, float (&f)[8]);      // defined with the e4m3 kernels below
__device__ __forceinline__ void xrow_f32(const XRow<true>& r, float (&f)[8]) { e4m3x8_to_f32(r.d, f); }

template <bool DONE, bool KV8>
__device__ __forceinline__ void cross_attn_stream_body(
    const f16* __restrict__ q, const void* __restrict__ kc, const void* __restrict__ vc, const float* __restrict__ ks,
    const float* __restrict__ vs, f16* __restrict__ out, int H, int T, const int* __restrict__ done, int done_div) {
    constexpr int NG = 32, RPT = 8, CH = NG * RPT, NWV = 4;
    constexpr int EB = KV8 ? 1 : 2;                       // bytes per cached element
    __shared__ float wm[NWV], wl[NWV], wo[NWV][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bh = DONE ? blockIdx.y * H + blockIdx.x : blockIdx.x;
    if constexpr (DONE) {
        if (done[blockIdx.y / done_div]) return;          // an ended sequence: its token is fixed whatever this computes
    }
    const int rg = tid >> 3, dc = tid & 7;
    // workgroup-uniform bases and 32-bit offsets (a sequence's head is T * 64 < 2^31 elements)
    const char* kb = reinterpret_cast<const char*>(kc) + (long)bh * T * 64 * EB;
    const char* vb = reinterpret_cast<const char*>(vc) + (long)bh * T * 64 * EB;
    const float* ksb = KV8 ? ks + (long)bh * T : nullptr;
    const float* vsb = KV8 ? vs + (long)bh * T : nullptr;
    auto row_of = [&](int c, int i) -> int { return min(c * CH + rg + NG * i, T - 1); };
    auto fetch = [&](const char* base, const float* sb, int c, int i) {
        XRow<KV8> r;
        const int row = row_of(c, i);
        const char* src = base + (unsigned)((row * 64 + dc * 8) * EB);
        if constexpr (KV8) { r.d = load8<kNtXattn>(src); r.s = sb[row]; }
        else r.d = load_half8<kNtXattn>(src);
        return r;
    };
    XRow<KV8> kv[RPT], vv[RPT];
    // the query first (loads retire in order: requested last it would be waited for behind the whole first chunk),
    // then the first chunk: all K rows, then all V rows, in the order the loop issues them
    const half8 qh = *reinterpret_cast<const half8*>(q + (long)bh * 64 + dc * 8);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < RPT; ++i) kv[i] = fetch(kb, ksb, 0, i);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < RPT; ++i) vv[i] = fetch(vb, vsb, 0, i);
    __builtin_amdgcn_sched_barrier(0);
    float qv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) qv[j] = (float)qh[j];
    float m = -3.0e38f, l = 0.f;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int nch = (T + CH - 1) / CH;
    // MORE: another chunk follows; its K rows are requested as soon as the scores have freed kv, its V rows as soon
    // as the outputs have freed vv.  The last chunk is a copy of the body without those requests, so that the load
    // counters the compiler waits on are exact in both (one body for both would wait for the smaller count).
    auto chunk = [&](int c, auto more_c) {
        constexpr bool MORE = decltype(more_c)::value;
        // ---- scores of this chunk (rows past T are duplicates of row T-1 and are masked out)
        float sv[RPT];
        float cm = -3.0e38f;
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            float f[8];
            xrow_f32(kv[i], f);
            float d = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) d += qv[j] * f[j];
            d += __shfl_xor(d, 1);
            d += __shfl_xor(d, 2);
            d += __shfl_xor(d, 4);
            if constexpr (KV8) d *= kv[i].s;
            sv[i] = (c * CH + rg + NG * i < T) ? d : -3.0e38f;
            cm = fmaxf(cm, sv[i]);
        }
        if constexpr (MORE) {               // (pinned between the two halves: the scheduler would move them to the chunk's end)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < RPT; ++i) kv[i] = fetch(kb, ksb, c + 1, i);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- running softmax of this thread's rows (a thread whose rows are all masked keeps its state: exp(-huge) = 0)
        const float mn = fmaxf(m, cm);
        const float alpha = __expf(m - mn);
        l *= alpha;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] *= alpha;
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const float pr = __expf(sv[i] - mn);
            l += pr;
            float f[8];
            xrow_f32(vv[i], f);
            float w = pr;
            if constexpr (KV8) w *= vv[i].s;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += w * f[j];
        }
        m = mn;
        if constexpr (MORE) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < RPT; ++i) vv[i] = fetch(vb, vsb, c + 1, i);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int c = 0; c + 1 < nch; ++c) chunk(c, std::true_type{});
    chunk(nch - 1, std::false_type{});
    // ---- merge the 8 row groups of a wave (lanes that share dc): butterfly over lane bits 3..5
#pragma unroll
    for (int o = 8; o <= 32; o <<= 1) {
        const float m2 = __shfl_xor(m, o), l2 = __shfl_xor(l, o);
        const float mn = fmaxf(m, m2);
        const float a1 = __expf(m - mn), a2 = __expf(m2 - mn);
        l = l * a1 + l2 * a2;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = acc[j] * a1 + __shfl_xor(acc[j], o) * a2;
        m = mn;
    }
    if (lane < 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) wo[wave][lane * 8 + j] = acc[j];
        if (lane == 0) { wm[wave] = m; wl[wave] = l; }
    }
    __syncthreads();
    if (wave != 0) return;
    // ---- merge the four waves in index order; lane = output element
    float gm = wm[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) gm = fmaxf(gm, wm[w]);
    float L = 0.f, O = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) {
        const float a = __expf(wm[w] - gm);
        L += a * wl[w];
        O += a * wo[w][lane];
    }
    out[(long)bh * 64 + lane] = (f16)(O / L);
}

// fp16 cache: 116 registers, four workgroups per CU.  e4m3 cache: its row scales and code expansion need more than
// 128 registers (held to 128 it spills), three per CU.
template <bool DONE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void cross_attn_stream_kernel(
    const f16* __restrict__ q, const f16* __restrict__ kc, const f16* __restrict__ vc, f16* __restrict__ out, int H, int T,
    const int* __restrict__ done, int done_div) {
    cross_attn_stream_body<DONE, false>(q, kc, vc, nullptr, nullptr, out, H, T, done, done_div);
}
template <bool DONE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void cross_attn8_stream_kernel(
    const f16* __restrict__ q, const uint8_t* __restrict__ kc, const uint8_t* __restrict__ vc, const float* __restrict__ ks,
    const float* __restrict__ vs, f16* __restrict__ out, int H, int T, const int* __restrict__ done, int done_div) {
    cross_attn_stream_body<DONE, true>(q, kc, vc, ks, vs, out, H, T, done, done_div);
}

// ------------------------------------------------------------------ cross-attention over an e4m3 cache
// The cross-attention of dec_attn_kernel (S key ranges per (query row, head), ticket-ordered merge) over K/V kept
// as e4m3 codes [B][H][T][64] with one fp32 scale per row: 64 + 4 bytes per row instead of 128.
// score = (q . codes) * k_scale, output += p * v_scale * codes, everything in fp32 after an exact in-register
// expansion of the codes.
typedef float float2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void e4m3x8_to_f32(const uint2& w, float (&f)[8]) {
    const unsigned u[2] = {w.x, w.y};
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const float2v lo = __builtin_amdgcn_cvt_pk_f32_fp8(u[d], false);
        const float2v hi = __builtin_amdgcn_cvt_pk_f32_fp8(u[d], true);
        f[4 * d] = lo[0]; f[4 * d + 1] = lo[1]; f[4 * d + 2] = hi[0]; f[4 * d + 3] = hi[1];
    }
}

// Thread layout as in dec_attn_kernel: 8 adjacent lanes per row (8 codes = 8 bytes each), RPT rows per thread
// (row = group + NG*i), the same number of load instructions as the fp16 form at half the bytes each; the
// register budget (<= 96) keeps five workgroups per CU, so the 5 * 256 workgroups of a batch-16 launch are all
// resident at once.
template <int NT, int RPT, int S, bool GRP, bool DONE>
__global__ __launch_bounds__(NT, 5) void cross_attn8_kernel(const f16* __restrict__ q, const uint8_t* __restrict__ kc,
                                                         const uint8_t* __restrict__ vc, const float* __restrict__ ks,
                                                         const float* __restrict__ vs, f16* __restrict__ out, int H, int T,
                                                         float* __restrict__ partial, int* __restrict__ tickets, int kv_div,
                                                         int row0, const int* __restrict__ done, int done_div) {
    constexpr int NG = NT / 8;            // row groups
    constexpr int NWV = NT / 64;
    __shared__ float part[NWV][64];
    __shared__ float red[2 * NWV];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // DONE: grid (H * S, sequences), the sequence comes without a division; otherwise grid (sequences * H * S)
    const int split = blockIdx.x % S;
    const int bh = DONE ? blockIdx.y * H + blockIdx.x / S : blockIdx.x / S;
    const int seq = DONE ? blockIdx.y : (GRP ? bh / H : 0);
    const int head = bh - seq * H;          // used by GRP only
    if constexpr (DONE) { if (done[(seq + (GRP ? row0 : 0)) / done_div]) return; }     // ended: see dec_attn_kernel
    const int per = (T + S - 1) / S;
    const int t0 = split * per;
    const int n = min(T, t0 + per) - t0;
    const int rg = tid >> 3, dc = tid & 7;
    long bh_kv = bh;
    if constexpr (GRP) bh_kv = (long)((seq + row0) / kv_div) * H + head;
    const uint8_t* kb = kc + (bh_kv * T + t0) * 64 + dc * 8;
    const uint8_t* vb = vc + (bh_kv * T + t0) * 64 + dc * 8;
    const float* ksb = ks + bh_kv * T + t0;
    const float* vsb = vs + bh_kv * T + t0;

    uint2 kv[RPT];
    float rs[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = min(rg + NG * i, n - 1);
        kv[i] = load8<kNtXattn>(kb + (long)r * 64);
        rs[i] = ksb[r];
    }
    float qv[8];
    {
        const half8 qh = *reinterpret_cast<const half8*>(q + (long)bh * 64 + dc * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) qv[j] = (float)qh[j];
    }
    float sv[RPT];
    float mx = -3.0e38f;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        float f[8];
        e4m3x8_to_f32(kv[i], f);
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) d += qv[j] * f[j];
        d += __shfl_xor(d, 1);
        d += __shfl_xor(d, 2);
        d += __shfl_xor(d, 4);
        sv[i] = (rg + NG * i < n) ? d * rs[i] : -3.0e38f;
        mx = fmaxf(mx, sv[i]);
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = min(rg + NG * i, n - 1);
        kv[i] = load8<kNtXattn>(vb + (long)r * 64);
        rs[i] = vsb[r];
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) mx = fmaxf(mx, red[w]);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        sv[i] = __expf(sv[i] - mx);                 // masked rows: exp(-huge) = 0
        sum += sv[i];
    }
    sum = wave_sum(sum) * 0.125f;                   // each row's score is held by its 8 lanes
    if (lane == 0) red[NWV + wave] = sum;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        float f[8];
        e4m3x8_to_f32(kv[i], f);
        const float w = sv[i] * rs[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += w * f[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float a = acc[j];
        a += __shfl_xor(a, 8);
        a += __shfl_xor(a, 16);
        a += __shfl_xor(a, 32);
        acc[j] = a;
    }
    if (lane < 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) part[wave][lane * 8 + j] = acc[j];
    }
    __syncthreads();
    if (wave != 0) return;
    float o = 0.f, l = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) { o += part[w][lane]; l += red[NWV + w]; }
    // hand-off and merge: as in dec_attn_kernel
    float* mine = partial + ((long)bh * S + split) * kPartStride;
    __hip_atomic_store(mine + 2 + lane, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane == 0) {
        __hip_atomic_store(mine, mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + 1, l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(tickets + bh, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __shfl(ticket, 0);
    if (ticket != S - 1) return;
    float pm[S], pl[S], po[S];
    float gm = -3.0e38f;
#pragma unroll
    for (int i = 0; i < S; ++i) {
        const float* src = partial + ((long)bh * S + i) * kPartStride;
        pm[i] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pl[i] = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        po[i] = __hip_atomic_load(src + 2 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        gm = fmaxf(gm, pm[i]);
    }
    float L = 0.f, O = 0.f;
#pragma unroll
    for (int i = 0; i < S; ++i) {
        const float w = __expf(pm[i] - gm);
        L += w * pl[i];
        O += w * po[i];
    }
    out[(long)bh * 64 + lane] = (f16)(O / L);
    if (lane == 0) __hip_atomic_store(tickets + bh, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------ token bookkeeping
// dec_init: position 0, first forced token embedded.  dec_finalize (one workgroup per sequence):
// choose the next token (forced while inside the prompt, else the arg-max candidate), apply the
// end-of-text rule, record it, and embed it (token + learned position, modeling_whisper.py:737-762).
// The position itself is advanced once per step by the final LayerNorm launch, before the logits.
__global__ __launch_bounds__(256) void dec_init_kernel(FinalizeParams p, int B, int prompt_len, int max_new, int eot) {
    const int m = blockIdx.x, tid = threadIdx.x;
    const int tok = p.forced[(long)m * p.n_forced];
    for (int c = tid; c < p.D; c += 256) p.x[(long)m * p.D + c] = (float)p.tok_emb[(long)tok * p.D + c] + (float)p.pos_emb[c];
    for (int i = tid; i < max_new; i += 256) p.out_ids[(long)m * max_new + i] = eot;
    if (tid == 0) { p.out_len[m] = max_new; p.finished[m] = 0; }
    if (m == 0 && tid == 0) {
        p.st->pos = 0; p.st->prompt_len = prompt_len; p.st->max_new = max_new; p.st->eot = eot;
        p.st->n_finished = 0; p.st->batch = B;
    }
}

__global__ __launch_bounds__(256) void dec_finalize_kernel(FinalizeParams p, int B) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int next = p.st->pos, prompt_len = p.st->prompt_len, max_new = p.st->max_new, eot = p.st->eot;
    int tok;
    if (next < prompt_len) {
        tok = p.forced[(long)m * p.n_forced + next];
    } else {
        float best = -INFINITY;
        int bidx = 0x7fffffff;
        for (int i = tid; i < p.n_parts; i += 256) {
            const float v = p.part_val[(long)m * p.n_parts + i];
            const int ix = p.part_idx[(long)m * p.n_parts + i];
            if (v > best || (v == best && ix < bidx)) { best = v; bidx = ix; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o);
            const int oi = __shfl_xor(bidx, o);
            if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
        }
        if (lane == 0) { bv[wave] = best; bi[wave] = bidx; }
        __syncthreads();
        best = bv[0]; bidx = bi[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (bv[w] > best || (bv[w] == best && bi[w] < bidx)) { best = bv[w]; bidx = bi[w]; }
        tok = (bidx >= 0 && bidx < p.vocab) ? bidx : eot;      // all candidates masked or NaN
        const int sample = next - prompt_len;
        const int was_finished = p.finished[m];
        if (was_finished) tok = eot;
        __syncthreads();                                        // every thread has read finished[m]
        if (tid == 0 && sample < max_new) {
            p.out_ids[(long)m * max_new + sample] = tok;
            if (!was_finished && tok == eot) {
                p.out_len[m] = sample;
                p.finished[m] = 1;
                atomicAdd(&p.st->n_finished, 1);
            }
        }
    }
    if (next < p.n_text_ctx) {
        for (int c = tid; c < p.D; c += 256)
            p.x[(long)m * p.D + c] = (float)p.tok_emb[(long)tok * p.D + c] + (float)p.pos_emb[(long)next * p.D + c];
    }
}

// ------------------------------------------------------------------ beam search bookkeeping
// One workgroup per recording, once per step, in place of dec_finalize.  Restates, on the device,
// the update of OpenAI Whisper's BeamSearchDecoder (decoding.py of the openai-whisper package; the
// reference leaves decoding to whisperx's defaults, transcribe.py:111, and holds no such code):
// every live beam proposes its beam+1 most likely tokens under the suppress mask; proposals are
// ranked by cumulative log-probability (ties: beam order, then token rank); walking down, one that
// ends in end-of-text joins the recording's finished list while that has room, any other becomes a
// live beam until `beam` are chosen.  Beams that change parent take over the parent's token
// history and its row of `src_rows`; the K/V cache itself never moves.
// Before the first sampled token all beams of a recording are copies: only beam 0 carries weight
// (the others start at -inf), which is what the reference algorithm's dictionary of sequences does.
__global__ __launch_bounds__(256) void beam_init_kernel(BeamParams p, int prompt_len, int max_new, int eot) {
    const int a = blockIdx.x, tid = threadIdx.x;
    const int R = p.n_audio * p.beam;
    for (int j = 0; j < p.beam; ++j) {
        const int r = a * p.beam + j;
        const int tok = p.forced[(long)r * p.n_forced];
        for (int c = tid; c < p.D; c += 256) p.x[(long)r * p.D + c] = (float)p.tok_emb[(long)tok * p.D + c] + (float)p.pos_emb[c];
        for (int t = tid; t < p.n_text_ctx; t += 256) p.src_rows[(long)t * R + r] = r;
        for (int i = tid; i < max_new; i += 256) p.hist[(long)r * max_new + i] = eot;
        if (tid == 0) p.sum_lp[r] = j == 0 ? 0.f : -INFINITY;
    }
    if (tid == 0) { p.fin_count[a] = 0; p.completed[a] = 0; }
    if (a == 0 && tid == 0) {
        p.st->pos = 0; p.st->prompt_len = prompt_len; p.st->max_new = max_new; p.st->eot = eot;
        p.st->n_finished = 0; p.st->batch = R;
    }
}

// The scan half of a beam step: workgroup (part, sequence) goes over its sixteenth of the sequence's logits under
// the suppress mask and leaves the running maximum, the sum of exp(v - maximum) and its beam+1 best
// (logit, id) pairs (descending; ties: lowest id).  beam_step_kernel merges the parts.
__global__ __launch_bounds__(256) void beam_scan_kernel(BeamParams p) {
    constexpr int KMAX = kMaxBeam + 1;
    __shared__ float topv[KMAX][256];           // per-thread best-K lists (column = thread)
    __shared__ int topi[KMAX][256];
    __shared__ float wv[4]; __shared__ int wi[4]; __shared__ int wt[4];
    __shared__ float redm[4], reds[4];
    const int part = blockIdx.x, r = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = p.beam + 1;
    const int next = p.st->pos, prompt_len = p.st->prompt_len;
    if (next < prompt_len || p.completed[r / p.beam]) return;
    const uint8_t* mask = next == prompt_len ? p.mask_first : p.mask_rest;
    const int per = (p.vocab + kBeamParts - 1) / kBeamParts;
    const int lo = part * per, hi = min(p.vocab, lo + per);
    const float* lg = p.logits + (long)r * p.vocab;
    for (int c = 0; c < K; ++c) { topv[c][tid] = -INFINITY; topi[c][tid] = 0x7fffffff; }
    float m = -INFINITY, ssum = 0.f;
    for (int i = lo + tid; i < hi; i += 256) {
        if (mask[i]) continue;
        const float v = lg[i];
        if (v > m) { ssum = ssum * __expf(m - v) + 1.f; m = v; } else { ssum += __expf(v - m); }
        if (v > topv[K - 1][tid]) {             // indices arrive in increasing order: ties keep the earlier one
            int c = K - 1;
            while (c > 0 && v > topv[c - 1][tid]) { topv[c][tid] = topv[c - 1][tid]; topi[c][tid] = topi[c - 1][tid]; --c; }
            topv[c][tid] = v; topi[c][tid] = i;
        }
    }
    float bm = wave_max(m);
    if (lane == 0) redm[wave] = bm;
    __syncthreads();
    bm = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
    const float bs = wave_sum(m == -INFINITY ? 0.f : ssum * __expf(m - bm));
    if (lane == 0) reds[wave] = bs;
    __syncthreads();
    const long slot = (long)r * kBeamParts + part;
    if (tid == 0) { p.part_m[slot] = bm; p.part_s[slot] = reds[0] + reds[1] + reds[2] + reds[3]; }
    // K rounds: every thread offers the head of its list; the best (value, then lowest id) is popped
    int head = 0;
    for (int c = 0; c < K; ++c) {
        float v = topv[head < K ? head : K - 1][tid];
        int ix = topi[head < K ? head : K - 1][tid];
        if (head >= K) { v = -INFINITY; ix = 0x7fffffff; }
        int who = tid;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(v, o); const int oi = __shfl_xor(ix, o); const int ow = __shfl_xor(who, o);
            if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; who = ow; }
        }
        if (lane == 0) { wv[wave] = v; wi[wave] = ix; wt[wave] = who; }
        __syncthreads();
        float gv = wv[0]; int gi = wi[0], gt = wt[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (wv[w] > gv || (wv[w] == gv && wi[w] < gi)) { gv = wv[w]; gi = wi[w]; gt = wt[w]; }
        if (tid == gt) ++head;
        if (tid == 0) { p.part_v[slot * KMAX + c] = gv; p.part_i[slot * KMAX + c] = gi; }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void beam_step_kernel(BeamParams p) {
    constexpr int KMAX = kMaxBeam + 1, NC = kMaxBeam * KMAX;
    static_assert(kBeamParts * KMAX <= 256, "one thread per (part, rank) candidate");
    __shared__ float wv[4]; __shared__ int wi[4]; __shared__ int wt[4];
    __shared__ float cand_lp[NC]; __shared__ int cand_tok[NC]; __shared__ int cand_rank[NC];
    __shared__ int new_parent[kMaxBeam], new_tok[kMaxBeam]; __shared__ float new_lp[kMaxBeam];
    __shared__ int fin_parent[NC]; __shared__ float fin_score[NC]; __shared__ int n_fin_new;
    extern __shared__ int stage[];              // [beam][max(max_new, n_text_ctx)] history / cache-row staging

    const int a = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int beam = p.beam, K = beam + 1, R = p.n_audio * beam;
    const int next = p.st->pos, prompt_len = p.st->prompt_len, max_new = p.st->max_new, eot = p.st->eot;
    const int r0 = a * beam;
    if (next < prompt_len || p.completed[a]) {      // forced prompt, or this recording has all its hypotheses
        if (next < p.n_text_ctx)
            for (int j = 0; j < beam; ++j) {
                const int tok = next < prompt_len ? p.forced[(long)(r0 + j) * p.n_forced + next] : eot;
                for (int c = tid; c < p.D; c += 256)
                    p.x[(long)(r0 + j) * p.D + c] = (float)p.tok_emb[(long)tok * p.D + c] + (float)p.pos_emb[(long)next * p.D + c];
            }
        return;
    }
    const int sample = next - prompt_len;

    // ---- per beam: log-sum-exp over the unmasked vocabulary and the K best tokens, from the scan's parts
    for (int j = 0; j < beam; ++j) {
        const long row = (long)(r0 + j) * kBeamParts;
        float bm = -INFINITY;
        for (int q = 0; q < kBeamParts; ++q) bm = fmaxf(bm, p.part_m[row + q]);
        float bs = 0.f;
        for (int q = 0; q < kBeamParts; ++q) { const float mq = p.part_m[row + q]; if (mq != -INFINITY) bs += p.part_s[row + q] * __expf(mq - bm); }
        const float lse = bm + __logf(bs);
        // thread (part, rank) holds one candidate; K rounds pop the best (value, then lowest id)
        float mine = -INFINITY; int mine_i = 0x7fffffff;
        if (tid < kBeamParts * K) {
            const int q = tid / K, c = tid % K;
            mine = p.part_v[(row + q) * KMAX + c]; mine_i = p.part_i[(row + q) * KMAX + c];
        }
        for (int c = 0; c < K; ++c) {
            float v = mine; int ix = mine_i, who = tid;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(v, o); const int oi = __shfl_xor(ix, o); const int ow = __shfl_xor(who, o);
                if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; who = ow; }
            }
            if (lane == 0) { wv[wave] = v; wi[wave] = ix; wt[wave] = who; }
            __syncthreads();
            float gv = wv[0]; int gi = wi[0], gt = wt[0];
#pragma unroll
            for (int w = 1; w < 4; ++w)
                if (wv[w] > gv || (wv[w] == gv && wi[w] < gi)) { gv = wv[w]; gi = wi[w]; gt = wt[w]; }
            if (tid == gt) { mine = -INFINITY; mine_i = 0x7fffffff; }
            if (tid == 0) {
                cand_lp[j * K + c] = p.sum_lp[r0 + j] + (gv - lse);
                cand_tok[j * K + c] = (gi >= 0 && gi < p.vocab) ? gi : eot;
            }
            __syncthreads();
        }
    }
    // ---- rank the beam*K proposals (descending score; ties keep proposal order)
    const int nc = beam * K;
    if (tid < nc) {
        const float v = cand_lp[tid];
        int rk = 0;
        for (int k = 0; k < nc; ++k) rk += (cand_lp[k] > v || (cand_lp[k] == v && k < tid)) ? 1 : 0;
        cand_rank[rk] = tid;
    }
    __syncthreads();
    if (tid == 0) {
        int saved = 0, nf = 0, room = p.max_cand - p.fin_count[a];
        for (int k = 0; k < nc && saved < beam; ++k) {
            const int c = cand_rank[k];
            if (cand_tok[c] == eot) {
                if (nf < room) { fin_parent[nf] = c / K; fin_score[nf] = cand_lp[c]; ++nf; }
            } else {
                new_parent[saved] = c / K; new_tok[saved] = cand_tok[c]; new_lp[saved] = cand_lp[c]; ++saved;
            }
        }
        n_fin_new = nf;
    }
    __syncthreads();
    // ---- finished hypotheses: the parent's history as it stands (end-of-text itself is not stored)
    const int nf = n_fin_new, base = p.fin_count[a];
    for (int f = 0; f < nf; ++f) {
        const int* src = p.hist + (long)(r0 + fin_parent[f]) * max_new;
        int* dst = p.fin_tokens + ((long)a * p.max_cand + base + f) * max_new;
        for (int i = tid; i < sample; i += 256) dst[i] = src[i];
        if (tid == 0) { p.fin_lp[a * p.max_cand + base + f] = fin_score[f]; p.fin_len[a * p.max_cand + base + f] = sample; }
    }
    // ---- live beams take over their parents' histories and cache rows (staged: parents may repeat)
    const int width = max(max_new, p.n_text_ctx);
    for (int j = 0; j < beam; ++j)
        for (int i = tid; i < sample; i += 256) stage[j * width + i] = p.hist[(long)(r0 + j) * max_new + i];
    __syncthreads();
    for (int j = 0; j < beam; ++j) {
        int* dst = p.hist + (long)(r0 + j) * max_new;
        for (int i = tid; i < sample; i += 256) dst[i] = stage[new_parent[j] * width + i];
        if (tid == 0 && sample < max_new) dst[sample] = new_tok[j];
    }
    __syncthreads();
    for (int j = 0; j < beam; ++j)
        for (int t = tid; t < next; t += 256) stage[j * width + t] = p.src_rows[(long)t * R + r0 + j];
    __syncthreads();
    for (int j = 0; j < beam; ++j)
        for (int t = tid; t < next; t += 256) p.src_rows[(long)t * R + r0 + j] = stage[new_parent[j] * width + t];
    if (tid < beam) p.sum_lp[r0 + tid] = new_lp[tid];
    if (tid == 0) {
        p.fin_count[a] = base + nf;
        if (base + nf >= p.max_cand) { p.completed[a] = 1; atomicAdd(&p.st->n_finished, 1); }
    }
    if (next < p.n_text_ctx)
        for (int j = 0; j < beam; ++j) {
            const int tok = new_tok[j];
            for (int c = tid; c < p.D; c += 256)
                p.x[(long)(r0 + j) * p.D + c] = (float)p.tok_emb[(long)tok * p.D + c] + (float)p.pos_emb[(long)next * p.D + c];
        }
}

// ------------------------------------------------------------------ sampling at a temperature
// The decodes that OpenAI Whisper's decode_with_fallback (transcribe.py of the openai-whisper package, mirrored by
// faster-whisper's generate_with_fallback; the reference leaves them to whisperx, transcribe.py:111) runs when a
// window's text fails its compression-ratio or log-probability test: `best_of` independent samples per recording
// from softmax(logits / temperature) under the suppress mask (GreedyDecoder.update of decoding.py), each summing
// the log-probabilities (log-softmax of the untempered logits) of its sampled tokens up to and including
// end-of-text.  The draw is a Gumbel arg-max, token = argmax_i(logits_i / temperature - log(-log u_i)) with u_i a
// counter-based uniform of (seed, sequence, position, i): distributed as the categorical draw of the published code,
// independent of how the vocabulary is split over workgroups, and reproducible on the host.  inv_temp == 0 is the
// arg-max itself (temperature 0 with its log-probabilities).
__global__ __launch_bounds__(256) void sample_init_kernel(SampleParams p, int prompt_len, int max_new, int eot, float inv_temp, uint32_t seed,
                                                          int n_ended_at_start) {
    const int r = blockIdx.x, tid = threadIdx.x;
    const int tok = p.forced[(long)r * p.n_forced];
    for (int c = tid; c < p.D; c += 256) p.x[(long)r * p.D + c] = (float)p.tok_emb[(long)tok * p.D + c] + (float)p.pos_emb[c];
    for (int i = tid; i < max_new; i += 256) p.out_ids[(long)r * max_new + i] = eot;
    if (tid == 0) {
        const bool on = !p.active || p.active[r / p.best_of];
        p.out_len[r] = on ? max_new : 0;
        p.finished[r] = on ? 0 : 1;
        p.sum_lp[r] = 0.f;
    }
    if (r == 0 && tid == 0) {
        p.st->pos = 0; p.st->prompt_len = prompt_len; p.st->max_new = max_new; p.st->eot = eot;
        p.st->n_finished = n_ended_at_start; p.st->batch = p.n_audio * p.best_of;
        p.ss->inv_temp = inv_temp; p.ss->seed = seed;
    }
}

// Workgroup (part, sequence): its sixteenth of the sequence's logits under the suppress mask -> running maximum and
// sum of exp(v - maximum), and the best perturbed score with the raw logit and id behind it (ties: lowest id).
__global__ __launch_bounds__(256) void sample_scan_kernel(SampleParams p) {
    __shared__ float redm[4], reds[4], wv[4], wl[4];
    __shared__ int wi[4];
    const int part = blockIdx.x, r = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int next = p.st->pos, prompt_len = p.st->prompt_len;
    if (next < prompt_len || p.finished[r]) return;
    const uint8_t* mask = next == prompt_len ? p.mask_first : p.mask_rest;
    const float inv_temp = p.ss->inv_temp;
    const uint32_t seed = p.ss->seed, sample = (uint32_t)(next - prompt_len);
    const int per = (p.vocab + kBeamParts - 1) / kBeamParts;
    const int lo = part * per, hi = min(p.vocab, lo + per);
    const float* lg = p.logits + (long)r * p.vocab;
    float m = -INFINITY, ssum = 0.f, best = -INFINITY, braw = -INFINITY;
    int bidx = 0x7fffffff;
    for (int i = lo + tid; i < hi; i += 256) {
        if (mask[i]) continue;
        const float v = lg[i];
        if (v > m) { ssum = ssum * expf(m - v) + 1.f; m = v; } else { ssum += expf(v - m); }
        float sc = v;
        if (inv_temp > 0.f) sc = v * inv_temp - logf(-logf(sample_uniform(seed, (uint32_t)r, sample, (uint32_t)i)));
        if (sc > best) { best = sc; braw = v; bidx = i; }          // ids arrive in increasing order: ties keep the lower one
    }
    float bm = wave_max(m);
    if (lane == 0) redm[wave] = bm;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o), ol = __shfl_xor(braw, o);
        const int oi = __shfl_xor(bidx, o);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; braw = ol; bidx = oi; }
    }
    if (lane == 0) { wv[wave] = best; wl[wave] = braw; wi[wave] = bidx; }
    __syncthreads();
    bm = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
    const float bs = wave_sum(m == -INFINITY ? 0.f : ssum * expf(m - bm));
    if (lane == 0) reds[wave] = bs;
    __syncthreads();
    if (tid == 0) {
        float gv = wv[0], gl = wl[0]; int gi = wi[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (wv[w] > gv || (wv[w] == gv && wi[w] < gi)) { gv = wv[w]; gl = wl[w]; gi = wi[w]; }
        const long slot = (long)r * kBeamParts + part;
        p.part_m[slot] = bm; p.part_s[slot] = reds[0] + reds[1] + reds[2] + reds[3];
        p.part_v[slot] = gv; p.part_l[slot] = gl; p.part_i[slot] = gi;
    }
}

// One workgroup per sequence, in place of dec_finalize: merge the parts in index order, record the token and its
// log-probability, apply the end-of-text rule, embed the next input.
__global__ __launch_bounds__(256) void sample_step_kernel(SampleParams p) {
    __shared__ int s_tok;
    const int r = blockIdx.x, tid = threadIdx.x;
    const int next = p.st->pos, prompt_len = p.st->prompt_len, max_new = p.st->max_new, eot = p.st->eot;
    if (tid == 0) {
        int tok;
        if (next < prompt_len) {
            tok = p.forced[(long)r * p.n_forced + next];
        } else if (p.finished[r]) {
            tok = eot;
        } else {
            const long row = (long)r * kBeamParts;
            float bm = -INFINITY;
            for (int q = 0; q < kBeamParts; ++q) bm = fmaxf(bm, p.part_m[row + q]);
            float bs = 0.f, gv = -INFINITY, gl = 0.f;
            int gi = 0x7fffffff;
            for (int q = 0; q < kBeamParts; ++q) {
                const float mq = p.part_m[row + q];
                if (mq != -INFINITY) bs += p.part_s[row + q] * expf(mq - bm);
                const float v = p.part_v[row + q]; const int ix = p.part_i[row + q];
                if (v > gv || (v == gv && ix < gi)) { gv = v; gl = p.part_l[row + q]; gi = ix; }
            }
            const bool ok = gi >= 0 && gi < p.vocab;            // false: everything masked or NaN
            tok = ok ? gi : eot;
            const int sample = next - prompt_len;
            if (sample < max_new) {
                p.out_ids[(long)r * max_new + sample] = tok;
                if (ok) p.sum_lp[r] += gl - (bm + logf(bs));
                if (tok == eot) {
                    p.out_len[r] = sample;
                    p.finished[r] = 1;
                    atomicAdd(&p.st->n_finished, 1);
                }
            }
        }
        s_tok = tok;
    }
    __syncthreads();
    const int tok = s_tok;
    if (next < p.n_text_ctx)
        for (int c = tid; c < p.D; c += 256)
            p.x[(long)r * p.D + c] = (float)p.tok_emb[(long)tok * p.D + c] + (float)p.pos_emb[(long)next * p.D + c];
}

// ------------------------------------------------------------------ prompt prefill
// The forced prompt (start-of-previous + initial prompt + task tokens: several dozen ids with the
// reference's initial_prompt, transcribe.py:40,111) is known before decoding starts, so all of its
// positions go through the decoder in ONE pass of M = batch * P rows (row m = sequence m / P,
// position m % P) on the big GEMM kernels, instead of P latency-bound single-position steps.  These
// are the pieces of that pass that are not GEMMs or the (re-used) cross-attention kernel.

// x[m][:] = token embedding of forced[m / P][m % P] + learned position m % P (modeling_whisper.py:737-762)
__global__ __launch_bounds__(256) void prefill_embed_kernel(const int* __restrict__ forced, int n_forced, int P,
                                                            const f16* __restrict__ tok_emb, const f16* __restrict__ pos_emb,
                                                            float* __restrict__ x, int D) {
    const int m = blockIdx.x, b = m / P, j = m - b * P;
    const int tok = forced[(long)b * n_forced + j];
    for (int c = threadIdx.x; c < D; c += 256) x[(long)m * D + c] = (float)tok_emb[(long)tok * D + c] + (float)pos_emb[(long)j * D + c];
}

// Causal self-attention among the P prompt positions of one (sequence, head), one wave per query.
// qkv: [M][3D] fp16 rows of (scaled q | k | v); every wave also files its own position's k and v rows
// in the layer's cache ([B][H][Tctx][64]), where the sampled positions will find them.
// Scores: lane l owns keys l, l+64, ... (<= 7 of them: P <= 448); output: lane l owns dimension l.
// cache_seq_stride: sequence b files its rows under cache sequence b * cache_seq_stride (beam search: the
// prompt rows of a recording live once, in the cache rows of its first beam).
__global__ __launch_bounds__(256) void prefill_self_attn_kernel(const f16* __restrict__ qkv, f16* __restrict__ kcache,
                                                                f16* __restrict__ vcache, f16* __restrict__ out, int H, int P,
                                                                int Tctx, int D, int cache_seq_stride) {
    constexpr int KPL = 7;
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.y * 4 + (threadIdx.x >> 6);      // query position
    if (j >= P) return;                                      // whole wave
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const long ld = 3L * D;
    const f16* base = qkv + (long)b * P * ld + h * 64;      // row j' of this sequence: base + j' * ld
    const f16* qr = base + (long)j * ld;
    {   // cache rows of position j
        const long dst = (((long)b * cache_seq_stride * H + h) * Tctx + j) * 64 + lane;
        kcache[dst] = qr[D + lane];
        vcache[dst] = qr[2 * D + lane];
    }
    float q[64];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const half8 v = *reinterpret_cast<const half8*>(qr + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) q[c * 8 + e] = (float)v[e];
    }
    float sc[KPL];
    float mx = -3.0e38f;
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
        const int key = lane + 64 * i;
        sc[i] = -3.0e38f;
        if (key <= j) {
            const f16* kr = base + (long)key * ld + D;
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const half8 v = *reinterpret_cast<const half8*>(kr + c * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) d += q[c * 8 + e] * (float)v[e];
            }
            sc[i] = d;
        }
        mx = fmaxf(mx, sc[i]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < KPL; ++i) { sc[i] = (lane + 64 * i <= j) ? __expf(sc[i] - mx) : 0.f; sum += sc[i]; }
    sum = wave_sum(sum);
    float o = 0.f;
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
        if (64 * i > j) break;                               // wave-uniform
        const int nk = min(64, j + 1 - 64 * i);
        for (int k = 0; k < nk; ++k) o += __shfl(sc[i], k) * (float)base[(long)(64 * i + k) * ld + 2 * D + lane];
    }
    out[((long)b * P + j) * D + h * 64 + lane] = (f16)(o / sum);
}

// Cross-attention of ALL prompt positions of one (sequence, head) over the cached encoder K/V
// (modeling_whisper.py:486-492 for every prompt row at once).  One workgroup per (sequence, head, block of 128
// prompt rows): the 1500 keys pass through LDS once, in tiles of 64, and serve every row of the block on the
// matrix cores, where one single-query launch per row would pull the same 375 KB through the L2 for each of
// them (75 prompt rows: 75 times).  Each wave owns 32 query rows (two 16-column tiles of the swapped product
// S^T = K Q^T, so a lane's scores all belong to one query and the softmax statistics stay in the lane);
// O^T = V^T P^T takes V^T from the row-major V tile with the transposing LDS read (ds_read_b64_tr_b16: lane
// 4a+b of a 16-lane group addresses row a, columns 4b..4b+3 of a 4 x 16 block and receives column 4a+b).
// V rows are 80 halfs apart in LDS: the 8 rows a 32-lane half reads then start 40 banks apart, on 8 distinct
// 8-bank runs.  Waves whose rows lie past the prompt only help with the staging.
// KV8: the cache holds e4m3 codes with one fp32 scale per row (cross_attn8_kernel's format); the codes go to the
// matrix cores as they are (exact in fp16), the K scale multiplies the score and the V scale the probability, the
// places where the single-query kernels apply them.
// q: [sequences * P][H * 64] (scaled queries), out likewise; sequence b reads the K/V of recording b / kv_rep.
constexpr int PFQ = 128, PFK = 64, PFVS = 80;
constexpr float kPfLog2e = 1.4426950408889634f;
constexpr float kPfRescaleGap = 6.0f;

__device__ __forceinline__ int pf_swz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }
__device__ __forceinline__ float pf_column_max(float v) {          // over lanes l&15, +16, +32, +48
    v = fmaxf(v, __shfl_xor(v, 16));
    return fmaxf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ float pf_column_sum(float v) {
    v += __shfl_xor(v, 16);
    return v + __shfl_xor(v, 32);
}

// CAUSAL: the same kernel as the prompt's SELF-attention (modeling_whisper.py:473-478 for all prompt rows): q, k, v are the
// three thirds of the rows of `q` ([sequences * P][3 * H * 64]: scaled q | k | v), key t is visible to query rows >= t,
// key tiles past a workgroup's last row are never staged, and every workgroup files the k / v rows of its own 128 prompt
// positions in the layer's cache (kcache / vcache [cache sequences][H][Tctx][64]; sequence b under cache sequence
// b * cache_seq_stride), where the sampled positions will find them.
// QT = 1: at most 16 query rows per (sequence, head) (the hypotheses of a recording in beam search or best-of-n
// sampling, one decode position each).  One wave per workgroup then has rows at all, with one 16-column tile; which
// wave it is rotates with the workgroup id, so that the workgroups sharing a CU compute on different SIMDs.  Such a
// launch is nothing but a stream of K/V through the CUs, and what it reaches depends on the bytes in flight: four
// workgroups per CU (the registers of the second query tile are free) and TWO tiles on their way per workgroup
// instead of one (one set of staging registers is filed in LDS while the other is still in flight).
template <bool KV8, bool CAUSAL = false, int QT = 2>
__global__ __launch_bounds__(256, QT == 1 ? 4 : 3) void prefill_cross_attn_kernel(const f16* __restrict__ q, const void* __restrict__ kc,
                                                                    const void* __restrict__ vc, const float* __restrict__ ks,
                                                                    const float* __restrict__ vs, f16* __restrict__ out, int H,
                                                                    int P, int T, int kv_rep, f16* __restrict__ kcache = nullptr,
                                                                    f16* __restrict__ vcache = nullptr, int Tctx = 0,
                                                                    int cache_seq_stride = 1, const int* __restrict__ done = nullptr,
                                                                    int done_div = 1) {
    static_assert(!(KV8 && CAUSAL), "the self-attention cache is fp16");
    static_assert(QT == 2 || !CAUSAL, "a prompt's self-attention has rows for every wave");
    constexpr int DEPTH = QT == 1 ? 2 : 1;          // tiles in flight from global memory
    __shared__ __attribute__((aligned(16))) f16 ldsK[2][PFK * 64];
    __shared__ __attribute__((aligned(16))) f16 ldsV[2][PFK * PFVS];
    __shared__ __attribute__((aligned(16))) float ldsS[2][2][KV8 ? PFK : 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int nqb = (P + PFQ - 1) / PFQ;
    const int bh = blockIdx.x / nqb, qblock = blockIdx.x - bh * nqb;
    const int seq = bh / H, h = bh - seq * H;
    if (done) {      // rows r with done[r / done_div] set keep what `out` held: a workgroup all of whose rows are done has no work
        bool all_done = true;
        for (int r = seq * P + qblock * PFQ; r < min(seq * P + P, seq * P + (qblock + 1) * PFQ) && all_done; ++r)
            all_done = done[r / done_div] != 0;
        if (all_done) return;                   // (workgroup-uniform, before any barrier)
    }
    const long bh_kv = (long)(seq / kv_rep) * H + h;
    const int q0 = QT == 2 ? qblock * PFQ + wave * 32 : 0;
    const bool active = QT == 2 ? q0 < P : wave == (int)(blockIdx.x & 3);          // wave-uniform
    const int D = H * 64;
    const long ldq = CAUSAL ? 3L * D : D;
    const f16* qbase = q + (long)seq * P * ldq + h * 64;

    half8 qf[QT][2];
#pragma unroll
    for (int qs = 0; qs < QT; ++qs) {
        const int row = min(q0 + qs * 16 + fr, P - 1);
#pragma unroll
        for (int s = 0; s < 2; ++s) qf[qs][s] = *reinterpret_cast<const half8*>(qbase + (long)row * ldq + s * 32 + fg * 8);
    }
    if constexpr (CAUSAL) {          // this block's k / v rows go to the cache: 128 rows x 8 pieces of 16 bytes, each of k and v
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = qblock * PFQ + (tid >> 3) + 32 * j;
            if (row < P) {
                const f16* src = qbase + (long)row * ldq + D + (tid & 7) * 8;
                const long dst = (((long)seq * cache_seq_stride * H + h) * Tctx + row) * 64 + (tid & 7) * 8;
                *reinterpret_cast<uint4*>(kcache + dst) = *reinterpret_cast<const uint4*>(src);
                *reinterpret_cast<uint4*>(vcache + dst) = *reinterpret_cast<const uint4*>(src + D);
            }
        }
    }

    // staging: a thread moves two 8-element chunks of K and two of V per tile (16 bytes each, 8 in the e4m3 cache)
    typedef typename std::conditional<KV8, uint2, uint4>::type Chunk;
    Chunk rk[DEPTH][2], rv[DEPTH][2];
    float rscale[DEPTH] = {};
    const int ch = tid & 7;
    auto gload = [&](int kv0, Chunk (&rk)[2], Chunk (&rv)[2], float& rscale) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = kv0 + (tid >> 3) + 32 * i;
            const long off = (bh_kv * T + key) * 64 + ch * 8;
            if constexpr (CAUSAL) {
                const f16* src = qbase + (long)key * ldq + D + ch * 8;
                rk[i] = key < T ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
                rv[i] = key < T ? *reinterpret_cast<const uint4*>(src + D) : make_uint4(0, 0, 0, 0);
            } else if constexpr (KV8) {
                rk[i] = key < T ? *reinterpret_cast<const uint2*>(static_cast<const uint8_t*>(kc) + off) : make_uint2(0, 0);
                rv[i] = key < T ? *reinterpret_cast<const uint2*>(static_cast<const uint8_t*>(vc) + off) : make_uint2(0, 0);
            } else {
                rk[i] = key < T ? *reinterpret_cast<const uint4*>(static_cast<const f16*>(kc) + off) : make_uint4(0, 0, 0, 0);
                rv[i] = key < T ? *reinterpret_cast<const uint4*>(static_cast<const f16*>(vc) + off) : make_uint4(0, 0, 0, 0);
            }
        }
        if constexpr (KV8) {
            if (tid < 2 * PFK) {
                const int key = kv0 + (tid & (PFK - 1));
                rscale = key < T ? (tid < PFK ? ks : vs)[bh_kv * T + key] : 0.f;
            }
        }
    };
    auto as_f16 = [](const Chunk& c) -> uint4 {
        if constexpr (KV8) {
            float f[8];
            e4m3x8_to_f32(c, f);
            const half8 hv = half8{(f16)f[0], (f16)f[1], (f16)f[2], (f16)f[3], (f16)f[4], (f16)f[5], (f16)f[6], (f16)f[7]};
            return *reinterpret_cast<const uint4*>(&hv);
        } else {
            return c;
        }
    };
    auto lstore = [&](int buf, const Chunk (&rk)[2], const Chunk (&rv)[2], float rscale) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (tid >> 3) + 32 * i;
            *reinterpret_cast<uint4*>(&ldsK[buf][pf_swz(row, ch)]) = as_f16(rk[i]);
            *reinterpret_cast<uint4*>(&ldsV[buf][row * PFVS + ch * 8]) = as_f16(rv[i]);
        }
        if constexpr (KV8) {
            if (tid < 2 * PFK) ldsS[buf][tid >> 6][tid & (PFK - 1)] = rscale;
        }
    };

    float4v o[4][QT];          // [head-dim tile][query tile]: O^T
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int qs = 0; qs < QT; ++qs) o[i][qs] = float4v{0, 0, 0, 0};
    // reference exponent (log2 units) and sum per query; the reference rises only when a tile exceeds it by
    // more than kPfRescaleGap, so the accumulators are rarely rescaled after the first tile
    float mrun[QT] = {};
    float lrun[QT] = {};

    const int ntile = ((CAUSAL ? min(T, (qblock + 1) * PFQ) : T) + PFK - 1) / PFK;
    gload(0, rk[0], rv[0], rscale[0]);
    lstore(0, rk[0], rv[0], rscale[0]);
    if (DEPTH == 2 && ntile > 1) gload(PFK, rk[0], rv[0], rscale[0]);
    __syncthreads();
    // one key tile: `cur` names the staging registers that hold (or are about to receive) tile it + 1
    auto tile_step = [&](int it, auto cur_c) {
        constexpr int cur = decltype(cur_c)::value, nxt = (cur + DEPTH - 1) % DEPTH;
        const int buf = it & 1, kv0 = it * PFK;
        if (it + DEPTH < ntile) gload(kv0 + DEPTH * PFK, rk[nxt], rv[nxt], rscale[nxt]);
        if (active && !(CAUSAL && kv0 > q0 + 31)) {          // (wave-uniform) causal: tiles past the wave's last row hold nothing for it
            // ---- S^T tiles: [key tile i][query tile qs]
            float4v sc[4][QT];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int qs = 0; qs < QT; ++qs) sc[i][qs] = float4v{0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const half8 kf = *reinterpret_cast<const half8*>(&ldsK[buf][pf_swz(i * 16 + fr, s * 4 + fg)]);
#pragma unroll
                    for (int qs = 0; qs < QT; ++qs) sc[i][qs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[qs][s], sc[i][qs], 0, 0, 0);
                }
            }
            // exponents of 2 relative to the reference; lane holds keys kv0 + 16i + 4fg + r of query fr
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float4v kscale = float4v{kPfLog2e, kPfLog2e, kPfLog2e, kPfLog2e};
                if constexpr (KV8) kscale = *reinterpret_cast<const float4v*>(&ldsS[buf][0][i * 16 + 4 * fg]) * kPfLog2e;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kv0 + i * 16 + 4 * fg + r;
                    const bool past = key >= T;
#pragma unroll
                    for (int qs = 0; qs < QT; ++qs) {
                        const bool hid = CAUSAL && key > q0 + 16 * qs + fr;
                        sc[i][qs][r] = (past || hid) ? -1.0e30f : fmaf(sc[i][qs][r], kscale[r], -mrun[qs]);
                    }
                }
            }
            // ---- online softmax per query column
            float mx[QT];
            bool lift = it == 0;
#pragma unroll
            for (int qs = 0; qs < QT; ++qs) {
                float m = fmaxf(fmaxf(sc[0][qs][0], sc[0][qs][1]), fmaxf(sc[0][qs][2], sc[0][qs][3]));
#pragma unroll
                for (int i = 1; i < 4; ++i) m = fmaxf(m, fmaxf(fmaxf(sc[i][qs][0], sc[i][qs][1]), fmaxf(sc[i][qs][2], sc[i][qs][3])));
                mx[qs] = pf_column_max(m);
                lift = lift || mx[qs] > kPfRescaleGap;
            }
            if (__any(lift)) {      // wave-uniform
#pragma unroll
                for (int qs = 0; qs < QT; ++qs) {
                    const float delta = it == 0 ? mx[qs] : fmaxf(mx[qs], 0.f);
                    const float alpha = it == 0 ? 1.f : __builtin_amdgcn_exp2f(-delta);
                    mrun[qs] += delta;
                    lrun[qs] *= alpha;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        o[i][qs][0] *= alpha; o[i][qs][1] *= alpha; o[i][qs][2] *= alpha; o[i][qs][3] *= alpha;
                        sc[i][qs][0] -= delta; sc[i][qs][1] -= delta; sc[i][qs][2] -= delta; sc[i][qs][3] -= delta;
                    }
                }
            }
            half8 pf[QT][2];           // [query tile][step of 32 keys]
#pragma unroll
            for (int qs = 0; qs < QT; ++qs) {
                float ps = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float4v vscale = float4v{1.f, 1.f, 1.f, 1.f};
                    if constexpr (KV8) vscale = *reinterpret_cast<const float4v*>(&ldsS[buf][1][i * 16 + 4 * fg]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(sc[i][qs][r]);
                        ps += pv;
                        sc[i][qs][r] = KV8 ? pv * vscale[r] : pv;
                    }
                }
                lrun[qs] += ps;                              // partial over this lane's keys; reduced at the end
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    pf[qs][s] = half8{(f16)sc[2 * s][qs][0], (f16)sc[2 * s][qs][1], (f16)sc[2 * s][qs][2], (f16)sc[2 * s][qs][3],
                                      (f16)sc[2 * s + 1][qs][0], (f16)sc[2 * s + 1][qs][1], (f16)sc[2 * s + 1][qs][2],
                                      (f16)sc[2 * s + 1][qs][3]};
            }
            // ---- O^T += V^T P^T; element j of step s is key 32s + 16(j>>2) + 4fg + (j&3)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const f16* blk = &ldsV[buf][(32 * s + 4 * fg + (fr >> 2)) * PFVS + dt * 16 + 4 * (fr & 3)];
                    typedef short short4v __attribute__((ext_vector_type(4)));
                    typedef short short8v __attribute__((ext_vector_type(8)));
                    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)blk);
                    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)(blk + 16 * PFVS));
                    const short8v both = short8v{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    const half8 vf = __builtin_bit_cast(half8, both);
#pragma unroll
                    for (int qs = 0; qs < QT; ++qs) o[dt][qs] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[qs][s], o[dt][qs], 0, 0, 0);
                }
            }
        }
        if (it + 1 < ntile) lstore(buf ^ 1, rk[cur], rv[cur], rscale[cur]);
        __syncthreads();
    };
    for (int it = 0; it < ntile; it += DEPTH) {
        tile_step(it, std::integral_constant<int, 0>{});
        if (DEPTH == 2 && it + 1 < ntile) tile_step(it + 1, std::integral_constant<int, DEPTH - 1>{});
    }
    if (!active) return;

    // ---- normalise and store: lane holds O[query fr][16dt + 4fg .. +3]
#pragma unroll
    for (int qs = 0; qs < QT; ++qs) {
        const float inv = 1.0f / pf_column_sum(lrun[qs]);
        const int row = q0 + qs * 16 + fr;
        if (row < P) {
            f16* dst = out + ((long)seq * P + row) * D + h * 64 + 4 * fg;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *reinterpret_cast<half4*>(dst + dt * 16) =
                    half4{(f16)(o[dt][qs][0] * inv), (f16)(o[dt][qs][1] * inv), (f16)(o[dt][qs][2] * inv), (f16)(o[dt][qs][3] * inv)};
        }
    }
}

// dx[b][:] = x[b * P + P - 1][:] (the residual row of the last prompt position) and position = P - 1:
// the state a step-by-step run would be in just before the final LayerNorm of its P-th step.
// rep > 1 (beam search): every beam of recording b gets the row, and src_rows [Tctx][n_seq * rep] of the
// prompt positions point all beams at the cache rows of the recording's first beam.
__global__ __launch_bounds__(256) void prefill_last_rows_kernel(const float* __restrict__ x, float* __restrict__ dx, int P, int D,
                                                                DecState* __restrict__ st, int rep, int* __restrict__ src_rows) {
    const int b = blockIdx.x, R = gridDim.x * rep;
    for (int j = 0; j < rep; ++j)
        for (int c = threadIdx.x; c < D; c += 256) dx[((long)b * rep + j) * D + c] = x[((long)b * P + P - 1) * D + c];
    if (src_rows)
        for (int i = threadIdx.x; i < P * rep; i += 256) src_rows[(long)(i / rep) * R + b * rep + (i % rep)] = b * rep;
    if (b == 0 && threadIdx.x == 0) st->pos = P - 1;
}

template <int MT, int KPW, int XV, int R, bool W8>
void launch_gemv_r(const GemvParams& p, int nw, hipStream_t s) {
    size_t lds = (size_t)nw * R * MT * 64 * 16;
    if (XV > 0) lds += (size_t)2 * p.K * 4 + (size_t)16 * MT * (p.K + 8) * 2;
    hipLaunchKernelGGL((gemv_kernel<MT, KPW, XV, R, W8>), dim3((p.N + p.nv - 1) / p.nv, (p.B + 16 * MT - 1) / (16 * MT)), dim3(nw * 64), lds, s, p);
}
template <int MT, int KPW, int XV, bool W8>
void launch_gemv_t(const GemvParams& p, int nw, hipStream_t s) {
    // two row tiles per workgroup: the fused kernels (the MLP's first matrix), and the plain single-tile form when
    // several groups of sequences share a launch
    if constexpr (XV > 0 || MT == 1) {
        if (p.nv > 16) { launch_gemv_r<MT, KPW, XV, 2, W8>(p, nw, s); return; }
    }
    launch_gemv_r<MT, KPW, XV, 1, W8>(p, nw, s);
}

constexpr int kMaxLds = 160 * 1024;

template <int MT, int KPW, int XV, bool W8>
void prepare_gemv_t() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemv_kernel<MT, KPW, XV, 1, W8>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
    if constexpr (XV > 0)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemv_kernel<MT, KPW, XV, 2, W8>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
}
template <int MT>
void prepare_mt() {
    prepare_gemv_t<MT, 1, 1, false>(); prepare_gemv_t<MT, 2, 3, false>(); prepare_gemv_t<MT, 2, 4, false>(); prepare_gemv_t<MT, 3, 6, false>();
    prepare_gemv_t<MT, 4, 8, false>(); prepare_gemv_t<MT, 5, 10, false>();
    prepare_gemv_t<MT, 1, 0, false>(); prepare_gemv_t<MT, 2, 0, false>(); prepare_gemv_t<MT, 3, 0, false>(); prepare_gemv_t<MT, 4, 0, false>();
    prepare_gemv_t<MT, 5, 0, false>(); prepare_gemv_t<MT, 6, 0, false>(); prepare_gemv_t<MT, 8, 0, false>(); prepare_gemv_t<MT, 10, 0, false>();
    // e4m3 weights: k-steps of 64 columns, so half the steps per wave
    prepare_gemv_t<MT, 1, 1, true>(); prepare_gemv_t<MT, 1, 3, true>(); prepare_gemv_t<MT, 1, 4, true>(); prepare_gemv_t<MT, 2, 6, true>();
    prepare_gemv_t<MT, 2, 8, true>(); prepare_gemv_t<MT, 3, 10, true>();
    prepare_gemv_t<MT, 1, 0, true>(); prepare_gemv_t<MT, 2, 0, true>(); prepare_gemv_t<MT, 3, 0, true>(); prepare_gemv_t<MT, 4, 0, true>();
    prepare_gemv_t<MT, 5, 0, true>();
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(logits_kernel<MT>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
}

template <int MT, bool W8>
bool launch_gemv_mt(const GemvParams& p, hipStream_t s) {
    const int nks = p.K / (W8 ? 64 : 32);
    // Many groups: every workgroup of the fused form re-reads and re-normalises its group's fp32 rows (82 KB at
    // K = 1280), which is what bounds the launch once the weights come from L2; normalise once instead.
    static const int ln_split_from = getenv("MW_LN_SPLIT_FROM") ? atoi(getenv("MW_LN_SPLIT_FROM")) : 33;
    if (p.ln_g && p.B < ln_split_from) {                   // fused LayerNorm: K = d_model, 512 threads
        if constexpr (W8) {
            switch (p.K) {
                case 128: launch_gemv_t<MT, 1, 1, true>(p, 8, s); return true;
                case 384: launch_gemv_t<MT, 1, 3, true>(p, 8, s); return true;
                case 512: launch_gemv_t<MT, 1, 4, true>(p, 8, s); return true;
                case 768: launch_gemv_t<MT, 2, 6, true>(p, 8, s); return true;
                case 1024: launch_gemv_t<MT, 2, 8, true>(p, 8, s); return true;
                case 1280: launch_gemv_t<MT, 3, 10, true>(p, 8, s); return true;
                default: break;
            }
        } else {
            switch (p.K) {
                case 128: launch_gemv_t<MT, 1, 1, false>(p, 8, s); return true;
                case 384: launch_gemv_t<MT, 2, 3, false>(p, 8, s); return true;
                case 512: launch_gemv_t<MT, 2, 4, false>(p, 8, s); return true;
                case 768: launch_gemv_t<MT, 3, 6, false>(p, 8, s); return true;
                case 1024: launch_gemv_t<MT, 4, 8, false>(p, 8, s); return true;
                case 1280: launch_gemv_t<MT, 5, 10, false>(p, 8, s); return true;
                default: break;
            }
        }
    }
    if (p.ln_g) {
        // no fused instantiation for this width (or many groups): separate LayerNorm launch, then the plain form
        launch_layernorm(p.x32, p.ln_g, p.ln_b, p.ln_scratch, nullptr, p.B, p.K, s);
        GemvParams q = p;
        q.ln_g = nullptr; q.ln_b = nullptr; q.xin16 = p.ln_scratch;
        if (q.nv > 16 && MT != 1) q.nv = 16;    // the plain form has two row tiles for MT = 1 only
        return launch_gemv_mt<MT, W8>(q, s);
    }
    const int nw = nks >= 48 ? 16 : nks >= 16 ? 8 : 4;
    const int kpw = (nks + nw - 1) / nw;
    if constexpr (W8) {
        switch (kpw) {
            case 1: launch_gemv_t<MT, 1, 0, true>(p, nw, s); return true;
            case 2: launch_gemv_t<MT, 2, 0, true>(p, nw, s); return true;
            case 3: launch_gemv_t<MT, 3, 0, true>(p, nw, s); return true;
            case 4: launch_gemv_t<MT, 4, 0, true>(p, nw, s); return true;
            case 5: launch_gemv_t<MT, 5, 0, true>(p, nw, s); return true;
            default: return false;      // K > 5120
        }
    } else {
        switch (kpw) {
            case 1: launch_gemv_t<MT, 1, 0, false>(p, nw, s); return true;
            case 2: launch_gemv_t<MT, 2, 0, false>(p, nw, s); return true;
            case 3: launch_gemv_t<MT, 3, 0, false>(p, nw, s); return true;
            case 4: launch_gemv_t<MT, 4, 0, false>(p, nw, s); return true;
            case 5: launch_gemv_t<MT, 5, 0, false>(p, nw, s); return true;
            case 6: launch_gemv_t<MT, 6, 0, false>(p, nw, s); return true;
            case 7: case 8: launch_gemv_t<MT, 8, 0, false>(p, nw, s); return true;
            case 9: case 10: launch_gemv_t<MT, 10, 0, false>(p, nw, s); return true;
            default: return false;      // K > 5120
        }
    }
}

template <int MT, int R, bool W8>
bool launch_splitk_r(const GemvParams& p, int S, hipStream_t s) {
    const int nks = p.K / (W8 ? 64 : 32);
    const int ks_per = (nks + S - 1) / S;
    S = (nks + ks_per - 1) / ks_per;                    // drop ranges that would be empty
    const int kpw = (ks_per + kSplitKWaves - 1) / kSplitKWaves;
    const dim3 grid(p.N / (16 * R), S, (p.B + 16 * MT - 1) / (16 * MT)), block(kSplitKWaves * 64);
    switch (kpw) {
#define MW_SK_CASE(KPW) case KPW: hipLaunchKernelGGL((gemv_splitk_kernel<MT, KPW, R, W8>), grid, block, 0, s, p, ks_per); return true;
        MW_SK_CASE(1) MW_SK_CASE(2) MW_SK_CASE(3) MW_SK_CASE(4)
#undef MW_SK_CASE
        default: break;
    }
    if constexpr (!W8) {
        switch (kpw) {
#define MW_SK_CASE(KPW) case KPW: hipLaunchKernelGGL((gemv_splitk_kernel<MT, KPW, R, false>), grid, block, 0, s, p, ks_per); return true;
            MW_SK_CASE(5) MW_SK_CASE(6) MW_SK_CASE(7) MW_SK_CASE(8)
#undef MW_SK_CASE
            default: break;
        }
    }
    return false;
}

// Residual projection with K >= 2N (the MLP's second matrix): 32-row groups and as many K ranges as
// give about one workgroup per CU.  Square projections (K = N = d_model) stay on gemv_kernel:
// their activation re-reads are 4x smaller and the merge hand-off costs more than it saves
// (measured: 3 ranges of 16-row groups slowed a large-v3 step by 1.3 %).
// Returns false when the shape does not fit; the plain kernel takes over.
bool launch_splitk(const GemvParams& p, int mt, hipStream_t s) {
    static const bool off = getenv("MW_SPLITK_OFF") != nullptr;
    const int kstep = p.W8 ? 64 : 32;
    if (off || !p.sk_partial || !p.sk_tickets || p.K % kstep || p.K < 2 * p.N || p.N % 32) return false;
    const int groups = p.N / 32;
    const int S = max(1, min(240 / groups, min(kSplitKMax, p.K / kstep)));
    if (groups > p.sk_groups || (long)groups * S * 2 * mt * 256 > p.sk_partial_floats) return false;
    if (p.W8) return mt == 1 ? launch_splitk_r<1, 2, true>(p, S, s) : launch_splitk_r<2, 2, true>(p, S, s);
    return mt == 1 ? launch_splitk_r<1, 2, false>(p, S, s) : launch_splitk_r<2, 2, false>(p, S, s);
}


// gemm_dec_kernel for more than 32 sequences (fp16 input rows; beyond 128 as blocks of 128 in the grid's z).  Row tiles per workgroup: 4 for wide outputs and long K
// (fewer passes over the activations), 2 for narrow ones (more row groups); K ranges so that one round of
// workgroups about fills the CUs and a range fits the registers (KR k-steps) and LDS.  Returns false, having
// launched nothing, when the shape does not fit (the grouped gemv_kernel takes over).
template <int MT, int RT, int KR, bool W8>
void launch_gemm_dec_t(const GemvParams& p, int S, int ks_per, hipStream_t s) {
    constexpr size_t lds = dec_gemm_lds_bytes<MT, KR, W8>();
    const dim3 grid((p.N + 16 * RT - 1) / (16 * RT), S, (p.B + 16 * MT - 1) / (16 * MT)), block(kDecGemmWaves * 64);
    hipLaunchKernelGGL((gemm_dec_kernel<MT, RT, KR, W8>), grid, block, lds, s, p, ks_per);
}

template <int MT, bool W8>
bool launch_gemm_dec_mt(const GemvParams& p, hipStream_t s) {
    // the longest K range: 512 columns are 132 KB of LDS at 128 sequences; 112 sequences (MT = 7) leave room for 640
    constexpr int KSTEP = W8 ? 64 : 32, KR_MAX = MT == 7 ? (W8 ? 10 : 20) : (W8 ? 8 : 16);
    if (p.K % KSTEP || p.N % 4 || !p.xin16 || p.ln_g) return false;
    const int nks = p.K / KSTEP;
    const bool wide = p.N >= 2048 || p.K > 2048;
    // Row tiles per workgroup, twice as many (8 wide / 4 narrow) beyond 64 sequences when the caller says that other
    // engines share the GPU (p.dg_fat): every 16 * RT weight rows pull the whole activation matrix through the
    // L2 -> CU path again, and where the decode groups' K/V streams keep that path busy on every CU the bytes count,
    // not the number of workgroups (benchmark step 144.0 - 144.4 ms -> 140.7 - 141.3; with the smaller target below
    // 138.4 - 140.6).  A decode group that has the GPU to itself is 4 % faster with the smaller workgroups (one loop of
    // 112 sequences: 939 against 994 ms).  MW_DEC_GEMM_RT2X overrides: bit 0 wide, bit 1 narrow.
    static const int rt_env = getenv("MW_DEC_GEMM_RT2X") ? atoi(getenv("MW_DEC_GEMM_RT2X")) : -1;
    const bool fat = p.dg_fat && MT > 4;
    const int rt_scale = rt_env >= 0 ? rt_env : (fat ? 3 : 0);
    const bool twice = wide ? (rt_scale & 1) : (rt_scale & 2);
    const int rt = MT == 7 ? 8 : (wide ? 4 : 2) * (twice ? 2 : 1);
    const int groups = (p.N + 16 * rt - 1) / (16 * rt);
    // Workgroups aimed at: about one per CU (240) is what a launch alone on the GPU wants, but every K range writes
    // and re-reads a partial tile of all sequences through memory, and in a run where several decode groups and the
    // encoders share the GPU that traffic costs more than the shorter launch gains: with 120 (three ranges instead of
    // four to six for K = 1280) the benchmark step is 1.5-2 % shorter (156.3-157.1 ms against 158.7-160.2).
    static const int target_env = getenv("MW_DEC_GEMM_WGS") ? atoi(getenv("MW_DEC_GEMM_WGS")) : 0;
    const int target = target_env > 0 ? target_env : MT == 7 ? 20 : (MT > 4 ? (fat ? 90 : 120) : 240);
    int S = max((nks + KR_MAX - 1) / KR_MAX, min(nks, (target + groups / 2) / groups));
    S = max(1, S);
    int ks_per = (nks + S - 1) / S;
    S = (nks + ks_per - 1) / ks_per;
    const int blocks = (p.B + 16 * MT - 1) / (16 * MT);                 // sequence blocks (grid z), each with its own slab and tickets
    const long slab = (long)groups * S * 16 * rt * 16 * MT;
    if ((long)groups * blocks > p.dg_groups || (S > 1 && (!p.dg_partial || !p.sk_tickets || slab * blocks > p.dg_partial_floats)))
        return false;
    GemvParams q = p;
    q.sk_partial = p.dg_partial;                 // this kernel's own, larger slab; the tickets are shared (always left at zero)
    q.sk_partial_floats = slab; q.sk_groups = groups;                   // extents of one sequence block (group_view)
#define MW_DG(RT, KR) launch_gemm_dec_t<MT, RT, KR, W8>(q, S, ks_per, s)
    if constexpr (MT == 7) {
        if (ks_per <= KR_MAX / 2) MW_DG(8, KR_MAX / 2); else MW_DG(8, KR_MAX);
    } else if (twice) {
        if (ks_per <= KR_MAX / 2) { if (wide) MW_DG(8, KR_MAX / 2); else MW_DG(4, KR_MAX / 2); }
        else { if (wide) MW_DG(8, KR_MAX); else MW_DG(4, KR_MAX); }
    } else if (ks_per <= KR_MAX / 2) { if (wide) MW_DG(4, KR_MAX / 2); else MW_DG(2, KR_MAX / 2); }
    else { if (wide) MW_DG(4, KR_MAX); else MW_DG(2, KR_MAX); }
#undef MW_DG
    return true;
}

template <int MT, bool W8>
void prepare_gemm_dec_mt() {
    constexpr int kDecGemmLds = 144 * 1024;      // the kernel also holds a static word: the whole 160 KiB cannot be dynamic
    constexpr int KR_MAX = MT == 7 ? (W8 ? 10 : 20) : (W8 ? 8 : 16);
    static_assert(dec_gemm_lds_bytes<MT, KR_MAX, W8>() <= kDecGemmLds, "the longest K range fits the LDS");
    if constexpr (MT == 7) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_dec_kernel<MT, 8, KR_MAX / 2, W8>), hipFuncAttributeMaxDynamicSharedMemorySize, kDecGemmLds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_dec_kernel<MT, 8, KR_MAX, W8>), hipFuncAttributeMaxDynamicSharedMemorySize, kDecGemmLds);
    } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_dec_kernel<MT, 2, KR_MAX / 2, W8>), hipFuncAttributeMaxDynamicSharedMemorySize, kDecGemmLds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_dec_kernel<MT, 4, KR_MAX / 2, W8>), hipFuncAttributeMaxDynamicSharedMemorySize, kDecGemmLds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_dec_kernel<MT, 2, KR_MAX, W8>), hipFuncAttributeMaxDynamicSharedMemorySize, kDecGemmLds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_dec_kernel<MT, 4, KR_MAX, W8>), hipFuncAttributeMaxDynamicSharedMemorySize, kDecGemmLds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_dec_kernel<MT, 8, KR_MAX / 2, W8>), hipFuncAttributeMaxDynamicSharedMemorySize, kDecGemmLds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_dec_kernel<MT, 8, KR_MAX, W8>), hipFuncAttributeMaxDynamicSharedMemorySize, kDecGemmLds);
    }
}

bool launch_gemm_dec(const GemvParams& p, hipStream_t s) {
    if (p.B <= 32) return false;
    if (p.B <= 64) return p.W8 ? launch_gemm_dec_mt<4, true>(p, s) : launch_gemm_dec_mt<4, false>(p, s);
    // up to 112 sequences beside other engines: seven sequence tiles leave LDS for K ranges of 640 columns, i.e. two
    // ranges instead of three for K = 1280 and 40 % fewer partial tiles through memory (benchmark step 140.3 -> 137.9 ms;
    // MW_DEC_GEMM_MT7=0 switches it off, 2 keeps it to the wide outputs)
    static const int mt7 = getenv("MW_DEC_GEMM_MT7") ? atoi(getenv("MW_DEC_GEMM_MT7")) : 1;
    // (larger groups: blocks of 112 sequences wherever that takes no more blocks than 128 would: 129..224, 257..336, ...)
    if ((p.B + 111) / 112 == (p.B + 127) / 128 && p.dg_fat && mt7 && (mt7 == 1 || p.N >= 2048 || p.K > 2048))
        return p.W8 ? launch_gemm_dec_mt<7, true>(p, s) : launch_gemm_dec_mt<7, false>(p, s);
    return p.W8 ? launch_gemm_dec_mt<8, true>(p, s) : launch_gemm_dec_mt<8, false>(p, s);
}

}  // namespace

int gemv_logits_parts(int N) { return ((N + 15) / 16 + 7) / 8; }

void dec_kernels_prepare() {
    prepare_mt<1>(); prepare_mt<2>();
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(logits_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLogitsWideLds);
    prepare_gemm_dec_mt<4, false>(); prepare_gemm_dec_mt<4, true>(); prepare_gemm_dec_mt<8, false>(); prepare_gemm_dec_mt<8, true>(); prepare_gemm_dec_mt<7, false>(); prepare_gemm_dec_mt<7, true>();
}

void launch_gemv(const GemvParams& p0, hipStream_t s) {
    GemvParams p = p0;
    // up to 16 sequences: one row tile; up to 32: two row tiles per workgroup; beyond: groups of 32 (see group_view)
    static const int group_from = getenv("MW_GROUP_FROM") ? atoi(getenv("MW_GROUP_FROM")) : 33;
    static const int group_mt = getenv("MW_GROUP_MT") ? atoi(getenv("MW_GROUP_MT")) : 2;
    const int mt = p.B <= 16 ? 1 : p.B < group_from ? 2 : group_mt;
    if (p.out_mode == GV_LOGITS) {
        static const bool wide_off = getenv("MW_LOGITS_WIDE") && !atoi(getenv("MW_LOGITS_WIDE"));
        if (p.B > 32 && !wide_off && p.K % 32 == 0) {      // decode groups: one pass over the embedding per block of 128 sequences
            const int parts = gemv_logits_parts(p.N), n_groups = (p.B + LW_ROWS - 1) / LW_ROWS;
            hipLaunchKernelGGL(logits_wide_kernel, dim3(n_groups > 1 ? (parts + 7) / 8 * 8 * n_groups : parts), dim3(512), kLogitsWideLds, s, p);
            return;
        }
        const int parts = gemv_logits_parts(p.N), n_groups = (p.B + 16 * mt - 1) / (16 * mt);
        dim3 grid(n_groups > 1 ? (parts + 7) / 8 * 8 * n_groups : parts);      // several groups: see the id mapping in logits_kernel
        const size_t lds = (size_t)16 * mt * (p.K + 8) * 2 + (size_t)8 * 16 * mt * 8;
        if (mt == 1) hipLaunchKernelGGL(logits_kernel<1>, grid, dim3(512), lds, s, p);
        else hipLaunchKernelGGL(logits_kernel<2>, grid, dim3(512), lds, s, p);
        return;
    }
    if (p.B > 32) {
        // one pass over the weights for all sequences; the LayerNorm (when there is one) as its own launch
        static const bool dg_off = getenv("MW_DEC_GEMM_OFF") != nullptr;
        if (!dg_off && p.out_mode != GV_LOGITS) {
            GemvParams q = p;
            if (p.ln_g) { q.ln_g = nullptr; q.ln_b = nullptr; q.xin16 = p.ln_scratch; }
            // shape check first (launch_gemm_dec returns false before launching anything), then the LayerNorm
            if (q.K % (q.W8 ? 64 : 32) == 0 && q.N % 16 == 0) {
                if (p.ln_g) launch_layernorm(p.x32, p.ln_g, p.ln_b, p.ln_scratch, nullptr, p.B, p.K, s);
                if (launch_gemm_dec(q, s)) return;
                if (p.ln_g) { p = q; }          // the rows are normalised already: the grouped form below reads them
            }
        }
    }
    // narrow outputs: 8 rows per workgroup so that at least ~160 workgroups stream weights
    static const int nv_narrow = getenv("MW_NV_NARROW") ? atoi(getenv("MW_NV_NARROW")) : 8;
    static const int nv_wide = getenv("MW_NV_WIDE") ? atoi(getenv("MW_NV_WIDE")) : 16;
    // several groups (see group_view) already multiply the workgroups: 16 rows each, so that a workgroup's weight
    // slice is not smaller than the activation rows it has to fetch beside it
    static const bool grp_nv_off = getenv("MW_GROUP_NV_OFF") != nullptr;
    const int n_groups = (p.B + 16 * mt - 1) / (16 * mt);
    p.nv = (p.N <= 2048 && (n_groups < 2 || grp_nv_off)) ? nv_narrow : nv_wide;
    static const bool grp_r2 = getenv("MW_GROUP_R2_OFF") == nullptr;      // wide outputs of grouped launches: 32 rows per workgroup
    if (grp_r2 && n_groups >= 2 && mt == 1 && p.N >= 3072 && p.N % 32 == 0) p.nv = 32;
    if (p.ln_g && p.nv == 16 && p.N > 256 * 16 && p.N <= 256 * 32)        // at most one workgroup per CU
        p.nv = ((p.N + 255) / 256 + 3) / 4 * 4;
    if (p.out_mode == GV_RESID && !p.ln_g && launch_splitk(p, mt, s)) return;
    if (p.W8) { if (mt == 1) (void)launch_gemv_mt<1, true>(p, s); else (void)launch_gemv_mt<2, true>(p, s); }
    else { if (mt == 1) (void)launch_gemv_mt<1, false>(p, s); else (void)launch_gemv_mt<2, false>(p, s); }
}

template <int NT, int RPT>
static void launch_self_attn_rpt(const f16* q, const f16* kc, const f16* vc, f16* out, int B, int H, int Tctx, const DecState* st,
                                 hipStream_t s, const int* src_rows) {
    if (src_rows)
        hipLaunchKernelGGL((dec_attn_kernel<NT, RPT, 1, true>), dim3(B * H), dim3(NT), 0, s, q, kc, vc, out, H, Tctx, 0, st,
                           (float*)nullptr, (int*)nullptr, 1, src_rows, B);
    else
        hipLaunchKernelGGL((dec_attn_kernel<NT, RPT, 1>), dim3(B * H), dim3(NT), 0, s, q, kc, vc, out, H, Tctx, 0, st,
                           (float*)nullptr, (int*)nullptr, 1, (const int*)nullptr, B);
}

int dec_self_attn_bucket(int n_keys) { return n_keys <= 64 ? 0 : n_keys <= 128 ? 1 : n_keys <= 256 ? 2 : 3; }

void launch_dec_self_attn(const f16* q, const f16* kc, const f16* vc, f16* out, int B, int H, int Tctx,
                          const DecState* st, hipStream_t s, const int* src_rows, int max_keys) {
    // NT threads = NT/8 row groups; a thread owns RPT cached positions.  The launch is sized for the
    // keys that can exist (`max_keys` >= position + 1 of every step it will serve): rows past the
    // context only cost duplicate loads, and at the start of a transcript that is most of them.
    // Few (sequence, head) pairs: 512 threads each, so that the chip holds them all at once with the fewest rows
    // per thread.  Many (a decode group of 33..128 sequences): 128-thread workgroups with more rows per thread,
    // 16 of them per CU, so that the whole launch is still resident in one round (2560 workgroups of 512 threads
    // were two and a half rounds: 15 us per launch at 128 sequences).
    const bool many = (long)B * H > 1024;
    switch (dec_self_attn_bucket(max_keys)) {
        case 0: if (many) launch_self_attn_rpt<128, 4>(q, kc, vc, out, B, H, Tctx, st, s, src_rows);
                else launch_self_attn_rpt<512, 1>(q, kc, vc, out, B, H, Tctx, st, s, src_rows);
                break;
        case 1: if (many) launch_self_attn_rpt<128, 8>(q, kc, vc, out, B, H, Tctx, st, s, src_rows);
                else launch_self_attn_rpt<512, 2>(q, kc, vc, out, B, H, Tctx, st, s, src_rows);
                break;
        case 2: if (many) launch_self_attn_rpt<256, 8>(q, kc, vc, out, B, H, Tctx, st, s, src_rows);
                else launch_self_attn_rpt<512, 4>(q, kc, vc, out, B, H, Tctx, st, s, src_rows);
                break;
        default: launch_self_attn_rpt<512, 7>(q, kc, vc, out, B, H, Tctx, st, s, src_rows); break;   // 448 positions
    }
}

// LDS request that holds a kernel without LDS needs of its own to `per_cu` workgroups per CU (160 KiB of LDS per CU)
static size_t lds_request_for(int per_cu) { return per_cu <= 0 ? 0 : per_cu == 1 ? 96 * 1024 : (size_t)(160 * 1024 / (per_cu + 1) + 1024); }

void launch_dec_cross_attn(const f16* q, const f16* kc, const f16* vc, f16* out, int B, int H, int T, float* partial,
                           int* tickets, hipStream_t s, int kv_div, int row0, const int* done, int done_div, int stream_from_arg,
                           int stream_per_cu) {
    // 4 key ranges of 375 frames per (sequence, head); 256 threads = 32 row groups x 12 rows = 384 >= 375
    const dim3 g1(B * H * kXattnSplit), g2(H * kXattnSplit, B);
    const DecState* nost = nullptr;
    const int* norows = nullptr;
    // Launches of many rounds (decode groups): an LDS request that nothing uses holds the kernel to three workgroups
    // per CU instead of five.  Three workgroups' loads in flight already cover the HBM round trip; a deeper queue
    // only delays everyone's requests, this kernel's own included: 6.18 -> 6.44 TB/s alone at 112 sequences
    // (two per CU: 5.80), the benchmark step 160.5 -> 159.0 ms.  Launches that fit the chip in one round keep five.
    static const int xlds_env = getenv("MW_XATTN_LDS") ? atoi(getenv("MW_XATTN_LDS")) : -1;
    const size_t xlds = xlds_env >= 0 ? (size_t)xlds_env : ((long)B * H * kXattnSplit >= 2 * 1280 ? 40000 : 0);
    // Decode groups (at least ~4 (sequence, head) pairs per CU): one workgroup per pair walks all the keys, no key
    // ranges and no ticket hand-off (cross_attn_stream_kernel); three workgroups per CU as above.
    static const int stream_from_env = getenv("MW_XSTREAM_FROM") ? atoi(getenv("MW_XSTREAM_FROM")) : -1;     // A/B runs
    const int stream_from = stream_from_env >= 0 ? stream_from_env : stream_from_arg;
    static const int stream_lds_env = getenv("MW_XSTREAM_LDS") ? atoi(getenv("MW_XSTREAM_LDS")) : -1;      // A/B runs
    const size_t stream_lds = stream_lds_env >= 0 ? (size_t)stream_lds_env : lds_request_for(stream_per_cu);
    if (kv_div == 1 && stream_from > 0 && (long)B * H >= stream_from) {
        if (done) hipLaunchKernelGGL(cross_attn_stream_kernel<true>, dim3(H, B), dim3(256), stream_lds, s, q, kc, vc, out, H, T, done, done_div);
        else hipLaunchKernelGGL(cross_attn_stream_kernel<false>, dim3(B * H), dim3(256), stream_lds, s, q, kc, vc, out, H, T, done, done_div);
        return;
    }
    if (kv_div > 1) {
        if (done) hipLaunchKernelGGL((dec_attn_kernel<256, 12, kXattnSplit, false, true, true>), g2, dim3(256), xlds, s, q, kc, vc, out, H, T, T,
                                     nost, partial, tickets, kv_div, norows, row0, done, done_div);
        else hipLaunchKernelGGL((dec_attn_kernel<256, 12, kXattnSplit, false, true>), g1, dim3(256), xlds, s, q, kc, vc, out, H, T, T, nost,
                                partial, tickets, kv_div, norows, row0, done, done_div);
    } else {
        if (done) hipLaunchKernelGGL((dec_attn_kernel<256, 12, kXattnSplit, false, false, true>), g2, dim3(256), xlds, s, q, kc, vc, out, H, T, T,
                                     nost, partial, tickets, 1, norows, B, done, done_div);
        else hipLaunchKernelGGL((dec_attn_kernel<256, 12, kXattnSplit>), g1, dim3(256), xlds, s, q, kc, vc, out, H, T, T, nost, partial,
                                tickets, 1, norows, B, done, done_div);
    }
}

void launch_prefill_embed(const int* forced, int n_forced, int B, int P, const f16* tok_emb, const f16* pos_emb, float* x, int D,
                          hipStream_t s) {
    hipLaunchKernelGGL(prefill_embed_kernel, dim3(B * P), dim3(256), 0, s, forced, n_forced, P, tok_emb, pos_emb, x, D);
}

void launch_prefill_self_attn(const f16* qkv, f16* kcache, f16* vcache, f16* out, int B, int H, int P, int Tctx, int D, hipStream_t s,
                              int cache_seq_stride) {
    // a handful of positions: one wave per query; a real prompt: all rows of a (sequence, head) on the matrix cores
    static const int mfma_from = getenv("MW_PREFILL_SELF_MFMA_FROM") ? atoi(getenv("MW_PREFILL_SELF_MFMA_FROM")) : 16;
    if (P < mfma_from || D != H * 64) {
        hipLaunchKernelGGL(prefill_self_attn_kernel, dim3(B * H, (P + 3) / 4), dim3(256), 0, s, qkv, kcache, vcache, out, H, P, Tctx, D,
                           cache_seq_stride);
        return;
    }
    const float* none = nullptr;
    hipLaunchKernelGGL((prefill_cross_attn_kernel<false, true>), dim3(B * H * ((P + PFQ - 1) / PFQ)), dim3(256), 0, s, qkv,
                       (const void*)nullptr, (const void*)nullptr, none, none, out, H, P, P, 1, kcache, vcache, Tctx, cache_seq_stride);
}

void launch_prefill_cross_attn(const f16* q, const f16* kc, const f16* vc, f16* out, int B, int P, int H, int T, int kv_rep,
                               hipStream_t s, const int* done, int done_div) {
    const float* none = nullptr;
    f16* nocache = nullptr;
    static const int narrow = getenv("MW_PREFILL_NARROW") ? atoi(getenv("MW_PREFILL_NARROW")) : 1;
    if (P <= 16 && narrow)      // a decode position of a few hypotheses per recording: one query tile, deeper staging
        hipLaunchKernelGGL((prefill_cross_attn_kernel<false, false, 1>), dim3(B * H), dim3(256), 0, s, q, kc, vc, none, none, out, H, P, T,
                           kv_rep, nocache, nocache, 0, 1, done, done_div);
    else
        hipLaunchKernelGGL(prefill_cross_attn_kernel<false>, dim3(B * H * ((P + PFQ - 1) / PFQ)), dim3(256), 0, s, q, kc, vc, none, none,
                           out, H, P, T, kv_rep, nocache, nocache, 0, 1, done, done_div);
}

void launch_prefill_cross_attn8(const f16* q, const uint8_t* k8, const uint8_t* v8, const float* ks, const float* vs, f16* out, int B,
                                int P, int H, int T, int kv_rep, hipStream_t s, const int* done, int done_div) {
    f16* nocache = nullptr;
    static const int narrow = getenv("MW_PREFILL_NARROW") ? atoi(getenv("MW_PREFILL_NARROW")) : 1;
    if (P <= 16 && narrow)
        hipLaunchKernelGGL((prefill_cross_attn_kernel<true, false, 1>), dim3(B * H), dim3(256), 0, s, q, k8, v8, ks, vs, out, H, P, T, kv_rep,
                           nocache, nocache, 0, 1, done, done_div);
    else
        hipLaunchKernelGGL(prefill_cross_attn_kernel<true>, dim3(B * H * ((P + PFQ - 1) / PFQ)), dim3(256), 0, s, q, k8, v8, ks, vs, out, H,
                           P, T, kv_rep, nocache, nocache, 0, 1, done, done_div);
}

void launch_prefill_last_rows(const float* x, float* dx, int B, int P, int D, DecState* st, hipStream_t s, int rep, int* src_rows) {
    hipLaunchKernelGGL(prefill_last_rows_kernel, dim3(B), dim3(256), 0, s, x, dx, P, D, st, rep, src_rows);
}

void launch_dec_cross_attn8(const f16* q, const uint8_t* k8, const uint8_t* v8, const float* ks, const float* vs, f16* out, int B,
                            int H, int T, float* partial, int* tickets, hipStream_t s, int kv_div, int row0, const int* done, int done_div,
                            int stream_from_arg, int stream_per_cu) {
    // decode groups: one workgroup per (sequence, head) walking all keys, as launch_dec_cross_attn
    static const int stream_from_env = getenv("MW_XSTREAM_FROM") ? atoi(getenv("MW_XSTREAM_FROM")) : -1;
    const int stream_from = stream_from_env >= 0 ? stream_from_env : stream_from_arg;
    static const int stream_lds_env = getenv("MW_XSTREAM_LDS") ? atoi(getenv("MW_XSTREAM_LDS")) : -1;      // A/B runs
    const size_t stream_lds = stream_lds_env >= 0 ? (size_t)stream_lds_env : lds_request_for(stream_per_cu);
    if (kv_div == 1 && stream_from > 0 && (long)B * H >= stream_from) {
        if (done) hipLaunchKernelGGL(cross_attn8_stream_kernel<true>, dim3(H, B), dim3(256), stream_lds, s, q, k8, v8, ks, vs, out, H, T, done, done_div);
        else hipLaunchKernelGGL(cross_attn8_stream_kernel<false>, dim3(B * H), dim3(256), stream_lds, s, q, k8, v8, ks, vs, out, H, T, done, done_div);
        return;
    }
    // 4 key ranges of 375 frames; 256 threads = 32 row groups x 12 rows = 384 >= 375
    const dim3 g1(B * H * kXattnSplit), g2(H * kXattnSplit, B);
#define MW_X8(GRP, DONE, GRID, KVD, R0) hipLaunchKernelGGL((cross_attn8_kernel<256, 12, kXattnSplit, GRP, DONE>), GRID, dim3(256), 0, s, q, k8, \
                                                           v8, ks, vs, out, H, T, partial, tickets, KVD, R0, done, done_div)
    if (kv_div > 1) { if (done) MW_X8(true, true, g2, kv_div, row0); else MW_X8(true, false, g1, kv_div, row0); }
    else { if (done) MW_X8(false, true, g2, 1, 0); else MW_X8(false, false, g1, 1, 0); }
#undef MW_X8
}

void launch_dec_init(const FinalizeParams& p, int B, int prompt_len, int max_new, int eot, hipStream_t s) {
    hipLaunchKernelGGL(dec_init_kernel, dim3(B), dim3(256), 0, s, p, B, prompt_len, max_new, eot);
}

void launch_dec_finalize(const FinalizeParams& p, int B, hipStream_t s) {
    hipLaunchKernelGGL(dec_finalize_kernel, dim3(B), dim3(256), 0, s, p, B);
}

void launch_beam_init(const BeamParams& p, int prompt_len, int max_new, int eot, hipStream_t s) {
    hipLaunchKernelGGL(beam_init_kernel, dim3(p.n_audio), dim3(256), 0, s, p, prompt_len, max_new, eot);
}

void launch_beam_step(const BeamParams& p, hipStream_t s) {
    hipLaunchKernelGGL(beam_scan_kernel, dim3(kBeamParts, p.n_audio * p.beam), dim3(256), 0, s, p);
    const size_t lds = (size_t)p.beam * p.n_text_ctx * sizeof(int);      // n_text_ctx >= max_new
    hipLaunchKernelGGL(beam_step_kernel, dim3(p.n_audio), dim3(256), lds, s, p);
}

void launch_sample_init(const SampleParams& p, int prompt_len, int max_new, int eot, float inv_temp, uint32_t seed, int n_ended_at_start,
                        hipStream_t s) {
    hipLaunchKernelGGL(sample_init_kernel, dim3(p.n_audio * p.best_of), dim3(256), 0, s, p, prompt_len, max_new, eot, inv_temp, seed,
                       n_ended_at_start);
}

void launch_sample_step(const SampleParams& p, hipStream_t s) {
    hipLaunchKernelGGL(sample_scan_kernel, dim3(kBeamParts, p.n_audio * p.best_of), dim3(256), 0, s, p);
    hipLaunchKernelGGL(sample_step_kernel, dim3(p.n_audio * p.best_of), dim3(256), 0, s, p);
}

}  // namespace mw
