// Decode-step kernels for gfx950: every launch of one greedy step, sized for a batch of at most
// 32 sequences.  All of them read the current position from device memory (DecState), so one
// captured hipGraph of a step is replayed for every token.
//
// Roofline: each kernel streams its weights (or the K/V cache) exactly once from HBM and does
// O(batch) work per byte, so all are HBM/latency-bound; the matrix products still go through
// v_mfma_f32_16x16x32_f16 (batch on the MFMA column index) because 16 FMAs per weight element on
// the VALU would cost as much as the stream itself.
#include "mw_kernels.h"

namespace mw {

namespace {

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

// ------------------------------------------------------------------ skinny GEMM (weights streamed once)
// out[b][n] = epi( sum_k X[b][k] * W[n][k] ),  b < B <= 16*MT.
// Workgroup = 8 waves that split K; it owns NV (8 or 16) weight rows, so narrow outputs (N = d_model)
// still spread over >= 160 workgroups.  Weight fragments go straight from global memory to the
// MFMA A operand (no LDS: nothing is shared between waves); the activation fragment is the B
// operand, read from L2, optionally through a fused LayerNorm (statistics computed per workgroup).
// Partial sums of the 8 waves are combined through LDS in a fixed order: results are bit-for-bit
// reproducible (no atomics).
constexpr int GV_WAVES = 8;
constexpr int GV_CH = 10;       // k-steps (of 32) whose loads are issued together

template <int NV, int MT>
__global__ __launch_bounds__(512) void gemv_kernel(GemvParams p) {
    __shared__ float red[GV_WAVES][MT][64][4];
    __shared__ float stat[16 * MT][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int n0 = blockIdx.x * NV;
    int wrow = n0 + (NV == 16 ? fr : (fr & 7));
    if (wrow >= p.N) wrow = p.N - 1;
    const f16* wp = p.W + (long)wrow * p.K + fg * 8;

    const int nks = p.K / 32;
    const int kpw = (nks + GV_WAVES - 1) / GV_WAVES;
    const int ks0 = wave * kpw;
    const int ks1 = min(ks0 + kpw, nks);

    // issue the first group of weight loads before anything that depends on the activations
    uint4 wreg[GV_CH];
#pragma unroll
    for (int c = 0; c < GV_CH; ++c)
        wreg[c] = (ks0 + c < ks1) ? *reinterpret_cast<const uint4*>(wp + (ks0 + c) * 32) : make_uint4(0, 0, 0, 0);

    const bool ln = p.ln_g != nullptr;
    if (ln) {           // LayerNorm statistics of the B input rows (fp32, eps 1e-5)
        for (int r = wave; r < p.B; r += GV_WAVES) {
            const float* xr = p.x32 + (long)r * p.K;
            float s = 0.f;
            for (int c = lane * 4; c < p.K; c += 256) {
                const float4 v = *reinterpret_cast<const float4*>(xr + c);
                s += v.x + v.y + v.z + v.w;
            }
            const float mean = wave_sum(s) / p.K;
            float q = 0.f;
            for (int c = lane * 4; c < p.K; c += 256) {
                const float4 v = *reinterpret_cast<const float4*>(xr + c);
                const float a = v.x - mean, b = v.y - mean, cc = v.z - mean, d = v.w - mean;
                q += a * a + b * b + cc * cc + d * d;
            }
            const float rstd = rsqrtf(wave_sum(q) / p.K + 1e-5f);
            if (lane == 0) { stat[r][0] = mean; stat[r][1] = rstd; }
        }
        __syncthreads();
    }

    float4v acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = float4v{0, 0, 0, 0};

    for (int kb = ks0; kb < ks1; kb += GV_CH) {
        if (kb != ks0) {
#pragma unroll
            for (int c = 0; c < GV_CH; ++c)
                wreg[c] = (kb + c < ks1) ? *reinterpret_cast<const uint4*>(wp + (kb + c) * 32) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < GV_CH; ++c) {
            if (kb + c < ks1) {
                const int k = (kb + c) * 32 + fg * 8;
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const int m = t * 16 + fr;
                    half8 xf = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (m < p.B) {
                        if (ln) {
                            const float* xr = p.x32 + (long)m * p.K + k;
                            const float4 a = *reinterpret_cast<const float4*>(xr), b = *reinterpret_cast<const float4*>(xr + 4);
                            const float4 g0 = *reinterpret_cast<const float4*>(p.ln_g + k), g1 = *reinterpret_cast<const float4*>(p.ln_g + k + 4);
                            const float4 b0 = *reinterpret_cast<const float4*>(p.ln_b + k), b1 = *reinterpret_cast<const float4*>(p.ln_b + k + 4);
                            const float mu = stat[m][0], rs = stat[m][1];
                            xf = half8{(f16)((a.x - mu) * rs * g0.x + b0.x), (f16)((a.y - mu) * rs * g0.y + b0.y),
                                       (f16)((a.z - mu) * rs * g0.z + b0.z), (f16)((a.w - mu) * rs * g0.w + b0.w),
                                       (f16)((b.x - mu) * rs * g1.x + b1.x), (f16)((b.y - mu) * rs * g1.y + b1.y),
                                       (f16)((b.z - mu) * rs * g1.z + b1.z), (f16)((b.w - mu) * rs * g1.w + b1.w)};
                        } else {
                            xf = *reinterpret_cast<const half8*>(p.xin16 + (long)m * p.K + k);
                        }
                    }
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<half8*>(&wreg[c]), xf, acc[t], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) *reinterpret_cast<float4v*>(&red[wave][t][lane][0]) = acc[t];
    __syncthreads();
    if (wave != 0) return;

    // ---- epilogue: lane holds out[m][n..n+3], n = n0 + 4*fg (rows 8..15 are duplicates when NV == 8)
    const int nl = 4 * fg;
    if (nl >= NV) return;
    const int n = n0 + nl;
    if (n >= p.N) return;                          // N is a multiple of 4 for every skinny GEMM
    const int pos = p.st ? p.st->pos : 0;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = t * 16 + fr;
        if (m >= p.B) continue;
        float4v v = {0, 0, 0, 0};
#pragma unroll
        for (int w = 0; w < GV_WAVES; ++w) v += *reinterpret_cast<float4v*>(&red[w][t][lane][0]);
        if (p.bias) {
            const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
            v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
        }
        if (n < p.scale_ncols) v *= p.scale;
        if (p.gelu) { v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]); }
        if (p.out_mode == GV_RESID) {
            float4* c = reinterpret_cast<float4*>(p.out32 + (long)m * p.N + n);
            float4 o = *c;
            o.x += v[0]; o.y += v[1]; o.z += v[2]; o.w += v[3];
            *c = o;
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

// ------------------------------------------------------------------ logits + partial argmax
// logits[b][n] = xn[b] . E[n] over the tied token embedding (modeling_whisper.py:1080); one wave
// per 16 vocabulary rows with the full K, so no cross-wave reduction.  Each wave also reduces its
// 16 columns to one (max, lowest index) candidate per sequence under the suppress mask that
// applies to this position; dec_finalize picks the winner.
template <int MT>
__global__ __launch_bounds__(256) void logits_kernel(GemvParams p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int tile = blockIdx.x * 4 + wave;
    const int n0 = tile * 16;
    if (n0 >= p.N) return;
    int wrow = n0 + fr;
    if (wrow >= p.N) wrow = p.N - 1;
    const f16* wp = p.W + (long)wrow * p.K + fg * 8;
    float4v acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = float4v{0, 0, 0, 0};
    const int nks = p.K / 32;
    for (int kb = 0; kb < nks; kb += 8) {
        uint4 wreg[8];
#pragma unroll
        for (int c = 0; c < 8; ++c)
            wreg[c] = (kb + c < nks) ? *reinterpret_cast<const uint4*>(wp + (kb + c) * 32) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (kb + c < nks) {
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const int m = t * 16 + fr;
                    half8 xf = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (m < p.B) xf = *reinterpret_cast<const half8*>(p.xin16 + (long)m * p.K + (kb + c) * 32 + fg * 8);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<half8*>(&wreg[c]), xf, acc[t], 0, 0, 0);
                }
            }
        }
    }
    const int sample = p.st->pos + 1 - p.st->prompt_len;       // index of the token chosen from these logits
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
        if (fg == 0 && m < p.B) {
            p.part_val[(long)m * p.n_parts + tile] = best;
            p.part_idx[(long)m * p.n_parts + tile] = bidx;
        }
    }
}

// ------------------------------------------------------------------ single-query attention over a cache
// One workgroup per (sequence, head): scores over n keys ([n][64] fp16, contiguous), softmax, P.V.
// Each 128-byte key/value row is fetched by 8 adjacent lanes (16 B each), 8 rows per wave
// instruction, so every wave instruction reads 1 KiB of contiguous memory.
// Self-attention (modeling_whisper.py:473-478): n = position + 1, every cached key is visible to the
// newest query.  Cross-attention (:486-492): n = 1500 encoder frames, K/V cached once (:313-326).
template <int MAXT>
__global__ __launch_bounds__(256) void dec_attn_kernel(const f16* __restrict__ q, const f16* __restrict__ kc,
                                                       const f16* __restrict__ vc, f16* __restrict__ out, int H,
                                                       int tstride, int nfixed, const DecState* __restrict__ st) {
    __shared__ float sc[MAXT];
    __shared__ float part[4][64];
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bh = blockIdx.x;
    const int b = bh / H, h = bh - b * H;
    const int n = st ? st->pos + 1 : nfixed;
    const int rg = tid >> 3, dc = tid & 7;
    const f16* kb = kc + (long)bh * tstride * 64 + dc * 8;
    const f16* vb = vc + (long)bh * tstride * 64 + dc * 8;

    float qv[8];
    {
        const half8 qh = *reinterpret_cast<const half8*>(q + ((long)b * H + h) * 64 + dc * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) qv[j] = (float)qh[j];
    }
    // ---- scores
    for (int t0 = rg; t0 < n; t0 += 128) {
        half8 kk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + 32 * u;
            kk[u] = t < n ? *reinterpret_cast<const half8*>(kb + (long)t * 64) : half8{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float d = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) d += qv[j] * (float)kk[u][j];
            d += __shfl_xor(d, 1);
            d += __shfl_xor(d, 2);
            d += __shfl_xor(d, 4);
            const int t = t0 + 32 * u;
            if (dc == 0 && t < n) sc[t] = d;
        }
    }
    __syncthreads();
    // ---- softmax over sc[0..n)
    float mx = -3.0e38f;
    for (int t = tid; t < n; t += 256) mx = fmaxf(mx, sc[t]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int t = tid; t < n; t += 256) {
        const float e = __expf(sc[t] - mx);
        sc[t] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    sum = red[4] + red[5] + red[6] + red[7];
    // ---- P.V
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int t0 = rg; t0 < n; t0 += 128) {
        half8 vv[4];
        float pp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + 32 * u;
            const bool ok = t < n;
            vv[u] = ok ? *reinterpret_cast<const half8*>(vb + (long)t * 64) : half8{0, 0, 0, 0, 0, 0, 0, 0};
            pp[u] = ok ? sc[t] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += pp[u] * (float)vv[u][j];
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
    if (tid < 64) {
        const float o = (part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid]) / sum;
        out[((long)b * H + h) * 64 + tid] = (f16)o;
    }
}

// ------------------------------------------------------------------ token bookkeeping
// dec_init: position 0, first forced token embedded.  dec_finalize: choose the next token
// (forced while inside the prompt, else the arg-max candidate), apply the end-of-text rule,
// record it, embed it (token + learned position, modeling_whisper.py:737-762) and advance.
// One workgroup: wave w serves sequences w, w+16.
__global__ __launch_bounds__(1024) void dec_init_kernel(FinalizeParams p, int B, int prompt_len, int max_new, int eot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int m = wave; m < B; m += 16) {
        const int tok = p.forced[(long)m * p.n_forced];
        for (int c = lane; c < p.D; c += 64) p.x[(long)m * p.D + c] = (float)p.tok_emb[(long)tok * p.D + c] + (float)p.pos_emb[c];
        for (int i = lane; i < max_new; i += 64) p.out_ids[(long)m * max_new + i] = eot;
        if (lane == 0) { p.out_len[m] = max_new; p.finished[m] = 0; }
    }
    if (threadIdx.x == 0) {
        p.st->pos = 0; p.st->prompt_len = prompt_len; p.st->max_new = max_new; p.st->eot = eot;
        p.st->n_finished = 0; p.st->batch = B;
    }
}

__global__ __launch_bounds__(1024) void dec_finalize_kernel(FinalizeParams p, int B) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pos = p.st->pos, prompt_len = p.st->prompt_len, max_new = p.st->max_new, eot = p.st->eot;
    const int next = pos + 1;
    for (int m = wave; m < B; m += 16) {
        int tok;
        if (next < prompt_len) {
            tok = p.forced[(long)m * p.n_forced + next];
        } else {
            float best = -INFINITY;
            int bidx = 0x7fffffff;
            for (int i = lane; i < p.n_parts; i += 64) {
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
            tok = (bidx >= 0 && bidx < p.vocab) ? bidx : eot;      // all candidates masked or NaN
            const int sample = next - prompt_len;
            if (p.finished[m]) {
                tok = eot;
            } else if (tok == eot && lane == 0 && sample < max_new) {
                p.out_len[m] = sample;
                atomicAdd(&p.st->n_finished, 1);
            }
            const bool fin_now = tok == eot;
            if (lane == 0 && sample < max_new) p.out_ids[(long)m * max_new + sample] = tok;
            __builtin_amdgcn_wave_barrier();
            if (lane == 0 && fin_now) p.finished[m] = 1;
        }
        if (next < p.n_text_ctx) {
            for (int c = lane; c < p.D; c += 64)
                p.x[(long)m * p.D + c] = (float)p.tok_emb[(long)tok * p.D + c] + (float)p.pos_emb[(long)next * p.D + c];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) p.st->pos = next;
}

}  // namespace

int gemv_logits_parts(int N) { return (N + 15) / 16; }

void launch_gemv(const GemvParams& p, hipStream_t s) {
    const int mt = p.B <= 16 ? 1 : 2;
    if (p.out_mode == GV_LOGITS) {
        const int tiles = gemv_logits_parts(p.N);
        dim3 grid((tiles + 3) / 4);
        if (mt == 1) hipLaunchKernelGGL(logits_kernel<1>, grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL(logits_kernel<2>, grid, dim3(256), 0, s, p);
        return;
    }
    // narrow outputs: 8 rows per workgroup so that at least ~160 workgroups stream weights
    const bool narrow = p.N <= 2048;
    if (narrow) {
        dim3 grid((p.N + 7) / 8);
        if (mt == 1) hipLaunchKernelGGL((gemv_kernel<8, 1>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((gemv_kernel<8, 2>), grid, dim3(512), 0, s, p);
    } else {
        dim3 grid((p.N + 15) / 16);
        if (mt == 1) hipLaunchKernelGGL((gemv_kernel<16, 1>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((gemv_kernel<16, 2>), grid, dim3(512), 0, s, p);
    }
}

void launch_dec_self_attn(const f16* q, const f16* kc, const f16* vc, f16* out, int B, int H, int Tctx,
                          const DecState* st, hipStream_t s) {
    hipLaunchKernelGGL(dec_attn_kernel<448>, dim3(B * H), dim3(256), 0, s, q, kc, vc, out, H, Tctx, 0, st);
}

void launch_dec_cross_attn(const f16* q, const f16* kc, const f16* vc, f16* out, int B, int H, int T, hipStream_t s) {
    hipLaunchKernelGGL(dec_attn_kernel<1504>, dim3(B * H), dim3(256), 0, s, q, kc, vc, out, H, T, T, (const DecState*)nullptr);
}

void launch_dec_init(const FinalizeParams& p, int B, int prompt_len, int max_new, int eot, hipStream_t s) {
    hipLaunchKernelGGL(dec_init_kernel, dim3(1), dim3(1024), 0, s, p, B, prompt_len, max_new, eot);
}

void launch_dec_finalize(const FinalizeParams& p, int B, hipStream_t s) {
    hipLaunchKernelGGL(dec_finalize_kernel, dim3(1), dim3(1024), 0, s, p, B);
}

}  // namespace mw
