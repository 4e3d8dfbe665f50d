// Kernels of the wav2vec2 CTC model (forced alignment) that the Whisper path does not already have: the first
// convolution on the raw waveform, LayerNorm (+GELU) over fp16 rows, the group-major copy that turns the grouped
// positional convolution into plain GEMMs, and the log-softmax of the CTC logits.  Everything else of that model
// runs on the encoder's GEMM, LayerNorm and attention kernels.
// Algorithm: transformers models/wav2vec2/modeling_wav2vec2.py:275-299 (conv -> LayerNorm -> GELU), :326-379
// (positional convolution), :1742 (logits); torch.log_softmax as whisperx's align applies it to them.
#include "mw_kernels.h"

namespace mw {

namespace {

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
// GELU(x) = x * Phi(x), erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the fp16 rounding of the
// result), the negative side through 1 + erf(z) = poly * exp(-z^2) so that nothing cancels: a dozen instructions
// where erff takes about fifty.  The feature extractor applies it to 0.2 G elements per 30 s of audio.
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float q = fmaf(1.061405429f, t, -1.453152027f);
    q = fmaf(q, t, 1.421413741f);
    q = fmaf(q, t, -0.284496736f);
    q = fmaf(q, t, 0.254829592f);
    const float h = 0.5f * q * t * __expf(-z * z);       // 0.5 * (1 - erf(z))
    return x * (x >= 0.f ? 1.0f - h : h);
}

// One wave per output frame (grid-stride over frames); lane l owns channels l, l + 64, ...  The C x K0 weights
// sit in LDS, the K0 samples of a frame are the same address for every lane (one broadcast load each).
// HBM: 4 bytes per sample in, 2 * C bytes per frame out (the output dominates: 98 MB for 30 s at C = 512).
constexpr int kConv0MaxCPL = 8, kConv0MaxK = 16;
// REGW: the lane's own weights (its C / 64 channels x K0 taps) live in registers for the whole launch (the usual
// geometry, 512 channels x 10 taps, is 80 of them); otherwise they are read from LDS per frame.
template <int CPL, int K0C>
__global__ __launch_bounds__(256) void w2v_conv0_reg_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, const float* __restrict__ g,
                                                            const float* __restrict__ beta, f16* __restrict__ out, int T0, int S0) {
    constexpr int C = CPL * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float wr[CPL][K0C], bb[CPL], gg[CPL], be[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int c = lane + 64 * i;
        bb[i] = b[c]; gg[i] = g[c]; be[i] = beta[c];
#pragma unroll
        for (int j = 0; j < K0C; ++j) wr[i][j] = w[c * K0C + j];
    }
    const int stride = gridDim.x * 4;
    for (int t = blockIdx.x * 4 + wave; t < T0; t += stride) {
        float xv[K0C];
#pragma unroll
        for (int j = 0; j < K0C; ++j) xv[j] = x[(long)t * S0 + j];
        float y[CPL];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            float a = bb[i];
#pragma unroll
            for (int j = 0; j < K0C; ++j) a += wr[i][j] * xv[j];          // same order of additions as the LDS form
            y[i] = a;
            s += a;
        }
        const float mean = wave_sum(s) / C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < CPL; ++i) { y[i] -= mean; q += y[i] * y[i]; }
        const float rstd = rsqrtf(wave_sum(q) / C + 1e-5f);
#pragma unroll
        for (int i = 0; i < CPL; ++i) out[(long)t * C + lane + 64 * i] = (f16)gelu_erf(y[i] * rstd * gg[i] + be[i]);
    }
}

__global__ __launch_bounds__(256) void w2v_conv0_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, const float* __restrict__ g,
                                                        const float* __restrict__ beta, f16* __restrict__ out, int T0, int C,
                                                        int K0, int S0) {
    extern __shared__ float wl[];                                  // [C][K0]
    for (int i = threadIdx.x; i < C * K0; i += blockDim.x) wl[i] = w[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cpl = C >> 6;
    float bb[kConv0MaxCPL], gg[kConv0MaxCPL], be[kConv0MaxCPL];
#pragma unroll
    for (int i = 0; i < kConv0MaxCPL; ++i) {
        const int c = min(lane + 64 * i, C - 1);
        bb[i] = b[c]; gg[i] = g[c]; be[i] = beta[c];
    }
    const int stride = gridDim.x * 4;
    for (int t = blockIdx.x * 4 + wave; t < T0; t += stride) {
        float xv[kConv0MaxK];
#pragma unroll
        for (int j = 0; j < kConv0MaxK; ++j) xv[j] = j < K0 ? x[(long)t * S0 + j] : 0.f;
        float y[kConv0MaxCPL];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < kConv0MaxCPL; ++i) {
            y[i] = 0.f;
            if (i < cpl) {
                const float* wr = wl + (lane + 64 * i) * K0;
                float a = bb[i];
                for (int j = 0; j < K0; ++j) a += wr[j] * xv[j];
                y[i] = a;
                s += a;
            }
        }
        const float mean = wave_sum(s) / C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < kConv0MaxCPL; ++i)
            if (i < cpl) { y[i] -= mean; q += y[i] * y[i]; }
        const float rstd = rsqrtf(wave_sum(q) / C + 1e-5f);
#pragma unroll
        for (int i = 0; i < kConv0MaxCPL; ++i)
            if (i < cpl) out[(long)t * C + lane + 64 * i] = (f16)gelu_erf(y[i] * rstd * gg[i] + be[i]);
    }
}

// One wave per row; a lane owns the 8-half pieces lane, lane + 64, ... of the row (statistics in fp32).  NV: pieces per
// lane, D <= 512 * NV (the feature extractor's 512 channels: one piece, nothing to skip).
constexpr int kLn16MaxV = 8;       // D <= 4096
template <int NV>
__global__ __launch_bounds__(256) void ln16_kernel(f16* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                   int rows, int D, float eps, int gelu) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    f16* xr = x + (long)row * D;
    float v[NV][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < D) {
            const half8 h = *reinterpret_cast<const half8*>(xr + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[i][e] = (float)h[e]; s += v[i][e]; }
        }
    }
    const float mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if ((lane + 64 * i) * 8 < D) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[i][e] -= mean; q += v[i][e] * v[i][e]; }
        }
    const float rstd = rsqrtf(wave_sum(q) / D + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < D) {
            const float4 g0 = *reinterpret_cast<const float4*>(g + c), g1 = *reinterpret_cast<const float4*>(g + c + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(b + c), b1 = *reinterpret_cast<const float4*>(b + c + 4);
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            half8 h;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float o = v[i][e] * rstd * gg[e] + bb[e];
                if (gelu) o = gelu_erf(o);
                h[e] = (f16)o;
            }
            *reinterpret_cast<half8*>(xr + c) = h;
        }
    }
}

// pack[g][r][c] = x[r - K/2][g * W + c] for 0 <= r - K/2 < T, else 0; W = D / G.  One thread per 8 output halfs.
__global__ __launch_bounds__(256) void w2v_pos_pack_kernel(const float* __restrict__ x, f16* __restrict__ pack, int T, int D, int G,
                                                           int K) {
    const int W = D / G, R = T + K;
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i >= (long)G * R * W) return;
    const int c = (int)(i % W);
    const long gr = i / W;
    const int r = (int)(gr % R), g = (int)(gr / R);
    const int t = r - K / 2;
    half8 h = {0, 0, 0, 0, 0, 0, 0, 0};
    if (t >= 0 && t < T) {
        const float* src = x + (long)t * D + g * W + c;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = (f16)src[e];
    }
    *reinterpret_cast<half8*>(pack + i) = h;
}

// One workgroup per row.
__global__ __launch_bounds__(256) void log_softmax_rows_kernel(float* __restrict__ z, int ld, int V) {
    __shared__ float red[4];
    float* zr = z + (long)blockIdx.x * ld;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float m = -3.0e38f;
    for (int i = tid; i < V; i += 256) m = fmaxf(m, zr[i]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int i = tid; i < V; i += 256) s += expf(zr[i] - m);
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float lse = m + logf(red[0] + red[1] + red[2] + red[3]);
    for (int i = tid; i < V; i += 256) zr[i] -= lse;
}

// One workgroup per row: the best non-blank entry (for wildcard tokens), then the gather.
__global__ __launch_bounds__(256) void ctc_gather_kernel(const float* __restrict__ z, int ld, int V, const int32_t* __restrict__ tokens, int J,
                                                         int blank, float* __restrict__ out) {
    __shared__ float red[4];
    const float* zr = z + (long)blockIdx.x * ld;
    float* o = out + (long)blockIdx.x * (J + 1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float m = -3.0e38f;
    for (int i = tid; i < V; i += 256) if (i != blank) m = fmaxf(m, zr[i]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (tid == 0) o[0] = zr[blank];
    for (int j = tid; j < J; j += 256) {
        const int t = tokens[j];
        o[1 + j] = t < 0 ? m : zr[t];
    }
}

}  // namespace

void launch_ctc_gather(const float* z, int rows, int ld, int V, const int32_t* tokens, int J, int blank, float* out, hipStream_t s) {
    hipLaunchKernelGGL(ctc_gather_kernel, dim3(rows), dim3(256), 0, s, z, ld, V, tokens, J, blank, out);
}

void launch_w2v_conv0(const float* x, long n, const float* w, const float* b, const float* g, const float* beta, f16* out, int T0, int C,
                      int K0, int S0, hipStream_t s) {
    (void)n;
    const int grid = T0 < 4096 ? (T0 + 3) / 4 : 1024;
    const dim3 gd(grid > 0 ? grid : 1);
    if (C == 512 && K0 == 10)           // wav2vec2's feature extractor: the lane's 8 x 10 weights stay in registers
        hipLaunchKernelGGL((w2v_conv0_reg_kernel<8, 10>), gd, dim3(256), 0, s, x, w, b, g, beta, out, T0, S0);
    else
        hipLaunchKernelGGL(w2v_conv0_kernel, gd, dim3(256), (size_t)C * K0 * 4, s, x, w, b, g, beta, out, T0, C, K0, S0);
}

void launch_ln16(f16* x, const float* g, const float* b, int rows, int D, float eps, int gelu, hipStream_t s) {
    const dim3 grid((rows + 3) / 4);
    if (D <= 512) hipLaunchKernelGGL(ln16_kernel<1>, grid, dim3(256), 0, s, x, g, b, rows, D, eps, gelu);
    else if (D <= 1024) hipLaunchKernelGGL(ln16_kernel<2>, grid, dim3(256), 0, s, x, g, b, rows, D, eps, gelu);
    else hipLaunchKernelGGL(ln16_kernel<kLn16MaxV>, grid, dim3(256), 0, s, x, g, b, rows, D, eps, gelu);
}

void launch_w2v_pos_pack(const float* x, f16* pack, int T, int D, int G, int K, hipStream_t s) {
    const long n8 = (long)G * (T + K) * (D / G) / 8;
    hipLaunchKernelGGL(w2v_pos_pack_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, x, pack, T, D, G, K);
}

void launch_log_softmax_rows(float* z, int rows, int ld, int V, hipStream_t s) {
    hipLaunchKernelGGL(log_softmax_rows_kernel, dim3(rows), dim3(256), 0, s, z, ld, V);
}

}  // namespace mw
