// fp16 x fp16 -> fp32 MFMA GEMM with fused epilogues, for the encoder and the cross-attention
// K/V projection (M = batch*1500 rows, K and N in the thousands: MFMA-bound).
//
//   C[m][n] = epi( sum_k A[m][k] * W[n][k] )      A: activations, W: nn.Linear weight [out][in]
//
// Tile (MI*32)x128x64, MI = 4 in use (128 rows), 256 threads = 4 waves in a 2x2 grid, each wave
// (MI*16)x64 as MI x 4 tiles of v_mfma_f32_16x16x32_f16; two workgroups per CU.  Both operands are K-contiguous, so every fragment is one
// ds_read_b128 from an XOR-swizzled LDS image (16-byte chunk index ^ ((row>>1)&7): the 16 rows a
// ds_read_b128 lane group touches land on 16 distinct slots of the 256-byte bank row).
// Global->LDS staging: when K is a multiple of 64 the next K-tile is fetched with
// global_load_lds_dwordx4 (no VGPR round trip, no ds_write) while the MFMAs run on the current one;
// a wave instruction writes 1 KiB of LDS linearly (8 tile rows), so the XOR swizzle is applied to
// the per-lane SOURCE address and again on the fragment reads.  Rows past M or N load a clamped,
// valid row and are never stored.  Other K go through registers with zero fill.
// The weight fragment is the MFMA's A operand and the activation fragment its B operand, so each
// lane ends up with 4 consecutive output columns of one row: bias loads and stores are vector.
//
// A and C rows may be "batched with gaps": row m lives at (m / rows_per_batch) * batch_stride +
// (m % rows_per_batch) * ld.  With ld smaller than K consecutive rows overlap, which is how the
// k=3 convolutions of the encoder stem run as plain GEMMs over a time-major padded buffer.
//
// F8 form (both operands OCP e4m3, one byte per element): the staging code is the f16 code run on
// "pseudo halfs" (the launcher halves K and the leading dimensions, so a 128-byte tile row is 128
// fp8 elements instead of 64 halfs); only the MFMA and the epilogue differ.  A lane's two 16-byte
// fragments of a tile row go to ONE v_mfma_f32_16x16x128_f8f6f4 (the gfx950 form that runs fp8 at
// twice the f16 rate; zero scale operands select its unscaled encoding): which k a fragment byte
// stands for is irrelevant as long as both operands are read through the same (row, chunk) map,
// which they are.  The accumulator then holds the integer-exact sum of e4m3 products; the epilogue
// multiplies by the activation row's scale and the weight row's scale before the bias.
#include "mw_kernels.h"

#include <stdlib.h>
#include <mutex>
#include <type_traits>

namespace mw {

namespace {

constexpr int BN = 128, BK = 64;

typedef int int4v __attribute__((ext_vector_type(4)));
typedef int int8v __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float4v mfma_f16(int4v a, int4v b, float4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}
// 128 fp8 (e4m3) products per output element: a lane passes 32 bytes of its row per operand
__device__ __forceinline__ float4v mfma_f8(int4v a0, int4v a1, int4v b0, int4v b1, float4v c) {
    const int8v a = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
    const int8v b = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
}

__device__ __forceinline__ int swz(int row, int chunk) {      // offset in halfs inside a [rows][64] tile
    return row * BK + ((chunk ^ ((row >> 1) & 7)) << 3);
}

// GELU(x) = x * Phi(x) with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the
// fp16 rounding of the result); the negative side uses 1 + erf(z) = poly * exp(-z^2) directly, so
// there is no cancellation.  12 instructions instead of the ~50 of erff: the epilogue is straight
// line code executed once per tile, and its size is what it costs.
__device__ __forceinline__ float gelu_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float q = fmaf(1.061405429f, t, -1.453152027f);
    q = fmaf(q, t, 1.421413741f);
    q = fmaf(q, t, -0.284496736f);
    q = fmaf(q, t, 0.254829592f);
    const float h = 0.5f * q * t * __expf(-z * z);       // 0.5 * (1 - erf(z))
    return x * (x >= 0.f ? 1.0f - h : h);
}

// Fused epilogue: lane holds C[m][n..n+3], m = m0 + wm*MI*16 + j*16 + fr, n = n0 + wn*64 + i*16 + 4*fg.
// Output mode and GELU are compile-time here (one uniform switch at the call site): the unrolled
// MI x 4 body must stay small, or instruction fetch dominates the tile.
// SC: 0 = no scales; 1 = columns scaled by w_scale[n] (fp16 product over unscaled weight codes);
// 2 = a_scale[m] * w_scale[n] (fp8 product).
template <int MI, int OUT, bool GELU, int SC = 0>
__device__ __forceinline__ void gemm_epilogue_t(const GemmParams& p, float4v (&acc)[4][MI], int m0, int n0, int wm, int wn,
                                                int fr, int fg) {
    float4 bv[4], wsc[4];
    float sc[4];
    bool n_ok[4];
    long col[4];            // column part of the output offset
    int sec[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + wn * 64 + i * 16 + 4 * fg;
        n_ok[i] = n < p.N;                              // N is a multiple of 4
        const int nc = n_ok[i] ? n : 0;
        bv[i] = p.bias ? *reinterpret_cast<const float4*>(p.bias + nc) : make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (SC > 0) wsc[i] = *reinterpret_cast<const float4*>(p.w_scale + nc);
        sc[i] = n < p.scale_ncols ? p.scale : 1.0f;
        if constexpr (OUT == OUT_F16 || OUT == OUT_RESID_F32 || OUT == OUT_KV_SPLIT8) {
            col[i] = nc; sec[i] = 0;
        } else {
            sec[i] = nc / p.D;
            const int nn = nc - sec[i] * p.D;
            const int hh = nn >> 6, e = nn & 63;
            if (OUT == OUT_QKV_SPLIT && sec[i] == 2) col[i] = ((long)hh * 64 + e) * p.Tpad;       // V^T [B][H][64][Tpad]
            else col[i] = (long)hh * p.T * 64 + e;                                                // [B][H][T][64]
        }
    }
    if constexpr (OUT == OUT_KV_SPLIT8) {
        // The wave's 64 columns n0 + wn*64 .. +63 are one head of K (first D columns) or of V: the row of 64 gets
        // one scale (max|.| / 448 over the 4 subtiles of this lane and the 4 lanes that share the row).
        const int nb = n0 + wn * 64;
        const bool head_ok = nb < p.N;                          // N = 2D is a multiple of 64
        const int nc = head_ok ? nb : 0;
        const int secw = nc / p.D, hh = (nc - secw * p.D) >> 6;
        uint8_t* cb = secw == 0 ? p.k8 : p.v8;
        float* sb = secw == 0 ? p.ks : p.vs;
#pragma unroll
        for (int j = 0; j < MI; ++j) {
            const int m = m0 + wm * MI * 16 + j * 16 + fr;
            const bool m_ok = m < p.M;
            const float asc = p.a_scale[m_ok ? m : 0];
            float4v v[4];
            float mx = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = acc[i][j];
                v[i][0] = v[i][0] * (asc * wsc[i].x) + bv[i].x; v[i][1] = v[i][1] * (asc * wsc[i].y) + bv[i].y;
                v[i][2] = v[i][2] * (asc * wsc[i].z) + bv[i].z; v[i][3] = v[i][3] * (asc * wsc[i].w) + bv[i].w;
                mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[i][0]), fabsf(v[i][1]))), fmaxf(fabsf(v[i][2]), fabsf(v[i][3])));
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float scl = mx > 0.f ? mx / 448.f : 1.f;
            if (m_ok && head_ok) {
                const int bb = m / p.T, tt = m - bb * p.T;
                const long rowi = ((long)bb * p.H + hh) * p.T + tt;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<unsigned*>(cb + rowi * 64 + i * 16 + 4 * fg) =
                        pack4_e4m3(v[i][0] / scl, v[i][1] / scl, v[i][2] / scl, v[i][3] / scl);
                if (fg == 0) sb[rowi] = scl;
            }
        }
        return;
    }
    const bool flat = p.c_rows_per_batch >= p.M;
#pragma unroll
    for (int j = 0; j < MI; ++j) {
        const int m = m0 + wm * MI * 16 + j * 16 + fr;
        const bool m_ok = m < p.M;
        float asc = 1.f;
        if constexpr (SC == 2) asc = p.a_scale[m_ok ? m : 0];
        long row, rowT = 0;     // row part of the output offset (rowT: the same for the transposed V)
        if constexpr (OUT == OUT_F16 || OUT == OUT_RESID_F32) {
            row = flat ? (long)m * p.ldc : (long)(m / p.c_rows_per_batch) * p.c_batch_stride + (long)(m % p.c_rows_per_batch) * p.ldc;
        } else {
            const int bb = m / p.T, tt = m - bb * p.T;
            row = ((long)bb * p.H * p.T + tt) * 64;
            rowT = (long)bb * p.H * 64 * p.Tpad + tt;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (!(m_ok && n_ok[i])) continue;
            float4v v = acc[i][j];
            if constexpr (SC > 0) { v[0] *= asc * wsc[i].x; v[1] *= asc * wsc[i].y; v[2] *= asc * wsc[i].z; v[3] *= asc * wsc[i].w; }
            v[0] = (v[0] + bv[i].x) * sc[i]; v[1] = (v[1] + bv[i].y) * sc[i];
            v[2] = (v[2] + bv[i].z) * sc[i]; v[3] = (v[3] + bv[i].w) * sc[i];
            if constexpr (GELU) { v[0] = gelu_fast(v[0]); v[1] = gelu_fast(v[1]); v[2] = gelu_fast(v[2]); v[3] = gelu_fast(v[3]); }
            if constexpr (OUT == OUT_RESID_F32) {
                float* c = reinterpret_cast<float*>(p.C) + row + col[i];
                float4 o = *reinterpret_cast<float4*>(c);
                o.x += v[0]; o.y += v[1]; o.z += v[2]; o.w += v[3];
                *reinterpret_cast<float4*>(c) = o;
            } else {
                const half4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                if constexpr (OUT == OUT_F16) {
                    *reinterpret_cast<half4*>(reinterpret_cast<f16*>(p.C) + row + col[i]) = h;
                } else if (OUT == OUT_QKV_SPLIT && sec[i] == 2) {
                    f16* c = p.v + rowT + col[i];
                    c[0] = h[0]; c[p.Tpad] = h[1]; c[2L * p.Tpad] = h[2]; c[3L * p.Tpad] = h[3];
                } else {
                    f16* base = (OUT == OUT_QKV_SPLIT) ? (sec[i] == 0 ? p.q : p.k) : (sec[i] == 0 ? p.k : p.v);
                    *reinterpret_cast<half4*>(base + row + col[i]) = h;
                }
            }
        }
    }
}

template <int MI, bool F8 = false>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, float4v (&acc)[4][MI], int m0, int n0, int wm, int wn, int fr,
                                              int fg) {
    if constexpr (F8) {         // the three products that run in fp8: encoder QKV, MLP in (GELU), cross K/V
        switch (p.out_mode) {
            case OUT_F16: gemm_epilogue_t<MI, OUT_F16, true, 2>(p, acc, m0, n0, wm, wn, fr, fg); break;
            case OUT_QKV_SPLIT: gemm_epilogue_t<MI, OUT_QKV_SPLIT, false, 2>(p, acc, m0, n0, wm, wn, fr, fg); break;
            case OUT_KV_SPLIT8: gemm_epilogue_t<MI, OUT_KV_SPLIT8, false, 2>(p, acc, m0, n0, wm, wn, fr, fg); break;
            default: gemm_epilogue_t<MI, OUT_KV_SPLIT, false, 2>(p, acc, m0, n0, wm, wn, fr, fg); break;
        }
        return;
    }
    if (p.w_scale) {            // fp16 product over fp8 weight codes expanded to fp16: residual projections, prompt prefill
        if (p.out_mode == OUT_RESID_F32) gemm_epilogue_t<MI, OUT_RESID_F32, false, 1>(p, acc, m0, n0, wm, wn, fr, fg);
        else if (p.gelu) gemm_epilogue_t<MI, OUT_F16, true, 1>(p, acc, m0, n0, wm, wn, fr, fg);
        else gemm_epilogue_t<MI, OUT_F16, false, 1>(p, acc, m0, n0, wm, wn, fr, fg);
        return;
    }
    switch (p.out_mode) {
        case OUT_F16:
            if (p.gelu) gemm_epilogue_t<MI, OUT_F16, true>(p, acc, m0, n0, wm, wn, fr, fg);
            else gemm_epilogue_t<MI, OUT_F16, false>(p, acc, m0, n0, wm, wn, fr, fg);
            break;
        case OUT_RESID_F32:
            if (p.gelu) gemm_epilogue_t<MI, OUT_RESID_F32, true>(p, acc, m0, n0, wm, wn, fr, fg);
            else gemm_epilogue_t<MI, OUT_RESID_F32, false>(p, acc, m0, n0, wm, wn, fr, fg);
            break;
        case OUT_QKV_SPLIT: gemm_epilogue_t<MI, OUT_QKV_SPLIT, false>(p, acc, m0, n0, wm, wn, fr, fg); break;
        default: gemm_epilogue_t<MI, OUT_KV_SPLIT, false>(p, acc, m0, n0, wm, wn, fr, fg); break;
    }
}

// ------------------------------------------------------------------ epilogues of the 256x256 fp16 kernel
// That kernel files the weight rows of a wave's 64 columns in its LDS slots in a permuted order (see its DMA source
// setup), so that a lane's accumulators hold EIGHT consecutive columns of a row instead of four:
//     m = m0 + wr*128 + j*16 + fr,   n = n0 + wc*64 + 32*i2 + 8*fg + [0, 8):  +0..3 in acc[2*i2][j], +4..7 in acc[2*i2 + 1][j]
// An fp16 result is then one 16-byte store per (i2, j), 16 per lane instead of 32 of 8 bytes: the output burst of a
// tile is bound by store instructions issued, not by bytes.
template <int OUT, bool GELU, int SC>
__device__ __forceinline__ void gemm256_epilogue_perm(const GemmParams& p, float4v (&acc)[4][8], int m0, int n0, int wr, int wc,
                                                      int fr, int fg) {
    float4 bv[2][2], wsc[2][2];
    float sc[2];
    bool n_ok[2];
    long col[2];
    int sec[2];
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2) {
        const int n = n0 + wc * 64 + 32 * i2 + 8 * fg;
        n_ok[i2] = n < p.N;                              // N is a multiple of 8 (checked by the launcher)
        const int nc = n_ok[i2] ? n : 0;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            bv[i2][g] = p.bias ? *reinterpret_cast<const float4*>(p.bias + nc + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (SC > 0) wsc[i2][g] = *reinterpret_cast<const float4*>(p.w_scale + nc + 4 * g);
        }
        sc[i2] = n < p.scale_ncols ? p.scale : 1.0f;
        if constexpr (OUT == OUT_F16 || OUT == OUT_RESID_F32) {
            col[i2] = nc; sec[i2] = 0;
        } else {
            sec[i2] = nc / p.D;
            const int nn = nc - sec[i2] * p.D;
            const int hh = nn >> 6, e = nn & 63;
            if (OUT == OUT_QKV_SPLIT && sec[i2] == 2) col[i2] = ((long)hh * 64 + e) * p.Tpad;     // V^T [B][H][64][Tpad]
            else col[i2] = (long)hh * p.T * 64 + e;                                              // [B][H][T][64]
        }
    }
    const bool flat = p.c_rows_per_batch >= p.M;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int m = m0 + wr * 128 + j * 16 + fr;
        const bool m_ok = m < p.M;
        long row, rowT = 0;
        if constexpr (OUT == OUT_F16 || OUT == OUT_RESID_F32) {
            row = flat ? (long)m * p.ldc : (long)(m / p.c_rows_per_batch) * p.c_batch_stride + (long)(m % p.c_rows_per_batch) * p.ldc;
        } else {
            const int bb = m / p.T, tt = m - bb * p.T;
            row = ((long)bb * p.H * p.T + tt) * 64;
            rowT = (long)bb * p.H * 64 * p.Tpad + tt;
        }
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
            if (!(m_ok && n_ok[i2])) continue;
            float v[8];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const float4v a = acc[2 * i2 + g][j];
                const float4 b = bv[i2][g];
                float w0 = 1.f, w1 = 1.f, w2 = 1.f, w3 = 1.f;
                if constexpr (SC > 0) { w0 = wsc[i2][g].x; w1 = wsc[i2][g].y; w2 = wsc[i2][g].z; w3 = wsc[i2][g].w; }
                v[4 * g + 0] = (a[0] * w0 + b.x) * sc[i2]; v[4 * g + 1] = (a[1] * w1 + b.y) * sc[i2];
                v[4 * g + 2] = (a[2] * w2 + b.z) * sc[i2]; v[4 * g + 3] = (a[3] * w3 + b.w) * sc[i2];
            }
            if constexpr (GELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = gelu_fast(v[e]);
            }
            if constexpr (OUT == OUT_RESID_F32) {
                float* c = reinterpret_cast<float*>(p.C) + row + col[i2];
                float4 o0 = *reinterpret_cast<float4*>(c), o1 = *reinterpret_cast<float4*>(c + 4);
                o0.x += v[0]; o0.y += v[1]; o0.z += v[2]; o0.w += v[3];
                o1.x += v[4]; o1.y += v[5]; o1.z += v[6]; o1.w += v[7];
                *reinterpret_cast<float4*>(c) = o0;
                *reinterpret_cast<float4*>(c + 4) = o1;
            } else {
                const half8 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3], (f16)v[4], (f16)v[5], (f16)v[6], (f16)v[7]};
                if constexpr (OUT == OUT_F16) {
                    *reinterpret_cast<half8*>(reinterpret_cast<f16*>(p.C) + row + col[i2]) = h;
                } else if (OUT == OUT_QKV_SPLIT && sec[i2] == 2) {       // V^T from an untransposed tile: element stores
                    f16* c = p.v + rowT + col[i2];
#pragma unroll
                    for (int e = 0; e < 8; ++e) c[(long)e * p.Tpad] = h[e];
                } else {
                    f16* base = (OUT == OUT_QKV_SPLIT) ? (sec[i2] == 0 ? p.q : p.k) : (sec[i2] == 0 ? p.k : p.v);
                    *reinterpret_cast<half8*>(base + row + col[i2]) = h;
                }
            }
        }
    }
}

// A V tile of the QKV product computed with the operands' roles exchanged (the kernel feeds weight rows where the
// activations go and activation rows, permuted as above, where the weights go), so the accumulators hold the
// TRANSPOSED tile: a lane has eight consecutive frames of one output column,
//     n = n0 + wr*128 + j*16 + fr,   m = m0 + wc*64 + 32*i2 + 8*fg + [0, 8)
// and V^T [B][H][64][Tpad] is written with 8-byte stores (frames in fours never straddle a chunk: T % 4 == 0).
__device__ __forceinline__ void gemm256_epilogue_vT(const GemmParams& p, float4v (&acc)[4][8], int m0, int n0, int wr, int wc, int fr,
                                                    int fg) {
    long moff[2][2];
    bool m_ok[2][2];
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int m = m0 + wc * 64 + 32 * i2 + 8 * fg + 4 * g;
            m_ok[i2][g] = m < p.M;
            const int mc = m_ok[i2][g] ? m : 0;
            const int bb = mc / p.T, tt = mc - bb * p.T;
            moff[i2][g] = (long)bb * p.H * 64 * p.Tpad + tt;
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int n = n0 + wr * 128 + j * 16 + fr;
        if (n >= p.N) continue;
        const float b = p.bias ? p.bias[n] : 0.f;
        const int nn = n - 2 * p.D;
        f16* rowp = p.v + (long)nn * p.Tpad;                   // (head * 64 + e) * Tpad, nn = head * 64 + e
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                if (!m_ok[i2][g]) continue;
                const float4v a = acc[2 * i2 + g][j];
                const half4 h = {(f16)(a[0] + b), (f16)(a[1] + b), (f16)(a[2] + b), (f16)(a[3] + b)};
                *reinterpret_cast<half4*>(rowp + moff[i2][g]) = h;
            }
    }
}

__device__ __forceinline__ void gemm256_epilogue_f16(const GemmParams& p, float4v (&acc)[4][8], int m0, int n0, int wr, int wc, int fr,
                                                     int fg, bool v_tile) {
    if (v_tile) { gemm256_epilogue_vT(p, acc, m0, n0, wr, wc, fr, fg); return; }
    if (p.w_scale) {            // fp16 product over fp8 weight codes expanded to fp16: residual projections, prompt prefill
        if (p.out_mode == OUT_RESID_F32) gemm256_epilogue_perm<OUT_RESID_F32, false, 1>(p, acc, m0, n0, wr, wc, fr, fg);
        else if (p.gelu) gemm256_epilogue_perm<OUT_F16, true, 1>(p, acc, m0, n0, wr, wc, fr, fg);
        else gemm256_epilogue_perm<OUT_F16, false, 1>(p, acc, m0, n0, wr, wc, fr, fg);
        return;
    }
    switch (p.out_mode) {
        case OUT_F16:
            if (p.gelu) gemm256_epilogue_perm<OUT_F16, true, 0>(p, acc, m0, n0, wr, wc, fr, fg);
            else gemm256_epilogue_perm<OUT_F16, false, 0>(p, acc, m0, n0, wr, wc, fr, fg);
            break;
        case OUT_RESID_F32:
            if (p.gelu) gemm256_epilogue_perm<OUT_RESID_F32, true, 0>(p, acc, m0, n0, wr, wc, fr, fg);
            else gemm256_epilogue_perm<OUT_RESID_F32, false, 0>(p, acc, m0, n0, wr, wc, fr, fg);
            break;
        case OUT_QKV_SPLIT: gemm256_epilogue_perm<OUT_QKV_SPLIT, false, 0>(p, acc, m0, n0, wr, wc, fr, fg); break;
        default: gemm256_epilogue_perm<OUT_KV_SPLIT, false, 0>(p, acc, m0, n0, wr, wc, fr, fg); break;
    }
}

// MI = 8 (256x128 tile, 96 KiB of LDS, one workgroup per CU) is held to 256 registers: one wave per SIMD that leaves
// half of the register file, 64 KiB of LDS and 12 wave slots per CU to kernels of other streams.
template <int MI, bool GLDS, bool F8 = false>
__global__ __launch_bounds__(256, MI == 8 ? 2 : 1) void gemm_kernel(GemmParams p) {
    constexpr int BM = MI * 32;
    constexpr int STAGE = (BM + BN) * BK;                        // halfs per buffer: A tile then W tile
    extern __shared__ __attribute__((aligned(16))) f16 lds[];   // [2][STAGE]
    if (gridDim.y > 1) {                                        // grouped launch: blockIdx.y picks the product
        const long g = blockIdx.y;
        p.A += g * p.a_group_stride;
        p.W += g * p.w_group_stride;
        if (p.bias) p.bias += g * p.N;
        p.C = p.out_mode == OUT_RESID_F32 ? static_cast<void*>(reinterpret_cast<float*>(p.C) + g * p.c_group_stride)
                                          : static_cast<void*>(reinterpret_cast<f16*>(p.C) + g * p.c_group_stride);
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // Tile order.  (1) XCD-aware: blocks b and b+8 share an XCD (round-robin dispatch), so each XCD
    // gets a contiguous run of tile ids.  (2) Inside a run, ids walk GROUP_M row-tiles for one
    // column-tile before moving to the next column-tile, so the tiles an XCD runs at once form a
    // patch that shares a few A panels and W panels in its 4 MiB L2 (a row-major walk re-streams
    // all of W from the Infinity Cache for every row-tile).
    const int nt_n = (p.N + BN - 1) / BN;
    const int nt_m = (p.M + BM - 1) / BM;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * nt_n;
    const int group = bid / per_group;
    const int gm0 = group * GROUP_M;
    const int gsz = min(GROUP_M, nt_m - gm0);
    const int in_group = bid - group * per_group;
    const int m0 = (gm0 + in_group % gsz) * BM, n0 = (in_group / gsz) * BN;

    float4v acc[4][MI];   // [n subtile][m subtile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};
    const int nk = (p.K + BK - 1) / BK;
    const int fr = lane & 15, fg = lane >> 4;

    auto compute = [&](int buf) {
        const f16* at = lds + buf * STAGE;
        const f16* wt = at + BM * BK;
        if constexpr (F8) {
            int4v wf[4][2], af[MI][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[i][ks] = *reinterpret_cast<const int4v*>(&wt[swz(wn * 64 + i * 16 + fr, ks * 4 + fg)]);
#pragma unroll
                for (int j = 0; j < MI; ++j) af[j][ks] = *reinterpret_cast<const int4v*>(&at[swz(wm * MI * 16 + j * 16 + fr, ks * 4 + fg)]);
            }
#pragma unroll
            for (int j = 0; j < MI; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = mfma_f8(wf[i][0], wf[i][1], af[j][0], af[j][1], acc[i][j]);
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                half8 wf[4], af[MI];
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const half8*>(&wt[swz(wn * 64 + i * 16 + fr, ks * 4 + fg)]);
#pragma unroll
                for (int j = 0; j < MI; ++j) af[j] = *reinterpret_cast<const half8*>(&at[swz(wm * MI * 16 + j * 16 + fr, ks * 4 + fg)]);
#pragma unroll
                for (int j = 0; j < MI; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], af[j], acc[i][j], 0, 0, 0);
            }
        }
    };

    if constexpr (GLDS) {
        // one LDS-DMA instruction fills 8 tile rows (1 KiB): lane -> row += lane>>3, physical chunk = lane&7
        const f16* a_src[MI];
        const f16* w_src[4];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row = (wave * MI + i) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);            // logical chunk that belongs in this slot
            const int m = min(m0 + row, p.M - 1);
            a_src[i] = p.A + (long)(m / p.a_rows_per_batch) * p.a_batch_stride + (long)(m % p.a_rows_per_batch) * p.lda + chunk * 8;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (wave * 4 + i) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            w_src[i] = p.W + (long)min(n0 + row, p.N - 1) * p.ldw + chunk * 8;
        }
        auto stage = [&](int buf, int k0) {
            f16* at = lds + buf * STAGE;
            f16* wt = at + BM * BK;
#pragma unroll
            for (int i = 0; i < MI; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + k0),
                                                 (__attribute__((address_space(3))) void*)(&at[(wave * MI + i) * 8 * BK]), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src[i] + k0),
                                                 (__attribute__((address_space(3))) void*)(&wt[(wave * 4 + i) * 8 * BK]), 16, 0, 0);
        };
        if constexpr (MI == 8) {
            // three buffers, two K-tiles in flight while one is multiplied: a wave waits for its own DMA of tile kt
            // (all but the MI + 4 instructions of tile kt + 1 retired), the barrier makes every wave's part of it
            // visible and says that everyone is done reading tile kt - 1, whose buffer tile kt + 2 then takes
            stage(0, 0);
            if (nk > 1) stage(1, BK);
            int buf = 0;
            for (int kt = 0; kt < nk; ++kt) {
                if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (kt + 2 < nk) stage(buf >= 1 ? buf - 1 : 2, (kt + 2) * BK);
                compute(buf);
                buf = buf == 2 ? 0 : buf + 1;
            }
        } else {
        stage(0, 0);
        __syncthreads();                       // drains the LDS-DMA (vmcnt(0)) before anyone reads
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) stage((kt + 1) & 1, (kt + 1) * BK);
            compute(kt & 1);
            __syncthreads();
        }
        }
    } else {
        // staging through registers (any K that is a multiple of 8, zero fill): MI = 4 only
        static_assert(GLDS || MI == 4, "register staging is written for the 128-row tile");
        const f16* a_ptr[4];
        const f16* w_ptr[4];
        bool a_ok[4], w_ok[4];
        const int ch = tid & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (tid >> 3) + 32 * i;
            const int m = m0 + row, n = n0 + row;
            a_ok[i] = m < p.M;
            w_ok[i] = n < p.N;
            const int mm = a_ok[i] ? m : 0, nn = w_ok[i] ? n : 0;
            a_ptr[i] = p.A + (long)(mm / p.a_rows_per_batch) * p.a_batch_stride + (long)(mm % p.a_rows_per_batch) * p.lda + ch * 8;
            w_ptr[i] = p.W + (long)nn * p.ldw + ch * 8;
        }
        uint4 ra[4], rw[4];
        auto gload = [&](int k0) {
            const bool k_ok = (k0 + ch * 8) < p.K;      // K is a multiple of 8: a chunk is all in or all out
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i] = (a_ok[i] && k_ok) ? *reinterpret_cast<const uint4*>(a_ptr[i] + k0) : make_uint4(0, 0, 0, 0);
                rw[i] = (w_ok[i] && k_ok) ? *reinterpret_cast<const uint4*>(w_ptr[i] + k0) : make_uint4(0, 0, 0, 0);
            }
        };
        auto lstore = [&](int buf) {
            f16* at = lds + buf * STAGE;
            f16* wt = at + BM * BK;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = (tid >> 3) + 32 * i;
                *reinterpret_cast<uint4*>(&at[swz(row, ch)]) = ra[i];
                *reinterpret_cast<uint4*>(&wt[swz(row, ch)]) = rw[i];
            }
        };
        gload(0);
        lstore(0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) gload((kt + 1) * BK);
            compute(kt & 1);
            if (kt + 1 < nk) lstore((kt & 1) ^ 1);
            __syncthreads();
        }
    }
    gemm_epilogue<MI, F8>(p, acc, m0, n0, wm, wn, fr, fg);
}

template <int MI, bool GLDS, bool F8 = false>
void launch_t(const GemmParams& p, hipStream_t s) {
    constexpr int BM = MI * 32;
    // MW_GEMM_LDS_PAD (tuning runs): extra bytes of LDS request that nothing uses, to hold the kernel to one workgroup per CU
    static const size_t pad = getenv("MW_GEMM_LDS_PAD") ? (size_t)atoi(getenv("MW_GEMM_LDS_PAD")) : 0;
    const size_t lds = (size_t)(MI == 8 ? 3 : 2) * (BM + BN) * BK * sizeof(f16) + pad;
    static std::once_flag prepared;     // raise the dynamic-LDS limit once, whichever engine thread gets here first
    std::call_once(prepared, [lds] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<MI, GLDS, F8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    const int nt = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    hipLaunchKernelGGL((gemm_kernel<MI, GLDS, F8>), dim3(nt, p.groups > 1 ? p.groups : 1), dim3(256), lds, s, p);
}

// ------------------------------------------------------------------ 256x256x64 tile, 8 waves
// For the big encoder products (>= one 256x256 tile per CU).  One workgroup per CU: 8 waves as
// 2 (M) x 4 (N), each 128x64 of C in 32 accumulator tiles.  LDS holds two K-tiles, each as four
// 16 KiB half-tiles (A0, A1, B0, B1): half h of A is rows wr*128 + h*64 + [0,64) of both wave rows,
// half h of B is columns wc*64 + h*32 + [0,32) of all four wave columns, so that a quadrant
// (ha, hb) of every wave's output needs exactly half-tiles A[ha] and B[hb].
//
// The K loop (two phases of 32 MFMAs per K-tile, the two wave rows one barrier apart, LDS-DMA prefetch with counted
// vmcnt) is described at the `ktile` lambda.  Measured with in-kernel stamps (-DMW_GEMM_STAMPS): 1.48 us per K-tile
// of 64, where the MFMAs alone take 1.05 us at the ~1.95 GHz the chip holds under this load: a wave's read
// section (24 ds_read_b128 and 8 LDS-DMA instructions per K-tile, the latter 60-185 cycles each to issue) is as
// long as its partner's MFMA section.  Four phases of 16 MFMAs (twice the barriers) were 2 % slower.  Issuing the
// fp16 form's DMA like the fp8 form's (matrix base in scalar registers, one 32-bit lane offset per instruction) took
// 24 registers less and exactly the same time: the address form is not what a read section costs.  Timing builds
// with parts removed (results meaningless), per K-tile: MFMAs and barriers alone 0.95-1.0 us, + LDS-DMA 1.16,
// + LDS reads instead 1.36-1.48, everything 1.5-1.6; SQ_LDS_BANK_CONFLICT is 0 and the LDS array is busy 17 % of the
// kernel (SQ_LDS_IDX_ACTIVE = 4 cycles per ds_read_b128), so the reads cost latency beside the MFMAs, not bandwidth.
constexpr int kHalfSlot = 128 * BK;          // halfs per half-tile slot
enum { SLOT_A0 = 0, SLOT_A1 = 1, SLOT_B0 = 2, SLOT_B1 = 3 };

template <int N>
__device__ __forceinline__ void wait_vm() {
    if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#ifdef MW_GEMM_STAMPS                // timing experiments only: s_memrealtime (100 MHz) at four points of every workgroup
__device__ unsigned long long g_gemm_stamps[8192 * 4];
#define MW_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_gemm_stamps[blockIdx.x * 4 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MW_STAMP(k) do { } while (0)
#endif

template <bool F8>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) f16 lds[];   // [2 K-tiles][4 half-tiles][128][64], the only LDS object
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fg = lane >> 4;

    const int nt_n = (p.N + 255) / 256;
    const int nt_m = (p.M + 255) / 256;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    #ifdef MW_GROUP_M                     // tuning builds
    constexpr int GROUP_M = MW_GROUP_M;
#else
    constexpr int GROUP_M = 4;        // M = 24000, K = 1280: 4 row tiles per column tile beat 8 by 2-4 % (N = 1280, 3840; K = 5120), 2 and 16 lose
#endif
    const int per_group = GROUP_M * nt_n;
    const int group = bid / per_group;
    const int gm0 = group * GROUP_M;
    const int gsz = min(GROUP_M, nt_m - gm0);
    const int in_group = bid - group * per_group;
    const int m0 = (gm0 + in_group % gsz) * 256, n0 = (in_group / gsz) * 256;
    // a tile that lies entirely in the V third of a QKV product is computed transposed (workgroup-uniform)
    const bool v_tile = !F8 && p.out_mode == OUT_QKV_SPLIT && n0 >= 2 * p.D && p.D % 256 == 0 && p.T % 4 == 0;

    // per-lane DMA sources: instruction i of a half-tile fills slot rows (wave*2+i)*8 + (lane>>3).
    // f16 form: 64-bit pointers through the builtin.  fp8 form: the 8-register MFMA operands leave no room for
    // eight pointers and their per-issue 64-bit adds (34 registers spilled, and every reload drains the DMA
    // queue), so it keeps 32-bit byte offsets and issues the load itself with the matrix base + k offset in
    // scalar registers (both matrices span < 4 GiB: checked by the launcher).
    const f16* a_src[2][2];
    const f16* b_src[2][2];
    // fp8 form: ONE lane offset per matrix (instruction 0 of half-tile 0).  Half-tile 1 lies a whole number of rows
    // further (a scalar), instruction 1 eight rows further with the swizzle key's bit 2 flipped (offset ^ 64: the
    // row pitch is a multiple of 128 bytes, so bits 4..6 of the offset are the chunk alone).  No clamping: the
    // launcher sends only flat, N % 256 == 0 problems here and A is readable for 255 rows past M.
    unsigned a_off0b = 0, b_off0b = 0;
    if constexpr (F8) {
        const int r = wave * 16 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        a_off0b = (unsigned)(((long)(m0 + (r >> 6) * 128 + (r & 63)) * p.lda + chunk * 8) * 2);
        b_off0b = (unsigned)(((long)(n0 + (r >> 5) * 64 + (r & 31)) * p.ldw + chunk * 8) * 2);
    } else {
        // Slot row r of an A half-tile holds tile row (r>>6)*128 + h*64 + (r&63).  Slot row r of a B half-tile holds,
        // of its wave column's 32 rows, row 8*(x>>2) + 4*i + (x&3) where i = (r>>4)&1 is the MFMA subtile and x = r&15 the
        // operand row inside it: output column 4*fg + c of subtile i is then tile column 8*fg + 4*i + c, and a lane's two
        // subtiles are eight consecutive columns (gemm256_epilogue_perm).  In a V tile the roles are exchanged: weight
        // rows fill the A slots and (permuted) activation rows the B slots, which computes the transposed tile.
        auto act_row = [&](int m) {
            m = min(m, p.M - 1);
            return p.A + (long)(m / p.a_rows_per_batch) * p.a_batch_stride + (long)(m % p.a_rows_per_batch) * p.lda;
        };
        auto w_row = [&](int n) { return p.W + (long)min(n, p.N - 1) * p.ldw; };
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = (wave * 2 + i) * 8 + (lane >> 3);
                const int chunk = (lane & 7) ^ ((r >> 1) & 7);
                const int ar = (r >> 6) * 128 + h * 64 + (r & 63);
                const int x = r & 15;
                const int br = (r >> 5) * 64 + h * 32 + 8 * (x >> 2) + 4 * ((r >> 4) & 1) + (x & 3);
                a_src[h][i] = (v_tile ? w_row(n0 + ar) : act_row(m0 + ar)) + chunk * 8;
                b_src[h][i] = (v_tile ? act_row(m0 + br) : w_row(n0 + br)) + chunk * 8;
            }
    }
    const unsigned lds_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) f16*)lds;
    auto issue = [&](int buf, int slot, bool is_a, int h, int tile) {
        if constexpr (F8) {
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(((buf * 4 + slot) * kHalfSlot + wave * 2 * 8 * BK) * 2));
            const long pitch = (is_a ? p.lda : p.ldw) * 2;                                    // bytes per row
            const char* sbase = reinterpret_cast<const char*>(is_a ? p.A : p.W) + (long)tile * (BK * 2) + (long)h * (is_a ? 64 : 32) * pitch;
            const unsigned v0 = is_a ? a_off0b : b_off0b;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst), "v"(v0), "s"(sbase) : "memory");
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :: "s"(dst + (unsigned)(8 * BK * 2)), "v"(v0 ^ 64u), "s"(sbase + 8 * pitch) : "memory");
        } else {
            const f16* const (&src)[2] = is_a ? a_src[h] : b_src[h];
            f16* dst = lds + (buf * 4 + slot) * kHalfSlot + wave * 2 * 8 * BK;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (long)tile * BK),
                                                 (__attribute__((address_space(3))) void*)(dst + i * 8 * BK), 16, 0, 0);
        }
    };
    // fragment addresses: row r of a slot, 16-byte chunk c -> r*64 + ((c ^ ((r>>1)&7)) << 3); the XOR term
    // depends on fr only (all row offsets are multiples of 16)
    const int sx = (fr >> 1) & 7;
    const int a_off0 = (wr * 64 + fr) * BK + ((fg ^ sx) << 3);          // + j*16*BK; k-step 1: chunk ^ 4
    const int b_off0 = (wc * 32 + fr) * BK + ((fg ^ sx) << 3);
    // fragments of both k-steps of a tile row: 8 registers per (sub)tile, the f16 form uses them as two
    // 4-register operands, the fp8 form as one 8-register operand
    int8v af[4], b0f[2], b1f[2];
    auto read_a = [&](const f16* slot) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            af[j].lo = *reinterpret_cast<const int4v*>(slot + a_off0 + j * 16 * BK);
            af[j].hi = *reinterpret_cast<const int4v*>(slot + (a_off0 ^ 32) + j * 16 * BK);
        }
    };
    auto read_b = [&](const f16* slot, int8v (&bf)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bf[i].lo = *reinterpret_cast<const int4v*>(slot + b_off0 + i * 16 * BK);
            bf[i].hi = *reinterpret_cast<const int4v*>(slot + (b_off0 ^ 32) + i * 16 * BK);
        }
    };
    float4v acc[4][8];   // [n subtile][m subtile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};
    auto quadrant = [&](auto hac, auto hbc, int8v (&bf)[2]) {
        constexpr int HA = decltype(hac)::value, HB = decltype(hbc)::value;
        __builtin_amdgcn_s_setprio(1);
        if constexpr (F8) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    acc[HB * 2 + i][HA * 4 + j] =
                        __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bf[i], af[j], acc[HB * 2 + i][HA * 4 + j], 0, 0, 0, 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    acc[HB * 2 + i][HA * 4 + j] = mfma_f16(bf[i].lo, af[j].lo, acc[HB * 2 + i][HA * 4 + j]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    acc[HB * 2 + i][HA * 4 + j] = mfma_f16(bf[i].hi, af[j].hi, acc[HB * 2 + i][HA * 4 + j]);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // A K-tile is TWO phases of 32 MFMAs: Q1 = A0 x (B0, B1), Q2 = A1 x (B1, B0) (B stays in registers).  A phase is
    //     ds_read what it adds | issue LDS-DMA prefetch | counted vmcnt | lgkmcnt(0) | barrier | 32 MFMA | barrier
    // and the two wave rows run one barrier apart, so the wave that shares a SIMD with an MFMA-ing wave is the one
    // doing LDS reads and prefetch.  Prefetch: Q1 of tile t issues A1(t+1) (its slot was read in Q2(t-1)), Q2 of
    // tile t issues A0, B0, B1 of tile t+2 into the slots Q1(t) has just read: a slot is refilled in the phase
    // after its last read, which is safe because every wave drains its LDS reads (lgkmcnt(0)) before the barrier
    // that ends its read section, and both wave rows have passed that barrier before anyone's next read section.
    // A half-tile is read one phase after the phase whose vmcnt retired it (every wave waits for its own DMA, then a
    // barrier); 8 DMA instructions per wave stay in flight across the barriers, hence vmcnt(8), never 0 in the loop.
    // MODE 0: steady state; 1: second-to-last K-tile (no tile t+2); 2: last K-tile (nothing to issue).
    auto ktile = [&](auto bufc, auto modec, int t) {
        constexpr int BUF = decltype(bufc)::value, MODE = decltype(modec)::value;
        // fp8 form: the second buffer lies past the 64 KB reach of a ds_read offset, and left alone the compiler keeps
        // a second set of lane addresses for it live across the whole loop (spilled, each reload draining the DMA
        // queue).  An offset it cannot see through is added per K-tile instead.
        int buf_off = BUF * 4 * kHalfSlot;
        if constexpr (F8 && BUF == 1) asm volatile("" : "+s"(buf_off));
        const f16* cur = lds + buf_off;
        // Q1
        read_b(cur + SLOT_B0 * kHalfSlot, b0f);
        read_a(cur + SLOT_A0 * kHalfSlot);
        read_b(cur + SLOT_B1 * kHalfSlot, b1f);
        if constexpr (MODE < 2) issue(BUF ^ 1, SLOT_A1, true, 1, t + 1);
        wait_vm<MODE < 2 ? 8 : 0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        quadrant(I0{}, I0{}, b0f);
        quadrant(I0{}, I1{}, b1f);
        __builtin_amdgcn_s_barrier();
        // Q2
        read_a(cur + SLOT_A1 * kHalfSlot);
        if constexpr (MODE == 0) {
            issue(BUF, SLOT_A0, true, 0, t + 2);
            issue(BUF, SLOT_B0, false, 0, t + 2);
            issue(BUF, SLOT_B1, false, 1, t + 2);
        }
        if constexpr (MODE < 2) wait_vm<MODE == 0 ? 8 : 2>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        quadrant(I1{}, I1{}, b1f);
        quadrant(I1{}, I0{}, b0f);
        __builtin_amdgcn_s_barrier();
    };

    const int nk = p.K / BK;                         // even and >= 2 (checked by the launcher)
    MW_STAMP(0);
    issue(0, SLOT_A0, true, 0, 0); issue(0, SLOT_B0, false, 0, 0); issue(0, SLOT_B1, false, 1, 0);
    issue(0, SLOT_A1, true, 1, 0);
    issue(1, SLOT_A0, true, 0, 1); issue(1, SLOT_B0, false, 0, 1); issue(1, SLOT_B1, false, 1, 1);
    wait_vm<8>();                                    // A0, B0, B1 of tile 0 have landed
    __builtin_amdgcn_s_barrier();
    MW_STAMP(1);
    if (wr == 1) __builtin_amdgcn_s_barrier();       // second wave row runs one barrier behind
    for (int t = 0; t < nk - 2; t += 2) {
        ktile(I0{}, I0{}, t);
        ktile(I1{}, I0{}, t + 1);
    }
    ktile(I0{}, I1{}, nk - 2);
    ktile(I1{}, std::integral_constant<int, 2>{}, nk - 1);
    if (wr == 0) __builtin_amdgcn_s_barrier();       // same barrier count for every wave
#ifdef MW_GEMM_SKIP_EPILOGUE          // timing experiments only: the main loop alone (the test is false at run time)
    if (p.scale != 12345.f) return;
#endif
    MW_STAMP(2);
    if constexpr (F8) gemm_epilogue<8, true>(p, acc, m0, n0, wr, wc, fr, fg);
    else gemm256_epilogue_f16(p, acc, m0, n0, wr, wc, fr, fg, v_tile);
#ifdef MW_GEMM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MW_STAMP(3);
#endif
}

template <bool F8>
void launch_256(const GemmParams& p, hipStream_t s) {
    constexpr size_t lds = (size_t)8 * kHalfSlot * sizeof(f16);     // 128 KiB
    static std::once_flag prepared;
    std::call_once(prepared, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<F8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    const int nt = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    hipLaunchKernelGGL(gemm256_kernel<F8>, dim3(nt), dim3(512), lds, s, p);
}

}  // namespace

#ifdef MW_GEMM_STAMPS
void gemm_debug_stamps(unsigned long long* dst, int n) { (void)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_gemm_stamps), (size_t)n * 8); }
#endif

void launch_gemm(const GemmParams& p0, hipStream_t s) {
    if (p0.f8) {
        // fp8 operands: one byte per element.  Re-express the problem in 2-byte units for the staging code
        // (the launcher's callers guarantee K % 128 == 0 and even leading dimensions).
        GemmParams p = p0;
        p.K /= 2; p.lda /= 2; p.ldw /= 2; p.a_batch_stride /= 2;
        const int nk = p.K / BK;
        const long tiles256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
        // 256x256 form: 1200-1290 TFLOP/s at M = 24000..48000, K = 1280 against 1010-1070 for the 128x128 form.
        static const bool use256 = getenv("MW_GEMM_F8_NO256") == nullptr;
        const bool flat = p.a_rows_per_batch >= p.M;      // the fp8 256x256 form does not clamp rows (see its DMA offsets)
        // ... and addresses both matrices with 32-bit byte offsets (units here are 2 bytes; A is read up to 255 rows past M)
        const bool fits32 = ((long)(p.M + 256) * p.lda + p.K) * 2 < (1L << 32) && ((long)p.N * p.ldw + p.K) * 2 < (1L << 32);
        if (use256 && flat && fits32 && p.N % 256 == 0 && nk >= 2 && nk % 2 == 0 && tiles256 >= 256) launch_256<true>(p, s);
        else launch_t<4, true, true>(p, s);
        return;
    }
    const GemmParams& p = p0;
    // MI = 8 (256-row tile, one workgroup per CU) measured 35 % slower than MI = 4 (two per CU): with a
    // single barrier-synchronised workgroup on a CU nothing covers its waits.
    static const bool no256 = getenv("MW_GEMM_NO256") != nullptr;
    // fewest 256x256 tiles for which that kernel is taken (one workgroup per CU, 256 CUs).  176 (two thirds of the CUs
    // busy in one round) still beats the 128x128 form's round and a half: the aligner's 12000 x 1024 products (188
    // tiles) took align() from 10.9 k to 11.5 k audio-s/s.  MW_GEMM_256_MIN_TILES: tuning runs.
    static const long min256 = getenv("MW_GEMM_256_MIN_TILES") ? atol(getenv("MW_GEMM_256_MIN_TILES")) : 176;
    static const bool share = getenv("MW_GEMM_CS") != nullptr;          // tuning runs: the CU-sharing 256x128 form for every big product
    const int nk = p.K / BK;
    const long tiles256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
    if (p.K % BK != 0) launch_t<4, false>(p, s);
    else if (p.groups > 1) launch_t<4, true>(p, s);                  // grouped products: the 128x128 form only
    else if (share && tiles256 >= 256) launch_t<8, true>(p, s);
    else if (!no256 && nk >= 2 && nk % 2 == 0 && tiles256 >= min256 && p.N % 8 == 0) launch_256<false>(p, s);
    else launch_t<4, true>(p, s);
}

}  // namespace mw
