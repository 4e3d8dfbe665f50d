// Encoder-side kernels other than the GEMM: LayerNorm, position-table broadcast, the
// non-causal flash attention over 1500 frames, and the weight import conversions.
#include "mw_kernels.h"

namespace mw {

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ------------------------------------------------------------------ LayerNorm
// One wave per row (fp32 in, fp32 statistics, eps 1e-5), 4 rows per workgroup; the row stays in
// registers between the two passes so HBM sees one read and one fp16 write.
// modeling_whisper.py:392,402,642 (nn.LayerNorm defaults).
constexpr int LN_MAXV = 8;     // D <= 2048
template <int NV>               // NV = ceil(D / 256): float4 pieces per lane
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                        const float* __restrict__ b, f16* __restrict__ y,
                                                        float* __restrict__ y32, int rows, int D,
                                                        int* __restrict__ advance) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    // Only the decode step passes `advance` (its position counter): no other workgroup of this launch
    // reads it, and the launches after this one see the new value across the kernel boundary.
    if (advance && blockIdx.x == 0 && threadIdx.x == 0) *advance += 1;
    if (row >= rows) return;
    const float* xr = x + (long)row * D;
    // every load of the wave is issued here (row, gain, offset): one round trip, the reductions run behind it
    float4 v[NV], gg[NV], bb[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = min((lane + 64 * i) * 4, D - 4);          // past the row's end: a duplicate of its last piece, never used
        v[i] = *reinterpret_cast<const float4*>(xr + c);
        gg[i] = *reinterpret_cast<const float4*>(g + c);
        bb[i] = *reinterpret_cast<const float4*>(b + c);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if ((lane + 64 * i) * 4 < D) s += v[i].x + v[i].y + v[i].z + v[i].w;
    const float mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < D) {
            v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
            q += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / D + 1e-5f);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < D) {
            const float o0 = v[i].x * rstd * gg[i].x + bb[i].x, o1 = v[i].y * rstd * gg[i].y + bb[i].y;
            const float o2 = v[i].z * rstd * gg[i].z + bb[i].z, o3 = v[i].w * rstd * gg[i].w + bb[i].w;
            if (y) *reinterpret_cast<half4*>(y + (long)row * D + c) = half4{(f16)o0, (f16)o1, (f16)o2, (f16)o3};
            if (y32) *reinterpret_cast<float4*>(y32 + (long)row * D + c) = make_float4(o0, o1, o2, o3);
        }
    }
}

// LayerNorm whose output feeds an fp8 product: the normalised row is quantised to e4m3 with one
// scale per row, scale = max|y| / 448 (1 for an all-zero row), y8 = round(y / scale).  Same
// statistics as layernorm_kernel; the row never leaves the registers.
__global__ __launch_bounds__(256) void layernorm_fp8_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                            const float* __restrict__ b, uint8_t* __restrict__ y8,
                                                            float* __restrict__ yscale, float* __restrict__ y32, int rows, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (long)row * D;
    float4 v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < D) {
            v[i] = *reinterpret_cast<const float4*>(xr + c);
            s += v[i].x + v[i].y + v[i].z + v[i].w;
        }
    }
    const float mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < D) {
            v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
            q += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / D + 1e-5f);
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < D) {
            const float4 gg = *reinterpret_cast<const float4*>(g + c);
            const float4 bb = *reinterpret_cast<const float4*>(b + c);
            v[i].x = v[i].x * rstd * gg.x + bb.x; v[i].y = v[i].y * rstd * gg.y + bb.y;
            v[i].z = v[i].z * rstd * gg.z + bb.z; v[i].w = v[i].w * rstd * gg.w + bb.w;
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[i].x), fabsf(v[i].y))), fmaxf(fabsf(v[i].z), fabsf(v[i].w)));
            if (y32) *reinterpret_cast<float4*>(y32 + (long)row * D + c) = v[i];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const float sc = mx > 0.f ? mx / 448.f : 1.f;
    if (lane == 0) yscale[row] = sc;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < D)
            *reinterpret_cast<unsigned*>(y8 + (long)row * D + c) = pack4_e4m3(v[i].x / sc, v[i].y / sc, v[i].z / sc, v[i].w / sc);
    }
}

__global__ void bcast_rows_kernel(const float* __restrict__ table, float* __restrict__ x, long per_batch, long total) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i < total) *reinterpret_cast<float4*>(x + i) = *reinterpret_cast<const float4*>(table + (i % per_batch));
}

// ------------------------------------------------------------------ encoder attention
// softmax(Q K^T) V per (batch, head), non-causal, no mask (modeling_whisper.py:228-235 with
// scaling already applied to q, :309).  Workgroup = 128 queries (4 waves x 32), KV tiles of 64.
//
// S^T = K Q^T is computed with keys on the MFMA row index, so a lane holds, for ONE query
// (column lane&15), 4 keys per 16-key tile: the running max / sum of the online softmax need only
// the 4 lanes that share a column (v_permlane16_swap + v_permlane32_swap: plain VALU, no LDS round
// trip).  Those same registers, converted to fp16, are directly the B operand of O^T = V^T P^T
// (k index = key).  V arrives already transposed ([64][T]) from the QKV GEMM; the rows of the K tile are
// staged in a permuted order (see the kernel) so that the 8 keys a lane contracts over are contiguous in V^T:
// every V^T fragment is one ds_read_b128, like the K fragments.  Output O^T has the query on the
// lane again, so the rescale factors are per-lane scalars.
//
// The query fragments are multiplied by log2(e) once, as they are loaded (q arrives scaled by head_dim^-0.5, a power
// of two, from the QKV product: a factor that is not one would round differently from tile position to tile
// position there), so the scores are exponents of 2, and the score accumulators start at minus the query's
// reference exponent: per score the softmax costs one v_exp_f32, one add, a third of a max3 and half a convert, no
// multiply or subtract.  The
// reference is the first tile's maximum and is only raised when some score of the wave outgrows it (see the kernel),
// so the O / l rescale is skipped for almost every tile.  (This file is compiled with -fno-honor-nans: the maxima of
// MFMA results then need no canonicalising v_max x, x each.)
constexpr int AQ = 128, AKV = 64;
constexpr float kLog2e = 1.4426950408889634f;

__device__ __forceinline__ int swz64(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }

// max over the 4 lanes lane&15, +16, +32, +48 (every lane gets the result)
__device__ __forceinline__ float quad_column_max(float v) {
    const unsigned u = __float_as_uint(v);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const unsigned w = __float_as_uint(v);
    auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float quad_column_sum(float v) {
    const unsigned u = __float_as_uint(v);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const unsigned w = __float_as_uint(v);
    auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// Counters show the vector issue port as the busiest unit of this kernel (about 960 of its cycles per wave and key
// tile in its first form, 256 of them the exponentials, against 512 of the matrix pipe), and on this chip the vector
// and matrix instructions of the waves of a SIMD add up rather than overlap, so the loop is written to issue as few
// vector instructions as it can:
//  * K and V^T tiles go from global memory to LDS by LDS-DMA (see the staging below): no staging registers, no
//    ds_write, no predicate; with the registers that frees, four workgroups share a CU (two: 401 us at batch 16,
//    three: 316, four with the rest of this list: 247).
//  * no per-tile maximum.  The scores come out of the MFMAs relative to the running reference exponent; whether that
//    reference still holds is read off the lane's own sum of the 16 weights it just formed (needed anyway): a sum
//    above kSumLimit means some score outgrew the reference, and only then (and for the first tile, which sets the
//    reference) is the tile's maximum taken, the reference raised, O and l rescaled and the tile's scores formed
//    again from the K tile still in LDS.  Weights stay below kSumLimit, far inside fp16 range.
// VARLEN (a packed batch of recordings of different lengths, the forced aligner's): batch element b has keys_of[b] <= T
// keys (and as many queries that matter); its rows still lie T apart.  T below is then the element's own length
// wherever keys are counted; workgroups whose queries all lie past it have nothing to do.
constexpr float kSumLimit = 1024.0f;
// LDS buffers of a workgroup (a K and a V^T tile each).  Two, the next tile requested one ahead, and four workgroups
// per CU (125 registers): 246.6 us at batch 16; three, two tiles ahead, three workgroups per CU: 256.
#ifndef MW_ATTN_BUFFERS
#define MW_ATTN_BUFFERS 2
#endif
constexpr int kAttnBuffers = MW_ATTN_BUFFERS;
typedef unsigned u32x4e __attribute__((ext_vector_type(4)));

template <bool VARLEN>
__global__ __launch_bounds__(256, kAttnBuffers == 2 ? 4 : 3) void enc_attention_kernel(const f16* __restrict__ q, const f16* __restrict__ k,
                                                            const f16* __restrict__ vT, f16* __restrict__ out, int H,
                                                            int Tstride, int Tpad, const int* __restrict__ keys_of) {
    __shared__ __attribute__((aligned(16))) f16 lds[kAttnBuffers][2][AKV * 64];     // [buffer][K | V^T][64 x 64]: 16 KB each
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    // 1-D grid, XCD-aware: blocks b and b+8 share an XCD, so give each XCD a contiguous run of
    // (head, query-tile) pairs; the query tiles of one head then read its K/V through one L2.
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, qq = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + idx;
    }
    const int nqt = (Tstride + AQ - 1) / AQ;
    const int bh = bid / nqt;
    const int q0 = (bid - bh * nqt) * AQ + wave * 32;
    const int T = VARLEN ? keys_of[bh / H] : Tstride;
    if constexpr (VARLEN) {
        if ((bid - bh * nqt) * AQ >= T) return;                 // (workgroup-uniform, before any barrier)
    }
    const f16* qb = q + (long)bh * Tstride * 64;

    // Q fragments (B operand: column = query, k = head-dim), straight from global.
    half8 qf[2][2];
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
        int row = q0 + qs * 16 + fr;
        if (row >= T) row = T - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const half8 raw = *reinterpret_cast<const half8*>(qb + (long)row * 64 + ks * 32 + fg * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[qs][ks][e] = (f16)((float)raw[e] * kLog2e);     // scores become exponents of 2
        }
    }

    // Staging by LDS-DMA, kAttnBuffers - 1 tiles ahead.  A wave instruction fills
    // 1 KB = 8 rows of 128 bytes; which 16 bytes of global memory a lane fetches is free, so the row swizzle costs
    // nothing, and the key order is set by the K side: LDS row rho of the K tile holds key
    //     pi(rho) = 32 (rho >> 5) + 8 ((rho >> 2) & 3) + 4 ((rho >> 4) & 1) + (rho & 3),
    // so that the eight scores a lane holds per 32 keys (S^T rows 16 (2s + h) + 4 fg + r) belong to the CONSECUTIVE
    // keys 32 s + 8 fg + 4 h + r: as fp16 they are directly the B operand of O^T = V^T P^T over an unpermuted V^T
    // tile.  Nothing is read out of range: K rows past the end repeat the last row (their scores are masked), V^T
    // chunks past the end repeat the row's last chunk (weight 0 exactly).
    const __amdgpu_buffer_rsrc_t krs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(k + (long)bh * Tstride * 64), 0, T * 128, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(vT + (long)bh * 64 * Tpad), 0, 64 * Tpad * 2, 0x00020000);
    int k_key[2], k_chunk[2], v_row[2], v_chunk[2];       // per lane and DMA instruction: K key inside the tile, byte offsets
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rho = (i * 4 + wave) * 8 + (lane >> 3);           // LDS row this lane fills (K: score row; V^T: head-dim row)
        const int c = (lane & 7) ^ ((rho >> 1) & 7);                // logical 16-byte chunk that lands at position lane & 7
        k_key[i] = 32 * (rho >> 5) + 8 * ((rho >> 2) & 3) + 4 * ((rho >> 4) & 1) + (rho & 3);
        k_chunk[i] = c * 16;
        v_row[i] = rho * Tpad * 2;
        v_chunk[i] = c * 8;
    }
    auto stage = [&](int kv0, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (__attribute__((address_space(3))) void*)&lds[buf][0][(i * 4 + wave) * 512], 16,
                                                     min(kv0 + k_key[i], T - 1) * 128 + k_chunk[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (__attribute__((address_space(3))) void*)&lds[buf][1][(i * 4 + wave) * 512], 16,
                                                     v_row[i] + min(kv0 + v_chunk[i], Tpad - 8) * 2, 0, 0, 0);
    };

    float4v o[4][2];           // [head-dim tile][query tile]: O^T
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[i][0] = float4v{0, 0, 0, 0}; o[i][1] = float4v{0, 0, 0, 0}; }
    // Reference exponent per query (log2 units).  The score accumulators START at -mrun, so the MFMAs deliver
    // score - mrun and the exponentials take them as they come.
    float mrun[2] = {0.f, 0.f};
    float lrun[2] = {0.f, 0.f};
    float4v sc[4][2];          // S^T tiles: [key tile i][query tile qs], relative to the reference exponent

    auto scores = [&](int buf, int kv0) {
        const float n0 = -mrun[0], n1 = -mrun[1];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sc[i][0] = float4v{n0, n0, n0, n0}; sc[i][1] = float4v{n1, n1, n1, n1};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const half8 kf = *reinterpret_cast<const half8*>(&lds[buf][0][swz64(i * 16 + fr, ks * 4 + fg)]);
                sc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[0][ks], sc[i][0], 0, 0, 0);
                sc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[1][ks], sc[i][1], 0, 0, 0);
            }
        }
        if (kv0 + AKV > T) {      // last tile: keys past the end contribute nothing
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kv0 + 32 * (i >> 1) + 8 * fg + 4 * (i & 1) + r >= T) { sc[i][0][r] = -1.0e30f; sc[i][1][r] = -1.0e30f; }      // key pi(16 i + 4 fg + r)
        }
    };
    // the tile's maxima become part of the reference (first tile: they ARE the reference, nothing to rescale yet)
    auto raise_reference = [&](bool first) {
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
            float m = fmaxf(fmaxf(sc[0][qs][0], sc[0][qs][1]), fmaxf(sc[0][qs][2], sc[0][qs][3]));
#pragma unroll
            for (int i = 1; i < 4; ++i) m = fmaxf(m, fmaxf(fmaxf(sc[i][qs][0], sc[i][qs][1]), fmaxf(sc[i][qs][2], sc[i][qs][3])));
            const float mx = quad_column_max(m);
            const float delta = first ? mx : fmaxf(mx, 0.f);
            const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
            mrun[qs] += delta;
            lrun[qs] *= alpha;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o[i][qs][0] *= alpha; o[i][qs][1] *= alpha; o[i][qs][2] *= alpha; o[i][qs][3] *= alpha;
                sc[i][qs][0] -= delta; sc[i][qs][1] -= delta; sc[i][qs][2] -= delta; sc[i][qs][3] -= delta;
            }
        }
    };
    float ps[2];
    auto weights = [&]() {      // scores -> weights in place; ps = this lane's sums
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(sc[i][qs][r]);
                    sc[i][qs][r] = pv;
                    a += pv;
                }
            ps[qs] = a;
        }
    };

    const int ntile = (T + AKV - 1) / AKV;
    constexpr int NB = kAttnBuffers, AHEAD = NB - 1;      // tiles requested ahead of the one being worked on
    stage(0, 0);
    if (AHEAD == 2 && ntile > 1) { stage(AKV, 1); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }      // (the compiler does not count the LDS-DMA)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    for (int it = 0; it < ntile; ++it) {
        const int kv0 = it * AKV;
        const bool ahead = it + AHEAD < ntile;
        if (ahead) stage(kv0 + AHEAD * AKV, buf == 0 ? NB - 1 : buf - 1);      // into the buffer tile it - 1 has left

        scores(buf, kv0);
        if (it == 0) raise_reference(true);
        weights();
        if (__any(ps[0] > kSumLimit || ps[1] > kSumLimit)) {     // wave-uniform, rare: some score outgrew the reference
            scores(buf, kv0);
            raise_reference(false);
            weights();
        }
        lrun[0] += ps[0];                                // partial over this lane's keys; reduced at the end
        lrun[1] += ps[1];
        half8 pf[2][2];            // [query tile][k step of 32 keys]
#pragma unroll
        for (int qs = 0; qs < 2; ++qs)
#pragma unroll
            for (int s = 0; s < 2; ++s)
                pf[qs][s] = half8{(f16)sc[2 * s][qs][0], (f16)sc[2 * s][qs][1], (f16)sc[2 * s][qs][2], (f16)sc[2 * s][qs][3],
                                  (f16)sc[2 * s + 1][qs][0], (f16)sc[2 * s + 1][qs][1], (f16)sc[2 * s + 1][qs][2],
                                  (f16)sc[2 * s + 1][qs][3]};
        // ---- O^T += V^T P^T ; element j of k-step s is key 32s + 16(j>>2) + 4*fg + (j&3) = permuted column 32s + 8fg + j
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const half8 vf = *reinterpret_cast<const half8*>(&lds[buf][1][swz64(dt * 16 + fr, 4 * s + fg)]);
                o[dt][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[0][s], o[dt][0], 0, 0, 0);
                o[dt][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[1][s], o[dt][1], 0, 0, 0);
            }
        }
        // tile it + 1 (requested an iteration ago, before the four requests of this one) has to be in LDS for everybody
        if (ahead && AHEAD == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        buf = buf == NB - 1 ? 0 : buf + 1;
    }

    // ---- normalise and store: lane holds O[query][16dt + 4fg .. +3]
    const int b = bh / H, h = bh - b * H;
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
        const float inv = 1.0f / quad_column_sum(lrun[qs]);
        const int row = q0 + qs * 16 + fr;
        if (row < T) {
            f16* dst = out + ((long)b * Tstride + row) * (H * 64) + h * 64 + 4 * fg;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *reinterpret_cast<half4*>(dst + dt * 16) =
                    half4{(f16)(o[dt][qs][0] * inv), (f16)(o[dt][qs][1] * inv), (f16)(o[dt][qs][2] * inv), (f16)(o[dt][qs][3] * inv)};
        }
    }
}

// ------------------------------------------------------------------ weight import
__device__ __forceinline__ float load_as_f32(const void* src, int dt, long i) {
    if (dt == 0) return reinterpret_cast<const float*>(src)[i];
    if (dt == 1) return (float)reinterpret_cast<const f16*>(src)[i];
    const unsigned int u = (unsigned int)reinterpret_cast<const unsigned short*>(src)[i] << 16;   // bf16
    return __uint_as_float(u);
}

__global__ void convert_kernel(const void* __restrict__ src, int dt, void* __restrict__ dst, int to_f16, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = load_as_f32(src, dt, i);
    if (to_f16) reinterpret_cast<f16*>(dst)[i] = (f16)v; else reinterpret_cast<float*>(dst)[i] = v;
}

__global__ void conv_permute_kernel(const void* __restrict__ src, int dt, f16* __restrict__ dst, int D, int C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;     // index into dst [D][3][C]
    if (i >= (long)D * 3 * C) return;
    const int c = i % C, j = (i / C) % 3, d = i / (3L * C);
    dst[i] = (f16)load_as_f32(src, dt, ((long)d * C + c) * 3 + j);
}

// fp8 weight import: one workgroup per weight row; scale = max|w| / 448 per row (1 for a zero row),
// w8 = round_to_e4m3(w / scale).  cols is a multiple of 4.
__global__ __launch_bounds__(256) void quant_rows_kernel(const void* __restrict__ src, int dt, uint8_t* __restrict__ dst,
                                                         float* __restrict__ scale, int cols) {
    __shared__ float red[4];
    const long base = (long)blockIdx.x * cols;
    const int tid = threadIdx.x;
    float mx = 0.f;
    for (int c = tid; c < cols; c += 256) mx = fmaxf(mx, fabsf(load_as_f32(src, dt, base + c)));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float sc = mx > 0.f ? mx / 448.f : 1.f;
    if (tid == 0) scale[blockIdx.x] = sc;
    for (int c = tid * 4; c < cols; c += 1024)
        *reinterpret_cast<unsigned*>(dst + base + c) =
            pack4_e4m3(load_as_f32(src, dt, base + c) / sc, load_as_f32(src, dt, base + c + 1) / sc,
                       load_as_f32(src, dt, base + c + 2) / sc, load_as_f32(src, dt, base + c + 3) / sc);
}

// e4m3 bytes -> the same values in fp16 (every e4m3 value is an fp16 value): 16 elements per thread
__global__ __launch_bounds__(256) void fp8_to_f16_kernel(const uint8_t* __restrict__ src, f16* __restrict__ dst, long n16) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    const uint4 w = *reinterpret_cast<const uint4*>(src + i * 16);
    const unsigned u[4] = {w.x, w.y, w.z, w.w};
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    half2v h[8];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        h[2 * d] = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(u[d], 1.0f, false);
        h[2 * d + 1] = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(u[d], 1.0f, true);
    }
    uint4* o = reinterpret_cast<uint4*>(dst + i * 16);
    o[0] = *reinterpret_cast<uint4*>(&h[0]);
    o[1] = *reinterpret_cast<uint4*>(&h[4]);
}

}  // namespace

void launch_layernorm(const float* x, const float* g, const float* b, f16* y, float* y32, int rows, int D, hipStream_t s,
                      int* advance) {
    const dim3 grid((rows + 3) / 4), block(256);
    switch ((D + 255) / 256) {
#define MW_LN_CASE(NV) case NV: hipLaunchKernelGGL(layernorm_kernel<NV>, grid, block, 0, s, x, g, b, y, y32, rows, D, advance); break;
        MW_LN_CASE(1) MW_LN_CASE(2) MW_LN_CASE(3) MW_LN_CASE(4) MW_LN_CASE(5) MW_LN_CASE(6) MW_LN_CASE(7)
#undef MW_LN_CASE
        default: hipLaunchKernelGGL(layernorm_kernel<LN_MAXV>, grid, block, 0, s, x, g, b, y, y32, rows, D, advance); break;
    }
}

void launch_layernorm_fp8(const float* x, const float* g, const float* b, uint8_t* y8, float* yscale, float* y32, int rows, int D,
                          hipStream_t s) {
    hipLaunchKernelGGL(layernorm_fp8_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, g, b, y8, yscale, y32, rows, D);
}

void launch_quant_rows(const void* src, int src_dtype, uint8_t* dst, float* scale, int rows, int cols, hipStream_t s) {
    hipLaunchKernelGGL(quant_rows_kernel, dim3(rows), dim3(256), 0, s, src, src_dtype, dst, scale, cols);
}

void launch_fp8_to_f16(const uint8_t* src, f16* dst, long n, hipStream_t s) {
    const long n16 = n / 16;
    hipLaunchKernelGGL(fp8_to_f16_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, s, src, dst, n16);
}

void launch_bcast_rows(const float* table, float* x, int B, int rows, int D, hipStream_t s) {
    const long per = (long)rows * D, total = per * B;
    hipLaunchKernelGGL(bcast_rows_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, s, table, x, per, total);
}

void launch_enc_attention(const f16* q, const f16* k, const f16* vT, f16* out, int B, int H, int T, int Tpad, hipStream_t s,
                          const int* keys_of) {
    const dim3 grid(((T + AQ - 1) / AQ) * B * H);
    if (keys_of) hipLaunchKernelGGL(enc_attention_kernel<true>, grid, dim3(256), 0, s, q, k, vT, out, H, T, Tpad, keys_of);
    else hipLaunchKernelGGL(enc_attention_kernel<false>, grid, dim3(256), 0, s, q, k, vT, out, H, T, Tpad, keys_of);
}

void launch_convert(const void* src, int src_dtype, void* dst, int dst_is_f16, long n, hipStream_t s) {
    hipLaunchKernelGGL(convert_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, src_dtype, dst, dst_is_f16, n);
}

void launch_conv_permute(const void* src, int src_dtype, f16* dst, int D, int C, hipStream_t s) {
    const long n = (long)D * 3 * C;
    hipLaunchKernelGGL(conv_permute_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, src_dtype, dst, D, C);
}

}  // namespace mw
