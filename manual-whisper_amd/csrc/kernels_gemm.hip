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
#include "mw_kernels.h"

namespace mw {

namespace {

constexpr int BN = 128, BK = 64;

__device__ __forceinline__ int swz(int row, int chunk) {      // offset in halfs inside a [rows][64] tile
    return row * BK + ((chunk ^ ((row >> 1) & 7)) << 3);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

// Fused epilogue: lane holds C[m][n..n+3], m = m0 + wm*MI*16 + j*16 + fr, n = n0 + wn*64 + i*16 + 4*fg.
template <int MI>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, float4v (&acc)[4][MI], int m0, int n0, int wm, int wn,
                                              int fr, int fg) {
#pragma unroll
    for (int j = 0; j < MI; ++j) {
        const int m = m0 + wm * MI * 16 + j * 16 + fr;
        if (m >= p.M) continue;
        int bb = 0, tt = 0;
        if (p.out_mode >= OUT_QKV_SPLIT) { bb = m / p.T; tt = m - bb * p.T; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + i * 16 + 4 * fg;
            if (n >= p.N) continue;                     // N is a multiple of 4
            float4v v = acc[i][j];
            if (p.bias) {
                const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
                v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
            }
            if (n < p.scale_ncols) { v[0] *= p.scale; v[1] *= p.scale; v[2] *= p.scale; v[3] *= p.scale; }
            if (p.gelu) { v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]); }
            if (p.out_mode == OUT_RESID_F32) {
                float* c = reinterpret_cast<float*>(p.C) + (long)(m / p.c_rows_per_batch) * p.c_batch_stride +
                           (long)(m % p.c_rows_per_batch) * p.ldc + n;
                float4 o = *reinterpret_cast<float4*>(c);
                o.x += v[0]; o.y += v[1]; o.z += v[2]; o.w += v[3];
                *reinterpret_cast<float4*>(c) = o;
            } else {
                half4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                if (p.out_mode == OUT_F16) {
                    f16* c = reinterpret_cast<f16*>(p.C) + (long)(m / p.c_rows_per_batch) * p.c_batch_stride +
                             (long)(m % p.c_rows_per_batch) * p.ldc + n;
                    *reinterpret_cast<half4*>(c) = h;
                } else {
                    const int sec = n / p.D, nn = n - sec * p.D;
                    const int hh = nn >> 6, e = nn & 63;
                    const long bh = (long)bb * p.H + hh;
                    if (p.out_mode == OUT_QKV_SPLIT && sec == 2) {           // V transposed: [B][H][64][Tpad]
                        f16* c = p.v + (bh * 64 + e) * p.Tpad + tt;
                        c[0] = h[0]; c[p.Tpad] = h[1]; c[2L * p.Tpad] = h[2]; c[3L * p.Tpad] = h[3];
                    } else {
                        f16* base = (p.out_mode == OUT_QKV_SPLIT) ? (sec == 0 ? p.q : p.k) : (sec == 0 ? p.k : p.v);
                        *reinterpret_cast<half4*>(base + (bh * p.T + tt) * 64 + e) = h;
                    }
                }
            }
        }
    }
}

template <int MI, bool GLDS>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
    constexpr int BM = MI * 32;
    constexpr int STAGE = (BM + BN) * BK;                        // halfs per buffer: A tile then W tile
    extern __shared__ __attribute__((aligned(16))) f16 lds[];   // [2][STAGE]
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
        stage(0, 0);
        __syncthreads();                       // drains the LDS-DMA (vmcnt(0)) before anyone reads
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) stage((kt + 1) & 1, (kt + 1) * BK);
            compute(kt & 1);
            __syncthreads();
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
    gemm_epilogue<MI>(p, acc, m0, n0, wm, wn, fr, fg);
}

template <int MI, bool GLDS>
void launch_t(const GemmParams& p, hipStream_t s) {
    constexpr int BM = MI * 32;
    constexpr size_t lds = (size_t)2 * (BM + BN) * BK * sizeof(f16);
    static bool prepared = false;
    if (!prepared) {      // raise the dynamic-LDS limit once (first use is never inside stream capture: the encoder runs first)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<MI, GLDS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        prepared = true;
    }
    const int nt = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    hipLaunchKernelGGL((gemm_kernel<MI, GLDS>), dim3(nt), dim3(256), lds, s, p);
}

}  // namespace

void launch_gemm(const GemmParams& p, hipStream_t s) {
    // MI = 8 (256-row tile, one workgroup per CU) measured 35 % slower than MI = 4 (two per CU): with a
    // single barrier-synchronised workgroup on a CU nothing covers its waits.
    if (p.K % BK != 0) launch_t<4, false>(p, s);
    else launch_t<4, true>(p, s);
}

}  // namespace mw
