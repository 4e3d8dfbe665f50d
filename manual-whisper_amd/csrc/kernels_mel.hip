// Log-mel front end for gfx950: framing, reflect padding, 400-point real DFT, power
// spectrum, mel filter bank and log10 in ONE kernel; clamp/scale/transposition in a second.
//
// Algorithm: transformers feature_extraction_whisper.py:140-165 (centred STFT, periodic Hann,
// last frame dropped, mel filters, log10 clamp 1e-10, floor at chunk max - 8, (x+4)/4).
//
// The 400-point DFT is not a power of two, so it is done as an exact-f32 matrix product on the
// f32 MFMA (v_mfma_f32_32x32x2_f32): [256 bins x 400 samples] * [400 samples x 32 frames], with
// the Hann window folded into the cos/sin tables.  The [201 x 3000] power spectrum never
// reaches HBM: it goes through LDS straight into the mel-filter product.
// HBM traffic per chunk: read 1.92 MB PCM, write n_mels*3000*4 B of log-mel (+ fp16 copy).
#include "mw_kernels.h"

namespace mw {

namespace {

constexpr int FR = 32;             // frames per workgroup
constexpr int FLD = kNfft + 1;     // 401: odd row stride -> conflict-free column reads
constexpr int PLD = FR + 1;        // 33

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// grid (94, B), 256 threads. LDS: frame matrix F[32][401] f32, later reused as P[256][33].
__global__ __launch_bounds__(256) void mel_dft_kernel(const float* __restrict__ pcm, MelTables t,
                                                      float* __restrict__ logmel, float* __restrict__ blockmax) {
    __shared__ float lds[FR * FLD];          // 51,328 B  (P needs 256*33*4 = 33,792 B)
    __shared__ float red[4];
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * FR;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const float* x = pcm + (long)b * kChunkSamples;

    // ---- frames with reflect padding of 200 samples (torch.stft center=True)
    for (int e = tid; e < FR * kNfft; e += 256) {
        const int i = e / kNfft, n = e - i * kNfft;
        const int tt = t0 + i;
        float v = 0.f;
        if (tt < kFrames) {
            int s = tt * kHop + n - kNfft / 2;
            if (s < 0) s = -s;
            if (s >= kChunkSamples) s = 2 * (kChunkSamples - 1) - s;
            v = x[s];
        }
        lds[i * FLD + n] = v;
    }
    __syncthreads();

    // ---- DFT: wave w owns bins [64w, 64w+64) as two 32-bin tiles, re and im accumulators
    const int j = lane & 31, kk = lane >> 5;
    float16v re0 = {0}, im0 = {0}, re1 = {0}, im1 = {0};
    const int f0 = wave * 64;
    const bool second = (f0 + 32) < 224;     // bins >= 224 are all padding
    if (f0 < 224) {
        const float* bc = t.basis_cos + f0 + j;
        const float* bs = t.basis_sin + f0 + j;
#pragma unroll 4
        for (int n0 = 0; n0 < kNfft; n0 += 2) {
            const int n = n0 + kk;
            const float fb = lds[j * FLD + n];
            const float c0 = bc[n * kFreqPad], s0 = bs[n * kFreqPad];
            re0 = __builtin_amdgcn_mfma_f32_32x32x2f32(c0, fb, re0, 0, 0, 0);
            im0 = __builtin_amdgcn_mfma_f32_32x32x2f32(s0, fb, im0, 0, 0, 0);
            if (second) {
                const float c1 = bc[n * kFreqPad + 32], s1 = bs[n * kFreqPad + 32];
                re1 = __builtin_amdgcn_mfma_f32_32x32x2f32(c1, fb, re1, 0, 0, 0);
                im1 = __builtin_amdgcn_mfma_f32_32x32x2f32(s1, fb, im1, 0, 0, 0);
            }
        }
    }
    __syncthreads();                          // every wave is done reading F
    // ---- power spectrum into P[bin][frame]; accumulator row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
        lds[(f0 + row) * PLD + j] = re0[r] * re0[r] + im0[r] * im0[r];
        lds[(f0 + 32 + row) * PLD + j] = re1[r] * re1[r] + im1[r] * im1[r];
    }
    __syncthreads();

    // ---- mel filters: wave w owns mel rows [32w, 32w+32)
    float mx = -3.0e38f;
    const int m0 = wave * 32;
    if (m0 < t.n_mels_pad) {
        float16v acc = {0};
        const float* fl = t.filt + m0 + j;
#pragma unroll 2
        for (int k0 = 0; k0 < 202; k0 += 2) {
            const int k = k0 + kk;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fl[k * t.n_mels_pad], lds[k * PLD + j], acc, 0, 0, 0);
        }
        const int tt = t0 + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
            if (m < t.n_mels && tt < kFrames) {
                const float v = log10f(fmaxf(acc[r], 1e-10f));
                logmel[((long)b * t.n_mels + m) * kFrames + tt] = v;
                mx = fmaxf(mx, v);
            }
        }
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    if (tid == 0) blockmax[b * gridDim.x + blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// grid (94, B), 256 threads: 32 frames x n_mels per block.
__global__ __launch_bounds__(256) void mel_norm_kernel(float* __restrict__ logmel, const float* __restrict__ blockmax,
                                                       int n_mels, int nblk, f16* __restrict__ melT, int do_clamp) {
    __shared__ float tile[128 * 33];
    __shared__ float gmax_s;
    const int b = blockIdx.y, t0 = blockIdx.x * 32, tid = threadIdx.x;
    if (do_clamp) {
        if (tid < 64) {
            float m = -3.0e38f;
            for (int i = tid; i < nblk; i += 64) m = fmaxf(m, blockmax[b * nblk + i]);
            m = wave_max(m);
            if (tid == 0) gmax_s = m;
        }
        __syncthreads();
    }
    const float floor_v = do_clamp ? gmax_s - 8.0f : 0.f;
    for (int e = tid; e < n_mels * 32; e += 256) {
        const int m = e >> 5, i = e & 31;
        const int tt = t0 + i;
        float v = 0.f;
        if (tt < kFrames) {
            const long idx = ((long)b * n_mels + m) * kFrames + tt;
            v = logmel[idx];
            if (do_clamp) {
                v = (fmaxf(v, floor_v) + 4.0f) / 4.0f;
                logmel[idx] = v;
            }
        }
        tile[m * 33 + i] = v;
    }
    __syncthreads();
    for (int e = tid; e < n_mels * 32; e += 256) {
        const int i = e / n_mels, m = e - i * n_mels;
        const int tt = t0 + i;
        if (tt < kFrames) melT[((long)b * (kFrames + 2) + tt + 1) * n_mels + m] = (f16)tile[m * 33 + i];
    }
}

}  // namespace

void launch_mel_dft(const float* pcm, int B, const MelTables& t, float* logmel, float* blockmax, hipStream_t s) {
    dim3 grid((kFrames + FR - 1) / FR, B);
    hipLaunchKernelGGL(mel_dft_kernel, grid, dim3(256), 0, s, pcm, t, logmel, blockmax);
}

void launch_mel_norm(float* logmel, const float* blockmax, int B, int n_mels, f16* melT, hipStream_t s) {
    dim3 grid((kFrames + 31) / 32, B);
    hipLaunchKernelGGL(mel_norm_kernel, grid, dim3(256), 0, s, logmel, blockmax, n_mels, (int)grid.x, melT, 1);
}

void launch_mel_import(const float* mel, int B, int n_mels, f16* melT, hipStream_t s) {
    dim3 grid((kFrames + 31) / 32, B);
    hipLaunchKernelGGL(mel_norm_kernel, grid, dim3(256), 0, s, const_cast<float*>(mel), (const float*)nullptr, n_mels,
                       (int)grid.x, melT, 0);
}

}  // namespace mw
