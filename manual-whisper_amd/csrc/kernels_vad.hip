// Speech-activity front end: frame log-energies of mono PCM, the input of the window search that
// replaces fixed 30 s chunks (the reference hands vad_options to whisperx, transcribe.py:43-46,112;
// whisperx runs a neural segmentation model there, which does not exist offline: an energy
// detector stands in for it, the hysteresis/merge logic above it is the host's).
//
//   out[f] = 10 * log10( mean_{i < frame, f*hop+i < n} x[f*hop + i]^2 + 1e-10 )
//
// HBM-bound and tiny: 4 B read per sample and hop, one wave per frame, 256-byte coalesced wave
// loads, shuffle reduction.  One hour of audio is 230 MB: ~50 us of streaming.
#include "mw_kernels.h"

namespace mw {

namespace {

__global__ __launch_bounds__(256) void frame_energy_kernel(const float* __restrict__ pcm, long n, int frame, int hop,
                                                           float* __restrict__ out, long n_frames) {
    const int lane = threadIdx.x & 63;
    const long f = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= n_frames) return;
    const long start = f * hop;
    const long avail = min((long)frame, n - start);          // >= 1: n_frames = ceil(n / hop)
    float s = 0.f;
    for (long i = lane; i < avail; i += 64) {
        const float v = pcm[start + i];
        s += v * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[f] = 10.0f * log10f(s / (float)avail + 1e-10f);
}

}  // namespace

void launch_frame_energy(const float* pcm, long n, int frame, int hop, float* out, long n_frames, hipStream_t s) {
    hipLaunchKernelGGL(frame_energy_kernel, dim3((unsigned)((n_frames + 3) / 4)), dim3(256), 0, s, pcm, n, frame, hop, out, n_frames);
}

}  // namespace mw
