// Encoder attention (launch_enc_attention) against a naive softmax(QK^T)V on a sample of query rows, and its time
// at the large-v3 batch-16 shape; built by manual_whisper_amd.build and run by tests/test_gpu_attention.py.
//   tools/attn_bench [B] [H] [T] [mode] [timed iterations]
// mode "rising": the scores of every query grow by ~12 (log2 units) per tile of 64 keys, so the kernel's reference
// exponent has to be raised at almost every tile; "varlen": batch element b has its own number of keys (the packed
// form the forced aligner uses); "plain" (default): uniform random operands.
#include "mw_kernels.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>

using namespace mw;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

// x[i] = uniform(-scale, scale) + offset + ramp * (row of a [.., T, 64] tensor) / T
__global__ void fill16(f16* x, long n, unsigned seed, float scale, float offset = 0.f, float ramp = 0.f, int T = 1) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned h = (unsigned)i ^ seed;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    x[i] = (f16)(((h >> 8) * (1.0f / 16777216.0f) - 0.5f) * 2.0f * scale + offset + ramp * (float)((i / 64) % T) / (float)T);
}

// one block per sampled (bh, query): naive attention in fp32
__global__ void ref_kernel(const f16* q, const f16* k, const f16* vT, float* out, int T, int Tpad, const int* rows, int H,
                           const int* keys_of) {
    __shared__ float p[2048];
    __shared__ float red[256];
    const int bh = rows[2 * blockIdx.x], qi = rows[2 * blockIdx.x + 1], tid = threadIdx.x;
    const f16* qr = q + ((long)bh * T + qi) * 64;
    const int Tstride = T;
    if (keys_of) T = keys_of[bh / H];           // rows stay Tstride apart
    k += (long)bh * (Tstride - T) * 64;         // so that k[(bh * T + t) * 64] below is row t of this (batch, head)
    float mx = -1e30f;
    for (int t = tid; t < T; t += 256) {
        float s = 0.f;
        for (int d = 0; d < 64; ++d) s += (float)qr[d] * (float)k[((long)bh * T + t) * 64 + d];
        p[t] = s;
        mx = fmaxf(mx, s);
    }
    red[tid] = mx; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]); __syncthreads(); }
    mx = red[0]; __syncthreads();
    float sum = 0.f;
    for (int t = tid; t < T; t += 256) { p[t] = expf(p[t] - mx); sum += p[t]; }
    red[tid] = sum; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    sum = red[0];
    if (tid < 64) {
        float o = 0.f;
        for (int t = 0; t < T; ++t) o += p[t] * (float)vT[((long)bh * 64 + tid) * Tpad + t];
        out[blockIdx.x * 64 + tid] = o / sum;
    }
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 16, H = argc > 2 ? atoi(argv[2]) : 20, T = argc > 3 ? atoi(argv[3]) : 1500;
    const std::string mode = argc > 4 ? argv[4] : "plain";
    const int iters = argc > 5 ? atoi(argv[5]) : 50;
    if (B < 1 || H < 1 || T < 8 || T > 2048 || iters < 0 || (mode != "plain" && mode != "rising" && mode != "varlen")) {
        printf("usage: attn_bench [B] [H] [T <= 2048] [plain|rising|varlen] [iterations]\n");
        return 1;
    }
    const int Tpad = (T + 7) / 8 * 8;
    const long nqk = (long)B * H * T * 64, nv = (long)B * H * 64 * Tpad;
    f16 *q, *k, *vT, *out; float* ref; int* rows_d;
    CK(hipMalloc(&q, nqk * 2)); CK(hipMalloc(&k, nqk * 2)); CK(hipMalloc(&vT, nv * 2 + 256)); CK(hipMalloc(&out, nqk * 2));
    const bool rising = mode == "rising";
    // q already carries the head_dim^-0.5 scale.  rising: q . k gains 64 * 0.25 * 12 * t / T, i.e. ~12 log2 units per 64 keys at T = 1500
    fill16<<<(unsigned)((nqk + 255) / 256), 256>>>(q, nqk, 1u, 0.7f, rising ? 0.25f : 0.f);
    fill16<<<(unsigned)((nqk + 255) / 256), 256>>>(k, nqk, 2u, 2.0f, 0.f, rising ? 12.0f * T / 1500.0f : 0.f, T);
    fill16<<<(unsigned)((nv + 255) / 256), 256>>>(vT, nv, 3u, 1.0f);
    CK(hipMemset(out, 0xff, nqk * 2));
    hipStream_t s; CK(hipStreamCreate(&s));
    std::vector<int> keys(B, T);
    int* keys_d = nullptr;
    if (mode == "varlen") {
        for (int b = 0; b < B; ++b) keys[b] = T - (int)((b * 97u + 13u) % (unsigned)(T / 2));
        CK(hipMalloc(&keys_d, B * 4));
        CK(hipMemcpy(keys_d, keys.data(), B * 4, hipMemcpyHostToDevice));
    }
    launch_enc_attention(q, k, vT, out, B, H, T, Tpad, s, keys_d);
    CK(hipStreamSynchronize(s));
    std::vector<int> rows;
    for (int i = 0; i < 96; ++i) {
        rows.push_back((int)((1103515245u * (unsigned)(i + 3) + 12345u) % (unsigned)(B * H)));
        const int Tb = keys[rows.back() / H];       // rows that exist for this batch element
        rows.push_back(i < 8 ? (i < 4 ? i : Tb - 1 - (i - 4)) : (int)((22695477u * (unsigned)(i + 11) + 1u) % (unsigned)Tb));
    }
    const int nr = (int)rows.size() / 2;
    CK(hipMalloc(&rows_d, rows.size() * 4)); CK(hipMalloc(&ref, (size_t)nr * 64 * 4));
    CK(hipMemcpy(rows_d, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
    ref_kernel<<<nr, 256>>>(q, k, vT, ref, T, Tpad, rows_d, H, keys_d);
    std::vector<float> href((size_t)nr * 64);
    std::vector<f16> hout((size_t)nqk);
    CK(hipMemcpy(href.data(), ref, href.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hout.data(), out, hout.size() * 2, hipMemcpyDeviceToHost));
    double maxerr = 0; long bad = 0;
    for (int r = 0; r < nr; ++r) {
        const int bh = rows[2 * r], qi = rows[2 * r + 1], b = bh / H, h = bh % H;
        for (int d = 0; d < 64; ++d) {
            const double want = href[(size_t)r * 64 + d], got = (double)(float)hout[((size_t)b * T + qi) * (H * 64) + h * 64 + d];
            const double err = fabs(want - got);
            if (!(err < 4e-3 + 4e-3 * fabs(want))) { if (bad < 5) printf("  mismatch bh=%d q=%d d=%d want %f got %f\n", bh, qi, d, want, got); ++bad; }
            maxerr = fmax(maxerr, err);
        }
    }
    printf("check: %d rows, max |err| %.5f, mismatches %ld\n", nr, maxerr, bad);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (iters == 0) return bad ? 2 : 0;
    for (int i = 0; i < 2; ++i) launch_enc_attention(q, k, vT, out, B, H, T, Tpad, s, keys_d);
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) launch_enc_attention(q, k, vT, out, B, H, T, Tpad, s, keys_d);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    printf("B=%d H=%d T=%d: %.1f us, %.1f TFLOP/s\n", B, H, T, ms * 1e3, 4.0 * B * H * (double)T * T * 64 / (ms * 1e-3) / 1e12);
    return bad ? 2 : 0;
}
