// Developer harness: checks launch_gemm against a naive GPU product on random fp16 data and times it.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I manual-whisper_amd/csrc tools/gemm_bench.hip \
//         manual-whisper_amd/csrc/kernels_gemm.o -o tools/gemm_bench
//   tools/gemm_bench M N K [iters] [f8|gelu|qkv|resid]
// "gelu", "qkv" (head split with transposed V, T = 1500) and "resid" (fp32 read-modify-write) time the encoder's other
// epilogues; only the plain and "gelu" forms are checked against the reference product.
// With "f8" both operands are random OCP e4m3 codes with random row scales, the epilogue is the MLP-in one
// (scales, bias, GELU), and the reference decodes the codes in software.
#include "mw_kernels.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

using namespace mw;
#ifdef MW_GEMM_STAMPS
namespace mw { void gemm_debug_stamps(unsigned long long* dst, int n); }
#endif

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__global__ void fill_kernel(f16* x, long n, unsigned seed, float scale) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned h = (unsigned)i ^ seed;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    x[i] = (f16)(((h >> 8) * (1.0f / 16777216.0f) - 0.5f) * 2.0f * scale);
}

__global__ void ref_kernel(const f16* A, const f16* W, const float* bias, float* C, int M, int N, int K, const int* rows, int nrows) {
    const int r = blockIdx.y, n = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows || n >= N) return;
    const int m = rows[r];
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += (float)A[(long)m * K + k] * (float)W[(long)n * K + k];
    C[(long)r * N + n] = s + bias[n];
}

__global__ void fill8_kernel(uint8_t* x, long n, unsigned seed) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned h = (unsigned)i ^ seed;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    unsigned c = h & 0xff;
    if ((c & 0x7f) == 0x7f) c ^= 1;          // 0x7f / 0xff are NaN in e4m3fn
    x[i] = (uint8_t)c;
}

__device__ float e4m3_decode(uint8_t c) {       // software decode: 1-4-3, bias 7, subnormals k * 2^-9
    const int e = (c >> 3) & 15, m = c & 7;
    const float v = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m * 0.125f, e - 7);
    return (c & 0x80) ? -v : v;
}

__global__ void ref8_kernel(const uint8_t* A, const uint8_t* W, const float* as, const float* ws, const float* bias, float* C,
                            int M, int N, int K, const int* rows, int nrows) {
    const int r = blockIdx.y, n = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows || n >= N) return;
    const int m = rows[r];
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += e4m3_decode(A[(long)m * K + k]) * e4m3_decode(W[(long)n * K + k]);
    const float x = s * (as[m] * ws[n]) + bias[n];
    C[(long)r * N + n] = 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 24000, N = argc > 2 ? atoi(argv[2]) : 1280, K = argc > 3 ? atoi(argv[3]) : 1280;
    const int iters = argc > 4 ? atoi(argv[4]) : 20;
    const bool f8 = argc > 5 && !strcmp(argv[5], "f8");
    const char* mode = argc > 5 ? argv[5] : "";
    f16 *A, *W, *C; float *bias, *Cref, *as = nullptr, *ws = nullptr; int* rows_d;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 2));
    CK(hipMalloc(&bias, (size_t)N * 4));
    if (f8) {
        fill8_kernel<<<(unsigned)(((long)M * K + 255) / 256), 256>>>((uint8_t*)A, (long)M * K, 1u);
        fill8_kernel<<<(unsigned)(((long)N * K + 255) / 256), 256>>>((uint8_t*)W, (long)N * K, 2u);
        std::vector<float> ha(M), hw(N);
        for (int i = 0; i < M; ++i) ha[i] = 0.002f + 1e-5f * (i % 97);
        for (int i = 0; i < N; ++i) hw[i] = 0.0005f + 1e-6f * (i % 53);
        CK(hipMalloc(&as, (size_t)M * 4)); CK(hipMalloc(&ws, (size_t)N * 4));
        CK(hipMemcpy(as, ha.data(), (size_t)M * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(ws, hw.data(), (size_t)N * 4, hipMemcpyHostToDevice));
    } else {
        fill_kernel<<<(unsigned)(((long)M * K + 255) / 256), 256>>>(A, (long)M * K, 1u, 1.0f);
        fill_kernel<<<(unsigned)(((long)N * K + 255) / 256), 256>>>(W, (long)N * K, 2u, 0.05f);
    }
    std::vector<float> hb(N);
    for (int i = 0; i < N; ++i) hb[i] = 0.01f * (i % 17);
    CK(hipMemcpy(bias, hb.data(), (size_t)N * 4, hipMemcpyHostToDevice));
    CK(hipMemset(C, 0xff, (size_t)M * N * 2));
    GemmParams p{};
    p.A = A; p.lda = K; p.a_rows_per_batch = M; p.a_batch_stride = 0; p.W = W; p.ldw = K; p.M = M; p.N = N; p.K = K;
    p.bias = bias; p.scale = 1.f; p.scale_ncols = 0; p.gelu = 0; p.out_mode = OUT_F16; p.C = C; p.ldc = N;
    p.c_rows_per_batch = M; p.c_batch_stride = 0;
    if (f8) { p.f8 = 1; p.a_scale = as; p.w_scale = ws; p.gelu = 1; }
    f16 *q = nullptr, *kk = nullptr, *vT = nullptr; float* resid = nullptr;
    const bool timing_only = false;
    const bool qkv = !strcmp(mode, "qkv"), rsd = !strcmp(mode, "resid");
    if (!strcmp(mode, "gelu")) p.gelu = 1;
    if (!strcmp(mode, "qkv")) {                       // N = 3 D, M = B * 1500
        const int D = N / 3, T = 1500, Tpad = 1504, B = (M + T - 1) / T;
        CK(hipMalloc(&q, (size_t)B * T * D * 2)); CK(hipMalloc(&kk, (size_t)B * T * D * 2)); CK(hipMalloc(&vT, (size_t)B * D * Tpad * 2));
        p.out_mode = OUT_QKV_SPLIT; p.q = q; p.k = kk; p.v = vT; p.T = T; p.Tpad = Tpad; p.H = D / 64; p.D = D; p.scale = 0.125f; p.scale_ncols = D;
    }
    if (!strcmp(mode, "resid")) {
        CK(hipMalloc(&resid, (size_t)M * N * 4)); CK(hipMemset(resid, 0, (size_t)M * N * 4));
        p.out_mode = OUT_RESID_F32; p.C = resid;
    }
    hipStream_t s; CK(hipStreamCreate(&s));
    if (getenv("MW_GB_LDA0")) p.lda = 0;             // every A row is row 0 (always cached): isolates the A operand's memory latency; the check then fails
    launch_gemm(p, s);
    CK(hipStreamSynchronize(s));
    long bad = 0;
    if (!timing_only) {
    // reference on a spread of rows (first, last, tile edges, random)
    std::vector<int> rows;
    for (int m : {0, 1, 63, 64, 127, 128, 255, 256, 257, M / 2, M - 257, M - 256, M - 129, M - 2, M - 1}) if (m >= 0 && m < M) rows.push_back(m);
    for (int i = 0; i < 113; ++i) rows.push_back((int)((1103515245u * (unsigned)(i + 7) + 12345u) % (unsigned)M));
    const int nrows = (int)rows.size();
    CK(hipMalloc(&rows_d, nrows * 4)); CK(hipMalloc(&Cref, (size_t)nrows * N * 4));
    CK(hipMemcpy(rows_d, rows.data(), nrows * 4, hipMemcpyHostToDevice));
    if (f8) ref8_kernel<<<dim3((N + 127) / 128, nrows), 128>>>((uint8_t*)A, (uint8_t*)W, as, ws, bias, Cref, M, N, K, rows_d, nrows);
    else ref_kernel<<<dim3((N + 127) / 128, nrows), 128>>>(A, W, bias, Cref, M, N, K, rows_d, nrows);
    std::vector<float> href((size_t)nrows * N);
    std::vector<f16> hc((size_t)M * N);
    CK(hipMemcpy(href.data(), Cref, href.size() * 4, hipMemcpyDeviceToHost));
    if (qkv) {                                            // gather q | k | v back into [M][N]
        const int D = N / 3, T = p.T, Tpad = p.Tpad, H = p.H, B = (M + T - 1) / T;
        std::vector<f16> hq((size_t)B * T * D), hk((size_t)B * T * D), hv((size_t)B * D * Tpad);
        CK(hipMemcpy(hq.data(), q, hq.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hk.data(), kk, hk.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hv.data(), vT, hv.size() * 2, hipMemcpyDeviceToHost));
        for (int m : rows) {
            const int b = m / T, t = m % T;
            for (int n = 0; n < N; ++n) {
                const int sec = n / D, nn = n % D, hh = nn / 64, e = nn % 64;
                f16 v;
                if (sec == 2) v = hv[((size_t)(b * H + hh) * 64 + e) * Tpad + t];
                else v = (sec == 0 ? hq : hk)[((size_t)(b * H + hh) * T + t) * 64 + e];
                hc[(size_t)m * N + n] = v;
            }
        }
    } else if (rsd) {                                     // the accumulator started at zero: one launch leaves the product
        std::vector<float> hr((size_t)M * N);
        CK(hipMemcpy(hr.data(), resid, hr.size() * 4, hipMemcpyDeviceToHost));
        for (int m : rows) for (int n = 0; n < N; ++n) hc[(size_t)m * N + n] = (f16)hr[(size_t)m * N + n];
    } else {
        CK(hipMemcpy(hc.data(), C, hc.size() * 2, hipMemcpyDeviceToHost));
    }
    double maxerr = 0, maxref = 0;
    for (int r = 0; r < nrows; ++r)
        for (int n = 0; n < N; ++n) {
            double ref = href[(size_t)r * N + n];
            if (!f8 && p.gelu) ref = 0.5 * ref * (1.0 + erf(ref * 0.70710678118654752));
            if (qkv && n < N / 3) ref *= p.scale;
            const double got = (double)(float)hc[(size_t)rows[r] * N + n];
            const double err = fabs(ref - got);
            if (!(err <= 2e-3 * fabs(ref) + 2e-3)) { if (bad < 5) printf("  mismatch m=%d n=%d ref=%f got=%f\n", rows[r], n, ref, got); ++bad; }
            maxerr = fmax(maxerr, err); maxref = fmax(maxref, fabs(ref));
        }
    printf("check: %d rows x %d cols, max |err| %.5f (max |ref| %.3f), mismatches %ld\n", nrows, N, maxerr, maxref, bad);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch_gemm(p, s);
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) launch_gemm(p, s);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
#ifdef MW_GEMM_STAMPS
    {   // per workgroup of the last launch: start -> first tile landed -> main loop done -> stores drained (10 ns ticks)
        const int nt = ((M + 255) / 256) * ((N + 255) / 256);
        std::vector<unsigned long long> st((size_t)nt * 4);
        gemm_debug_stamps(st.data(), nt * 4);
        unsigned long long t0 = ~0ull, t1 = 0;
        double pro = 0, mainl = 0, epi = 0;
        for (int i = 0; i < nt; ++i) {
            t0 = st[i * 4] < t0 ? st[i * 4] : t0; t1 = st[i * 4 + 3] > t1 ? st[i * 4 + 3] : t1;
            pro += (double)(st[i * 4 + 1] - st[i * 4]); mainl += (double)(st[i * 4 + 2] - st[i * 4 + 1]); epi += (double)(st[i * 4 + 3] - st[i * 4 + 2]);
        }
        printf("stamps: %d workgroups, kernel span %.1f us; mean prologue %.2f us, main loop %.2f us, epilogue + drain %.2f us\n", nt,
               (t1 - t0) * 0.01, pro / nt * 0.01, mainl / nt * 0.01, epi / nt * 0.01);
        // start times of the first 2 rounds, to see whether rounds run in lock step
        std::vector<double> starts;
        for (int i = 0; i < nt; ++i) starts.push_back((st[i * 4] - t0) * 0.01);
        std::sort(starts.begin(), starts.end());
        printf("  start time quantiles (us):");
        for (int q = 0; q <= 20; ++q) printf(" %.0f", starts[(size_t)(q * (nt - 1) / 20)]);
        printf("\n");
    }
#endif
    printf("M=%d N=%d K=%d: %.1f us, %.1f TFLOP/s\n", M, N, K, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
    if (getenv("MW_GB_DUAL") && !qkv) {
        // the same product on two streams at once (second output buffer): time per launch when two launches share the GPU
        GemmParams p2 = p;
        void* c2 = nullptr;
        CK(hipMalloc(&c2, (size_t)M * N * (rsd ? 4 : 2))); CK(hipMemset(c2, 0, (size_t)M * N * (rsd ? 4 : 2)));
        p2.C = (f16*)c2;
        hipStream_t s2; CK(hipStreamCreate(&s2));
        hipEvent_t f0, f1; CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
        for (int i = 0; i < 3; ++i) { launch_gemm(p, s); launch_gemm(p2, s2); }
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, s)); CK(hipEventRecord(f0, s2));
        for (int i = 0; i < iters; ++i) { launch_gemm(p, s); launch_gemm(p2, s2); }
        CK(hipEventRecord(e1, s)); CK(hipEventRecord(f1, s2));
        CK(hipDeviceSynchronize());
        float ma, mb, span; CK(hipEventElapsedTime(&ma, e0, e1)); CK(hipEventElapsedTime(&mb, f0, f1));
        CK(hipEventElapsedTime(&span, e0, f1));
        span = fmaxf(span, ma);
        printf("  two streams: %.1f / %.1f us per launch on each, %.1f us per launch overall (%.1f TFLOP/s)\n", ma / iters * 1e3, mb / iters * 1e3,
               span / (2 * iters) * 1e3, 2.0 * M * N * K / (span / (2 * iters) * 1e-3) / 1e12);
    }
    return bad ? 2 : 0;
}
