// Development aid: does an encoder product (MFMA-bound) overlap with the decode cross-attention of a decode group
// (HBM-bound) when both are in flight on one GPU, or do the two take turns?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I manual-whisper_amd/csrc tools/coreside_bench.hip \
//         manual-whisper_amd/csrc/kernels_gemm.o manual-whisper_amd/csrc/kernels_dec.o -o tools/coreside_bench
//   tools/coreside_bench [M N K [gemm_ms stream_ms [mode]]]
// The product is launch_gemm (MW_GEMM_NO256=1 / MW_GEMM_CS=1 pick the other tile kernels), mode "", "gelu" or "resid";
// the stream is launch_dec_cross_attn at 112 sequences x 20 heads x 1500 keys (cross_attn_stream_kernel, 860 MB per
// launch, MW_XSTREAM_LDS = its LDS request) cycling over 4 layers' caches.  Printed: each alone, each while the
// other runs all the time, and the makespan of `gemm_ms` + `stream_ms` of work enqueued on two streams at once against
// the sum of the two alone.
#include "mw_kernels.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

using namespace mw;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__global__ void fill_kernel(f16* x, long n, unsigned seed, float scale) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned h = (unsigned)i ^ seed;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    x[i] = (f16)(((h >> 8) * (1.0f / 16777216.0f) - 0.5f) * 2.0f * scale);
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int M = argc > 1 ? atoi(argv[1]) : 24000, N = argc > 2 ? atoi(argv[2]) : 5120, K = argc > 3 ? atoi(argv[3]) : 1280;
    const float gemm_ms = argc > 4 ? atof(argv[4]) : 45.f, stream_ms = argc > 5 ? atof(argv[5]) : 100.f;
    const char* mode = argc > 6 ? argv[6] : "";
    const int B = 112, H = 20, T = 1500, D = H * 64, L = 4;
    f16 *A, *W, *C; float* bias; float* resid = nullptr;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 2));
    CK(hipMalloc(&bias, (size_t)N * 4)); CK(hipMemset(bias, 0, (size_t)N * 4));
    fill_kernel<<<(unsigned)(((long)M * K + 255) / 256), 256>>>(A, (long)M * K, 1u, 1.0f);
    fill_kernel<<<(unsigned)(((long)N * K + 255) / 256), 256>>>(W, (long)N * K, 2u, 0.05f);
    GemmParams p{};
    p.A = A; p.lda = K; p.a_rows_per_batch = M; p.W = W; p.ldw = K; p.M = M; p.N = N; p.K = K;
    p.bias = bias; p.scale = 1.f; p.out_mode = OUT_F16; p.C = C; p.ldc = N; p.c_rows_per_batch = M;
    if (!strcmp(mode, "gelu")) p.gelu = 1;
    if (!strcmp(mode, "resid")) {
        CK(hipMalloc(&resid, (size_t)M * N * 4)); CK(hipMemset(resid, 0, (size_t)M * N * 4));
        p.out_mode = OUT_RESID_F32; p.C = resid;
    }
    const size_t kv_elems = (size_t)B * H * T * 64;
    f16 *q, *kc, *vc, *out;
    CK(hipMalloc(&q, (size_t)B * D * 2)); CK(hipMalloc(&out, (size_t)B * D * 2));
    CK(hipMalloc(&kc, kv_elems * 2 * L)); CK(hipMalloc(&vc, kv_elems * 2 * L));
    fill_kernel<<<(unsigned)(((long)B * D + 255) / 256), 256>>>(q, (long)B * D, 3u, 1.0f);
    fill_kernel<<<(unsigned)((kv_elems * L + 255) / 256), 256>>>(kc, (long)(kv_elems * L), 4u, 1.0f);
    fill_kernel<<<(unsigned)((kv_elems * L + 255) / 256), 256>>>(vc, (long)(kv_elems * L), 5u, 1.0f);
    CK(hipDeviceSynchronize());
    dec_kernels_prepare();
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t a0, a1, b0, b1;
    CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
    int layer = 0;
    auto gemm = [&](int n) { for (int i = 0; i < n; ++i) launch_gemm(p, s1); };
    auto stream = [&](int n) {
        for (int i = 0; i < n; ++i, layer = (layer + 1) % L)
            launch_dec_cross_attn(q, kc + layer * kv_elems, vc + layer * kv_elems, out, B, H, T, nullptr, nullptr, s2);
    };
    const double flop = 2.0 * M * N * K, sbytes = 2.0 * kv_elems * 2;
    float ms;
    // ---- each alone (sustained: the clock settles over the first tens of ms)
    gemm(100); CK(hipStreamSynchronize(s1));
    CK(hipEventRecord(a0, s1)); gemm(300); CK(hipEventRecord(a1, s1)); CK(hipStreamSynchronize(s1));
    CK(hipEventElapsedTime(&ms, a0, a1));
    const double g_us = ms * 1e3 / 300;
    printf("product alone   M=%d N=%d K=%d %s: %.1f us, %.1f TFLOP/s\n", M, N, K, mode, g_us, flop / g_us / 1e6);
    stream(50); CK(hipStreamSynchronize(s2));
    CK(hipEventRecord(b0, s2)); stream(300); CK(hipEventRecord(b1, s2)); CK(hipStreamSynchronize(s2));
    CK(hipEventElapsedTime(&ms, b0, b1));
    const double s_us = ms * 1e3 / 300;
    printf("stream alone    %d x %d x %d keys: %.1f us, %.1f GB/s\n", B, H, T, s_us, sbytes / s_us / 1e3);
    // ---- the product while the stream runs all the time, and the reverse
    {
        const int ng = (int)(60e3 / g_us) + 1, ns = (int)(400e3 / s_us) + 1;
        stream(20);
        CK(hipEventRecord(a0, s1)); gemm(ng); CK(hipEventRecord(a1, s1));
        stream(ns);
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, a0, a1));
        printf("product beside the stream: %.1f us per launch (%.2fx alone), %.1f TFLOP/s\n", ms * 1e3 / ng, ms * 1e3 / ng / g_us,
               flop / (ms * 1e3 / ng) / 1e6);
    }
    {
        const int ns = (int)(60e3 / s_us) + 1, ng = (int)(400e3 / g_us) + 1;
        gemm(20);
        CK(hipEventRecord(b0, s2)); stream(ns); CK(hipEventRecord(b1, s2));
        gemm(ng);
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, b0, b1));
        printf("stream beside the product: %.1f us per launch (%.2fx alone), %.1f GB/s\n", ms * 1e3 / ns, ms * 1e3 / ns / s_us,
               sbytes / (ms * 1e3 / ns) / 1e3);
    }
    // ---- a step's worth of each, enqueued at once: makespan against the sum of the two alone
    for (int rep = 0; rep < 2; ++rep) {
        const int ng = (int)(gemm_ms * 1e3 / g_us + 0.5), ns = (int)(stream_ms * 1e3 / s_us + 0.5);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a0, s1));
        CK(hipStreamWaitEvent(s2, a0, 0));
        // interleave the enqueues so that neither queue starts long before the other
        int ig = 0, is = 0;
        while (ig < ng || is < ns) {
            const int cg = ng - ig < 8 ? ng - ig : 8, cs = ns - is < 8 ? ns - is : 8;
            if (cg > 0) { gemm(cg); ig += cg; }
            if (cs > 0) { stream(cs); is += cs; }
        }
        CK(hipEventRecord(a1, s1)); CK(hipEventRecord(b1, s2));
        CK(hipDeviceSynchronize());
        float mg, msn; CK(hipEventElapsedTime(&mg, a0, a1)); CK(hipEventElapsedTime(&msn, a0, b1));
        const double alone = (ng * g_us + ns * s_us) * 1e-3, span = mg > msn ? mg : msn;
        printf("together: %d products + %d stream launches: product queue done at %.1f ms, stream queue at %.1f ms; makespan %.1f ms "
               "against %.1f ms one after the other (%.2fx)\n", ng, ns, mg, msn, span, alone, span / alone);
    }
    return 0;
}
