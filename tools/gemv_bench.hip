// Developer harness: times one decode projection (launch_gemv) the way a decode step runs it:
// 32 back-to-back launches in a captured graph, each with its own weight matrix (cold in cache).
//   tools/gemv_bench N K mode [B]      mode: 0 plain->fp16, 1 plain->residual, 2 LN-fused->fp16, 3 LN-fused + GELU
#include "mw_kernels.h"

#include <stdio.h>
#include <stdlib.h>
#include <vector>

using namespace mw;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__global__ void fill16(f16* x, long n, unsigned seed, float scale) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned h = (unsigned)i ^ seed;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    x[i] = (f16)(((h >> 8) * (1.0f / 16777216.0f) - 0.5f) * 2.0f * scale);
}
__global__ void fill32(float* x, long n, unsigned seed, float scale, float offset) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned h = (unsigned)i ^ seed;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    x[i] = ((h >> 8) * (1.0f / 16777216.0f) - 0.5f) * 2.0f * scale + offset;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 1280, K = argc > 2 ? atoi(argv[2]) : 1280, mode = argc > 3 ? atoi(argv[3]) : 0;
    const int B = argc > 4 ? atoi(argv[4]) : 16;
    const int L = 32;                                     // launches per graph
    const int NWT = getenv("HOT") ? 1 : L;               // HOT=1: every launch reads the same matrix (cache-resident weights)
    dec_kernels_prepare();
    f16 *W, *xin16, *out16, *lnscr; float *x32, *out32, *bias, *g, *b, *skp, *dgp; int* skt; DecState* st;
    CK(hipMalloc(&W, (size_t)L * N * K * 2)); CK(hipMalloc(&xin16, (size_t)B * K * 2));
    CK(hipMalloc(&out16, (size_t)B * N * 2)); CK(hipMalloc(&lnscr, (size_t)B * K * 2));
    CK(hipMalloc(&x32, (size_t)B * K * 4)); CK(hipMalloc(&out32, (size_t)B * N * 4)); CK(hipMalloc(&bias, (size_t)N * 4));
    CK(hipMalloc(&g, (size_t)K * 4)); CK(hipMalloc(&b, (size_t)K * 4));
    const int groups16 = (B + 15) / 16;
    CK(hipMalloc(&skp, (size_t)splitk_partial_floats(N) * groups16 * 4)); CK(hipMalloc(&skt, (size_t)splitk_groups(N) * groups16 * 4)); CK(hipMalloc(&st, sizeof(DecState)));
    CK(hipMalloc(&dgp, (size_t)decgemm_partial_floats(N, N) * 4));
    CK(hipMemset(skt, 0, (size_t)splitk_groups(N) * groups16 * 4)); CK(hipMemset(st, 0, sizeof(DecState))); CK(hipMemset(out32, 0, (size_t)B * N * 4));
    fill16<<<(unsigned)(((long)L * N * K + 255) / 256), 256>>>(W, (long)L * N * K, 1u, 0.03f);
    fill16<<<(B * K + 255) / 256, 256>>>(xin16, (long)B * K, 2u, 1.0f);
    fill32<<<(B * K + 255) / 256, 256>>>(x32, (long)B * K, 3u, 1.0f, 0.f);
    fill32<<<(N + 255) / 256, 256>>>(bias, N, 4u, 0.1f, 0.f);
    fill32<<<(K + 255) / 256, 256>>>(g, K, 5u, 0.1f, 1.f);
    fill32<<<(K + 255) / 256, 256>>>(b, K, 6u, 0.1f, 0.f);
    CK(hipDeviceSynchronize());
    hipStream_t s; CK(hipStreamCreate(&s));
    auto enqueue = [&]() {
        for (int l = 0; l < L; ++l) {
            GemvParams p{};
            p.B = B; p.N = N; p.K = K; p.W = W + (size_t)(l % NWT) * N * K; p.bias = bias; p.scale = 1.f; p.nv = 16; p.st = st;
            p.H = N / 64; p.D = N; p.Tctx = 448; p.ln_scratch = lnscr;
            p.sk_partial = skp; p.sk_partial_floats = splitk_partial_floats(N); p.sk_tickets = skt; p.sk_groups = splitk_groups(N);
            p.dg_partial = dgp; p.dg_partial_floats = decgemm_partial_floats(N, N); p.dg_groups = splitk_groups(N) * groups16;
            if (mode >= 2) { p.x32 = x32; p.ln_g = g; p.ln_b = b; p.gelu = mode == 3; p.out_mode = GV_F16; p.out16 = out16; }
            else if (mode == 1) { p.xin16 = xin16; p.out_mode = GV_RESID; p.out32 = out32; }
            else { p.xin16 = xin16; p.out_mode = GV_F16; p.out16 = out16; }
            launch_gemv(p, s);
        }
    };
    enqueue();
    CK(hipStreamSynchronize(s));
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    enqueue();
    CK(hipStreamEndCapture(s, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(exec, s));
    const int reps = 20;
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(exec, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / (reps * L);
    printf("N=%d K=%d mode=%d B=%d: %.2f us per launch (incl. boundary), weights %.2f TB/s\n", N, K, mode, B, us, (double)N * K * 2 / (us * 1e-6) / 1e12);
    return 0;
}
