// Developer harness: what does an in-launch producer -> consumer hand-off cost against a kernel boundary?
// A chain of 32 (producer, projection) pairs, as separate launches and as one launch per pair in which the
// projection workgroups load their weights, then wait for the producers' arrival counter, then read the
// activations with agent-scope loads.  Shapes of the decoder's self-attention -> output projection pair
// at batch 16 (320 producers of 128 bytes each, N = K = 1280).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 f16;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr int B = 16, N = 1280, K = 1280, NV = 8, NW = 8, KPW = 5, NPROD = 320;
constexpr int NCTR = 16;       // arrival counters per hand-off (one 64-byte line): a single word serialises 320 atomics

__global__ void fill16(f16* x, long n, unsigned seed, float scale) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned h = (unsigned)i ^ seed;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    x[i] = (f16)(((h >> 8) * (1.0f / 16777216.0f) - 0.5f) * 2.0f * scale);
}

// producer: (sequence, head) block writes its 64 fp16 outputs; `src` stands for the attention result
__device__ __forceinline__ void producer_body(const f16* src, f16* act, int bh, bool publish, int* counter) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave != 0) return;
    // two halves per lane pair: lanes 0..31 store 4 bytes each
    const float v = (float)src[bh * 64 + lane] * 1.0009765625f;
    const float hi = __shfl_down(v, 1);
    if (publish) {
        if ((lane & 1) == 0) {
            union { f16 h[2]; unsigned u; } pk;
            pk.h[0] = (f16)v; pk.h[1] = (f16)hi;
            __hip_atomic_store(reinterpret_cast<unsigned*>(act + bh * 64 + lane), pk.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(counter + (bh % NCTR) * 32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        act[bh * 64 + lane] = (f16)v;
    }
}

// projection: x[b][n] += sum_k act[b][k] W[n][k]; 8 rows per workgroup, 8 waves split K
template <bool WAIT>
__device__ __forceinline__ void consumer_body(const f16* W, const f16* act, float* x, int blk, int* counter, int* timeouts) {
    __shared__ float4v red[NW][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int n0 = blk * NV;
    const f16* wp = W + (long)(n0 + (fr & 7)) * K + fg * 8;
    uint4 wreg[KPW];
#pragma unroll
    for (int c = 0; c < KPW; ++c) wreg[c] = *reinterpret_cast<const uint4*>(wp + (wave + NW * c) * 32);
    const int nl = 4 * fg, n = n0 + nl;
    const bool owner = wave == 0 && nl < NV;
    float4 res4 = make_float4(0, 0, 0, 0);
    if (owner) res4 = *reinterpret_cast<const float4*>(x + (long)fr * N + n);
    half8 xf[KPW];
    const f16* xr = act + (long)fr * K + fg * 8;
    if constexpr (WAIT) {
        if (tid < NCTR) {         // one wave polls the 16 words of the counter line
            int spins = 0;
            while (!__all(__hip_atomic_load(counter + tid * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= NPROD / NCTR)) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 22)) { if (tid == 0) atomicAdd(timeouts, 1); break; }
            }
        }
        __syncthreads();
        // 16-byte loads that bypass this CU's L1 (agent scope).  hipcc neither counts nor waits for loads inside
        // asm, so the loads and their wait are ONE statement: the results exist for the compiler only after it.
        static_assert(KPW == 5, "five loads spelled out");
        uint4 u0, u1, u2, u3, u4;
        const f16* a0 = xr + (wave + NW * 0) * 32; const f16* a1 = xr + (wave + NW * 1) * 32; const f16* a2 = xr + (wave + NW * 2) * 32;
        const f16* a3 = xr + (wave + NW * 3) * 32; const f16* a4 = xr + (wave + NW * 4) * 32;
        asm volatile(
            "global_load_dwordx4 %0, %5, off sc1\n\t"
            "global_load_dwordx4 %1, %6, off sc1\n\t"
            "global_load_dwordx4 %2, %7, off sc1\n\t"
            "global_load_dwordx4 %3, %8, off sc1\n\t"
            "global_load_dwordx4 %4, %9, off sc1\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3), "=&v"(u4)
            : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4)
            : "memory");
        xf[0] = *reinterpret_cast<half8*>(&u0); xf[1] = *reinterpret_cast<half8*>(&u1); xf[2] = *reinterpret_cast<half8*>(&u2);
        xf[3] = *reinterpret_cast<half8*>(&u3); xf[4] = *reinterpret_cast<half8*>(&u4);
    } else {
#pragma unroll
        for (int c = 0; c < KPW; ++c) xf[c] = *reinterpret_cast<const half8*>(xr + (wave + NW * c) * 32);
    }
    float4v acc = {0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < KPW; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<half8*>(&wreg[c]), xf[c], acc, 0, 0, 0);
    red[wave][lane] = acc;
    __syncthreads();
    if (!owner) return;
    float4v v = red[0][lane];
    for (int w = 1; w < NW; ++w) v += red[w][lane];
    *reinterpret_cast<float4*>(x + (long)fr * N + n) = make_float4(res4.x + v[0], res4.y + v[1], res4.z + v[2], res4.w + v[3]);
}

__global__ __launch_bounds__(512) void producer_kernel(const f16* src, f16* act) { producer_body(src, act, blockIdx.x, false, nullptr); }
__global__ __launch_bounds__(512) void consumer_kernel(const f16* W, const f16* act, float* x) { consumer_body<false>(W, act, x, blockIdx.x, nullptr, nullptr); }
__global__ __launch_bounds__(512) void fused_kernel(const f16* src, f16* act, const f16* W, float* x, int* counter, int* timeouts) {
    if (blockIdx.x < NPROD) producer_body(src, act, blockIdx.x, true, counter);
    else consumer_body<true>(W, act, x, blockIdx.x - NPROD, counter, timeouts);
}

int main() {
    const int L = 32;
    f16 *W, *src, *act; float *x, *xref; int *counters, *timeouts;
    CK(hipMalloc(&W, (size_t)L * N * K * 2)); CK(hipMalloc(&src, (size_t)B * K * 2)); CK(hipMalloc(&act, (size_t)B * K * 2));
    CK(hipMalloc(&x, (size_t)B * N * 4)); CK(hipMalloc(&xref, (size_t)B * N * 4)); CK(hipMalloc(&counters, L * 4 * 16 * 32)); CK(hipMalloc(&timeouts, 4));
    fill16<<<(unsigned)(((long)L * N * K + 255) / 256), 256>>>(W, (long)L * N * K, 1u, 0.02f);
    fill16<<<(B * K + 255) / 256, 256>>>(src, (long)B * K, 2u, 1.0f);
    CK(hipMemset(timeouts, 0, 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    auto separate = [&](float* out) {
        for (int l = 0; l < L; ++l) {
            producer_kernel<<<NPROD, 512, 0, s>>>(src, act);
            consumer_kernel<<<N / NV, 512, 0, s>>>(W + (size_t)l * N * K, act, out);
        }
    };
    auto fused = [&](float* out) {
        (void)hipMemsetAsync(counters, 0, L * 4 * 16 * 32, s);
        for (int l = 0; l < L; ++l)
            fused_kernel<<<NPROD + N / NV, 512, 0, s>>>(src, act, W + (size_t)l * N * K, out, counters + l * 16 * 32, timeouts);
    };
    CK(hipMemsetAsync(xref, 0, (size_t)B * N * 4, s)); separate(xref);
    CK(hipMemsetAsync(x, 0, (size_t)B * N * 4, s)); fused(x);
    CK(hipStreamSynchronize(s));
    static float ha[B * N], hb[B * N];
    CK(hipMemcpy(ha, xref, sizeof(ha), hipMemcpyDeviceToHost)); CK(hipMemcpy(hb, x, sizeof(hb), hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < B * N; ++i) bad += ha[i] != hb[i];
    int tmo; CK(hipMemcpy(&tmo, timeouts, 4, hipMemcpyDeviceToHost));
    printf("fused == separate: %s (%d differing), spin timeouts %d\n", bad ? "NO" : "yes", bad, tmo);
    for (int mode = 0; mode < 2; ++mode) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        if (mode) fused(x); else separate(x);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e0, s));
        const int reps = 20;
        for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: %.2f us per (producer, projection) pair\n", mode ? "one launch, in-launch hand-off" : "two launches", ms * 1e3 / (reps * L));
    }
    CK(hipMemcpy(&tmo, timeouts, 4, hipMemcpyDeviceToHost));
    printf("spin timeouts after timing: %d\n", tmo);
    return 0;
}
