// Development aid: what a kernel that reads L2-resident data (as a matrix product's tile loads do) and a kernel that
// only issues MFMAs lose beside a saturating HBM read stream, and what the stream loses beside them.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/fifo_probe.hip -o tools/fifo_probe
// Three kernels, each timed alone and beside each of the others (two streams):
//   stream : non-temporal read of 860 MB, 2240 workgroups x 256 threads, 16 loads of 16 bytes in flight per thread
//   l2read : WG workgroups per CU x 256 threads re-reading a buffer of argv[2] MiB (default 2: resident in every XCD's
//            L2; 64: resident in the Infinity Cache only), DEPTH loads of 16 bytes in flight per thread, plain loads
//            or LDS-DMA
//   mfma   : one workgroup per CU, 4 waves, back-to-back v_mfma_f32_16x16x32_f16 on registers
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void stream_kernel(const u32x4* __restrict__ src, unsigned* out, long per_wg) {
    const u32x4* p = src + (long)blockIdx.x * per_wg + threadIdx.x;
    unsigned acc = 0;
    u32x4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = __builtin_nontemporal_load(p + u * 256);
    for (long i = 16 * 256; i < per_wg; i += 16 * 256) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const u32x4 x = v[u];
            v[u] = __builtin_nontemporal_load(p + i + u * 256);
            acc += x[0] ^ x[1] ^ x[2] ^ x[3];
        }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

template <int DEPTH, bool DMA>
__global__ __launch_bounds__(256) void l2read_kernel(const u32x4* __restrict__ src, unsigned* out, int iters, int n_vec) {
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
    unsigned acc = 0;
    int off = (int)(((long)blockIdx.x * 40993 * 256 + threadIdx.x) % n_vec);
    if constexpr (DMA) {
        // LDS-DMA: a wave instruction fills 1 KiB of LDS; DEPTH of them in flight per wave, nothing reads them back
        const int wave = threadIdx.x >> 6;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off),
                                                 (__attribute__((address_space(3))) void*)(lds + (wave * DEPTH + d) * 64), 16, 0, 0);
                off += 256; if (off >= n_vec) off -= n_vec;
            }
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH / 2) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc = lds[threadIdx.x][0];
    } else {
        u32x4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) { v[d] = src[off]; off += 256; if (off >= n_vec) off -= n_vec; }
        for (int it = 1; it < iters; ++it) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const u32x4 x = v[d];
                v[d] = src[off]; off += 256; if (off >= n_vec) off -= n_vec;
                acc += x[0] ^ x[1] ^ x[2] ^ x[3];
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc += v[d][0] ^ v[d][1] ^ v[d][2] ^ v[d][3];
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

__global__ __launch_bounds__(256) void mfma_kernel(float* out, int iters) {
    half8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x ^ i)); }
    float4v c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = float4v{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    if (s == 1234.5f) out[blockIdx.x] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Job { const char* name; void (*launch)(hipStream_t); double units; const char* unit; };   // units per launch (bytes or flops)
static u32x4 *g_big, *g_small; static unsigned* g_out; static float* g_fout;
static long g_per_wg; static int g_l2_iters, g_l2_wg, g_nvec, g_mfma_iters;
template <int DEPTH, bool DMA> static void launch_l2(hipStream_t s) {
    hipLaunchKernelGGL((l2read_kernel<DEPTH, DMA>), dim3(256 * g_l2_wg), dim3(256), DMA ? 4 * DEPTH * 1024 : 0, s, g_small, g_out, g_l2_iters, g_nvec);
}
static void launch_stream(hipStream_t s) { hipLaunchKernelGGL(stream_kernel, dim3(2240), dim3(256), 0, s, g_big, g_out, g_per_wg); }
static void launch_mfma(hipStream_t s) { hipLaunchKernelGGL(mfma_kernel, dim3(256), dim3(256), 0, s, g_fout, g_mfma_iters); }

static double time_alone(Job& j, hipStream_t s, int n) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) j.launch(s);
    CK(hipEventRecord(e0, s)); for (int i = 0; i < n; ++i) j.launch(s); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / n;
}
// a's launches timed while b runs all the time
static double time_beside(Job& a, double a_us, Job& b, double b_us, hipStream_t sa, hipStream_t sb) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int na = (int)(40e3 / a_us) + 1, nb = (int)(300e3 / b_us) + 1;
    for (int i = 0; i < 10; ++i) b.launch(sb);
    CK(hipEventRecord(e0, sa)); for (int i = 0; i < na; ++i) a.launch(sa); CK(hipEventRecord(e1, sa));
    for (int i = 0; i < nb; ++i) b.launch(sb);
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / na;
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    g_l2_wg = argc > 1 ? atoi(argv[1]) : 1;
    const size_t big = (size_t)2240 * 384 * 1024;            // 860 MB
    g_per_wg = 384 * 1024 / 16;
    const size_t small_bytes = (size_t)(argc > 2 ? atoi(argv[2]) : 2) * 1024 * 1024;
    g_nvec = (int)(small_bytes / 16);
    CK(hipMalloc(&g_big, big)); CK(hipMalloc(&g_small, small_bytes)); CK(hipMalloc(&g_out, 1 << 20)); CK(hipMalloc(&g_fout, 1 << 20));
    CK(hipMemset(g_big, 1, big)); CK(hipMemset(g_small, 1, small_bytes));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(l2read_kernel<8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(l2read_kernel<16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    g_l2_iters = 2000; g_mfma_iters = 20000;
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    Job stream{"stream (HBM, nt loads)", launch_stream, (double)big, "GB/s"};
    Job mfma{"mfma (registers only)", launch_mfma, 256.0 * 4 * g_mfma_iters * 8 * 2.0 * 16 * 16 * 32, "TFLOP/s x1000"};
    Job readers[] = {
        {"l2read plain depth 2", launch_l2<2, false>, 256.0 * g_l2_wg * 256 * 16 * 2 * g_l2_iters, "GB/s"},
        {"l2read plain depth 8", launch_l2<8, false>, 256.0 * g_l2_wg * 256 * 16 * 8 * g_l2_iters, "GB/s"},
        {"l2read LDS-DMA depth 8", launch_l2<8, true>, 256.0 * g_l2_wg * 256 * 16 * 8 * g_l2_iters, "GB/s"},
        {"l2read LDS-DMA depth 16", launch_l2<16, true>, 256.0 * g_l2_wg * 256 * 16 * 16 * g_l2_iters, "GB/s"},
    };
    const double st_us = time_alone(stream, s1, 100);
    printf("%-28s alone: %8.1f us  %8.1f %s\n", stream.name, st_us, stream.units / st_us / 1e3, stream.unit);
    const double mf_us = time_alone(mfma, s1, 20);
    printf("%-28s alone: %8.1f us  %8.1f %s\n", mfma.name, mf_us, mfma.units / mf_us / 1e3, mfma.unit);
    {
        const double a = time_beside(mfma, mf_us, stream, st_us, s1, s2), b = time_beside(stream, st_us, mfma, mf_us, s1, s2);
        printf("  mfma beside stream: %8.1f us (%.2fx)   stream beside mfma: %8.1f us (%.2fx) %8.1f GB/s\n", a, a / mf_us, b, b / st_us, stream.units / b / 1e3);
    }
    for (Job& r : readers) {
        const double r_us = time_alone(r, s1, 20);
        printf("%-28s alone (%d per CU): %8.1f us  %8.1f %s\n", r.name, g_l2_wg, r_us, r.units / r_us / 1e3, r.unit);
        const double a = time_beside(r, r_us, stream, st_us, s1, s2), b = time_beside(stream, st_us, r, r_us, s1, s2);
        printf("  reader beside stream: %8.1f us (%.2fx) %8.1f GB/s   stream beside reader: %8.1f us (%.2fx) %8.1f GB/s   sum %8.1f GB/s\n",
               a, a / r_us, r.units / a / 1e3, b, b / st_us, stream.units / b / 1e3, r.units / a / 1e3 + stream.units / b / 1e3);
        const double c = time_beside(r, r_us, mfma, mf_us, s1, s2);
        printf("  reader beside mfma:   %8.1f us (%.2fx)\n", c, c / r_us);
    }
    return 0;
}
