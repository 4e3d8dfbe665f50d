// Development aid: which compute-unit masks does hipExtStreamCreateWithCUMask accept on this device, and how
// fast does a bandwidth-bound copy run on a stream limited to them?   hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>

__global__ void copy_kernel(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

static float run(hipStream_t s, const float4* a, float4* b, size_t n) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, s, a, b, n);
    hipEventRecord(e0, s);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, s, a, b, n);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    printf("CUs %d\n", n_cu);
    const size_t n = (size_t)1 << 26;       // 1 GiB each way
    float4 *a, *b; hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMemset(a, 1, n * 16);
    hipStream_t s0; hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
    float ms = run(s0, a, b, n);
    printf("plain stream: %.3f ms  %.0f GB/s\n", ms, 2.0 * n * 16 / ms / 1e6);
    struct Case { const char* name; int words; int div; int mod; int lo; int hi; };
    // bit b set when lo <= ((b / div) % mod) < hi.  Bit b is CU b / 8 of XCD b % 8: every XCD must keep some CUs
    // (a kernel's workgroups are dealt to all XCDs in turn).
    const Case cases[] = {{"all 256 bits", 8, 1, 1, 0, 1}, {"quarter of every XCD (CUs 0-7 of 32)", 8, 8, 32, 0, 8},
                          {"half of every XCD (CUs 0-15)", 8, 8, 32, 0, 16}, {"every fourth CU of every XCD", 8, 8, 4, 0, 1}};
    for (const Case& c : cases) {
        std::vector<uint32_t> m(c.words, 0u);
        for (int bit = 0; bit < c.words * 32; ++bit) { int r = (bit / c.div) % c.mod; if (r >= c.lo && r < c.hi) m[bit / 32] |= 1u << (bit % 32); }
        hipStream_t s = nullptr;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, c.words, m.data());
        if (e != hipSuccess) { printf("%s: create failed: %s\n", c.name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        ms = run(s, a, b, n);
        printf("%s: %.3f ms  %.0f GB/s\n", c.name, ms, 2.0 * n * 16 / ms / 1e6);
        hipStreamDestroy(s);
    }
    return 0;
}
