// Development aid: what does a bandwidth-saturating read stream do to the latency of a dependent load issued by
// another kernel at the same time, as a function of the bytes the stream keeps in flight?  (The decode groups'
// short launches are chains of dependent round trips that run beside other groups' cross-attention streams.)
//   stream kernel: workgroups of 256 threads walk items of 192 KB (item, item + grid, ...), D loads of 16 bytes
//                  per thread in flight (non-temporal), grid and D chosen on the command line
//   probe kernel:  one wave, lane 0 chases pointers through a 1 GiB table of random links (every load misses the
//                  caches), 4000 links, timed with the 100 MHz counter
//   hipcc --offload-arch=gfx950 -O3 tools/lat_under_load.hip -o tools/lat_under_load
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr long ITEM16 = 192 * 1024 / 16;          // 16-byte pieces per item
constexpr int U = 48;                              // pieces per thread and item (256 threads)

template <int D>
__global__ __launch_bounds__(256) void stream_kernel(const u32x4* __restrict__ src, unsigned* sink, int n_items, int rounds) {
    unsigned acc = 0;
    for (int r = 0; r < rounds; ++r)
        for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
            const u32x4* p = src + (long)item * ITEM16 + threadIdx.x;
            u32x4 a[D];
#pragma unroll
            for (int d = 0; d < D; ++d) a[d] = __builtin_nontemporal_load(p + d * 256);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const u32x4 x = a[u % D];
                if (u + D < U) a[u % D] = __builtin_nontemporal_load(p + (u + D) * 256);
                acc += x[0] ^ x[1] ^ x[2] ^ x[3];
            }
        }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0 && acc == 0x12345678u) sink[0] = acc;
}

__global__ void probe_kernel(const unsigned* __restrict__ links, int n, unsigned long long* out, unsigned start) {
    if (threadIdx.x != 0) return;
    unsigned i = start;          // every probe continues where the last one ended: lines nobody has touched yet
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int k = 0; k < n; ++k) i = __builtin_nontemporal_load(links + (long)i * 32);      // one link per 128-byte line
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    out[0] = t1 - t0; out[1] = i;
}

static unsigned g_start = 0;
template <int D>
static void run(const u32x4* src, unsigned* sink, int n_items, int grid, const unsigned* links, unsigned long long* out_d, hipStream_t s1, hipStream_t s2) {
    const int n_links = 800, rounds = 20;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(stream_kernel<D>, dim3(grid), dim3(256), 0, s1, src, sink, n_items, 1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, s1);
    hipLaunchKernelGGL(stream_kernel<D>, dim3(grid), dim3(256), 0, s1, src, sink, n_items, rounds);
    (void)hipEventRecord(e1, s1);
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, s2, links, n_links, out_d, g_start);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long out[2];
    (void)hipMemcpy(out, out_d, 16, hipMemcpyDeviceToHost);
    g_start = (unsigned)out[1];
    const double probe_us = out[0] * 0.01;
    printf("D=%2d grid %5d (%5.1f MB in flight at most): stream %7.1f GB/s over %6.0f us; probe %7.1f us for %d links = %6.0f ns per dependent load%s\n",
           D, grid, (double)grid * 256 * D * 16 / 1e6, (double)n_items * rounds * 196608.0 / ms / 1e6, ms * 1e3, probe_us, n_links, probe_us * 1e3 / n_links,
           probe_us > ms * 1e3 ? "  (probe outlasted the stream)" : "");
}

int main(int argc, char** argv) {
    const int n_items = 2240 * 4;                                  // 1.76 GB: four launches' worth of a 112-sequence group
    u32x4* src; unsigned* sink; unsigned* links; unsigned long long* out_d;
    if (hipMalloc(&src, (size_t)n_items * 196608) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(src, 1, (size_t)n_items * 196608);
    (void)hipMalloc(&sink, 64); (void)hipMalloc(&out_d, 64);
    const long n_lines = 1L << 23;                                 // 8 M lines of 128 bytes = 1 GiB
    std::vector<unsigned> perm(n_lines);
    for (long i = 0; i < n_lines; ++i) perm[i] = (unsigned)i;
    unsigned long long st = 88172645463325252ull;
    for (long i = n_lines - 1; i > 0; --i) {                       // one cycle through all lines (Sattolo)
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        const long j = (long)(st % (unsigned long long)i);
        const unsigned t = perm[i]; perm[i] = perm[j]; perm[j] = t;
    }
    std::vector<unsigned> host((size_t)n_lines * 32, 0u);
    for (long i = 0; i < n_lines; ++i) host[(size_t)i * 32] = perm[i];
    (void)hipMalloc(&links, (size_t)n_lines * 128);
    (void)hipMemcpy(links, host.data(), (size_t)n_lines * 128, hipMemcpyHostToDevice);
    hipStream_t s1, s2; (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    // the probe alone
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, s2, links, 4000, out_d, 0u);
    (void)hipDeviceSynchronize();
    unsigned long long out[2];
    (void)hipMemcpy(out, out_d, 16, hipMemcpyDeviceToHost);
    printf("probe alone: %.0f ns per dependent load\n", out[0] * 10.0 / 4000);
    g_start = (unsigned)out[1];
    const int grids[] = {256, 512, 768, 1024, 2048};
    for (int g : grids) {
        run<2>(src, sink, n_items, g, links, out_d, s1, s2);
        run<4>(src, sink, n_items, g, links, out_d, s1, s2);
        run<8>(src, sink, n_items, g, links, out_d, s1, s2);
        run<16>(src, sink, n_items, g, links, out_d, s1, s2);
    }
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
