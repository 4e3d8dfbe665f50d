// Development aid: does the per-workgroup tail of the decode cross-attention (two dependent load phases, a barrier, a
// published partial, a returning atomic) explain the distance between its 6.3-6.4 TB/s and the 7.1 TB/s of a plain
// read stream?  Items of 96 KB (two halves of 48 KB), 256 threads, 12 loads of 16 bytes per thread and half.
//   mode 0: one item per workgroup, second half issued after the first is consumed, then the tail
//   mode 1: the same with both halves issued up front
//   mode 2: workgroups that walk items (item, item + grid, ...) and issue the next item's halves as soon as the
//           registers of the current one are consumed; the tail runs under those loads
//   hipcc --offload-arch=gfx950 -O3 tools/read_bw2.hip -o tools/read_bw2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int U = 12;

__device__ __forceinline__ unsigned fold(const u32x4 (&v)[U]) {
    unsigned a = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) a += v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
    return a;
}
__device__ __forceinline__ void issue(u32x4 (&v)[U], const u32x4* p) {
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + u * 256);
}
template <int TAIL = 3>
__device__ __forceinline__ void tail(unsigned acc, float* partial, int* tickets, int item, float* out) {
    __shared__ unsigned red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);      // every lane's loads count (or the compiler sinks them into lane 0's branch)
    float* mine = partial + (long)item * 68;
    if (TAIL == 0) { if (lane == 0) mine[wave] = (float)acc; return; }
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (wave != 0) return;
    const float o = (float)(red[0] + red[1] + red[2] + red[3]);
    if (TAIL == 1) { mine[2 + lane] = o; return; }
    __hip_atomic_store(mine + 2 + lane, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (TAIL == 2) return;
    int ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(tickets + item / 4, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __shfl(ticket, 0);
    if ((ticket & 3) != 3) return;
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += __hip_atomic_load(partial + (long)(item / 4 * 4 + i) * 68 + 2 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    out[(long)(item / 4) * 64 + lane] = s;
}

// one item per workgroup; both halves streamed side by side, D loads of each in flight per thread, then the tail
template <int D, int TAIL = 3>
__global__ __launch_bounds__(256) void kstream(const u32x4* __restrict__ src, float* partial, int* tickets, float* out, int n_items) {
    constexpr long HALF = 48 * 1024 / 16;
    const int item = blockIdx.x;
    const u32x4* p = src + (long)item * 2 * HALF + threadIdx.x;
    u32x4 a[D], b[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { a[d] = __builtin_nontemporal_load(p + d * 256); b[d] = __builtin_nontemporal_load(p + HALF + d * 256); }
    unsigned acc = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const u32x4 x = a[u % D], y = b[u % D];
        if (u + D < U) { a[u % D] = __builtin_nontemporal_load(p + (u + D) * 256); b[u % D] = __builtin_nontemporal_load(p + HALF + (u + D) * 256); }
        acc += x[0] ^ x[1] ^ x[2] ^ x[3];
        acc += (y[0] ^ y[1]) + (y[2] ^ y[3]);
    }
    tail<TAIL>(acc, partial, tickets, item, out);
}
template <int D, int TAIL = 3>
static void run_stream(const u32x4* src, float* partial, int* tickets, float* out, int n_items, int lds, const char* name) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kstream<D, TAIL>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((kstream<D, TAIL>), dim3(n_items), dim3(256), lds, 0, src, partial, tickets, out, n_items);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((kstream<D, TAIL>), dim3(n_items), dim3(256), lds, 0, src, partial, tickets, out, n_items);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-38s D=%2d  grid %6d lds %5d : %7.1f us  %7.1f GB/s  (%s)\n", name, D, n_items, lds, ms * 100, n_items * 98304.0 * 10 / ms / 1e6, hipGetErrorString(hipGetLastError()));
}

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 4))) void k(const u32x4* __restrict__ src, float* partial, int* tickets,
                                                                                       float* out, int n_items) {
    constexpr long HALF = 48 * 1024 / 16;
    u32x4 a[U], b[U];
    if constexpr (MODE < 2) {
        const int item = blockIdx.x;
        const u32x4* p = src + (long)item * 2 * HALF + threadIdx.x;
        issue(a, p);
        if (MODE == 1) issue(b, p + HALF);
        unsigned acc = fold(a);
        if (MODE == 0) issue(a, p + HALF);
        acc += MODE == 1 ? fold(b) : fold(a);
        tail(acc, partial, tickets, item, out);
    } else {
        int item = blockIdx.x;
        if (item >= n_items) return;
        const u32x4* p = src + (long)item * 2 * HALF + threadIdx.x;
        issue(a, p); issue(b, p + HALF);
        for (;;) {
            const int next = item + gridDim.x;
            const bool more = next < n_items;
            const u32x4* pn = src + (long)next * 2 * HALF + threadIdx.x;
            unsigned acc = fold(a);
            if (more) issue(a, pn);
            acc += fold(b);
            if (more) issue(b, pn + HALF);
            {   // the tail, inlined: wave 0 publishes while every wave's next loads are in flight
                __shared__ unsigned red[2][4];
                const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, par = (item / (int)gridDim.x) & 1;
                for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
                if (lane == 0) red[par][wave] = acc;
                __syncthreads();
                if (wave == 0) {
                    const float o = (float)(red[par][0] + red[par][1] + red[par][2] + red[par][3]);
                    float* mine = partial + (long)item * 68;
                    __hip_atomic_store(mine + 2 + lane, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    int ticket = 0;
                    if (lane == 0) ticket = __hip_atomic_fetch_add(tickets + item / 4, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ticket = __shfl(ticket, 0);
                    if ((ticket & 3) == 3) {
                        float s = 0.f;
                        for (int i = 0; i < 4; ++i) s += __hip_atomic_load(partial + (long)(item / 4 * 4 + i) * 68 + 2 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        out[(long)(item / 4) * 64 + lane] = s;
                    }
                }
            }
            if (!more) break;
            item = next;
        }
    }
}

template <int MODE>
static void run(const u32x4* src, float* partial, int* tickets, float* out, int n_items, int grid, int lds, const char* name) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), lds, 0, src, partial, tickets, out, n_items);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), lds, 0, src, partial, tickets, out, n_items);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s grid %6d lds %5d : %7.1f us  %7.1f GB/s  (%s)\n", name, grid, lds, ms * 100, n_items * 98304.0 * 10 / ms / 1e6, hipGetErrorString(hipGetLastError()));
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int n_items = 8960;                          // 112 sequences x 20 heads x 4 ranges: 860 MB per launch
    const size_t bytes = (size_t)n_items * 98304;
    u32x4* src; float *partial, *out; int* tickets;
    // several copies so that consecutive launches do not find their data in the Infinity Cache: walk 8 GiB
    if (hipMalloc(&src, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMalloc(&partial, (size_t)n_items * 68 * 4); (void)hipMalloc(&out, (size_t)n_items * 64); (void)hipMalloc(&tickets, n_items);
    (void)hipMemset(src, 1, bytes); (void)hipMemset(tickets, 0, n_items);
    for (int lds : {0, 40000}) {
        run_stream<4, 0>(src, partial, tickets, out, n_items, lds, "streamed, tail 0: store per wave");
        run_stream<4, 1>(src, partial, tickets, out, n_items, lds, "streamed, tail 1: barrier, wave 0 stores");
        run_stream<4, 2>(src, partial, tickets, out, n_items, lds, "streamed, tail 2: agent stores, drain");
        run_stream<4, 3>(src, partial, tickets, out, n_items, lds, "streamed, tail 3: ticket and merge");
        run_stream<12, 0>(src, partial, tickets, out, n_items, lds, "streamed, tail 0: store per wave");
        run_stream<12, 3>(src, partial, tickets, out, n_items, lds, "streamed, tail 3: ticket and merge");
    }
    for (int lds : {0, 40000}) {
        run<0>(src, partial, tickets, out, n_items, n_items, lds, "one item per workgroup, halves in turn");
        run<1>(src, partial, tickets, out, n_items, n_items, lds, "one item per workgroup, halves together");
        for (int grid : {512, 768, 1024, 1280}) run<2>(src, partial, tickets, out, n_items, grid, lds, "walking workgroups, next item prefetched");
    }
    return 0;
}
