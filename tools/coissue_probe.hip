// Development aid: do the MFMAs of one wave and the vector instructions of another wave ON THE SAME SIMD run side by
// side?  A workgroup of 512 threads puts two waves on every SIMD; waves 0-3 run a chain-free loop of
// v_mfma_f32_16x16x32_f16, waves 4-7 a loop of v_exp_f32 (or v_fma_f32).  One workgroup per CU, timed with events:
// "both" close to the larger of the two alone = they overlap; close to the sum = they take turns.
//   hipcc --offload-arch=gfx950 -O3 tools/coissue_probe.hip -o tools/coissue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

// mode bit 0: waves 0-3 do MFMAs; bit 1: waves 4-7 do exponentials; bit 2: waves 4-7 do FMAs instead;
// bit 3: the MFMAs are v_mfma_f32_32x32x16_f16 (half as many, twice the work each)
__global__ __launch_bounds__(512) void probe(float* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if ((mode & 9) == 9) {
            half8 a, b;
            for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
            float16v c[4];
            for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) c[i][e] = 0.f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[i], 0, 0, 0);
            }
            for (int i = 0; i < 4; ++i) r += c[i][0] + c[i][5] + c[i][10] + c[i][15];
        } else if (mode & 1) {
            half8 a, b;
            for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
            float4v c[8];
            for (int i = 0; i < 8; ++i) c[i] = float4v{0, 0, 0, 0};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[i], 0, 0, 0);
            }
            for (int i = 0; i < 8; ++i) r += c[i][0] + c[i][1] + c[i][2] + c[i][3];
        }
    } else if (mode & 2) {
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = -0.001f * (threadIdx.x + i);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_amdgcn_exp2f(x[i]) - 1.0001f;
        }
        for (int i = 0; i < 8; ++i) r += x[i];
    } else if (mode & 4) {
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = 0.001f * (threadIdx.x + i);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { x[i] = fmaf(x[i], 0.999f, 0.001f); x[i] = fmaf(x[i], 1.001f, -0.001f); x[i] = fmaf(x[i], 0.999f, 0.001f); x[i] = fmaf(x[i], 1.001f, -0.001f); }
        }
        for (int i = 0; i < 8; ++i) r += x[i];
    }
    if (r == 12345.678f) out[threadIdx.x] = r;
}

static float run(float* out, int iters, int mode) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, out, iters, mode);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, out, iters, mode);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1e3f;
}

int main() {
    float* out; (void)hipMalloc(&out, 4096);
    const int iters = 20000;            // 160 k MFMAs (16 cycles each) per wave: ~1.3 ms at 2 GHz
    for (int rep = 0; rep < 2; ++rep) {
        const float m = run(out, iters, 1), e = run(out, iters, 2), f = run(out, iters, 4);
        const float me = run(out, iters, 3), mf = run(out, iters, 5);
        printf("MFMA alone %.0f us (%.1f cycles per MFMA at 1.95 GHz), exp alone %.0f us (%.1f cycles per v_exp + v_sub), 4 FMA alone %.0f us\n", m,
               m * 1950.0 / (iters * 8.0), e, e * 1950.0 / (iters * 8.0), f);
        printf("  MFMA wave + exp wave on every SIMD: %.0f us (max %.0f, sum %.0f);  MFMA wave + FMA wave: %.0f us (max %.0f, sum %.0f)\n", me,
               m > e ? m : e, m + e, mf, m > f ? m : f, m + f);
    }
    for (int rep = 0; rep < 2; ++rep) {
        const float m = run(out, iters, 9), e = run(out, iters, 2), f = run(out, iters, 4);
        const float me = run(out, iters, 11), mf = run(out, iters, 13);
        printf("32x32x16: MFMA alone %.0f us (%.1f cycles per MFMA at 1.95 GHz); + exp wave %.0f us (max %.0f, sum %.0f); + FMA wave %.0f us (max %.0f, sum %.0f)\n",
               m, m * 1950.0 / (iters * 4.0), me, m > e ? m : e, m + e, mf, m > f ? m : f, m + f);
    }
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
