// Developer harness: what the gfx950 fp8 conversion instructions do, against a software e4m3fn codec.
//   hipcc -O3 --offload-arch=gfx950 tools/fp8_probe.hip -o tools/fp8_probe && tools/fp8_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));

__global__ void enc_kernel(const float* x, uint8_t* lo, uint8_t* hi, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int a = __builtin_amdgcn_cvt_pk_fp8_f32(x[i], 0.f, 0, false);      // byte 0 of the low word
    const int b = __builtin_amdgcn_cvt_pk_fp8_f32(0.f, x[i], 0, true);       // byte 3 (second of the high word)
    lo[i] = (uint8_t)(a & 0xff);
    hi[i] = (uint8_t)((unsigned)b >> 24);
}

__global__ void dec_kernel(float* f16out, float* f32out) {       // [256][4]: code placed in byte j of the word
    const int c = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned w = (unsigned)c << (8 * j);
        half2v h; float2v f;
        if (j >= 2) { h = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w, 1.0f, true); f = __builtin_amdgcn_cvt_pk_f32_fp8(w, true); }
        else { h = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w, 1.0f, false); f = __builtin_amdgcn_cvt_pk_f32_fp8(w, false); }
        f16out[c * 4 + j] = (float)h[j & 1];
        f32out[c * 4 + j] = f[j & 1];
    }
}

static float sw_decode(int c) {
    const int e = (c >> 3) & 15, m = c & 7;
    if ((c & 0x7f) == 0x7f) return NAN;
    const float v = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m * 0.125f, e - 7);
    return (c & 0x80) ? -v : v;
}
static int sw_encode(float x) {                      // RNE, saturating at 448
    float a = fminf(fabsf(x), 448.f);
    int e; frexpf(a, &e); e = e - 1 < -6 ? -6 : e - 1;
    const float step = ldexpf(1.f, e - 3);
    const float r = nearbyintf(a / step) * step;
    int best = 0;
    for (int c = 0; c < 127; ++c) if (sw_decode(c) == r) { best = c; break; }
    return best | (signbit(x) ? 0x80 : 0);
}

int main() {
    std::vector<float> xs;
    for (int c = 0; c < 126; ++c) {
        const float a = sw_decode(c), b = sw_decode(c + 1);
        for (float t : {0.f, 0.25f, 0.4999f, 0.5f, 0.5001f, 0.75f}) { xs.push_back(a + (b - a) * t); xs.push_back(-(a + (b - a) * t)); }
    }
    for (float v : {448.f, 449.f, 463.9f, 464.f, 480.f, 1000.f, 1e-10f, 0.0009765625f, 0.00048828125f}) { xs.push_back(v); xs.push_back(-v); }
    const int n = (int)xs.size();
    float* dx; uint8_t *dlo, *dhi; float *d16, *d32;
    hipMalloc(&dx, n * 4); hipMalloc(&dlo, n); hipMalloc(&dhi, n); hipMalloc(&d16, 1024 * 4); hipMalloc(&d32, 1024 * 4);
    hipMemcpy(dx, xs.data(), n * 4, hipMemcpyHostToDevice);
    enc_kernel<<<(n + 255) / 256, 256>>>(dx, dlo, dhi, n);
    dec_kernel<<<1, 256>>>(d16, d32);
    std::vector<uint8_t> lo(n), hi(n); std::vector<float> h16(1024), h32(1024);
    hipMemcpy(lo.data(), dlo, n, hipMemcpyDeviceToHost); hipMemcpy(hi.data(), dhi, n, hipMemcpyDeviceToHost);
    hipMemcpy(h16.data(), d16, 4096, hipMemcpyDeviceToHost); hipMemcpy(h32.data(), d32, 4096, hipMemcpyDeviceToHost);
    int bad_enc = 0, bad_dec = 0;
    for (int i = 0; i < n; ++i) {
        const int want = sw_encode(xs[i]);
        const bool zero_ok = (want & 0x7f) == 0 && (lo[i] & 0x7f) == 0;      // sign of zero is free
        if ((lo[i] != want && !zero_ok) || lo[i] != hi[i]) {
            if (bad_enc < 12) printf("encode %.9g: hw lo %02x hi %02x (%g), sw %02x (%g)\n", xs[i], lo[i], hi[i], sw_decode(lo[i]), want, sw_decode(want));
            ++bad_enc;
        }
    }
    for (int c = 0; c < 256; ++c)
        for (int j = 0; j < 4; ++j) {
            const float want = sw_decode(c);
            const bool ok16 = isnan(want) ? isnan(h16[c * 4 + j]) : h16[c * 4 + j] == want;
            const bool ok32 = isnan(want) ? isnan(h32[c * 4 + j]) : h32[c * 4 + j] == want;
            if (!ok16 || !ok32) { if (bad_dec < 12) printf("decode %02x byte %d: f16 %g f32 %g, sw %g\n", c, j, h16[c * 4 + j], h32[c * 4 + j], want); ++bad_dec; }
        }
    printf("encode: %d inputs, %d differ from RNE-saturating e4m3fn; decode: %d of 1024 differ\n", n, bad_enc, bad_dec);
    return 0;
}
