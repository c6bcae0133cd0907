// microbenchmark: fp32-equivalent contraction on the bf16 matrix cores by operand splitting (x = hi + mid + lo, three bf16 pieces, exact for a
// 24-bit mantissa; 6 of the 9 partial products kept: hi.hi hi.mid mid.hi mid.mid hi.lo lo.hi, dropped terms <= 2^-23 relative).
// Per k16 step one wave splits 8 fresh f32 values per lane (one B operand of v_mfma_f32_32x32x16_bf16) on the VALU — and-mask / subtract chain + packs —
// and issues NT x 6 MFMAs (NT output tiles share the B operand; the A pieces are pre-split, as weights staged once per launch would be).
// Question: do the split's VALU instructions hide behind the bf16 MFMAs (they do NOT behind v_mfma_f32_32x32x2_f32: profiles/r01_mfma_valu_microbench.md),
// and what fp32-equivalent rate results?   f32 MFMA reference: 157.3 TFLOP/s peak = 32 cycles per k per 32x32 tile.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split8(const float (&x)[8], u32x4& hi, u32x4& mid, u32x4& lo) {
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned xb = __float_as_uint(x[i]);
        h[i] = xb & 0xffff0000u;
        const float r = x[i] - __uint_as_float(h[i]);
        m[i] = __float_as_uint(r) & 0xffff0000u;
        l[i] = __float_as_uint(r - __uint_as_float(m[i]));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // pack the high halves of two f32 words into one register: {b[31:16], a[31:16]}
        hi[i] = __builtin_amdgcn_perm(h[2 * i + 1], h[2 * i], 0x07060302u);
        mid[i] = __builtin_amdgcn_perm(m[2 * i + 1], m[2 * i], 0x07060302u);
        lo[i] = __builtin_amdgcn_perm(l[2 * i + 1], l[2 * i], 0x07060302u);
    }
}

template <int NT, int WAVES, bool SPLIT>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, int iters) {
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 0.001f * threadIdx.x + 0.37f * i + 1.0f;
    u32x4 ah[NT], am[NT], al[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { float w[8]; for (int i = 0; i < 8; ++i) w[i] = 0.5f + 0.01f * (i + t) + 0.0001f * threadIdx.x; split8(w, ah[t], am[t], al[t]); }
    u32x4 bh, bm, bl; split8(x, bh, bm, bl);
    for (int it = 0; it < iters; ++it) {
        if (SPLIT) {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = x[i] * 1.0001f + 0.001f;   // fresh values (stands for the tanh output of the previous layer): +8 v_fma
            split8(x, bh, bm, bl);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bf16x8 Ah = __builtin_bit_cast(bf16x8, ah[t]), Am = __builtin_bit_cast(bf16x8, am[t]), Al = __builtin_bit_cast(bf16x8, al[t]);
            const bf16x8 Bh = __builtin_bit_cast(bf16x8, bh), Bm = __builtin_bit_cast(bf16x8, bm), Bl = __builtin_bit_cast(bf16x8, bl);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, acc[t], 0, 0, 0);   // small terms first
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bm, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, acc[t], 0, 0, 0);
        }
    }
    float s = 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + x[0];
}

// accuracy: C = A.B over K = 64 with 3-piece operands vs the f32 MFMA and vs a double reference, one wave
__global__ void accuracy(const float* A, const float* B, float* Csplit, float* Cf32) {   // A [32][64] row-major, B [64][32]
    const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
    f32x16 acc, ref;
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; ref[r] = 0.f; }
    for (int k0 = 0; k0 < 64; k0 += 16) {
        float a[8], b[8];
        for (int i = 0; i < 8; ++i) { a[i] = A[c * 64 + k0 + 8 * h + i]; b[i] = B[(k0 + 8 * h + i) * 32 + c]; }
        u32x4 ah, am, al, bh, bm, bl; split8(a, ah, am, al); split8(b, bh, bm, bl);
        const bf16x8 Ah = __builtin_bit_cast(bf16x8, ah), Am = __builtin_bit_cast(bf16x8, am), Al = __builtin_bit_cast(bf16x8, al);
        const bf16x8 Bh = __builtin_bit_cast(bf16x8, bh), Bm = __builtin_bit_cast(bf16x8, bm), Bl = __builtin_bit_cast(bf16x8, bl);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bm, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, acc, 0, 0, 0);
    }
    for (int k = 0; k < 64; k += 2) ref = __builtin_amdgcn_mfma_f32_32x32x2f32(A[c * 64 + k + h], B[(k + h) * 32 + c], ref, 0, 0, 0);
    for (int r = 0; r < 16; ++r) { const int row = (r & 3) + 8 * (r >> 2) + 4 * h; Csplit[row * 32 + c] = acc[r]; Cf32[row * 32 + c] = ref[r]; }
}

template <int NT, int WAVES, bool SPLIT> void run(float* out) {
    const int iters = 4000;
    k<NT, WAVES, SPLIT><<<256, 64 * WAVES>>>(out, iters); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<NT, WAVES, SPLIT><<<256, 64 * WAVES>>>(out, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double eq = 256.0 * WAVES * iters * NT * 2.0 * 32 * 32 * 16;        // fp32-equivalent FLOP
    printf("%d tile(s) per split, %d wave(s)/SIMD, split %-3s: %.3f ms  %7.1f TFLOP/s fp32-equivalent (%5.1f cycles per k16 step and tile at 2.4 GHz)\n", NT, WAVES / 4,
           SPLIT ? "on" : "off", ms, eq / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (iters * (double)NT) / (WAVES / 4.0));
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    run<1, 4, false>(out); run<1, 4, true>(out); run<2, 4, true>(out); run<4, 4, true>(out);
    run<1, 8, false>(out); run<1, 8, true>(out); run<2, 8, true>(out); run<4, 8, true>(out);
    float hA[32 * 64], hB[64 * 32], hS[1024], hF[1024]; double worst_s = 0, worst_f = 0, scale = 0;
    unsigned s = 12345; auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) / 16777216.0f) * 2.0f - 1.0f; };
    for (auto& v : hA) v = rnd(); for (auto& v : hB) v = rnd();
    float *dA, *dB, *dS, *dF; hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dS, 4096); hipMalloc(&dF, 4096);
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    accuracy<<<1, 64>>>(dA, dB, dS, dF); hipMemcpy(hS, dS, 4096, hipMemcpyDeviceToHost); hipMemcpy(hF, dF, 4096, hipMemcpyDeviceToHost);
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
        double r = 0, ab = 0; for (int k = 0; k < 64; ++k) { r += (double)hA[m * 64 + k] * hB[k * 32 + n]; ab += fabs((double)hA[m * 64 + k] * hB[k * 32 + n]); }
        worst_s = fmax(worst_s, fabs(hS[m * 32 + n] - r) / ab); worst_f = fmax(worst_f, fabs(hF[m * 32 + n] - r) / ab); scale = fmax(scale, ab);
    }
    printf("accuracy over K = 64 (max |C - C_f64| / sum|a b|): 3-piece bf16 split %.3g, f32 MFMA %.3g  (fp32 epsilon 5.96e-08)\n", worst_s, worst_f);
    return 0;
}
