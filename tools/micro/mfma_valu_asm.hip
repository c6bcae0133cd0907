// microbenchmark (round 5, second form): does vector arithmetic run beside v_mfma_f32_32x32x16_f16 on gfx950?
// The instruction streams are INLINE ASM, one block per group, so that neither LLVM's IR passes nor the machine scheduler can bunch or pack them
// (the first form, mfma_bf16_valu.hip, let the fillers sink behind the MFMAs and be SLP-packed into v_pk_fma_f32: its figures are of that stream).
//   mode 0: every wave runs  [MFMA, N x v_fma_f32]  per group                                  (same-wave co-issue)
//   mode 1: two waves per SIMD; the even wave of a SIMD runs MFMAs only, the odd one runs N x v_fma_f32 per group  (cross-wave co-issue)
// reports shader cycles per group (s_memtime of wave 0) and the kernel time.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// x = x * c + x with c in an SGPR: one VGPR per filler, so that no operand-bank conflict is part of the filler's price
#define FMA1(r) "v_fma_f32 %" #r ", %" #r ", %8, %" #r "\n\t"
template <int N> __device__ __forceinline__ void fillers(float (&v)[8], float c, float d) {
    if (N == 0) return;
    if (N == 2) asm volatile(FMA1(0) FMA1(1) : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "s"(c));
    if (N == 4) asm volatile(FMA1(0) FMA1(1) FMA1(2) FMA1(3) : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "s"(c));
    if (N == 5) asm volatile(FMA1(0) FMA1(1) FMA1(2) FMA1(3) FMA1(4) : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "s"(c));
    if (N == 6) asm volatile(FMA1(0) FMA1(1) FMA1(2) FMA1(3) FMA1(4) FMA1(5) : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "s"(c));
    if (N == 8) asm volatile(FMA1(0) FMA1(1) FMA1(2) FMA1(3) FMA1(4) FMA1(5) FMA1(6) FMA1(7) : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "s"(c));
    if (N == 12) { fillers<8>(v, c, d); fillers<4>(v, c, d); }
    if (N == 16) { fillers<8>(v, c, d); fillers<8>(v, c, d); }
}

template <int N, int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, unsigned long long* ticks, int iters) {
    f32x16 acc[2];
    for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float v[8]; for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    float c = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(iters * 1e-9f - 0.5f))), d = 0.f;
    f16x8 a, b; for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.5f + i); b[i] = (_Float16)(1.0f + i); }
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // waves w and w + 4 of an 8-wave workgroup share a SIMD
    //   mode 2: as mode 1 with the roles exchanged (the OLDER wave of a SIMD runs the fillers, the younger one the MFMAs);  mode 3: mode 2 with s_setprio 3 in the MFMA wave
    const bool does_mfma = MODE == 0 || (MODE == 1 ? w < 4 : w >= 4), does_valu = !does_mfma;
    if (MODE == 3 && does_mfma) __builtin_amdgcn_s_setprio(3);
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (MODE == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[u & 1]) : "v"(a), "v"(b));
                fillers<N>(v, c, d);
            }
        }
    } else if (does_mfma) {                       // the branch is outside the loops: nothing but the group's instructions and the loop's own s_cbranch per 8 groups
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[u & 1]) : "v"(a), "v"(b));
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) fillers<N>(v, c, d);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0; for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r]; for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
    if (threadIdx.x == 256 && blockIdx.x == 0) ticks[1] = t1 - t0;
}
template <int N, int MODE, int WAVES> void run(float* out, unsigned long long* ticks) {
    const int iters = 2000;
    hipMemset(ticks, 0, 16);
    k<N, MODE, WAVES><<<256, 64 * WAVES>>>(out, ticks, iters); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<N, MODE, WAVES><<<256, 64 * WAVES>>>(out, ticks, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t[2]; hipMemcpy(t, ticks, 16, hipMemcpyDeviceToHost);
    const double groups = iters * 8.0;
    printf("%-36s waves/SIMD %d  v_fma_f32 x %2d per group : wave 0 %6.1f cycles per group, wave 4 %6.1f   (kernel %.3f ms)\n",
           MODE == 0 ? "same wave: MFMA + fillers" : MODE == 1 ? "older wave MFMA, younger fillers" : MODE == 2 ? "older wave fillers, younger MFMA" : "same, s_setprio 3 in the MFMA wave", WAVES / 4, N, (double)t[0] / groups, (double)t[1] / groups, ms);
}
int main() {
    float* out; unsigned long long* ticks; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&ticks, 16);
    run<0, 0, 4>(out, ticks); run<2, 0, 4>(out, ticks); run<4, 0, 4>(out, ticks); run<5, 0, 4>(out, ticks); run<6, 0, 4>(out, ticks); run<8, 0, 4>(out, ticks); run<12, 0, 4>(out, ticks); run<16, 0, 4>(out, ticks);
    run<0, 0, 8>(out, ticks); run<2, 0, 8>(out, ticks); run<4, 0, 8>(out, ticks); run<8, 0, 8>(out, ticks);
    run<0, 1, 8>(out, ticks); run<2, 1, 8>(out, ticks); run<4, 1, 8>(out, ticks); run<5, 1, 8>(out, ticks); run<6, 1, 8>(out, ticks); run<8, 1, 8>(out, ticks); run<12, 1, 8>(out, ticks); run<16, 1, 8>(out, ticks);
    run<0, 2, 8>(out, ticks); run<4, 2, 8>(out, ticks); run<6, 2, 8>(out, ticks); run<8, 2, 8>(out, ticks); run<16, 2, 8>(out, ticks);
    run<4, 3, 8>(out, ticks); run<6, 3, 8>(out, ticks); run<8, 3, 8>(out, ticks); run<16, 3, 8>(out, ticks);
    return 0;
}
