// microbenchmark: can ONE wave (one wave per SIMD) run VALU work under its own v_mfma_f32_32x32x16_bf16 chain?
// per group: 1 MFMA (chain over NACC accumulators, round robin) + N independent VALU ops of one class; reports ticks per group (s_memtime) and kernel time.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int N, int NACC, int CLS, int WAVES, bool AG = false>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, unsigned long long* ticks, int iters) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float v[16]; for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
    bf16x8 a, b; for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.5f + i); b[i] = (__bf16)(1.0f + i); }
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (AG) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[u % NACC]) : "v"(a), "v"(b));      // accumulator in AGPRs (round 5: does the VGPR-form accumulator cost the fillers their slots?)
            else acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u % NACC], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                float& x = v[i % 16];
                if (CLS == 0) x = __builtin_fmaf(x, 1.0001f, 0.5f);
                else if (CLS == 1) x = __builtin_amdgcn_exp2f(x);
                else if (CLS == 2) x = __uint_as_float(__float_as_uint(x) & 0xffff0ff0u);
                else x = __uint_as_float(__builtin_amdgcn_perm(__float_as_uint(x), __float_as_uint(v[(i + 1) % 16]), 0x07060302u));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0; for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r]; for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}
template <int N, int NACC, int CLS, int WAVES, bool AG = false> void run(float* out, unsigned long long* ticks) {
    const int iters = 2000;
    k<N, NACC, CLS, WAVES, AG><<<256, 64 * WAVES>>>(out, ticks, iters); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<N, NACC, CLS, WAVES, AG><<<256, 64 * WAVES>>>(out, ticks, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    const double groups = iters * 8.0;
    const char* cls[] = {"v_fma_f32", "v_exp_f32", "v_and_b32", "v_perm_b32"};
    printf("%s waves/SIMD %d  accs %d  %-10s x %2d per MFMA : %6.1f ticks per group   %.1f ns per group per wave  (kernel %.3f ms)\n", AG ? "AGPR-acc" : "VGPR-acc", WAVES / 4, NACC, cls[CLS], N,
           (double)t / groups, ms * 1e6 / groups, ms);
}
int main() {
    float* out; unsigned long long* ticks; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&ticks, 8);
    run<0, 1, 0, 4>(out, ticks); run<0, 2, 0, 4>(out, ticks); run<0, 4, 0, 4>(out, ticks);
    run<2, 1, 0, 4>(out, ticks); run<4, 1, 0, 4>(out, ticks); run<6, 1, 0, 4>(out, ticks); run<8, 1, 0, 4>(out, ticks); run<12, 1, 0, 4>(out, ticks); run<16, 1, 0, 4>(out, ticks);
    run<4, 2, 0, 4>(out, ticks); run<6, 2, 0, 4>(out, ticks); run<8, 2, 0, 4>(out, ticks); run<16, 2, 0, 4>(out, ticks);
    run<4, 2, 1, 4>(out, ticks); run<8, 2, 1, 4>(out, ticks);
    run<4, 2, 2, 4>(out, ticks); run<8, 2, 2, 4>(out, ticks);
    run<4, 2, 3, 4>(out, ticks); run<8, 2, 3, 4>(out, ticks);
    run<0, 2, 0, 8>(out, ticks); run<8, 2, 0, 8>(out, ticks); run<16, 2, 0, 8>(out, ticks);
    // the same with the accumulators in AGPRs (inline asm): round 5
    run<0, 2, 0, 4, true>(out, ticks); run<2, 2, 0, 4, true>(out, ticks); run<4, 2, 0, 4, true>(out, ticks); run<6, 2, 0, 4, true>(out, ticks); run<8, 2, 0, 4, true>(out, ticks); run<16, 2, 0, 4, true>(out, ticks);
    run<4, 2, 1, 4, true>(out, ticks); run<0, 2, 0, 8, true>(out, ticks); run<8, 2, 0, 8, true>(out, ticks); run<16, 2, 0, 8, true>(out, ticks);
    return 0;
}
