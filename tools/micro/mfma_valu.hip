// microbenchmark: can ONE wave overlap its own VALU work with its own v_mfma_f32_32x32x2_f32 chain?
// per iteration: 1 dependent MFMA + N independent v_fma_f32; reports cycles per iteration (s_memtime) for one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int N, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, unsigned long long* ticks, int iters) {
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float v[16]; for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
    float a = threadIdx.x * 0.5f, b = 1.0001f;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < N; ++i) v[i % 16] = __builtin_fmaf(v[i % 16], 1.0001f, 0.5f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0; for (int r = 0; r < 16; ++r) s += acc[r]; for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}
template <int N, int WAVES> void run(float* out, unsigned long long* ticks) {
    const int iters = 2000;
    k<N, WAVES><<<256, 64 * WAVES>>>(out, ticks, iters); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<N, WAVES><<<256, 64 * WAVES>>>(out, ticks, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    const double flops = 256.0 * WAVES * iters * 8.0 * 2.0 * 32 * 32 * 2;
    printf("waves/SIMD %d  VALU per MFMA %2d : %6.1f ticks per (MFMA + VALU group)   kernel %.3f ms  %.1f TFLOP/s  (%.2f GHz if a tick is a cycle)\n", WAVES / 4, N,
           (double)t / (iters * 8.0), ms, flops / (ms * 1e-3) / 1e12, (double)t / (ms * 1e-3) / 1e9);
}
int main() {
    float* out; unsigned long long* ticks; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&ticks, 8);
    run<0, 4>(out, ticks); run<4, 4>(out, ticks); run<8, 4>(out, ticks); run<12, 4>(out, ticks); run<16, 4>(out, ticks); run<24, 4>(out, ticks); run<32, 4>(out, ticks);
    run<0, 8>(out, ticks); run<8, 8>(out, ticks); run<16, 8>(out, ticks); run<32, 8>(out, ticks);
    return 0;
}
