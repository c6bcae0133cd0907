// microbenchmark: what can one wave overlap with its own v_mfma_f32_32x32x2_f32 chain?  Per MFMA: N ops of one class.
//   class 0 v_fma_f32   1 v_pk_fma_f32 (2 floats)   2 v_exp_f32   3 v_add_u32 (integer)   4 ds_read_b128   5 v_mov_b32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CLS, int N, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, int iters) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i * 0.001f;
    __syncthreads();
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float v[16]; f32x2 p[16]; unsigned u[16]; f32x4 l[4];
    for (int i = 0; i < 16; ++i) { v[i] = threadIdx.x * 0.001f + i; p[i] = f32x2{v[i], v[i] + 1}; u[i] = threadIdx.x + i; }
    for (int i = 0; i < 4; ++i) l[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 0.5f, b = 1.0001f;
    const f32x2 c1 = {1.0001f, 0.9999f}, c2 = {0.5f, 0.25f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                if (CLS == 0) v[i % 16] = __builtin_fmaf(v[i % 16], 1.0001f, 0.5f);
                else if (CLS == 1) p[i % 16] = p[i % 16] * c1 + c2;
                else if (CLS == 2) v[i % 16] = __builtin_amdgcn_exp2f(v[i % 16]);
                else if (CLS == 3) u[i % 16] = u[i % 16] * 3u + 7u;
                else if (CLS == 4) l[i % 4] += *reinterpret_cast<const f32x4*>(lds + ((threadIdx.x * 4 + 16 * i + 64 * g) & 4092));
                else asm volatile("v_mov_b32 %0, %1" : "=v"(v[i % 16]) : "v"(v[(i + 1) % 16]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0; for (int r = 0; r < 16; ++r) s += acc[r];
    for (int i = 0; i < 16; ++i) s += v[i] + p[i][0] + p[i][1] + (float)u[i];
    for (int i = 0; i < 4; ++i) s += l[i][0] + l[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CLS, int N, int WAVES> void run(float* out, const char* name) {
    const int iters = 2000;
    k<CLS, N, WAVES><<<256, 64 * WAVES>>>(out, iters); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<CLS, N, WAVES><<<256, 64 * WAVES>>>(out, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * WAVES * iters * 8.0 * 2.0 * 32 * 32 * 2;
    printf("%-14s x%2d per MFMA, %d wave(s)/SIMD: %.3f ms  %6.1f TFLOP/s (MFMA only)\n", name, N, WAVES / 4, ms, flops / (ms * 1e-3) / 1e12);
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    run<0, 0, 8>(out, "baseline");
    run<0, 8, 8>(out, "v_fma_f32"); run<1, 8, 8>(out, "v_pk_fma_f32"); run<1, 4, 8>(out, "v_pk_fma_f32"); run<2, 8, 8>(out, "v_exp_f32"); run<3, 8, 8>(out, "v_mad_u32");
    run<4, 4, 8>(out, "ds_read_b128"); run<4, 8, 8>(out, "ds_read_b128"); run<5, 8, 8>(out, "v_mov_b32");
    run<0, 8, 4>(out, "v_fma_f32"); run<1, 8, 4>(out, "v_pk_fma_f32"); run<2, 8, 4>(out, "v_exp_f32"); run<3, 8, 4>(out, "v_mad_u32"); run<4, 4, 4>(out, "ds_read_b128"); run<5, 8, 4>(out, "v_mov_b32");
    return 0;
}
