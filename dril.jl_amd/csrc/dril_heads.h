// dril_heads.h — Categorical / DiagGaussian device functions shared by the forward kernels (dril_kernels.hip) and the loss head of the update kernels (dril_grad_common.h)
#pragma once
#include "dril_internal.h"

namespace dril {

// =============================================================================================
// distribution heads shared by policy_kernel / rollout_kernel / ppo_grad_kernel
// =============================================================================================
// Lux.softmax + Categorical: layer_forward.jl:141-149, categorical.jl:20-52
template <int A> __device__ __forceinline__ void softmax_n(const float (&z)[A], float (&p)[A]) {
    float m = z[0];
#pragma unroll
    for (int i = 1; i < A; ++i) m = fmaxf(m, z[i]);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < A; ++i) { p[i] = fexp(z[i] - m); s += p[i]; }
    const float inv = frcp(s);
#pragma unroll
    for (int i = 0; i < A; ++i) p[i] = p[i] * inv;
}
template <int A> __device__ __forceinline__ int categorical_sample(const float (&p)[A], double u) {
    float cs = 0.f; int a = A - 1; bool found = false;
#pragma unroll
    for (int i = 0; i < A; ++i) { cs += p[i]; if (!found && (double)cs >= u) { a = i; found = true; } }
    return a;
}
template <int A> __device__ __forceinline__ float pick(const float (&p)[A], int a) {
    float v = p[0];
#pragma unroll
    for (int i = 1; i < A; ++i) v = (a == i) ? p[i] : v;
    return v;
}
template <int A> __device__ __forceinline__ float categorical_entropy(const float (&p)[A]) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < A; ++i) s += p[i] * flog(p[i]);
    return -s;
}
constexpr float kLog2Pi = 1.8378770664093453f;
// DiagGaussian logpdf / entropy: diagGaussian.jl:25-43
template <int A> __device__ __forceinline__ float gauss_logpdf(const float (&x)[A], const float (&mu)[A], const float* ls) {
    float lss = 0.f, dss = 0.f;
#pragma unroll
    for (int i = 0; i < A; ++i) { lss += ls[i]; const float d = x[i] - mu[i]; dss += d * d * fexp(-2.0f * ls[i]); }
    return -0.5f * (2.0f * lss + dss + (float)A * kLog2Pi);
}
template <int A> __device__ __forceinline__ float gauss_entropy(const float* ls) {
    float lss = 0.f;
#pragma unroll
    for (int i = 0; i < A; ++i) lss += ls[i];
    return 0.5f * (float)A * (1.0f + kLog2Pi) + lss;
}


}  // namespace dril
