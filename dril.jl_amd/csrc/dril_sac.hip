// dril_sac.hip — the off-policy (SAC) path of libdril_hip.so: kernels + the C ABI of include/dril_sac.h.
//
// Reference: src/algorithms/sac.jl (losses :93-150, update! :299-404, train! :406-549), src/buffers/replay_buffer.jl,
// src/buffers/off_policy_collection.jl, src/DRiLDistributions/squashedDiagGaussian.jl.  BASELINE.json configs[4]:
// Pendulum-v1, 4096 device envs, SACLayer [512,512] relu, batch 256.
//
// Shape of the work (docs/sac.md): one gradient step is ~25 small dense contractions (256 samples x 512 x 512) separated by
// per-sample head math, all on one stream with no host round trip; one env step of the collection is the actor forward over
// 4096 envs + the env kernels of the on-policy path + a ring write.  The contractions are fp32 MFMA (v_mfma_f32_32x32x2_f32:
// exact fp32 products, fp32 accumulate) in ONE generic strided kernel: a workgroup owns one 32x32 output tile and its four
// waves split the contraction axis (in-workgroup split-K, summed through LDS in fixed wave order => deterministic), because
// with only 128..272 output tiles per contraction the chip is filled by K-parallelism, not by tiles.  Bias rides in the forward
// epilogue; [dW | db] come out of one contraction by appending a ones column to the activation operand (the flat parameter
// layout stores b right behind the column-major W, i.e. [W | b] is one (out x (in+1)) column-major matrix).
#ifndef DRIL_SAC_ADAM_BLOCKS
#define DRIL_SAC_ADAM_BLOCKS 1024
#endif
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/dril_sac.h"
#include "dril_internal.h"
#include "dril_gemm.h"

using namespace dril;

#define DRIL_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

thread_local std::string g_sac_create_error;


// =================================================================================================================
// SquashedDiagGaussian (squashedDiagGaussian.jl:24-46) — per-sample scalar math, accurate libm
// =================================================================================================================
constexpr int kMaxA = 16;
constexpr float kLog2Pi = 1.8378770664093453f;
__device__ inline float softplus_f(float x) { return log1pf(expf(-fabsf(x))) + (x > 0.f ? x : 0.f); }   // Lux.softplus
// a = tanh(mu + exp(ls) * noise); returns logpdf(d, a) (:36-46), g = atanh(clamp(a))
__device__ inline float squashed_sample_logp(const float* mu, const float* ls, const float* noise, int A, float* a, float* g) {
    const float eps = 1.0e-6f, lo = -1.0f + eps, hi = 1.0f - eps;
    float lss = 0.f, dss = 0.f, corr = 0.f;
    for (int i = 0; i < A; ++i) {
        a[i] = tanhf(mu[i] + expf(ls[i]) * noise[i]);
        const float xc = a[i] < lo ? lo : (a[i] > hi ? hi : a[i]);
        g[i] = atanhf(xc);
        corr += 2.0f * (logf(2.0f) - g[i] - softplus_f(-2.0f * g[i]));
        lss += ls[i]; const float d = g[i] - mu[i]; dss += d * d * expf(-2.0f * ls[i]);
    }
    return -0.5f * (2.0f * lss + dss + (float)A * kLog2Pi) - corr;       // diagGaussian.jl:25-36 minus the correction
}

// deterministic block reduction (blockDim.x == 256): pairwise tree in shared memory
__device__ inline double block_sum(double v, double* sh) {
    const int t = threadIdx.x;
    sh[t] = v; __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (t < s) sh[t] += sh[t + s]; __syncthreads(); }
    const double r = sh[0]; __syncthreads();
    return r;
}

struct SacRng { uint64_t key; uint64_t u; };
__device__ inline float sac_noise(SacRng r, int stream, int i, int a) {
    uint32_t o[4];
    philox4x32_10((uint32_t)r.key, (uint32_t)(r.key >> 32), (uint32_t)r.u, (uint32_t)(r.u >> 32), (uint32_t)stream, (uint32_t)(i * 4 + a / 2), o);
    return (a & 1) ? randn_f32(o[2], o[3]) : randn_f32(o[0], o[1]);
}

// ---- get_data_loader (replay_buffer.jl:116-157): gather one batch + its three noise draws -------------------------------
struct GatherArgs {
    int B, D, A; long long cap, head, size;
    const float *rb_obs, *rb_next, *rb_act, *rb_rew; const uint8_t* rb_term;
    const long long* inj_idx; const float *inj_ne, *inj_nn, *inj_np;     // injected batch (tests), per sample; null = Philox
    SacRng rng;
    float* xa;        // [2B][D]: rows [0,B) observations, [B,2B) next observations (actor input)
    float* xq;        // [B][D+A]: (obs, stored action)
    float *rew, *ne, *nn, *np; uint8_t* term;
};
__global__ void sac_gather_kernel(GatherArgs g) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.B) return;
    long long j;
    if (g.inj_idx) j = g.inj_idx[i];
    else {
        uint32_t o[4];
        philox4x32_10((uint32_t)g.rng.key, (uint32_t)(g.rng.key >> 32), (uint32_t)g.rng.u, (uint32_t)(g.rng.u >> 32), 4u, (uint32_t)i, o);
        j = (long long)(u01_f64(o[0], o[1]) * (double)g.size); if (j >= g.size) j = g.size - 1;
    }
    const long long slot = (g.head + j) % g.cap;
    for (int d = 0; d < g.D; ++d) {
        const float o = g.rb_obs[slot * g.D + d];
        g.xa[(size_t)i * g.D + d] = o; g.xq[(size_t)i * (g.D + g.A) + d] = o;
        g.xa[(size_t)(g.B + i) * g.D + d] = g.rb_next[slot * g.D + d];
    }
    for (int a = 0; a < g.A; ++a) {
        g.xq[(size_t)i * (g.D + g.A) + g.D + a] = g.rb_act[slot * g.A + a];
        g.ne[i * g.A + a] = g.inj_ne ? g.inj_ne[i * g.A + a] : sac_noise(g.rng, 5, i, a);
        g.nn[i * g.A + a] = g.inj_nn ? g.inj_nn[i * g.A + a] : sac_noise(g.rng, 6, i, a);
        g.np[i * g.A + a] = g.inj_np ? g.inj_np[i * g.A + a] : sac_noise(g.rng, 7, i, a);
    }
    g.rew[i] = g.rb_rew[slot]; g.term[i] = g.rb_term[slot];
}

// one sample's first layer h1[u] = act(W1[u, :] . x + b1[u]) for all H1 units by the 256 threads of a block; W1 (H1 x in) column-major; x in shared memory
constexpr int kFusedL1MaxIn = 16;    // the per-sample form re-reads W1 per block: only for small inputs (Pendulum: 3 / 4); wider nets keep the contraction
__device__ __forceinline__ void first_layer_row(const float* __restrict__ W1, const float* __restrict__ b1, const float* x, int in, int H1, int relu, float* __restrict__ h1) {
    for (int u = threadIdx.x; u < H1; u += blockDim.x) {
        float acc = b1[u];
        for (int k = 0; k < in; ++k) acc = fmaf(W1[u + (size_t)k * H1], x[k], acc);
        h1[u] = relu ? relu_nan(acc) : tanhf(acc);
    }
}
// the same with the weights requested EARLY (before the barrier behind which the sample's inputs appear: one round of memory latency less in kernels that are
// nothing but dependent round trips): in <= 4 features, H1 <= 512 (two units per thread of a 256-thread block); same fma order
constexpr int kL1PreMaxIn = 4, kL1PreMaxH = 512;
struct L1Pre { float w[2][kL1PreMaxIn]; float b[2]; };
__device__ __forceinline__ bool first_layer_pre_ok(int in, int H1) { return in <= kL1PreMaxIn && H1 <= kL1PreMaxH; }
__device__ __forceinline__ void first_layer_pre(L1Pre& r, const float* __restrict__ W1, const float* __restrict__ b1, int in, int H1) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int u = threadIdx.x + 256 * t; const bool ok = u < H1;
#pragma unroll
        for (int k = 0; k < kL1PreMaxIn; ++k) r.w[t][k] = (ok && k < in) ? W1[u + (size_t)k * H1] : 0.f;
        r.b[t] = ok ? b1[u] : 0.f;
    }
}
__device__ __forceinline__ void first_layer_post(const L1Pre& r, const float* x, int in, int H1, int relu, float* __restrict__ h1) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int u = threadIdx.x + 256 * t;
        if (u < H1) {
            float acc = r.b[t];
#pragma unroll
            for (int k = 0; k < kL1PreMaxIn; ++k) if (k < in) acc = fmaf(r.w[t][k], x[k], acc);
            h1[u] = relu ? relu_nan(acc) : tanhf(acc);
        }
    }
}
// =================================================================================================================
// The collection forward (off_policy_collection.jl:55: predict_actions_raw over all envs) for a narrow observation — BASELINE configs[4]: 4096 envs, [512,512].
// Its second layer, a 512 x 4096 x 512 contraction, is the largest kernel of a SAC iteration and the one place of the SAC path where the matrix pipe is the bound
// (31 us on v_mfma_f32_32x32x2_f32: 12 of MFMA + a 14 us staging skeleton, docs/sac.md).  Here it runs on v_mfma_f32_32x32x16_f16 with the fp32-equivalent two-piece
// operands of the PPO kernels (dril_device.h: x S = hi + lo, three products per k16 step, f32 accumulate: 2^-24 relative), and — unlike a split at staging time, which
// every workgroup would repeat for the rows it stages — the operands are split ONCE, by their producers:
//   * sac_collect_l1_kernel (first layer, elementwise) writes h1 as f32 AND as two f16 planes [n][H1] (kColActScale h1);
//   * the same launch (extra blocks) re-cuts the actor's W2 into two f16 planes [out][in] (kColWScale W2, k-contiguous rows) whenever the actor changed;
//   * sac_collect_l2_kernel: 64 (n) x 128 (m) output block per workgroup, 4 waves as 2 x 2 (one n-tile x two m-tiles each), 64-deep chunks of both operands'
//     planes through a double-buffered LDS image (coalesced 16-byte loads, rows of 144 bytes: conflict-free ds_read_b128 fragment reads), ONE barrier per chunk,
//     6 MFMAs per 6 fragment reads; D = h1 . W2' so that consecutive lanes hold consecutive output units: 128-byte runs in the store of h2 [n][H2].
// RANGE.  f16 pieces overflow beyond 65 504 / scale: |h1| >= 4 094 (relu activations are unbounded) or |W2| >= 1 023.  The producers check every value they cut and
// raise a tagged flag (atomicMax with the launch's tag: no clearing pass); the contraction reads the flags first and, if either is up, computes its block with f32
// MFMAs straight from the f32 operands instead (slow, exact) — decided on the device, no host round trip, never a silently wrong action.
// =================================================================================================================
constexpr float kColActScale = 16.0f, kColWScale = 64.0f;
constexpr float kF16Max = 65504.0f;
struct CollectL1Args {
    int E, D, H1, relu; const float* x; const float* W1; const float* b1; float* h1;
    unsigned* h1p; int* flag; int tag;                         // h1p != null: also the two f16 planes [2][E][H1 / 2] (packed pairs); overflow -> atomicMax(flag, tag)
    int nb_l1; const float* W2; int H2; unsigned* w2p; int* w2flag; int w2tag;   // blocks >= nb_l1 (present when the actor changed): W2 (H2 x H1, column-major) -> planes [2][H2][H1 / 2]
};
__global__ __launch_bounds__(256) void sac_collect_l1_kernel(CollectL1Args f) {
    __shared__ float xs[8][4];
    if ((int)blockIdx.x >= f.nb_l1) {                                                   // ---- the actor's W2 as two f16 planes, rows of one output unit contiguous in k ----
        const int words = f.H2 * (f.H1 / 2);                                            // one packed pair (k, k + 1) per word
        bool bad = false;
        for (int w = ((int)blockIdx.x - f.nb_l1) * 256 + threadIdx.x; w < words; w += ((int)gridDim.x - f.nb_l1) * 256) {
            const int o = w % f.H2, kp = w / f.H2;                                      // consecutive threads: consecutive output units (the column-major weight's unit stride)
            const float a = kColWScale * f.W2[o + (size_t)(2 * kp) * f.H2], b = kColWScale * f.W2[o + (size_t)(2 * kp + 1) * f.H2];
            bad |= !(fabsf(a) < kF16Max) || !(fabsf(b) < kF16Max);
            unsigned hi, lo; split2_pair(a, b, hi, lo);
            f.w2p[(size_t)o * (f.H1 / 2) + kp] = hi; f.w2p[(size_t)words + (size_t)o * (f.H1 / 2) + kp] = lo;
        }
        if (bad) atomicMax(f.w2flag, f.w2tag);
        return;
    }
    const int e0 = blockIdx.x * 8;
    if (threadIdx.x < 32) { const int r = threadIdx.x >> 2, d = threadIdx.x & 3, e = e0 + r; xs[r][d] = (d < f.D && e < f.E) ? f.x[(size_t)e * f.D + d] : 0.f; }
    __syncthreads();
    bool bad = false;
    for (int u2 = threadIdx.x; u2 < f.H1 / 2; u2 += 256) {                              // a thread owns units 2 u2, 2 u2 + 1 (one packed word per plane); H1 is even (host-checked)
        float wv[2][4], bv[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            bv[t] = f.b1[2 * u2 + t];
#pragma unroll
            for (int d = 0; d < 4; ++d) wv[t][d] = d < f.D ? f.W1[2 * u2 + t + (size_t)d * f.H1] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (e0 + r >= f.E) break;
            float v[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float acc = bv[t];
#pragma unroll
                for (int d = 0; d < 4; ++d) acc = fmaf(wv[t][d], xs[r][d], acc);        // (d >= D: zero weights) — the order of first_layer_row
                v[t] = f.relu ? relu_nan(acc) : tanhf(acc);
            }
            *reinterpret_cast<float2*>(f.h1 + (size_t)(e0 + r) * f.H1 + 2 * u2) = make_float2(v[0], v[1]);
            if (f.h1p) {
                const float a = kColActScale * v[0], b = kColActScale * v[1];
                bad |= !(fabsf(a) < kF16Max) || !(fabsf(b) < kF16Max);
                unsigned hi, lo; split2_pair(a, b, hi, lo);
                f.h1p[(size_t)(e0 + r) * (f.H1 / 2) + u2] = hi; f.h1p[(size_t)f.E * (f.H1 / 2) + (size_t)(e0 + r) * (f.H1 / 2) + u2] = lo;
            }
        }
    }
    if (bad) atomicMax(f.flag, f.tag);
}
constexpr int kL2TN = 64, kL2TM = 128, kL2KC = 64, kL2Row = kL2KC + 8;                  // output block (n x m), chunk depth, LDS row in f16 elements (144 bytes)
constexpr int kL2Plane = (kL2TN + kL2TM) * kL2Row;                                      // f16 elements of one plane (hi or lo) of one buffer: A rows then B rows
constexpr size_t kL2LdsBytes = sizeof(_Float16) * 2 * 2 * kL2Plane;                     // two buffers x two planes = 110 592 B
struct CollectL2Args {
    int E, H1, H2, relu; const unsigned* h1p; const unsigned* w2p; const float* b2; float* h2;
    const int* flag; int tag; const int* w2flag; int w2tag; const float* h1; const float* W2;   // out-of-range: exact-f32 path from the f32 operands
};
// loader of sac_collect_l2_kernel: per chunk and plane 512 sixteen-byte pieces of the activation rows (two per thread) and 1 024 of the weight rows (four per thread);
// piece p = (row p >> 3, seg p & 7): 8 f16 of that row at k = kc + 8 seg — eight consecutive lanes read one row's 128 bytes
__device__ __forceinline__ void l2_issue(const CollectL2Args& a, int tid, int n0, int m0, int kc, u32x4 (&la)[4], u32x4 (&lb)[8]) {
    const int hw = a.H1 / 2;                                                           // words per row in memory
    const size_t plane_a = (size_t)a.E * hw, plane_b = (size_t)a.H2 * hw;              // words per plane in memory
#pragma unroll
    for (int j = 0; j < 4; ++j) {                                                      // j = 2 plane + u
        const int p = tid + 256 * (j & 1), row = p >> 3, seg = p & 7; int gr = n0 + row; gr = gr < a.E ? gr : a.E - 1;   // rows past E re-read the last env (in bounds, never stored)
        la[j] = *reinterpret_cast<const u32x4*>(a.h1p + (size_t)(j >> 1) * plane_a + (size_t)gr * hw + (kc >> 1) + 4 * seg);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {                                                      // j = 4 plane + u
        const int p = tid + 256 * (j & 3), row = p >> 3, seg = p & 7;
        lb[j] = *reinterpret_cast<const u32x4*>(a.w2p + (size_t)(j >> 2) * plane_b + (size_t)(m0 + row) * hw + (kc >> 1) + 4 * seg);
    }
}
__device__ __forceinline__ void l2_commit(_Float16* l2s, int tid, int buf, const u32x4 (&la)[4], const u32x4 (&lb)[8]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int p = tid + 256 * (j & 1); *reinterpret_cast<u32x4*>(l2s + (size_t)(2 * buf + (j >> 1)) * kL2Plane + (size_t)(p >> 3) * kL2Row + 8 * (p & 7)) = la[j]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int p = tid + 256 * (j & 3); *reinterpret_cast<u32x4*>(l2s + (size_t)(2 * buf + (j >> 2)) * kL2Plane + (size_t)(kL2TN + (p >> 3)) * kL2Row + 8 * (p & 7)) = lb[j]; }
}
__global__ __launch_bounds__(256) void sac_collect_l2_kernel(CollectL2Args a) {
    extern __shared__ __attribute__((aligned(16))) _Float16 l2s[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, kh = lane >> 5, wn = wave & 1, wm = wave >> 1;
    const int n0 = blockIdx.x * kL2TN, m0 = blockIdx.y * kL2TM;
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float unscale = 1.0f / (kColActScale * kColWScale);
    if (*a.flag == a.tag || *a.w2flag == a.w2tag) {                                    // ---- out of f16's range somewhere this step: v_mfma_f32_32x32x2_f32 from the f32 operands ----
        const int n = n0 + 32 * wn + c; const int nn = n < a.E ? n : a.E - 1;
        for (int k0 = 0; k0 < a.H1; k0 += 8) {
            const int k = k0 + 4 * kh;
            const float4 av = *reinterpret_cast<const float4*>(a.h1 + (size_t)nn * a.H1 + k);          // H1 % 64 == 0 (host-checked)
            const float af[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int m = m0 + 64 * wm + 32 * t + c;
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[t] = mfma32(af[q], a.W2[m + (size_t)(k + q) * a.H2], acc[t]);
            }
        }
        unscale = 1.0f;
    } else {
        u32x4 la[4], lb[8];
        const int NC = a.H1 / kL2KC;
        l2_issue(a, tid, n0, m0, 0, la, lb); l2_commit(l2s, tid, 0, la, lb);
        __syncthreads();
        for (int ci = 0; ci < NC; ++ci) {
            const int buf = ci & 1;
            if (ci + 1 < NC) l2_issue(a, tid, n0, m0, (ci + 1) * kL2KC, la, lb);         // next chunk in flight under this chunk's MFMAs (two chunks ahead, in a second
            const _Float16* Ph = l2s + (size_t)(2 * buf) * kL2Plane; const _Float16* Pl = Ph + kL2Plane;   // register set, was measured: 18.0 vs 18.6 us — not what the kernel waits for)
#pragma unroll
            for (int st = 0; st < kL2KC / 16; ++st) {
                const int ko = 16 * st + 8 * kh;
                const f16x8 ah = *reinterpret_cast<const f16x8*>(Ph + (size_t)(32 * wn + c) * kL2Row + ko), al = *reinterpret_cast<const f16x8*>(Pl + (size_t)(32 * wn + c) * kL2Row + ko);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const size_t brow = (size_t)(kL2TN + 64 * wm + 32 * t + c) * kL2Row + ko;
                    const f16x8 bh = *reinterpret_cast<const f16x8*>(Ph + brow), bl = *reinterpret_cast<const f16x8*>(Pl + brow);
                    acc[t] = mfma_split3(ah, al, bh, bl, acc[t]);
                }
            }
            if (ci + 1 < NC) l2_commit(l2s, tid, buf ^ 1, la, lb);                       // (every wave left that buffer at the previous barrier)
            __syncthreads();
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int m = m0 + 64 * wm + 32 * t + c; const float b = a.b2[m];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + 32 * wn + rowfn(r, kh);
            if (n >= a.E) continue;
            const float v = fmaf(acc[t][r], unscale, b);
            a.h2[(size_t)n * a.H2 + m] = a.relu ? relu_nan(v) : tanhf(v);
        }
    }
}
// get_data_loader + the first layer of the actor on (obs | next obs) and of the two critics on (obs, stored action): sac_gather_kernel + the first launch of two
// net_forward calls.  One block per sample.
struct GatherL1Args {
    GatherArgs g; int H1, relu;
    const float *aW1, *ab1; float* ah1;                       // actor: W1 (H1 x D), b1; h1 [2B][H1]
    const float* P; int qw1, qb1; long long zP; long long zh; float* qh1;   // critic z: W1 at P + qw1 + z * zP; h1 [Z][nq][H1] with batch stride zh
};
__global__ __launch_bounds__(256) void sac_gather_l1_kernel(GatherL1Args f) {
    __shared__ float xs[2][kFusedL1MaxIn], xqs[kFusedL1MaxIn];
    const GatherArgs& g = f.g;
    const int i = blockIdx.x, W = g.D + g.A;
    const bool pre = first_layer_pre_ok(W, f.H1);                                    // weights of the three first layers requested before the sample is known
    L1Pre pa, pq[2];
    if (pre) { first_layer_pre(pa, f.aW1, f.ab1, g.D, f.H1); for (int z = 0; z < 2; ++z) first_layer_pre(pq[z], f.P + f.qw1 + z * f.zP, f.P + f.qb1 + z * f.zP, W, f.H1); }
    if (threadIdx.x == 0) {
        long long j;
        if (g.inj_idx) j = g.inj_idx[i];
        else {
            uint32_t o[4];
            philox4x32_10((uint32_t)g.rng.key, (uint32_t)(g.rng.key >> 32), (uint32_t)g.rng.u, (uint32_t)(g.rng.u >> 32), 4u, (uint32_t)i, o);
            j = (long long)(u01_f64(o[0], o[1]) * (double)g.size); if (j >= g.size) j = g.size - 1;
        }
        const long long slot = (g.head + j) % g.cap;
        for (int d = 0; d < g.D; ++d) {
            const float o = g.rb_obs[slot * g.D + d], n = g.rb_next[slot * g.D + d];
            g.xa[(size_t)i * g.D + d] = o; g.xq[(size_t)i * W + d] = o; g.xa[(size_t)(g.B + i) * g.D + d] = n;
            xs[0][d] = o; xs[1][d] = n; xqs[d] = o;
        }
        for (int a = 0; a < g.A; ++a) { const float act = g.rb_act[slot * g.A + a]; g.xq[(size_t)i * W + g.D + a] = act; xqs[g.D + a] = act; }
        g.rew[i] = g.rb_rew[slot]; g.term[i] = g.rb_term[slot];
    } else if (threadIdx.x >= 64 && threadIdx.x < 64 + 3 * g.A) {                    // the three noise arrays of the step on their own threads (a Philox block + a normal transform each), in another wave
        const int q = threadIdx.x - 64, which = q / g.A, a = q - which * g.A;
        const float* inj = which == 0 ? g.inj_ne : which == 1 ? g.inj_nn : g.inj_np;
        float* dst = which == 0 ? g.ne : which == 1 ? g.nn : g.np;
        dst[i * g.A + a] = inj ? inj[i * g.A + a] : sac_noise(g.rng, 5 + which, i, a);
    }
    __syncthreads();
    if (pre) {
        first_layer_post(pa, xs[0], g.D, f.H1, f.relu, f.ah1 + (size_t)i * f.H1);
        first_layer_post(pa, xs[1], g.D, f.H1, f.relu, f.ah1 + (size_t)(g.B + i) * f.H1);
#pragma unroll
        for (int z = 0; z < 2; ++z) first_layer_post(pq[z], xqs, W, f.H1, f.relu, f.qh1 + z * f.zh + (size_t)i * f.H1);
        return;
    }
    first_layer_row(f.aW1, f.ab1, xs[0], g.D, f.H1, f.relu, f.ah1 + (size_t)i * f.H1);
    first_layer_row(f.aW1, f.ab1, xs[1], g.D, f.H1, f.relu, f.ah1 + (size_t)(g.B + i) * f.H1);
#pragma unroll
    for (int z = 0; z < 2; ++z) first_layer_row(f.P + f.qw1 + z * f.zP, f.P + f.qb1 + z * f.zP, xqs, W, f.H1, f.relu, f.qh1 + z * f.zh + (size_t)i * f.H1);
}

// In-stream time stamps of dril_sac_iterate: the fps of every iteration's collection (off_policy_collection.jl:126-128) and the update / collection split of the
// profile need the time at the phase boundaries.  A HIP event per boundary costs the stream ~5 us each (three per iteration: 16 us of a 205 us iteration, seen as
// gaps in the rocprofv3 trace); instead the LAST kernel of a phase stores the constant-rate wall clock (s_memrealtime) when its highest-numbered workgroup is done —
// one store, no atomics, nothing waits.  Kernels of one stream run back to back, so a phase lasts from the previous phase's stamp to its own.
__device__ __forceinline__ void phase_stamp(unsigned long long* stamp) {
    if (stamp && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *stamp = wall_clock64();
}
__global__ void sac_stamp_kernel(unsigned long long* stamp) { phase_stamp(stamp); }
// device scalars shared by the kernels of one gradient step
struct SacScalars { float log_ent, ent_m, ent_v, alpha; };

// ---- entropy-coefficient step (sac.jl:313-343) + next actions for the critic target (:131) ----------------------------------
struct EntNextArgs {
    int B, D, A; const float* mu;    // [2B][A] actor means of (obs | next obs)
    const float* log_std; const float *ne, *nn; const float* xa;
    float* xq_next;   // [B][D+A] (next obs, next action)
    float* nlp;       // [B] next log-probs
    const float* np; float *xq_pi, *a_pi, *g_pi, *lp_pi;   // the actor-loss sample (sac_actor_loss :101): it only depends on the actor, which does not change before the actor step
    SacScalars* sc; float target_entropy, lr, b1, b2, eps, bt1, bt2; int auto_ent;
    float* stats;     // [8]: 0 actor_loss 1 critic_loss 2 entropy_loss 3 mean_q 4 ent_coef 5 |gc|^2 6 |ga|^2
};
__global__ __launch_bounds__(256) void sac_ent_next_kernel(EntNextArgs g) {
    __shared__ double sh[256];
    float ls[kMaxA]; for (int a = 0; a < g.A; ++a) ls[a] = g.log_std[a];
    double s = 0;
    for (int i = threadIdx.x; i < g.B; i += 256) {
        float a_[kMaxA], gg[kMaxA];
        if (g.auto_ent) s += (double)(squashed_sample_logp(g.mu + (size_t)i * g.A, ls, g.ne + (size_t)i * g.A, g.A, a_, gg) + g.target_entropy);
        g.nlp[i] = squashed_sample_logp(g.mu + (size_t)(g.B + i) * g.A, ls, g.nn + (size_t)i * g.A, g.A, a_, gg);
        for (int d = 0; d < g.D; ++d) g.xq_next[(size_t)i * (g.D + g.A) + d] = g.xa[(size_t)(g.B + i) * g.D + d];
        for (int a = 0; a < g.A; ++a) g.xq_next[(size_t)i * (g.D + g.A) + g.D + a] = a_[a];
        g.lp_pi[i] = squashed_sample_logp(g.mu + (size_t)i * g.A, ls, g.np + (size_t)i * g.A, g.A, a_, gg);
        for (int d = 0; d < g.D; ++d) g.xq_pi[(size_t)i * (g.D + g.A) + d] = g.xa[(size_t)i * g.D + d];
        for (int a = 0; a < g.A; ++a) { g.xq_pi[(size_t)i * (g.D + g.A) + g.D + a] = a_[a]; g.a_pi[i * g.A + a] = a_[a]; g.g_pi[i * g.A + a] = gg[a]; }
    }
    const double tot = block_sum(s, sh);
    if (threadIdx.x == 0) {
        float le = g.sc->log_ent;
        if (g.auto_ent) {
            const float cc = (float)(tot / g.B);
            g.stats[2] = -(le * cc);                                                          // loss = -(log_ent_coef * c), :330
            const float gr = -cc;
            const float m = g.b1 * g.sc->ent_m + (1.0f - g.b1) * gr, v = g.b2 * g.sc->ent_v + (1.0f - g.b2) * gr * gr;
            g.sc->ent_m = m; g.sc->ent_v = v;
            le -= m / (1.0f - g.bt1) / (sqrtf(v / (1.0f - g.bt2)) + g.eps) * g.lr;           // Optimisers.Adam
            g.sc->log_ent = le;
        }
        g.sc->alpha = expf(le);
        g.stats[4] = g.sc->alpha;                                                             // :391
    }
}

// ---- Bellman target + critic loss head (sac_critic_loss :136-150) ---------------------------------------------------------
struct CriticHeadArgs {
    int B; const float* q_next;  // [2][B] target-network values of (next obs, next action)
    const float* q;              // [2][B] current values of (obs, action)
    const float *rew, *nlp; const uint8_t* term; const SacScalars* sc; float gamma;
    float* dq;                   // [2][B] dL/dq
    float* stats;
};
__global__ __launch_bounds__(256) void sac_critic_head_kernel(CriticHeadArgs g) {
    __shared__ double sh[256];
    const float alpha = g.sc->alpha;
    double cl = 0, qs = 0;
    for (int i = threadIdx.x; i < g.B; i += 256) {
        const float n0 = g.q_next[i], n1 = g.q_next[g.B + i], mn = n0 < n1 ? n0 : n1;
        const float y = g.term[i] ? g.rew[i] : g.rew[i] + g.gamma * (mn - alpha * g.nlp[i]);
        for (int k = 0; k < 2; ++k) {
            const float qv = g.q[k * g.B + i], d = qv - y;
            cl += 0.5 * (double)d * d / g.B; qs += qv;
            g.dq[k * g.B + i] = d / (float)g.B;
        }
    }
    cl = block_sum(cl, sh); qs = block_sum(qs, sh);
    if (threadIdx.x == 0) { g.stats[1] = (float)cl; g.stats[3] = (float)(qs / (2.0 * g.B)); }
}

// actor loss head (:102-104): min over the critics, dL/dq
struct ActorHeadArgs { int B; const float* q_pi; const float* lp_pi; const SacScalars* sc; float* dq; float* stats; };
__global__ __launch_bounds__(256) void sac_actor_head_kernel(ActorHeadArgs g) {
    __shared__ double sh[256];
    const float alpha = g.sc->alpha;
    double al = 0;
    for (int i = threadIdx.x; i < g.B; i += 256) {
        const float q0 = g.q_pi[i], q1 = g.q_pi[g.B + i];
        const int km = q1 < q0 ? 1 : 0;
        al += ((double)alpha * g.lp_pi[i] - (km ? q1 : q0)) / g.B;
        g.dq[km * g.B + i] = -1.0f / (float)g.B; g.dq[(1 - km) * g.B + i] = 0.f;
    }
    al = block_sum(al, sh);
    if (threadIdx.x == 0) g.stats[0] = (float)al;
}
// reverse of the squashed sample (Zygote through tanh -> clamp -> atanh -> logpdf): dL/dmu per sample, dL/dlog_std summed
struct SquashBwdArgs {
    int B, D, A; const float* mu; const float* log_std; const float* np; const float *a_pi, *g_pi;
    const float* dxq;   // [2][B][D+A] input gradients of the two critics
    const SacScalars* sc; float* dmu; float* g_log_std;
};
__global__ __launch_bounds__(256) void sac_squash_bwd_kernel(SquashBwdArgs g) {
    __shared__ double sh[256];
    const float alpha = g.sc->alpha, eps = 1.0e-6f, lo = -1.0f + eps, hi = 1.0f - eps;
    double dls[kMaxA]; for (int a = 0; a < g.A; ++a) dls[a] = 0;
    const int W = g.D + g.A;
    for (int i = threadIdx.x; i < g.B; i += 256) {
        const float dlogp = alpha / (float)g.B;
        for (int a = 0; a < g.A; ++a) {
            const float da = g.dxq[(size_t)i * W + g.D + a] + g.dxq[((size_t)g.B + i) * W + g.D + a];
            const float ls = g.log_std[a], sig = expf(ls), e2 = expf(-2.0f * ls);
            const float x = g.a_pi[i * g.A + a], gg = g.g_pi[i * g.A + a], mu = g.mu[(size_t)i * g.A + a], d = gg - mu;
            const float inside = (x >= lo && x <= hi) ? 1.0f : 0.0f;
            const float dlp_dg = -d * e2 + 2.0f * tanhf(gg);
            const float du = dlogp * dlp_dg * inside + da * (1.0f - x * x);
            g.dmu[(size_t)i * g.A + a] = dlogp * (d * e2) + du;
            dls[a] += (double)(dlogp * (-1.0f + d * d * e2) + du * sig * g.np[i * g.A + a]);
        }
    }
    for (int a = 0; a < g.A; ++a) { const double t = block_sum(dls[a], sh); if (threadIdx.x == 0) g.g_log_std[a] = (float)t; }
}

// =================================================================================================================
// Fused heads.  A net's OUTPUT layer has one (Q nets) or A (actor) rows: as a contraction it is one launch of ~4.5 us for a few hundred dot
// products, and so is the first stage of the reverse pass (K = 1).  The kernels below do the output-layer dot products, the per-sample loss head and
// the first reverse stage dz2 = (W3' dOut) .* act'(h2) in ONE launch, one wave per sample (lanes stride the hidden units, fixed-order butterfly sum =>
// deterministic), per-block partial sums of the batch statistics folded in index order by the block that finishes last.  25 -> 19 dependent
// launches per update!.
// =================================================================================================================
constexpr int kHeadSamplesPerBlock = 1;   // one workgroup per sample: the dot products of a sample run on separate waves, so a head is ~3 dependent memory round trips
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
// dot of two contiguous, 16-byte-aligned rows of H floats (H % 4 == 0): every lane's float4 loads are issued before the first use
__device__ __forceinline__ float wave_dot(const float* __restrict__ w, const float* __restrict__ x, int H, int lane) {
    float s = 0.f;
    for (int u = lane; u < H / 4; u += 64) {
        const float4 a = reinterpret_cast<const float4*>(w)[u], b = reinterpret_cast<const float4*>(x)[u];
        s = fmaf(a.x, b.x, s); s = fmaf(a.y, b.y, s); s = fmaf(a.z, b.z, s); s = fmaf(a.w, b.w, s);
    }
    return wave_sum(s);
}
// dot of a strided weight row (element u at w[u * stride]) with a contiguous activation row
__device__ __forceinline__ float wave_dot_strided(const float* __restrict__ w, int stride, const float* __restrict__ x, int H, int lane) {
    float s = 0.f;
    for (int u = lane; u < H; u += 64) s = fmaf(w[(size_t)u * stride], x[u], s);
    return wave_sum(s);
}
__device__ __forceinline__ float act_deriv(float hval, int relu) { return relu ? (hval > 0.f ? 1.0f : 0.0f) : 1.0f - hval * hval; }
// last-block fold of per-block values (double, index order): `mine` is this block's value (valid in thread 0); returns true in the block that finished
// last, with the totals in tot[0..NV) (all threads)
template <int NV>
__device__ __forceinline__ bool fold_partials(const double (&mine)[NV], double* partials, unsigned int* counter, double (&tot)[NV], double* sh) {
    __shared__ bool last;
    const int t = threadIdx.x;
    if (t == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) partials[(size_t)blockIdx.x * NV + k] = mine[k];
        __threadfence(); last = atomicAdd(counter, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return false;
    __threadfence();
#pragma unroll
    for (int k = 0; k < NV; ++k) {                                             // all 256 threads fetch (one round trip), fixed-order tree: deterministic
        double a = 0;
        for (unsigned b = t; b < gridDim.x; b += 256) a += ((volatile const double*)partials)[(size_t)b * NV + k];
        tot[k] = block_sum(a, sh);
    }
    if (t == 0) *counter = 0u;
    return true;
}

// actor output layer on (obs | next obs) + entropy-coefficient step + next actions + the actor-loss sample: net_forward's third launch + sac_ent_next_kernel
struct ActorHeadFusedArgs {
    EntNextArgs e; const float* ah2; int H2; const float* W3; const float* b3;   // W3 (A x H2) column-major: row a strided by A
    float* mu_out; double* partials; unsigned int* counter;
    // first layer of the two TARGET critics on (next obs, next action), when the input is narrow (first_l1 != 0): z = 2, 3 of the four-net forward
    int first_l1, H1, relu; const float* P; int qw1, qb1; long long zP, zh; float* qh1;
};
// PRE (narrow spaces: D + A <= 4, H1 <= 512 when first_l1): everything the later phases read from memory and that does not depend on this kernel's results is
// requested at the top — the noise of the three samples and the two observation rows (lane 0 of waves 0 - 2), the first-layer weights of the two target critics
// (all threads) — and the (next obs, next action) row reaches the first layers through LDS: the kernel was four dependent round trips long, now two.
template <bool PRE>
__global__ __launch_bounds__(256) void sac_actor_out_ent_kernel(ActorHeadFusedArgs f) {
    __shared__ double sh[256];
    __shared__ float mus[2][kMaxA];
    __shared__ double ssum;
    __shared__ float xn[kFusedL1MaxIn];
    const EntNextArgs& g = f.e;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = blockIdx.x;
    float ls[kMaxA]; for (int a = 0; a < g.A; ++a) ls[a] = g.log_std[a];
    L1Pre pq[2];
    float nz[kL1PreMaxIn], xrow[kL1PreMaxIn];
    if constexpr (PRE) {
        if (f.first_l1) for (int z = 0; z < 2; ++z) first_layer_pre(pq[z], f.P + f.qw1 + (z + 2) * f.zP, f.P + f.qb1 + (z + 2) * f.zP, g.D + g.A, f.H1);
        if (lane == 0 && wave < 3) {
            const float* src = wave == 0 ? g.ne : wave == 1 ? g.nn : g.np;
            const float* xr = g.xa + (size_t)((wave == 1 ? g.B : 0) + i) * g.D;
#pragma unroll
            for (int a = 0; a < kL1PreMaxIn; ++a) { nz[a] = a < g.A ? src[(size_t)i * g.A + a] : 0.f; xrow[a] = a < g.D ? xr[a] : 0.f; }
        }
    }
    // phase 1: mu = W3 h2 + b3 for row i (obs) and row B + i (next obs): (row, a) pairs over the four waves
    for (int p = wave; p < 2 * g.A; p += 4) {
        const int row = p / g.A, a = p - row * g.A;
        const float d = wave_dot_strided(f.W3 + a, g.A, f.ah2 + (size_t)(row * g.B + i) * f.H2, f.H2, lane) + f.b3[a];
        if (lane == 0) { mus[row][a] = d; f.mu_out[(size_t)(row * g.B + i) * g.A + a] = d; }
    }
    if (threadIdx.x == 0) ssum = 0.0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && threadIdx.x < 192) {                          // phase 2: the three squashed samples of the row on lane 0 of three WAVES (libm-heavy scalar math: on three
        const int which = threadIdx.x >> 6;                                       // lanes of one wave the divergent branches ran one after the other — 3 x ~1.5 us of an 11 us kernel)
        float a_[kMaxA], gg[kMaxA];
        const float* noise = which == 0 ? g.ne + (size_t)i * g.A : which == 1 ? g.nn + (size_t)i * g.A : g.np + (size_t)i * g.A;
        const float* xr = g.xa + (size_t)((which == 1 ? g.B : 0) + i) * g.D;
        float lp;
        if constexpr (PRE) lp = squashed_sample_logp(mus[which == 1 ? 1 : 0], ls, nz, g.A, a_, gg);
        else lp = (which || g.auto_ent) ? squashed_sample_logp(mus[which == 1 ? 1 : 0], ls, noise, g.A, a_, gg) : 0.f;
        auto xv = [&](int d) { if constexpr (PRE) { float v = xrow[0]; for (int t = 1; t < kL1PreMaxIn; ++t) v = d == t ? xrow[t] : v; return v; } else return xr[d]; };
        if (which == 0) { if (g.auto_ent) ssum = (double)(lp + g.target_entropy); }
        else if (which == 1) {
            g.nlp[i] = lp;
            const bool to_lds = f.first_l1 != 0;                                 // (first_l1 implies D + A <= kFusedL1MaxIn)
            for (int d = 0; d < g.D; ++d) { const float v = xv(d); g.xq_next[(size_t)i * (g.D + g.A) + d] = v; if (to_lds) xn[d] = v; }
            for (int a = 0; a < g.A; ++a) { g.xq_next[(size_t)i * (g.D + g.A) + g.D + a] = a_[a]; if (to_lds) xn[g.D + a] = a_[a]; }
        } else {
            g.lp_pi[i] = lp;
            for (int d = 0; d < g.D; ++d) g.xq_pi[(size_t)i * (g.D + g.A) + d] = xv(d);
            for (int a = 0; a < g.A; ++a) { g.xq_pi[(size_t)i * (g.D + g.A) + g.D + a] = a_[a]; g.a_pi[i * g.A + a] = a_[a]; g.g_pi[i * g.A + a] = gg[a]; }
        }
    }
    __syncthreads();
    if (f.first_l1) {                                                             // lane 0 of wave 1 left this sample's (next obs, next action) row in xn
#pragma unroll
        for (int z = 2; z < 4; ++z) {
            if constexpr (PRE) first_layer_post(pq[z - 2], xn, g.D + g.A, f.H1, f.relu, f.qh1 + z * f.zh + (size_t)i * f.H1);
            else first_layer_row(f.P + f.qw1 + z * f.zP, f.P + f.qb1 + z * f.zP, xn, g.D + g.A, f.H1, f.relu, f.qh1 + z * f.zh + (size_t)i * f.H1);
        }
    }
    // the entropy-coefficient step itself (mean over the batch, scalar Adam) runs at the head of the next kernel that needs alpha (sac_q_out_head_kernel, mode 0):
    // every one of its blocks sums these B per-sample terms in the same order — cheaper than a grid-wide fold here (an atomic ticket, a fence and a second phase: ~7 us)
    if (threadIdx.x == 0) f.partials[i] = ssum;
}

// Q output layers (Z nets: the two critics, and with Z = 4 the two targets behind them) + loss head + dz2 of the two critics
struct QHeadFusedArgs {
    int B, H2, nq, relu, Z; long long zP, zh2;     // net z: W3 / b3 at P + w3 + z * zP, h2 at qh2 + z * zh2
    const float* P; int w3, b3; const float* qh2;
    float* q_out;          // [Z][nq]
    float* dz2;            // [2][nq][H2]
    // critic head (mode 0) / actor head (mode 1)
    int mode; const float *rew, *nlp, *lp_pi; const uint8_t* term; const SacScalars* sc; float gamma;
    float* dq; float* stats; double* partials; unsigned int* counter;
    // mode 0 also takes the entropy-coefficient step (sac.jl:313-343): `sc` is the state before it, `sc_next` receives the state after it (block 0), every block uses it in registers;
    // mode 1 and the later kernels are handed sc_next as their `sc`
    const double* ent_terms; SacScalars* sc_next; int auto_ent; float ent_lr, ent_b1, ent_b2, ent_eps, ent_bt1, ent_bt2;
};
__global__ __launch_bounds__(256) void sac_q_out_head_kernel(QHeadFusedArgs g) {
    __shared__ double sh[256];
    __shared__ float qs[4], dqs[2];
    __shared__ double part[2];
    __shared__ float alpha_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = blockIdx.x;
    if (g.mode == 0) {
        double t = 0;
        if (g.auto_ent) { for (int k = threadIdx.x; k < g.B; k += 256) t += g.ent_terms[k]; t = block_sum(t, sh); }   // fixed order: the same bits in every block
        if (threadIdx.x == 0) {
            float le = g.sc->log_ent, m = g.sc->ent_m, v = g.sc->ent_v, loss = 0.f;
            if (g.auto_ent) {
                const float cc = (float)(t / g.B);
                loss = -(le * cc);                                                                // loss = -(log_ent_coef * c), sac.jl:330
                const float gr = -cc;
                m = g.ent_b1 * m + (1.0f - g.ent_b1) * gr; v = g.ent_b2 * v + (1.0f - g.ent_b2) * gr * gr;
                le -= m / (1.0f - g.ent_bt1) / (sqrtf(v / (1.0f - g.ent_bt2)) + g.ent_eps) * g.ent_lr;   // Optimisers.Adam
            }
            alpha_s = expf(le);
            if (blockIdx.x == 0) { g.sc_next->log_ent = le; g.sc_next->ent_m = m; g.sc_next->ent_v = v; g.sc_next->alpha = alpha_s; if (g.auto_ent) g.stats[2] = loss; g.stats[4] = alpha_s; }   // :391
        }
    } else if (threadIdx.x == 0) alpha_s = g.sc->alpha;
    if (wave < g.Z) {                                                          // one wave per net: q_z = W3_z . h2_z + b3_z
        const float q = wave_dot(g.P + g.w3 + wave * g.zP, g.qh2 + wave * g.zh2 + (size_t)i * g.H2, g.H2, lane) + g.P[g.b3 + wave * g.zP];
        if (lane == 0) { qs[wave] = q; g.q_out[(size_t)wave * g.nq + i] = q; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float alpha = alpha_s;
        float dq0, dq1; double s0 = 0, s1 = 0;
        if (g.mode == 0) {                                                    // Bellman target + critic loss (sac_critic_loss :136-150)
            const float mn = qs[2] < qs[3] ? qs[2] : qs[3];
            const float y = g.term[i] ? g.rew[i] : g.rew[i] + g.gamma * (mn - alpha * g.nlp[i]);
            const float d0 = qs[0] - y, d1 = qs[1] - y;
            dq0 = d0 / (float)g.B; dq1 = d1 / (float)g.B;
            s0 = 0.5 * (double)d0 * d0 / g.B + 0.5 * (double)d1 * d1 / g.B; s1 = (double)qs[0] + (double)qs[1];
        } else {                                                              // actor loss (:102-104): min over the critics
            const int km = qs[1] < qs[0] ? 1 : 0;
            dq0 = km == 0 ? -1.0f / (float)g.B : 0.f; dq1 = km == 1 ? -1.0f / (float)g.B : 0.f;
            s0 = ((double)alpha * g.lp_pi[i] - (km ? qs[1] : qs[0])) / g.B;
        }
        dqs[0] = dq0; dqs[1] = dq1; g.dq[i] = dq0; g.dq[g.nq + i] = dq1; part[0] = s0; part[1] = s1;
    }
    __syncthreads();
    for (int u = threadIdx.x; u < 2 * g.H2; u += 256) {                        // dz2 = (W3' dq) .* act'(h2): the first stage of the reverse pass (K = 1)
        const int k = u >= g.H2, j = u - k * g.H2;
        g.dz2[((size_t)k * g.nq + i) * g.H2 + j] = g.P[g.w3 + k * g.zP + j] * dqs[k] * act_deriv(g.qh2[k * g.zh2 + (size_t)i * g.H2 + j], g.relu);
    }
    // the sums feed statistics only: per-block values, folded by sac_step_end_kernel (no atomics, no second phase here)
    if (threadIdx.x == 0) { g.partials[(size_t)blockIdx.x * 2] = part[0]; g.partials[(size_t)blockIdx.x * 2 + 1] = part[1]; }
}

// action columns of dx = W1' dz1 of both critics + reverse of the squashed sample + dz2 of the ACTOR: net_backward's last launch (dX), sac_squash_bwd_kernel
// and the first stage of the actor's reverse pass
struct SquashFusedArgs {
    SquashBwdArgs s; int H1, H2, relu, nq; const float* P; int qw1; long long zP;   // critic k: W1 (H1 x (D+A)) column-major at P + qw1 + k * zP
    const float* dz1;      // [2][nq][H1]
    const float* aW3; const float* ah2; float* adz2;   // actor: W3 (A x H2), h2 [.][H2], dz2 out [nq][H2]
    double* partials; unsigned int* counter;
};
__global__ __launch_bounds__(256) void sac_dx_squash_kernel(SquashFusedArgs f) {
    __shared__ double sh[256];
    __shared__ float das[2][kMaxA], dmus[kMaxA];
    __shared__ double dlss[kMaxA];
    const SquashBwdArgs& g = f.s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = blockIdx.x;
    const float alpha = g.sc->alpha, eps = 1.0e-6f, lo = -1.0f + eps, hi = 1.0f - eps;
    for (int p = wave; p < 2 * g.A; p += 4) {                                  // d Q_k / d action_a = W1_k[:, D + a] . dz1_k: (critic, a) pairs over the four waves
        const int kk = p / g.A, a = p - kk * g.A;
        const float d = wave_dot(f.P + f.qw1 + kk * f.zP + (size_t)(g.D + a) * f.H1, f.dz1 + ((size_t)kk * f.nq + i) * f.H1, f.H1, lane);
        if (lane == 0) das[kk][a] = d;
    }
    __syncthreads();
    if (threadIdx.x < g.A) {
        const int a = threadIdx.x;
        const float dlogp = alpha / (float)g.B, da = das[0][a] + das[1][a];
        const float ls = g.log_std[a], sig = expf(ls), e2 = expf(-2.0f * ls);
        const float x = g.a_pi[i * g.A + a], gg = g.g_pi[i * g.A + a], mu = g.mu[(size_t)i * g.A + a], d = gg - mu;
        const float inside = (x >= lo && x <= hi) ? 1.0f : 0.0f;
        const float dlp_dg = -d * e2 + 2.0f * tanhf(gg);
        const float du = dlogp * dlp_dg * inside + da * (1.0f - x * x);
        const float dm = dlogp * (d * e2) + du;
        dmus[a] = dm; g.dmu[(size_t)i * g.A + a] = dm;
        dlss[a] = (double)(dlogp * (-1.0f + d * d * e2) + du * sig * g.np[i * g.A + a]);
    }
    __syncthreads();
    for (int u = threadIdx.x; u < f.H2; u += 256) {                            // actor dz2 = (W3' dmu) .* act'(h2)
        float t = 0.f;
        for (int a = 0; a < g.A; ++a) t = fmaf(f.aW3[a + (size_t)u * g.A], dmus[a], t);
        f.adz2[(size_t)i * f.H2 + u] = t * act_deriv(f.ah2[(size_t)i * f.H2 + u], f.relu);
    }
    if (threadIdx.x < g.A) f.partials[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = dlss[threadIdx.x];   // log_std gradient: per-sample terms, summed by sac_step_end_kernel right before its Adam step
}

// ---- Optimisers.Adam on a parameter range; grads == nullptr applies ZERO gradients (zero_critic_grads! then apply_gradients,
// sac.jl:381-382: the moments decay and the parameters keep moving along the remaining momentum) ---------------------------
__device__ __forceinline__ float adam_one(float* p, float* m, float* v, int i, float gi, float lr, float b1, float b2, float eps, float bt1, float bt2) {
    const float mm = b1 * m[i] + (1.0f - b1) * gi, vv = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mm; v[i] = vv;
    const float np_ = p[i] - mm / (1.0f - bt1) / (sqrtf(vv / (1.0f - bt2)) + eps) * lr;
    p[i] = np_; return np_;
}
__device__ __forceinline__ float4 adam_vec(float* p, float* m, float* v, int i, float4 g, float lr, float b1, float b2, float eps, float bt1, float bt2) {
    float4 pp = *reinterpret_cast<float4*>(p + i), mm = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
    float* P = reinterpret_cast<float*>(&pp); float* M = reinterpret_cast<float*>(&mm); float* V = reinterpret_cast<float*>(&vv); const float* G = reinterpret_cast<const float*>(&g);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        M[t] = b1 * M[t] + (1.0f - b1) * G[t]; V[t] = b2 * V[t] + (1.0f - b2) * G[t] * G[t];
        P[t] -= M[t] / (1.0f - bt1) / (sqrtf(V[t] / (1.0f - bt2)) + eps) * lr;
    }
    *reinterpret_cast<float4*>(p + i) = pp; *reinterpret_cast<float4*>(m + i) = mm; *reinterpret_cast<float4*>(v + i) = vv;
    return pp;
}
// ---- the first layer's parameter gradients INSIDE the optimiser kernels ------------------------------------------------------
// [dW1 | db1] = dz1 . [x' | 1] of a narrow input (Pendulum: 3 / 4 features) is a (H1 x 5) x B contraction: 0.7 MFLOP in a launch of its own (7 us: the floor
// of a dependent launch), followed by the optimiser kernel that consumes it — and, on the critic side, by another small launch that re-evaluates the first layer
// with the stepped parameters for the actor loss.  Here extra blocks at the END of the optimiser kernel's grid own the first layer: a block takes kOptL1Units hidden
// units of one net; thread = (unit, one of 16 sample groups): per-thread sums over its samples (all loads in flight), the 16 groups folded through LDS in
// index order (deterministic), 5 x 16 threads then write the gradient (dril_sac_get_last_grads), take the Adam step and — x2 given — all threads evaluate the
// layer on x2 with the stepped parameters in the order of first_layer_row.  The elementwise loop of the kernel skips these parameters.
constexpr int kOptL1MaxIn = 4, kOptL1Units = 16, kOptL1Groups = 16, kOptL1MaxB = 1024;      // (2 B in floats of dynamic LDS: 32 KB at the cap)
struct FirstLayerOpt {
    int nblocks;                      // 0: off.  Z * ceil(H1 / kOptL1Units) blocks behind the elementwise ones
    int Z, in, H1, B, relu;
    int w1, b1; long long zP;         // W1 (H1 x in, column-major) / b1 of net z at index w1 / b1 + z zP of the kernel's parameter, moment and gradient arrays
    const float* dz1; long long zdz;  // [Z][B][H1]
    const float* x; int ldx;          // [B][ldx] rows (the same input for every z)
    const float* x2; float* h1; long long zh;   // x2 != null: h1[z][b][:] = act(W1 x2[b] + b1) with the stepped parameters; x2 [B][in]
};
__device__ __forceinline__ double first_layer_opt_block(const FirstLayerOpt& f, int blk, float* p, float* m, float* v, float* g,
                                                        float lr, float b1c, float b2c, float eps, float bt1, float bt2) {
    // ONE round of memory latency: the block's dz1 values (16 per thread at B = 256), both input matrices (into LDS) and the parameters / moments it will step are
    // all requested before anything waits
    extern __shared__ __attribute__((aligned(16))) float fl_x[];      // [B][4] x, then [B][4] x2 (launch: first_layer_opt_lds bytes)
    __shared__ float part[kOptL1Groups][kOptL1MaxIn + 1][kOptL1Units];
    __shared__ float wnew[kOptL1MaxIn + 1][kOptL1Units];
    const int nbu = (f.H1 + kOptL1Units - 1) / kOptL1Units, z = blk / nbu, u = threadIdx.x & (kOptL1Units - 1), grp = threadIdx.x / kOptL1Units;
    const int unit = (blk - z * nbu) * kOptL1Units + u;
    const bool live = unit < f.H1;
    const float* __restrict__ dz = f.dz1 + (size_t)z * f.zdz + (live ? unit : 0);
    float4* xs = reinterpret_cast<float4*>(fl_x); float4* x2s = xs + f.B;      // rows padded to four features (zeros): one ds_read_b128 per sample, no branches on `in`
    const int kk = threadIdx.x / kOptL1Units;                         // the parameter row this thread steps (threads < 5 x 16): k < in a column of W1, k == 4 the bias
    const bool steps = threadIdx.x < (kOptL1MaxIn + 1) * kOptL1Units && live && (kk < f.in || kk == kOptL1MaxIn);
    const int idx = (int)(z * f.zP) + (kk == kOptL1MaxIn ? f.b1 + unit : f.w1 + unit + kk * f.H1);
    float p0 = 0.f, m0 = 0.f, v0 = 0.f;
    if (steps) { p0 = p[idx]; m0 = m[idx]; v0 = v[idx]; }
    float acc[kOptL1MaxIn + 1];
#pragma unroll
    for (int k = 0; k <= kOptL1MaxIn; ++k) acc[k] = 0.f;
    for (int b0 = 0; b0 < f.B; b0 += kOptL1Groups * 16) {
        float d[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) { const int b = b0 + grp + kOptL1Groups * j; d[j] = dz[(size_t)(b < f.B ? b : f.B - 1) * f.H1]; }
        if (b0 == 0) {
            for (int i = threadIdx.x; i < f.B * kOptL1MaxIn; i += blockDim.x) {
                const int b = i / kOptL1MaxIn, k = i % kOptL1MaxIn;
                fl_x[i] = k < f.in ? f.x[(size_t)b * f.ldx + k] : 0.f;
                if (f.x2) fl_x[(size_t)f.B * kOptL1MaxIn + i] = k < f.in ? f.x2[(size_t)b * f.in + k] : 0.f;
            }
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int b = b0 + grp + kOptL1Groups * j;
            const float dd = b < f.B ? d[j] : 0.f;                    // (rows past B re-read the last one and count as zero)
            const float4 xv = xs[b < f.B ? b : f.B - 1];
            acc[0] = fmaf(dd, xv.x, acc[0]); acc[1] = fmaf(dd, xv.y, acc[1]); acc[2] = fmaf(dd, xv.z, acc[2]); acc[3] = fmaf(dd, xv.w, acc[3]);
            acc[kOptL1MaxIn] += dd;
        }
    }
#pragma unroll
    for (int k = 0; k <= kOptL1MaxIn; ++k) part[grp][k][u] = acc[k];
    __syncthreads();
    double ss = 0;
    if (threadIdx.x < (kOptL1MaxIn + 1) * kOptL1Units) {
        float gs = 0.f, np_ = 0.f;
#pragma unroll
        for (int q = 0; q < kOptL1Groups; ++q) gs += part[q][kk][u];
        if (steps) {                                                   // adam_one on the values requested at the top
            if (g) g[idx] = gs;
            const float mm = b1c * m0 + (1.0f - b1c) * gs, vv = b2c * v0 + (1.0f - b2c) * gs * gs;
            m[idx] = mm; v[idx] = vv;
            np_ = p0 - mm / (1.0f - bt1) / (sqrtf(vv / (1.0f - bt2)) + eps) * lr;
            p[idx] = np_;
            ss = (double)gs * gs;
        }
        wnew[kk][u] = np_;                                             // (rows k >= in: zero weights against the zero padding of x2)
    }
    if (!f.x2) return ss;
    __syncthreads();
    const float w0 = wnew[0][u], w1 = wnew[1][u], w2 = wnew[2][u], w3 = wnew[3][u], bb = wnew[kOptL1MaxIn][u];
    float* __restrict__ out = f.h1 + (size_t)z * f.zh + unit;
#pragma unroll 4
    for (int b = grp; b < f.B; b += kOptL1Groups) {
        const float4 xv = x2s[b];
        float a = fmaf(w0, xv.x, bb); a = fmaf(w1, xv.y, a); a = fmaf(w2, xv.z, a); a = fmaf(w3, xv.w, a);      // the order of first_layer_row (+ exact zero terms)
        if (live) out[(size_t)b * f.H1] = f.relu ? relu_nan(a) : tanhf(a);
    }
    return ss;
}
inline size_t first_layer_opt_lds(const FirstLayerOpt& f) { return f.nblocks ? (size_t)2 * f.B * kOptL1MaxIn * sizeof(float) : 0; }
// is element i one of the first-layer parameters those blocks own (net_off: b1 follows W1, so a net's run is (in + 1) H1 long — a multiple of 4 when H1 % 4 == 0)
__device__ __forceinline__ bool first_layer_opt_owns(const FirstLayerOpt& f, int i) {
    if (!f.nblocks) return false;
    for (int z = 0; z < f.Z; ++z) { const int o = i - (f.w1 + (int)(z * f.zP)); if (o >= 0 && o < (f.in + 1) * f.H1) return true; }
    return false;
}
constexpr int kSacAdamBlocks = DRIL_SAC_ADAM_BLOCKS;   // grid cap of the two elementwise optimiser kernels (measured: see the Makefile-free default below)
struct AdamRangeArgs { float* p; float* m; float* v; float* g; int n; float lr, b1, b2, eps, bt1, bt2; double* sumsq_partials; FirstLayerOpt fl; };
__global__ __launch_bounds__(256) void sac_adam_kernel(AdamRangeArgs a) {
    __shared__ double sh[256];
    double ss = 0;
    const int nfl = a.fl.nblocks, nb = gridDim.x - nfl, eb = (int)blockIdx.x - nfl;   // the first nfl blocks own the first layers (the longest dependent chain of the launch: dispatched first), the rest are elementwise
    const bool vec = (a.n & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.p) | reinterpret_cast<uintptr_t>(a.m) | reinterpret_cast<uintptr_t>(a.v) | reinterpret_cast<uintptr_t>(a.g)) & 15) == 0;
    if (eb < 0) ss = first_layer_opt_block(a.fl, blockIdx.x, a.p, a.m, a.v, a.g, a.lr, a.b1, a.b2, a.eps, a.bt1, a.bt2);
    else if (vec) {                                                                 // the padded device layout: 16-byte rows
        for (int i = eb * 256 + threadIdx.x; i < a.n / 4; i += nb * 256) {
            if (first_layer_opt_owns(a.fl, 4 * i)) continue;
            const float4 g = a.g ? *reinterpret_cast<const float4*>(a.g + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
            adam_vec(a.p, a.m, a.v, 4 * i, g, a.lr, a.b1, a.b2, a.eps, a.bt1, a.bt2);
            ss += (double)g.x * g.x + (double)g.y * g.y + (double)g.z * g.z + (double)g.w * g.w;
        }
    } else {
        for (int i = eb * 256 + threadIdx.x; i < a.n; i += nb * 256) {
            if (first_layer_opt_owns(a.fl, i)) continue;
            const float gi = a.g ? a.g[i] : 0.f;
            const float m = a.b1 * a.m[i] + (1.0f - a.b1) * gi, v = a.b2 * a.v[i] + (1.0f - a.b2) * gi * gi;
            a.m[i] = m; a.v[i] = v;
            a.p[i] -= m / (1.0f - a.bt1) / (sqrtf(v / (1.0f - a.bt2)) + a.eps) * a.lr;
            ss += (double)gi * gi;
        }
    }
    if (a.sumsq_partials) { ss = block_sum(ss, sh); if (threadIdx.x == 0) a.sumsq_partials[blockIdx.x] = ss; }
}
// End of one update!: apply_gradients(train_state, actor_loss_grad) (sac.jl:382 — actor_head and log_std with their gradients, the
// critic leaves with the ZERO arrays zero_critic_grads! left, :381), polyak_update! of the targets (:385-389, optimization_utils.jl:3-6)
// and the step's statistics, in ONE launch.  The block that finishes last sums the per-block partials in index order (deterministic).
struct StepEndArgs {
    float *p, *m, *v; const float* g_actor; int n_actor, ls_off, n_ls, q_off, n_q;
    float lr, b1, b2, eps, bt1_a, bt2_a, bt1_c, bt2_c;
    float* target; float tau; int do_polyak;
    double* ssq_a; float* stats; float* out;   // ssq_a: this update's row of squared-gradient partials, [end blocks] (the critic's partials sit in front of it)
    // deferred sums of the fused head kernels (nhead = 0: the unfused sequence wrote stats / the log_std gradient itself): per-sample rows [nhead][2] of the critic and
    // actor loss heads, [n_ls][nhead] of the log_std gradient
    int nhead, B; const double *hp_critic, *hp_actor, *hp_ls; float* g_ls;
    unsigned long long* stamp;   // phase_stamp (null: none)
    FirstLayerOpt fl;            // the actor's first layer: gradient + step in blocks of their own (nblocks = 0: its gradient is in g_actor like the rest)
    float* g_actor_w;            // (writable alias of g_actor for those blocks)
};
// n_actor and n_q are multiples of 4 (the device layout pads every net to 16 bytes; pads hold zero parameters and zero gradients)
__global__ __launch_bounds__(256) void sac_step_end_kernel(StepEndArgs a) {
    __shared__ double sh[256];
    const int va = a.n_actor / 4, vq = a.n_q / 4, nb = gridDim.x - a.fl.nblocks;
    double ss = 0;
    if ((int)blockIdx.x >= nb) ss = first_layer_opt_block(a.fl, blockIdx.x - nb, a.p, a.m, a.v, a.g_actor_w, a.lr, a.b1, a.b2, a.eps, a.bt1_a, a.bt2_a);
    else for (int i = blockIdx.x * 256 + threadIdx.x; i < va + vq; i += nb * 256) {
        if (i < va) {
            if (first_layer_opt_owns(a.fl, 4 * i)) continue;
            const float4 g = *reinterpret_cast<const float4*>(a.g_actor + 4 * i);
            adam_vec(a.p, a.m, a.v, 4 * i, g, a.lr, a.b1, a.b2, a.eps, a.bt1_a, a.bt2_a);
            ss += (double)g.x * g.x + (double)g.y * g.y + (double)g.z * g.z + (double)g.w * g.w;
        } else {
            const int k = 4 * (i - va);
            const float4 np_ = adam_vec(a.p, a.m, a.v, a.q_off + k, make_float4(0.f, 0.f, 0.f, 0.f), a.lr, a.b1, a.b2, a.eps, a.bt1_c, a.bt2_c);
            if (a.do_polyak) {
                float4 t = *reinterpret_cast<float4*>(a.target + k);
                t.x = a.tau * np_.x + (1.0f - a.tau) * t.x; t.y = a.tau * np_.y + (1.0f - a.tau) * t.y;
                t.z = a.tau * np_.z + (1.0f - a.tau) * t.z; t.w = a.tau * np_.w + (1.0f - a.tau) * t.w;
                *reinterpret_cast<float4*>(a.target + k) = t;
            }
        }
    }
    if (blockIdx.x == 0 && a.nhead) {                                                          // log_std gradient = sum of the per-sample terms (fixed order)
        for (int d = 0; d < a.n_ls; ++d) {
            double t = 0;
            for (int i = threadIdx.x; i < a.nhead; i += 256) t += a.hp_ls[(size_t)d * a.nhead + i];
            t = block_sum(t, sh);
            if (threadIdx.x == 0) a.g_ls[d] = (float)t;
            __syncthreads();
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < a.n_ls) {                                              // log_std: a handful of scalars
        const int j = a.ls_off + threadIdx.x; const float gi = a.nhead ? a.g_ls[threadIdx.x] : a.g_actor[j];
        adam_one(a.p, a.m, a.v, j, gi, a.lr, a.b1, a.b2, a.eps, a.bt1_a, a.bt2_a);
        ss += (double)gi * gi;
    }
    ss = block_sum(ss, sh);
    if (threadIdx.x == 0) a.ssq_a[blockIdx.x] = ss;          // squared-gradient partial of this block: summed on the host with the critic's (sac.jl:393) — no grid-wide fold for a statistic
    phase_stamp(a.stamp);
    if (blockIdx.x != 0) return;
    if (a.nhead) {                                                                              // statistics of the fused loss heads
        double c0 = 0, c1 = 0, p0 = 0;
        for (int i = threadIdx.x; i < a.nhead; i += 256) { c0 += a.hp_critic[2 * i]; c1 += a.hp_critic[2 * i + 1]; p0 += a.hp_actor[2 * i]; }
        c0 = block_sum(c0, sh); c1 = block_sum(c1, sh); p0 = block_sum(p0, sh);
        if (threadIdx.x == 0) { a.stats[1] = (float)c0; a.stats[3] = (float)(c1 / (2.0 * a.B)); a.stats[0] = (float)p0; }
    }
    if (threadIdx.x == 0) { a.out[0] = a.stats[0]; a.out[1] = a.stats[1]; a.out[2] = a.stats[2]; a.out[3] = a.stats[3]; a.out[4] = a.stats[4]; a.out[5] = 0.f; }
}

// ---- collection (off_policy_collection.jl:28-96) ---------------------------------------------------------------------------
struct CollectHeadArgs {
    int E, A, use_random; float* mu; const float* log_std; const float* inj_noise; const uint32_t* gstep; uint64_t seed0;
    float low, high; float* raw; float* envact;
    const float* h2; const float* w3; const float* b3; int H2;     // h2 != null: the actor's output layer mu = W3 h2 + b3 is evaluated HERE (sac_mu_rows) instead of in a launch of its own
};
// mu[e][a] = W3[a, :] . h2[e, :] + b3[a] for the kEnvsPerBlock envs of a 256-thread block (4096 envs = 256 blocks: every CU takes part in what is a latency-bound
// pass): each of the four waves takes kMuRows envs, the wave across the features — per 256-feature slice all rows' loads (one coalesced 1 KB line run each) are in
// flight together, then one butterfly sum per row (every lane ends with the total; fixed order => deterministic).  Result in mu_s[] (shared), valid after the
// caller's barrier.  Every thread of the block takes part whatever E is.
constexpr int kEnvsPerBlock = 16, kMuRows = kEnvsPerBlock / 4;
__device__ __forceinline__ void sac_mu_block(const CollectHeadArgs& g, int a, int e_block0, float* mu_s) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, e0 = e_block0 + kMuRows * wave;
    const float* __restrict__ w = g.w3 + a;                          // W3 column-major (A x H2): W3[a + k A]
    float acc[kMuRows];
#pragma unroll
    for (int r = 0; r < kMuRows; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < g.H2; k0 += 256) {                         // H2 % 4 == 0 (host-checked): a lane's four features are inside the row or all outside
        const int k = k0 + 4 * lane;
        const bool in = k < g.H2;
        const int kk = in ? k : 0;                                    // out-of-range lanes re-read the row's start and multiply by zero weights: no branch around the loads
        float wk[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { const float x = w[(size_t)(kk + t) * g.A]; wk[t] = in ? x : 0.f; }
        float4 v[kMuRows];
#pragma unroll
        for (int r = 0; r < kMuRows; ++r) {
            const int e = e0 + r < g.E ? e0 + r : g.E - 1;           // rows past E re-read the last one (in bounds, unused)
            v[r] = *reinterpret_cast<const float4*>(g.h2 + (size_t)e * g.H2 + kk);
        }
#pragma unroll
        for (int r = 0; r < kMuRows; ++r) { acc[r] = fmaf(v[r].x, wk[0], acc[r]); acc[r] = fmaf(v[r].y, wk[1], acc[r]); acc[r] = fmaf(v[r].z, wk[2], acc[r]); acc[r] = fmaf(v[r].w, wk[3], acc[r]); }
    }
    const float b = g.b3[a];
#pragma unroll
    for (int r = 0; r < kMuRows; ++r) {
        float s2 = acc[r];
#pragma unroll
        for (int o = 32; o; o >>= 1) s2 += __shfl_xor(s2, o);
        if (lane == r) mu_s[kMuRows * wave + r] = s2 + b;
    }
}
// 256 threads per kEnvsPerBlock envs: the output layer by all four waves (g.h2 set), then one thread per env
__global__ __launch_bounds__(256) void sac_collect_head_kernel(CollectHeadArgs g) {
    __shared__ float mu_s[kEnvsPerBlock];
    const int e = blockIdx.x * kEnvsPerBlock + threadIdx.x;
    if (g.h2 && !g.use_random) {                                     // the output layer first (whole block: no early return before it)
        for (int a = 0; a < g.A; ++a) {
            sac_mu_block(g, a, blockIdx.x * kEnvsPerBlock, mu_s);
            __syncthreads();
            if (threadIdx.x < kEnvsPerBlock && e < g.E) g.mu[(size_t)e * g.A + a] = mu_s[threadIdx.x];
            __syncthreads();
        }
    }
    if (threadIdx.x >= kEnvsPerBlock || e >= g.E) return;
    for (int a0 = 0; a0 < g.A; a0 += 2) {
        float z[2];
        if (g.inj_noise) { z[0] = g.inj_noise[e * g.A + a0]; z[1] = a0 + 1 < g.A ? g.inj_noise[e * g.A + a0 + 1] : 0.f; }
        else {
            uint32_t o[4]; const uint64_t k = g.seed0 + (uint64_t)e;
            philox4x32_10((uint32_t)k, (uint32_t)(k >> 32), g.gstep[e], 0u, 1u, (uint32_t)(a0 / 2), o);
            if (g.use_random) { z[0] = u01_f32(o[0]); z[1] = u01_f32(o[2]); } else { z[0] = randn_f32(o[0], o[1]); z[1] = randn_f32(o[2], o[3]); }
        }
        for (int t = 0; t < 2 && a0 + t < g.A; ++t) {
            const int a = a0 + t; float r, ev;
            if (g.use_random) { r = g.low + z[t] * (g.high - g.low); ev = r; }               // rand(rng, act_space): already env space, :50-53
            else {
                r = tanhf(g.mu[(size_t)e * g.A + a] + expf(g.log_std[a]) * z[t]);            // rand(SquashedDiagGaussian) squashedDiagGaussian.jl:24-27
                ev = tanhf(r) * (g.high - g.low) / 2.0f + (g.low + g.high) / 2.0f;           // to_env(TanhScaleAdapter) default_adapters.jl:13-21
            }
            g.raw[e * g.A + a] = r; g.envact[e * g.A + a] = ev;
        }
    }
}
struct PushArgs {
    int E, D, A; long long cap, tail; const float *obs, *raw, *rew, *tobs, *nobs; const uint8_t *term, *trunc;
    float *rb_obs, *rb_next, *rb_act, *rb_rew; uint8_t *rb_term, *rb_trunc;
    unsigned long long* stamp;   // phase_stamp (null: none)
};
__global__ void sac_push_kernel(PushArgs g) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    phase_stamp(g.stamp);                                    // (within a microsecond of the kernel's end: the stamp feeds a per-iteration fps statistic)
    if (e >= g.E) return;
    const long long slot = (g.tail + e) % g.cap;
    const bool tr = g.trunc[e] != 0;
    for (int d = 0; d < g.D; ++d) {
        g.rb_obs[slot * g.D + d] = g.obs[(size_t)e * g.D + d];
        g.rb_next[slot * g.D + d] = tr ? g.tobs[(size_t)e * g.D + d] : g.nobs[(size_t)e * g.D + d];   // truncated_observation | next observation
    }
    for (int a = 0; a < g.A; ++a) g.rb_act[slot * g.A + a] = g.raw[e * g.A + a];               // unprocessed action, :72
    g.rb_rew[slot] = g.rew[e]; g.rb_term[slot] = g.term[e]; g.rb_trunc[slot] = g.trunc[e];
}
// One env step of the collection for a DEVICE env in ONE launch: sac_collect_head_kernel -> env_step_kernel -> env_observe_kernel -> sac_push_kernel are all one thread per env
// on data of that env only (four dependent launches of 4 - 5 us and their boundaries per collected step).  Same device functions, same order of operations, and every
// intermediate buffer (e_raw, e_envact, e_rew, e_term, e_trunc, e_tobs, obs_nxt) still written: bit-identical to the four-launch sequence (A = 1: every device Box env).
struct CollectEnvArgs { CollectHeadArgs head; PushArgs push; uint64_t seed0; int episode_len; float* state; int32_t* step_count; uint32_t* episode; uint32_t* gstep;
                        float* rew; uint8_t* term; uint8_t* trunc; float* tobs; float* nobs; };
template <int KIND>
__global__ __launch_bounds__(256) void sac_collect_env_kernel(CollectEnvArgs c) {
    constexpr int S = EnvSpec<KIND>::S, D = EnvSpec<KIND>::D;
    __shared__ float mu_s[kEnvsPerBlock];
    const CollectHeadArgs& g = c.head;
    const int e = blockIdx.x * kEnvsPerBlock + threadIdx.x;          // 256 threads per kEnvsPerBlock envs: all four waves on the output layer, then one thread per env
    if (!g.use_random && g.h2) { sac_mu_block(g, 0, blockIdx.x * kEnvsPerBlock, mu_s); __syncthreads(); }   // the actor's output layer (A = 1), same device function as the head kernel
    if (threadIdx.x >= kEnvsPerBlock || e >= g.E) return;
    float mu_e = 0.f;
    if (!g.use_random) { if (g.h2) { mu_e = mu_s[threadIdx.x]; g.mu[e] = mu_e; } else mu_e = g.mu[e]; }
    // ---- the action (sac_collect_head_kernel, A = 1) ----
    float z;
    if (g.inj_noise) z = g.inj_noise[e];
    else {
        uint32_t o[4]; const uint64_t k = g.seed0 + (uint64_t)e;
        philox4x32_10((uint32_t)k, (uint32_t)(k >> 32), c.gstep[e], 0u, 1u, 0u, o);
        z = g.use_random ? u01_f32(o[0]) : randn_f32(o[0], o[1]);
    }
    float r, ev;
    if (g.use_random) { r = g.low + z * (g.high - g.low); ev = r; }
    else {
        r = tanhf(mu_e + expf(g.log_std[0]) * z);
        ev = tanhf(r) * (g.high - g.low) / 2.0f + (g.low + g.high) / 2.0f;
    }
    g.raw[e] = r; g.envact[e] = ev;
    // ---- act! with auto-reset (env_step_kernel) ----
    float st[S];
#pragma unroll
    for (int i = 0; i < S; ++i) st[i] = c.state[(size_t)e * S + i];
    bool t;
    const float rw = env_step<KIND>(st, ev, 0, false, &t);
    const int sc = c.step_count[e] + 1;
    const bool tr = sc >= c.episode_len;
    c.rew[e] = rw; c.term[e] = t; c.trunc[e] = tr; c.gstep[e] += 1;
    float to[D];
    env_obs<KIND>(st, to);                                                              // terminal_observation (stored where truncated)
    if (tr) {
#pragma unroll
        for (int i = 0; i < D; ++i) c.tobs[(size_t)e * D + i] = to[i];
    }
    if (t || tr) { const uint32_t ep = c.episode[e] + 1; c.episode[e] = ep; c.step_count[e] = 0; env_reset<KIND>(c.seed0 + (uint64_t)e, ep, st); }
    else c.step_count[e] = sc;
#pragma unroll
    for (int i = 0; i < S; ++i) c.state[(size_t)e * S + i] = st[i];
    // ---- observe (env_observe_kernel) ----
    float no[D];
    env_obs<KIND>(st, no);
#pragma unroll
    for (int i = 0; i < D; ++i) c.nobs[(size_t)e * D + i] = no[i];
    // ---- push! (sac_push_kernel) ----
    const PushArgs& q = c.push;
    const long long slot = (q.tail + e) % q.cap;
#pragma unroll
    for (int d = 0; d < D; ++d) { q.rb_obs[slot * D + d] = q.obs[(size_t)e * D + d]; q.rb_next[slot * D + d] = tr ? to[d] : no[d]; }
    q.rb_act[slot] = r; q.rb_rew[slot] = rw; q.rb_term[slot] = t; q.rb_trunc[slot] = tr;
    phase_stamp(q.stamp);                                               // thread 0 of the last workgroup owns a live env (grid = ceil(E / kEnvsPerBlock)): the end of its work ~ the end of the phase
}
// host-batch helpers
__global__ void sac_squash_eval_kernel(int B, int A, const float* mu, const float* log_std, const float* noise, int deterministic, float low, float high,
                                       float* actions, float* logp, float* envact) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    float ls[kMaxA], a_[kMaxA], gg[kMaxA], nz[kMaxA];
    for (int a = 0; a < A; ++a) { ls[a] = log_std[a]; nz[a] = deterministic ? 0.f : noise[i * A + a]; }
    const float lp = squashed_sample_logp(mu + (size_t)i * A, ls, nz, A, a_, gg);              // deterministic: mode(d) = tanh(mean) :48-50
    for (int a = 0; a < A; ++a) {
        if (actions) actions[i * A + a] = a_[a];
        if (envact) envact[i * A + a] = tanhf(a_[a]) * (high - low) / 2.0f + (low + high) / 2.0f;
    }
    if (logp) logp[i] = lp;
}
__global__ void sac_concat_kernel(int B, int D, int A, const float* obs, const float* act, float* xq) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    for (int d = 0; d < D; ++d) xq[(size_t)i * (D + A) + d] = obs[(size_t)i * D + d];
    for (int a = 0; a < A; ++a) xq[(size_t)i * (D + A) + D + a] = act[(size_t)i * A + a];
}
__global__ void sac_noise_fill_kernel(int n, int A, SacRng rng, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int a = 0; a < A; ++a) out[i * A + a] = sac_noise(rng, 8, i, a);
}

}  // namespace

// =================================================================================================================
// handle
// =================================================================================================================
struct dril_sac_handle {
    dril_sac_config cfg;
    int D = 0, A = 0, S = 0, H1 = 0, H2 = 0, nmax = 0;
    int P = 0, Pa = 0, Pq = 0;                       // host (ABI) layout: actor | q1 | q2 | log_std
    int Pd = 0, Pqd = 0, log_std_off = 0;            // DEVICE layout: every net starts on a 16-byte boundary (float4 operand loads); pads stay zero
    NetOff actor{}, q0{};                            // device offsets
    hipStream_t stream = nullptr;
    float *params = nullptr, *adam_m = nullptr, *adam_v = nullptr, *target = nullptr, *g_critic = nullptr, *g_actor = nullptr;
    SacScalars* sc = nullptr; float* stats = nullptr; float* stats_out = nullptr; int stats_cap = 0;
    double* ssq_rows = nullptr;   // [stats_cap][adam_blocks_c + end_blocks] squared-gradient partials per update, summed on the host (grad_norm statistic)
    unsigned int* counter = nullptr; int adam_blocks_c = 0, end_blocks = 0;
    SacScalars* sc_next = nullptr;   // ping-pong partner of `sc` (fused heads: the entropy step writes the new state here, then the two are swapped)
    double* head_partials = nullptr; unsigned int* head_counter = nullptr; unsigned* col_h1p = nullptr; unsigned* col_w2p = nullptr; int* col_flags = nullptr; int col_tag = 0, col_w2tag = 0; bool col_w2_dirty = true, f16_fwd = true, l2_attr_set = false;   // the f16-piece collection forward (sac_collect_l2_kernel)
    unsigned long long* it_stamps = nullptr; int it_stamps_cap = 0; double wall_hz = 1e8; bool fused_heads = true; bool fused_dw1 = false; int fl_c = 0, fl_a = 0; bool fused_collect = true; bool fused_fwd = true; bool trace_enqueue = false; std::vector<hipEvent_t> it_events; int iter_chunk = 64;   // fused output-layer + head kernels (DRIL_SAC_NO_FUSED_HEADS=1: the round-1 launch sequence, A/B)
    float bt_actor[2], bt_critic[2], bt_ent[2]; int64_t grad_updates = 0; uint64_t update_counter = 0, aux_counter = 0;
    float target_entropy = 0, act_lo = -2.0f, act_hi = 2.0f; bool external = false;   // bounds of the agent-facing action space: Box(-2,2), Box(-1,1) under ScalingWrapperEnv
    // env
    float* state = nullptr; int32_t* step_count = nullptr; uint32_t *episode = nullptr, *gstep = nullptr; float* disc_returns = nullptr;
    float *obs_cur = nullptr, *obs_nxt = nullptr, *e_rew = nullptr, *e_tobs = nullptr, *e_raw = nullptr, *e_envact = nullptr; uint8_t *e_term = nullptr, *e_trunc = nullptr;
    uint64_t env_seed0 = 0; bool env_ready = false, obs_valid = false;
    // replay ring
    long long cap = 0, size = 0, head = 0;
    float *rb_obs = nullptr, *rb_next = nullptr, *rb_act = nullptr, *rb_rew = nullptr; uint8_t *rb_term = nullptr, *rb_trunc = nullptr;
    // batch + activations
    float *xa = nullptr, *ah1 = nullptr, *ah2 = nullptr, *mu = nullptr;               // actor: [nmax][D], [nmax][H], [nmax][A]
    float *xq = nullptr, *xq_next = nullptr, *xq_pi = nullptr;                       // [B][D+A]
    float *qh1 = nullptr, *qh2 = nullptr, *th1 = nullptr, *th2 = nullptr;            // [2][nq][H]
    float *q_cur = nullptr, *q_next = nullptr, *q_pi = nullptr, *dq = nullptr;       // [2][nq]
    float *dz2 = nullptr, *dz1 = nullptr, *dxq = nullptr, *dmu = nullptr;            // [2][nq][H], [2][nq][D+A], [B][A]
    float *b_rew = nullptr, *b_ne = nullptr, *b_nn = nullptr, *b_np = nullptr, *b_nlp = nullptr, *a_pi = nullptr, *g_pi = nullptr, *lp_pi = nullptr; uint8_t* b_term = nullptr;
    int nq = 0;
    // injected inputs (tests)
    float* collect_noise = nullptr; size_t collect_noise_count = 0;
    int inj_updates = 0; long long* inj_idx = nullptr; float *inj_ne = nullptr, *inj_nn = nullptr, *inj_np = nullptr;
    // host-batch scratch
    float *s_in = nullptr, *s_act = nullptr, *s_noise = nullptr, *s_out = nullptr, *s_out2 = nullptr;
    // timing
    hipEvent_t ev_a = nullptr, ev_b = nullptr; double collect_ms = 0, update_ms = 0; int64_t collect_steps = 0, updates = 0;
    std::string err;
};

namespace {

int sfail(dril_sac_handle* h, int code, const std::string& msg) { if (h) h->err = msg; else g_sac_create_error = msg; return code; }
#define SHIP(h, expr)                                                                                        \
    do { hipError_t _e = (expr); if (_e != hipSuccess)                                                       \
        return sfail(h, DRIL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); } while (0)
#define SNEED(h) do { if (!(h)) return sfail(nullptr, DRIL_ERR_NOT_INITIALISED, "null handle"); (void)hipSetDevice((h)->cfg.device); } while (0)
#define SDO(expr) do { int _rc = (expr); if (_rc != DRIL_OK) return _rc; } while (0)
#define S_NOT_EXTERNAL(h, what) do { if ((h)->external) return sfail(h, DRIL_ERR_UNSUPPORTED, what ": the envs of DRIL_ENV_EXTERNAL live on the host (dril_sac_predict_actions + dril_sac_ext_push)"); } while (0)

template <typename T> hipError_t smalloc(T** p, size_t n) {
    hipError_t e = hipMalloc((void**)p, (n ? n : 1) * sizeof(T));
    if (e == hipSuccess) e = hipMemset(*p, 0, (n ? n : 1) * sizeof(T));
    return e;
}
int round4(int x) { return (x + 3) & ~3; }
int gemm_pair(dril_sac_handle* h, GemmArgs a, int Za, GemmArgs b, int Zb) {
    SHIP(h, launch_gemm_pair(a, Za, b, Zb, h->stream));
    return DRIL_OK;
}
int gemm(dril_sac_handle* h, GemmArgs g, int Z) {
    SHIP(h, launch_gemm(g, Z, h->stream));
    return DRIL_OK;
}

// One net = {W1 b1 W2 b2 W3 b3} at `P + off` (+ z * zP for the second critic); activations are (features x n) column-major
struct NetBufs { float* h1; float* h2; float* out; long long zh, zo, zh2; };   // [Z][n][H1], [Z][n][O], [Z][n][H2]: batch strides of h1 / out / h2 (zh2 == 0: H1 == H2 shapes, use zh)
int net_forward(dril_sac_handle* h, const float* P, NetOff off, long long zP, int in, int O, const float* X, int ldx, long long zX, int n,
                NetBufs b, int Z, int zdivX = 1, bool hidden_only = false, bool first_done = false) {
    const int H1 = h->H1, H2 = h->H2, act = h->cfg.activation ? EPI_RELU : EPI_TANH;
    GemmArgs g = gemm_args();                                                       // h1 = act(W1 x + b1)
    // a narrow input (Pendulum: 3 features) over many rows (the collection forward): an elementwise pass instead of a K = 3 contraction
    const bool elem_l1 = !first_done && h->fused_fwd && Z == 1 && in <= 4 && ldx == in && n >= 1024 && H1 % 2 == 0;
    if (elem_l1) {
        CollectL1Args l1{n, in, H1, h->cfg.activation ? 1 : 0, X, P + off.w1, P + off.b1, b.h1, nullptr, nullptr, 0, (n + 7) / 8, nullptr, 0, nullptr, nullptr, 0};
        hipLaunchKernelGGL(sac_collect_l1_kernel, dim3((n + 7) / 8), dim3(256), 0, h->stream, l1);
    }
    if (!first_done && !elem_l1) {
    g.A = P + off.w1; g.sAm = 1; g.sAk = H1; g.zA = zP; g.B = X; g.sBk = 1; g.sBn = ldx; g.zB = zX; g.zdivB = zdivX;
    g.C = b.h1; g.sCm = 1; g.sCn = H1; g.zC = b.zh; g.bias = P + off.b1; g.zBias = zP; g.M = H1; g.N = n; g.K = in; g.epi = act;
    SDO(gemm(h, g, Z));
    }
    g = gemm_args();                                                                // h2 = act(W2 h1 + b2)
    g.A = P + off.w2; g.sAm = 1; g.sAk = H2; g.zA = zP; g.B = b.h1; g.sBk = 1; g.sBn = H1; g.zB = b.zh;
    g.C = b.h2; g.sCm = 1; g.sCn = H2; g.zC = b.zh2 ? b.zh2 : b.zh; g.bias = P + off.b2; g.zBias = zP; g.M = H2; g.N = n; g.K = H1; g.epi = act;
    SDO(gemm(h, g, Z));
    if (hidden_only) return DRIL_OK;                                                // the output layer is folded into the fused head kernel that follows
    g = gemm_args();                                                                // out = W3 h2 + b3
    g.A = P + off.w3; g.sAm = 1; g.sAk = O; g.zA = zP; g.B = b.h2; g.sBk = 1; g.sBn = H2; g.zB = b.zh2 ? b.zh2 : b.zh;
    g.C = b.out; g.sCm = 1; g.sCn = O; g.zC = b.zo; g.bias = P + off.b3; g.zBias = zP; g.M = O; g.N = n; g.K = H2; g.epi = EPI_NONE;
    return gemm(h, g, Z);
}
// reverse pass given dOut [Z][n][O]: parameter gradients into G (same layout as P; null = skip) and/or dX [Z][n][in] (null = skip)
// have_dz2: the fused head kernel already wrote dz2 = (W3' dOut) .* act'(h2); then [dW3|db3], [dW2|db2] and dz1 share ONE launch (three independent contractions)
int net_backward(dril_sac_handle* h, const float* P, NetOff off, long long zP, int in, int O, const float* X, int ldx, long long zX, int n,
                 NetBufs b, const float* dOut, float* G, float* dX, int Z, bool have_dz2 = false, bool skip_w1 = false) {
    const int H1 = h->H1, H2 = h->H2, mask = h->cfg.activation ? EPI_MASK_RELU : EPI_MASK_TANH;
    const long long zd = (long long)h->nq * H1;   // dz buffers are [2][nq][H] (H1 == H2 layouts are separate buffers)
    const bool big = (long long)((H2 + 31) / 32) * ((n + 31) / 32) * Z >= 2048;          // large batches: one launch per contraction (the pair kernel is the split-K shape)
    GemmArgs w, g;
    w = gemm_args(); w.A = dOut; w.sAm = 1; w.sAk = O; w.zA = b.zo; w.B = b.h2; w.sBk = H2; w.sBn = 1; w.zB = b.zh2 ? b.zh2 : b.zh; w.ones_n = 1;      // [dW3 | db3] = dOut . [h2' | 1]
    w.C = G ? G + off.w3 : nullptr; w.sCm = 1; w.sCn = O; w.zC = zP; w.M = O; w.N = H2 + 1; w.K = n;
    g = gemm_args();                                                                // dz2 = (W3' dOut) .* act'(h2)
    g.A = P + off.w3; g.sAm = O; g.sAk = 1; g.zA = zP; g.B = dOut; g.sBk = 1; g.sBn = O; g.zB = b.zo;
    g.C = h->dz2; g.sCm = 1; g.sCn = H2; g.zC = (long long)h->nq * H2; g.aux = b.h2; g.zAux = b.zh2 ? b.zh2 : b.zh; g.M = H2; g.N = n; g.K = O; g.epi = mask;
    GemmArgs w3 = w;
    if (!have_dz2) { if (G && !big) SDO(gemm_pair(h, w, Z, g, Z)); else { if (G) SDO(gemm(h, w, Z)); SDO(gemm(h, g, Z)); } }
    w = gemm_args(); w.A = h->dz2; w.sAm = 1; w.sAk = H2; w.zA = (long long)h->nq * H2; w.B = b.h1; w.sBk = H1; w.sBn = 1; w.zB = b.zh; w.ones_n = 1;   // [dW2 | db2] = dz2 . [h1' | 1]
    w.C = G ? G + off.w2 : nullptr; w.sCm = 1; w.sCn = H2; w.zC = zP; w.M = H2; w.N = H1 + 1; w.K = n;
    g = gemm_args();                                                                // dz1 = (W2' dz2) .* act'(h1)
    g.A = P + off.w2; g.sAm = H2; g.sAk = 1; g.zA = zP; g.B = h->dz2; g.sBk = 1; g.sBn = H2; g.zB = (long long)h->nq * H2;
    g.C = h->dz1; g.sCm = 1; g.sCn = H1; g.zC = zd; g.aux = b.h1; g.zAux = b.zh; g.M = H1; g.N = n; g.K = H2; g.epi = mask;
    if (have_dz2 && G && !big) { const GemmArgs gs[3] = {w3, w, g}; const int zs[3] = {Z, Z, Z}; SHIP(h, launch_gemm_multi(gs, zs, 3, h->stream)); }
    else { if (have_dz2 && G) SDO(gemm(h, w3, Z)); if (G && !big) SDO(gemm_pair(h, w, Z, g, Z)); else { if (G) SDO(gemm(h, w, Z)); SDO(gemm(h, g, Z)); } }
    w = gemm_args(); w.A = h->dz1; w.sAm = 1; w.sAk = H1; w.zA = zd; w.B = X; w.sBk = ldx; w.sBn = 1; w.zB = zX; w.ones_n = 1;          // [dW1 | db1] = dz1 . [x' | 1]
    w.C = G ? G + off.w1 : nullptr; w.sCm = 1; w.sCn = H1; w.zC = zP; w.M = H1; w.N = in + 1; w.K = n;
    g = gemm_args(); g.A = P + off.w1; g.sAm = H1; g.sAk = 1; g.zA = zP; g.B = h->dz1; g.sBk = 1; g.sBn = H1; g.zB = zd;               // dx = W1' dz1
    g.C = dX; g.sCm = 1; g.sCn = in; g.zC = (long long)h->nq * in; g.M = in; g.N = n; g.K = H1;
    if (skip_w1) { if (dX) SDO(gemm(h, g, Z)); return DRIL_OK; }                     // [dW1 | db1] is computed by the optimiser kernel that follows (first_layer_opt_block)
    if (G && dX && !big) SDO(gemm_pair(h, w, Z, g, Z)); else { if (G) SDO(gemm(h, w, Z)); if (dX) SDO(gemm(h, g, Z)); }
    return DRIL_OK;
}
NetBufs actor_bufs(dril_sac_handle* h) { return NetBufs{h->ah1, h->ah2, h->mu, 0, 0, 0}; }
NetBufs q_bufs(dril_sac_handle* h, float* h1, float* h2, float* out) { return NetBufs{h1, h2, out, (long long)h->nq * h->H1, (long long)h->nq, (long long)h->nq * h->H2}; }

int ssync(dril_sac_handle* h) { SHIP(h, hipStreamSynchronize(h->stream)); return DRIL_OK; }

int adam_range(dril_sac_handle* h, int lo, int n, float* grads, const float* bt, double* ssq, int blocks, FirstLayerOpt fl = FirstLayerOpt{}) {
    AdamRangeArgs a{h->params + lo, h->adam_m + lo, h->adam_v + lo, grads ? grads + lo : nullptr, n, h->cfg.learning_rate, h->cfg.adam_beta1,
                    h->cfg.adam_beta2, h->cfg.adam_eps, bt[0], bt[1], ssq, fl};
    hipLaunchKernelGGL(sac_adam_kernel, dim3(blocks), dim3(256), first_layer_opt_lds(fl), h->stream, a);
    SHIP(h, hipGetLastError());
    return DRIL_OK;
}

// one update!(agent, alg, batch): sac.jl:299-404.  `slot` = index into the injected batches (-1 = Philox), `out` = device stats row
int sac_one_update(dril_sac_handle* h, int slot, float* out, unsigned long long* stamp = nullptr) {
    double* ssq_row = h->ssq_rows + (size_t)((out - h->stats_out) / 8) * (h->adam_blocks_c + h->end_blocks);   // this update's squared-gradient partials: [critic Adam blocks | end blocks]
    const int B = h->cfg.batch_size, D = h->D, A = h->A, W = D + A;
    const SacRng rng{h->cfg.seed ^ 0x5ac5ac5ac5ac5ac5ull, h->update_counter};
    GatherArgs ga{B, D, A, h->cap, h->head, h->size, h->rb_obs, h->rb_next, h->rb_act, h->rb_rew, h->rb_term,
                  slot >= 0 && h->inj_idx ? h->inj_idx + (size_t)slot * B : nullptr, slot >= 0 && h->inj_ne ? h->inj_ne + (size_t)slot * B * A : nullptr,
                  slot >= 0 && h->inj_nn ? h->inj_nn + (size_t)slot * B * A : nullptr, slot >= 0 && h->inj_np ? h->inj_np + (size_t)slot * B * A : nullptr,
                  rng, h->xa, h->xq, h->b_rew, h->b_ne, h->b_nn, h->b_np, h->b_term};
    const int relu_ = h->cfg.activation ? 1 : 0;
    const bool l1 = h->fused_heads && W <= kFusedL1MaxIn;      // narrow inputs: first layers inside the gather / head kernels (two launches less)
    if (l1) {
        GatherL1Args gl{ga, h->H1, relu_, h->params + h->actor.w1, h->params + h->actor.b1, h->ah1, h->params, h->q0.w1, h->q0.b1, h->Pqd, (long long)h->nq * h->H1, h->qh1};
        hipLaunchKernelGGL(sac_gather_l1_kernel, dim3(B), dim3(256), 0, h->stream, gl);
    } else hipLaunchKernelGGL(sac_gather_kernel, dim3((B + 255) / 256), dim3(256), 0, h->stream, ga);
    const int relu = h->cfg.activation ? 1 : 0, hb = (B + kHeadSamplesPerBlock - 1) / kHeadSamplesPerBlock;
    EntNextArgs en{B, D, A, h->mu, h->params + h->log_std_off, h->b_ne, h->b_nn, h->xa, h->xq_next, h->b_nlp, h->b_np, h->xq_pi, h->a_pi, h->g_pi, h->lp_pi, h->sc, h->target_entropy,
                   h->cfg.learning_rate, h->cfg.adam_beta1, h->cfg.adam_beta2, h->cfg.adam_eps, h->bt_ent[0], h->bt_ent[1], h->cfg.auto_ent_coef, h->stats};
    SquashBwdArgs sb{B, D, A, h->mu, h->params + h->log_std_off, h->b_np, h->a_pi, h->g_pi, h->dxq, h->sc, h->dmu, h->g_actor + h->log_std_off};
    const float ent_bt1 = h->bt_ent[0], ent_bt2 = h->bt_ent[1];
    if (h->fused_heads) {
        sb.sc = h->sc_next;                                                           // the state after this update's entropy step (written by the critic head kernel)
        // actor means of (obs | next obs): hidden layers as contractions, then output layer + entropy-coefficient step + next actions + the actor-loss sample in one launch
        SDO(net_forward(h, h->params, h->actor, 0, D, A, h->xa, D, 0, 2 * B, actor_bufs(h), 1, 1, true, l1));
        double* hp_critic = h->head_partials; double* hp_actor = hp_critic + 2 * (size_t)hb; double* hp_ls = hp_actor + 2 * (size_t)hb; double* hp_ent = hp_ls + (size_t)kMaxA * hb;   // one region per head kernel
        ActorHeadFusedArgs af{en, h->ah2, h->H2, h->params + h->actor.w3, h->params + h->actor.b3, h->mu, hp_ent, h->head_counter,
                              l1 ? 1 : 0, h->H1, relu, h->params, h->q0.w1, h->q0.b1, h->Pqd, (long long)h->nq * h->H1, h->qh1};
        if (W <= kL1PreMaxIn && (!l1 || h->H1 <= kL1PreMaxH)) hipLaunchKernelGGL(sac_actor_out_ent_kernel<true>, dim3(hb), dim3(256), 0, h->stream, af);
        else hipLaunchKernelGGL(sac_actor_out_ent_kernel<false>, dim3(hb), dim3(256), 0, h->stream, af);
        if (h->cfg.auto_ent_coef) { h->bt_ent[0] *= h->cfg.adam_beta1; h->bt_ent[1] *= h->cfg.adam_beta2; }
        // critic: all four Q nets' hidden layers in one pass (z = 0,1 the critics on (obs, action), z = 2,3 the targets on (next obs, next action)), then output
        // layers + Bellman target + loss head + dz2 of the critics in one launch; [dW3|db3], [dW2|db2], dz1 in one launch; [dW1|db1]; Adam (:362)
        SDO(net_forward(h, h->params, h->q0, h->Pqd, W, 1, h->xq, W, (long long)h->nq * W, B, q_bufs(h, h->qh1, h->qh2, h->q_cur), 4, 2, true, l1));
        QHeadFusedArgs qc{B, h->H2, h->nq, relu, 4, h->Pqd, (long long)h->nq * h->H2, h->params, h->q0.w3, h->q0.b3, h->qh2, h->q_cur, h->dz2,
                          0, h->b_rew, h->b_nlp, h->lp_pi, h->b_term, h->sc, h->cfg.gamma, h->dq, h->stats, hp_critic, h->head_counter,
                          hp_ent, h->sc_next, h->cfg.auto_ent_coef, h->cfg.learning_rate, h->cfg.adam_beta1, h->cfg.adam_beta2, h->cfg.adam_eps, ent_bt1, ent_bt2};
        hipLaunchKernelGGL(sac_q_out_head_kernel, dim3(hb), dim3(256), 0, h->stream, qc);
        const bool fl = h->fused_dw1;
        SDO(net_backward(h, h->params, h->q0, h->Pqd, W, 1, h->xq, W, 0, B, q_bufs(h, h->qh1, h->qh2, h->q_cur), h->dq, h->g_critic, nullptr, 2, true, fl));
        FirstLayerOpt flc{};
        if (fl) flc = FirstLayerOpt{h->fl_c, 2, W, h->H1, B, relu, 0, h->q0.b1 - h->q0.w1, h->Pqd, h->dz1, (long long)h->nq * h->H1, h->xq, W, h->xq_pi, h->qh1, (long long)h->nq * h->H1};
        SDO(adam_range(h, h->q0.w1, 2 * h->Pqd, h->g_critic, h->bt_critic, ssq_row, h->adam_blocks_c, flc));
        h->bt_critic[0] *= h->cfg.adam_beta1; h->bt_critic[1] *= h->cfg.adam_beta2;
        // actor (:93-105) with the UPDATED critics: hidden layers, then output layers + loss head + dz2 in one launch; dz1; then the action columns of W1' dz1, the
        // reverse of the squashed sample and the actor's dz2 in one launch; the actor's [dW3|db3], [dW2|db2], dz1 in one launch; [dW1|db1]
        SDO(net_forward(h, h->params, h->q0, h->Pqd, W, 1, h->xq_pi, W, 0, B, q_bufs(h, h->qh1, h->qh2, h->q_pi), 2, 1, true, fl));   // (fl: the critics' Adam launch already evaluated the first layer)
        QHeadFusedArgs qp{B, h->H2, h->nq, relu, 2, h->Pqd, (long long)h->nq * h->H2, h->params, h->q0.w3, h->q0.b3, h->qh2, h->q_pi, h->dz2,
                          1, h->b_rew, h->b_nlp, h->lp_pi, h->b_term, h->sc_next, h->cfg.gamma, h->dq, h->stats, hp_actor, h->head_counter,
                          nullptr, nullptr, 0, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        hipLaunchKernelGGL(sac_q_out_head_kernel, dim3(hb), dim3(256), 0, h->stream, qp);
        {   // dz1 = (W2' dz2) .* act'(h1) of both critics (no parameter gradients on this pass: Zygote differentiates the actor loss w.r.t. the actor only)
            const int H1 = h->H1, H2 = h->H2;
            GemmArgs g = gemm_args();
            g.A = h->params + h->q0.w2; g.sAm = H2; g.sAk = 1; g.zA = h->Pqd; g.B = h->dz2; g.sBk = 1; g.sBn = H2; g.zB = (long long)h->nq * H2;
            g.C = h->dz1; g.sCm = 1; g.sCn = H1; g.zC = (long long)h->nq * H1; g.aux = h->qh1; g.zAux = (long long)h->nq * H1; g.M = H1; g.N = B; g.K = H2;
            g.epi = relu ? EPI_MASK_RELU : EPI_MASK_TANH;
            SDO(gemm(h, g, 2));
        }
        SquashFusedArgs sf{sb, h->H1, h->H2, relu, h->nq, h->params, h->q0.w1, h->Pqd, h->dz1, h->params + h->actor.w3, h->ah2, h->dz2, hp_ls, h->head_counter};
        hipLaunchKernelGGL(sac_dx_squash_kernel, dim3(hb), dim3(256), 0, h->stream, sf);
        SDO(net_backward(h, h->params, h->actor, 0, D, A, h->xa, D, 0, B, actor_bufs(h), h->dmu, h->g_actor, nullptr, 1, true, fl));
        std::swap(h->sc, h->sc_next);                                                 // h->sc is the current state again for whoever reads it next
    } else {
        // actor means of (obs | next obs) in one pass: the entropy constant (:318-325) and the actor loss (:101) share the obs half,
        // the critic target (:131) uses the next-obs half; the actor parameters do not change until the actor step
        SDO(net_forward(h, h->params, h->actor, 0, D, A, h->xa, D, 0, 2 * B, actor_bufs(h), 1));
        hipLaunchKernelGGL(sac_ent_next_kernel, dim3(1), dim3(256), 0, h->stream, en);
        if (h->cfg.auto_ent_coef) { h->bt_ent[0] *= h->cfg.adam_beta1; h->bt_ent[1] *= h->cfg.adam_beta2; }
        // critic: target values with the target networks (:133-135), current values (:117), loss head, reverse pass, Adam (:362)
        // all four Q nets in one pass: z = 0,1 the critics on (obs, action), z = 2,3 the targets on (next obs, next action)
        SDO(net_forward(h, h->params, h->q0, h->Pqd, W, 1, h->xq, W, (long long)h->nq * W, B, q_bufs(h, h->qh1, h->qh2, h->q_cur), 4, 2));
        CriticHeadArgs ch{B, h->q_next, h->q_cur, h->b_rew, h->b_nlp, h->b_term, h->sc, h->cfg.gamma, h->dq, h->stats};
        hipLaunchKernelGGL(sac_critic_head_kernel, dim3(1), dim3(256), 0, h->stream, ch);
        SDO(net_backward(h, h->params, h->q0, h->Pqd, W, 1, h->xq, W, 0, B, q_bufs(h, h->qh1, h->qh2, h->q_cur), h->dq, h->g_critic, nullptr, 2));
        SDO(adam_range(h, h->q0.w1, 2 * h->Pqd, h->g_critic, h->bt_critic, ssq_row, h->adam_blocks_c));
        h->bt_critic[0] *= h->cfg.adam_beta1; h->bt_critic[1] *= h->cfg.adam_beta2;
        // actor (:93-105) with the UPDATED critics: sample, values, loss head, input gradients of the critics, squash reverse, actor reverse
        SDO(net_forward(h, h->params, h->q0, h->Pqd, W, 1, h->xq_pi, W, 0, B, q_bufs(h, h->qh1, h->qh2, h->q_pi), 2));
        ActorHeadArgs ah{B, h->q_pi, h->lp_pi, h->sc, h->dq, h->stats};
        hipLaunchKernelGGL(sac_actor_head_kernel, dim3(1), dim3(256), 0, h->stream, ah);
        SDO(net_backward(h, h->params, h->q0, h->Pqd, W, 1, h->xq_pi, W, 0, B, q_bufs(h, h->qh1, h->qh2, h->q_pi), h->dq, nullptr, h->dxq, 2));
        hipLaunchKernelGGL(sac_squash_bwd_kernel, dim3(1), dim3(256), 0, h->stream, sb);
        SDO(net_backward(h, h->params, h->actor, 0, D, A, h->xa, D, 0, B, actor_bufs(h), h->dmu, h->g_actor, nullptr, 1));
    }
    // apply_gradients(train_state, actor_loss_grad) :382 + target networks :385-389 + statistics: one launch
    const int do_polyak = h->grad_updates % h->cfg.target_update_interval == 0;
    StepEndArgs se{h->params, h->adam_m, h->adam_v, h->g_actor, round4(h->actor.end), h->log_std_off, A, h->q0.w1, 2 * h->Pqd,
                   h->cfg.learning_rate, h->cfg.adam_beta1, h->cfg.adam_beta2, h->cfg.adam_eps, h->bt_actor[0], h->bt_actor[1], h->bt_critic[0], h->bt_critic[1],
                   h->target, h->cfg.tau, do_polyak, ssq_row + h->adam_blocks_c, h->stats, out,
                   h->fused_heads ? hb : 0, B, h->head_partials, h->head_partials + 2 * (size_t)hb, h->head_partials + 4 * (size_t)hb, h->g_actor + h->log_std_off, stamp,
                   FirstLayerOpt{}, h->g_actor};
    if (h->fused_heads && h->fused_dw1)      // the actor's [dW1 | db1] = dz1 . [obs' | 1] and its step: blocks of this launch
        se.fl = FirstLayerOpt{h->fl_a, 1, D, h->H1, B, relu_, h->actor.w1, h->actor.b1, 0, h->dz1, 0, h->xa, D, nullptr, nullptr, 0};
    hipLaunchKernelGGL(sac_step_end_kernel, dim3(h->end_blocks), dim3(256), first_layer_opt_lds(se.fl), h->stream, se);
    SHIP(h, hipGetLastError());
    h->bt_actor[0] *= h->cfg.adam_beta1; h->bt_actor[1] *= h->cfg.adam_beta2;
    h->bt_critic[0] *= h->cfg.adam_beta1; h->bt_critic[1] *= h->cfg.adam_beta2;
    h->grad_updates += 1; h->update_counter += 1; h->col_w2_dirty = true;      // (the actor moved: its W2 planes are re-cut by the next collection step)
    return DRIL_OK;
}

int ensure_obs(dril_sac_handle* h) {
    if (!h->env_ready) return sfail(h, DRIL_ERR_NOT_INITIALISED, "dril_sac_env_reset has not been called");
    if (!h->obs_valid) { SHIP(h, launch_env_observe(h->cfg.env_kind, h->cfg.n_envs, h->state, h->obs_cur, h->stream)); h->obs_valid = true; }
    return DRIL_OK;
}
// one step of collect_trajectories (off_policy_collection.jl:42-93) for all envs
int collect_step(dril_sac_handle* h, int use_random, const float* inj_noise, unsigned long long* stamp = nullptr) {
    const int E = h->cfg.n_envs, D = h->D, A = h->A;
    // predict_actions_raw :55 — the hidden layers as contractions (the first one inside the second's staging when the input is narrow); the output layer inside the head /
    // env kernel that follows (fused_fwd; DRIL_SAC_NO_FUSED_FWD=1: three contractions and mu through memory, the round 1 - 3 form)
    const bool mu_in_head = h->fused_fwd && !use_random && h->H2 % 4 == 0;
    // the hidden layers: for a narrow observation and tile-sized widths the f16-piece form (sac_collect_l1_kernel cuts h1 and, when the actor changed, W2 into f16 planes;
    // sac_collect_l2_kernel contracts them; out-of-range values fall back to f32 MFMAs inside the kernel), else the generic contractions
    const bool f16_path = mu_in_head && h->f16_fwd && D <= 4 && h->H1 % kL2KC == 0 && h->H2 % kL2TM == 0 && E >= 1024 && h->col_h1p;
    if (f16_path) {
        const int relu = h->cfg.activation ? 1 : 0, nb_l1 = (E + 7) / 8, nb_w2 = h->col_w2_dirty ? 128 : 0;
        h->col_tag += 1; if (h->col_w2_dirty) h->col_w2tag += 1;
        CollectL1Args l1{E, D, h->H1, relu, h->obs_cur, h->params + h->actor.w1, h->params + h->actor.b1, h->ah1, h->col_h1p, h->col_flags, h->col_tag,
                         nb_l1, h->params + h->actor.w2, h->H2, h->col_w2p, h->col_flags + 1, h->col_w2tag};
        hipLaunchKernelGGL(sac_collect_l1_kernel, dim3(nb_l1 + nb_w2), dim3(256), 0, h->stream, l1);
        h->col_w2_dirty = false;
        CollectL2Args l2{E, h->H1, h->H2, relu, h->col_h1p, h->col_w2p, h->params + h->actor.b2, h->ah2, h->col_flags, h->col_tag, h->col_flags + 1, h->col_w2tag, h->ah1, h->params + h->actor.w2};
        if (!h->l2_attr_set) { SHIP(h, hipFuncSetAttribute((const void*)sac_collect_l2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kL2LdsBytes)); h->l2_attr_set = true; }   // per handle = per device
        hipLaunchKernelGGL(sac_collect_l2_kernel, dim3((E + kL2TN - 1) / kL2TN, h->H2 / kL2TM), dim3(256), kL2LdsBytes, h->stream, l2);
        SHIP(h, hipGetLastError());
    } else if (!use_random) SDO(net_forward(h, h->params, h->actor, 0, D, A, h->obs_cur, D, 0, E, actor_bufs(h), 1, 1, mu_in_head));
    CollectHeadArgs ca{E, A, use_random, h->mu, h->params + h->log_std_off, inj_noise, h->gstep, h->env_seed0, h->act_lo, h->act_hi, h->e_raw, h->e_envact,
                       mu_in_head ? h->ah2 : nullptr, h->params + h->actor.w3, h->params + h->actor.b3, h->H2};
    if (A == 1 && !h->external && h->fused_collect) {                                                            // every device Box env: head + act! + observe + push! in one launch
        const long long tail1 = (h->head + h->size) % h->cap;
        PushArgs pa1{E, D, A, h->cap, tail1, h->obs_cur, h->e_raw, h->e_rew, h->e_tobs, h->obs_nxt, h->e_term, h->e_trunc,
                     h->rb_obs, h->rb_next, h->rb_act, h->rb_rew, h->rb_term, h->rb_trunc, stamp};
        CollectEnvArgs ce{ca, pa1, h->env_seed0, h->cfg.episode_len, h->state, h->step_count, h->episode, h->gstep, h->e_rew, h->e_term, h->e_trunc, h->e_tobs, h->obs_nxt};
        const dim3 grid((E + kEnvsPerBlock - 1) / kEnvsPerBlock), block(256);
        if (h->cfg.env_kind == DRIL_ENV_PENDULUM) hipLaunchKernelGGL(sac_collect_env_kernel<1>, grid, block, 0, h->stream, ce);
        else if (h->cfg.env_kind == DRIL_ENV_PENDULUM_SCALED) hipLaunchKernelGGL(sac_collect_env_kernel<2>, grid, block, 0, h->stream, ce);
        else if (h->cfg.env_kind == DRIL_ENV_MOUNTAINCAR_CONTINUOUS_SCALED) hipLaunchKernelGGL(sac_collect_env_kernel<7>, grid, block, 0, h->stream, ce);
        else hipLaunchKernelGGL(sac_collect_env_kernel<4>, grid, block, 0, h->stream, ce);
        SHIP(h, hipGetLastError());
        const long long over1 = h->size + E - h->cap;
        if (over1 > 0) { h->head = (h->head + over1) % h->cap; h->size = h->cap; } else h->size += E;
        std::swap(h->obs_cur, h->obs_nxt);
        return DRIL_OK;
    }
    hipLaunchKernelGGL(sac_collect_head_kernel, dim3((E + kEnvsPerBlock - 1) / kEnvsPerBlock), dim3(256), 0, h->stream, ca);
    MonitorArgs mon{nullptr, nullptr, nullptr, nullptr, nullptr};
    SHIP(h, launch_env_step(h->cfg.env_kind, E, h->env_seed0, h->cfg.episode_len, 0, 0, h->e_envact, h->state, h->step_count, h->episode, h->gstep,
                            h->e_rew, h->e_term, h->e_trunc, h->e_tobs, mon, h->stream));                                     // act! :60
    SHIP(h, launch_env_observe(h->cfg.env_kind, E, h->state, h->obs_nxt, h->stream));                                         // observe :61
    const long long tail = (h->head + h->size) % h->cap;
    PushArgs pa{E, D, A, h->cap, tail, h->obs_cur, h->e_raw, h->e_rew, h->e_tobs, h->obs_nxt, h->e_term, h->e_trunc,
                h->rb_obs, h->rb_next, h->rb_act, h->rb_rew, h->rb_term, h->rb_trunc, stamp};
    hipLaunchKernelGGL(sac_push_kernel, dim3((E + 255) / 256), dim3(256), 0, h->stream, pa);
    SHIP(h, hipGetLastError());
    const long long over = h->size + E - h->cap;                                                  // CircularBuffer: overwrite the oldest
    if (over > 0) { h->head = (h->head + over) % h->cap; h->size = h->cap; } else h->size += E;
    std::swap(h->obs_cur, h->obs_nxt);
    return DRIL_OK;
}
int collect(dril_sac_handle* h, int n_steps, int use_random, double* fps) {
    if (n_steps <= 0) return sfail(h, DRIL_ERR_INVALID_ARG, "n_steps must be positive");
    if (h->collect_noise && h->collect_noise_count != (size_t)n_steps * h->cfg.n_envs * h->A)
        return sfail(h, DRIL_ERR_INVALID_ARG, "injected collect noise must hold n_steps * n_envs * action_dim values");
    SDO(ensure_obs(h));
    const auto t0 = std::chrono::steady_clock::now();
    if (h->cfg.profile_events) hipEventRecord(h->ev_a, h->stream);
    for (int t = 0; t < n_steps; ++t)
        SDO(collect_step(h, use_random, h->collect_noise ? h->collect_noise + (size_t)t * h->cfg.n_envs * h->A : nullptr));
    if (h->cfg.profile_events) hipEventRecord(h->ev_b, h->stream);
    SDO(ssync(h));
    if (h->cfg.profile_events) { float ms = 0; if (hipEventElapsedTime(&ms, h->ev_a, h->ev_b) == hipSuccess) { h->collect_ms += ms; h->collect_steps += n_steps; } }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (fps) *fps = (double)n_steps * h->cfg.n_envs / (dt > 0 ? dt : 1e-9);                      // :126-128
    if (h->collect_noise) { hipFree(h->collect_noise); h->collect_noise = nullptr; h->collect_noise_count = 0; }
    return DRIL_OK;
}
int ensure_stats(dril_sac_handle* h, int n) {
    if (n <= h->stats_cap) return DRIL_OK;
    if (h->stats_out) hipFree(h->stats_out);
    if (h->ssq_rows) hipFree(h->ssq_rows);
    h->stats_out = nullptr; h->ssq_rows = nullptr;
    SHIP(h, smalloc(&h->stats_out, (size_t)n * 8)); SHIP(h, smalloc(&h->ssq_rows, (size_t)n * (h->adam_blocks_c + h->end_blocks))); h->stats_cap = n;
    return DRIL_OK;
}
void fill_stats(const dril_sac_handle* h, const float* rows, int n, dril_sac_stats* out) {
    for (int k = 0; k < n; ++k) {
        const float* r = rows + (size_t)k * 8; dril_sac_stats& s = out[k]; memset(&s, 0, sizeof(s));
        s.actor_loss = r[0]; s.critic_loss = r[1]; s.entropy_loss = h->cfg.auto_ent_coef ? r[2] : 0.f; s.mean_q_values = r[3];
        s.entropy_coefficient = r[4]; s.grad_norm = r[5]; s.has_entropy_loss = h->cfg.auto_ent_coef ? 1 : 0;
    }
}
// the statistics rows of the last n gradient steps (stream drained): device rows + the squared-gradient partials summed here
int fetch_stats(dril_sac_handle* h, int n, dril_sac_stats* out) {
    std::vector<float> rows((size_t)n * 8);
    SHIP(h, hipMemcpy(rows.data(), h->stats_out, rows.size() * 4, hipMemcpyDeviceToHost));
    const int nb = h->adam_blocks_c + h->end_blocks;
    std::vector<double> ssq((size_t)n * nb);
    SHIP(h, hipMemcpy(ssq.data(), h->ssq_rows, ssq.size() * 8, hipMemcpyDeviceToHost));
    for (int k = 0; k < n; ++k) { double t = 0; for (int b = 0; b < nb; ++b) t += ssq[(size_t)k * nb + b]; rows[(size_t)k * 8 + 5] = (float)sqrt(t); }   // grad_norm, sac.jl:393 (index order: deterministic)
    fill_stats(h, rows.data(), n, out);
    return DRIL_OK;
}
int run_updates(dril_sac_handle* h, int n_updates, bool injected, dril_sac_stats* out) {
    if (h->size <= 0) return sfail(h, DRIL_ERR_NOT_INITIALISED, "the replay buffer is empty");
    SDO(ensure_stats(h, n_updates));
    if (h->cfg.profile_events) hipEventRecord(h->ev_a, h->stream);
    const auto t_enq0 = std::chrono::steady_clock::now();
    for (int k = 0; k < n_updates; ++k) SDO(sac_one_update(h, injected ? k : -1, h->stats_out + (size_t)k * 8));
    if (h->cfg.profile_events) hipEventRecord(h->ev_b, h->stream);
    const auto t_enq1 = std::chrono::steady_clock::now();
    SDO(ssync(h));
    if (h->trace_enqueue) {
        const auto t_done = std::chrono::steady_clock::now();
        fprintf(stderr, "[dril_sac] %d update(s): enqueue %.1f us, until drained %.1f us\n", n_updates,
                std::chrono::duration<double, std::micro>(t_enq1 - t_enq0).count(), std::chrono::duration<double, std::micro>(t_done - t_enq0).count());
    }
    if (h->cfg.profile_events) { float ms = 0; if (hipEventElapsedTime(&ms, h->ev_a, h->ev_b) == hipSuccess) { h->update_ms += ms; h->updates += n_updates; } }
    if (out) SDO(fetch_stats(h, n_updates, out));
    return DRIL_OK;
}
// train!'s loop body (sac.jl:464-535) for `count` iterations WITHOUT a host synchronisation between them: {train_freq env steps, n_upd gradient steps} enqueued back to
// back, one drain at the end.  Nothing the host decides depends on device results (the replay ring's head / size are host counters, batch indices and noise are
// device Philox streams keyed by the update counter), so the launches are the ones the step-by-step sequence issues, in the same order: bit-identical state.
// fps of an iteration = env steps / HIP-event time of its collection (the reference times the same span on the host clock, off_policy_collection.jl:126-128).
int run_iterations(dril_sac_handle* h, int count, int tf, int n_upd, dril_sac_stats* stats, int64_t stats_room, double* fps, int64_t fps_room) {
    if (count <= 0) return DRIL_OK;
    SDO(ensure_obs(h));
    if (n_upd > 0) { if (h->size <= 0 && tf <= 0) return sfail(h, DRIL_ERR_NOT_INITIALISED, "the replay buffer is empty"); SDO(ensure_stats(h, count * n_upd)); }
    const bool timed = h->cfg.profile_events || fps;
    const int n_st = 1 + 2 * count;                                                        // [loop start | per iteration: collection end, update end] (phase_stamp)
    if (timed) {
        if (n_st > h->it_stamps_cap) { if (h->it_stamps) hipFree(h->it_stamps); h->it_stamps = nullptr; SHIP(h, smalloc(&h->it_stamps, (size_t)n_st)); h->it_stamps_cap = n_st; }
        SHIP(h, hipMemsetAsync(h->it_stamps, 0, sizeof(unsigned long long) * n_st, h->stream));
        hipLaunchKernelGGL(sac_stamp_kernel, dim3(1), dim3(64), 0, h->stream, h->it_stamps);
    }
    for (int j = 0; j < count; ++j) {
        for (int t = 0; t < tf; ++t) SDO(collect_step(h, 0, nullptr, timed && t == tf - 1 ? h->it_stamps + 1 + 2 * j : nullptr));
        for (int k = 0; k < n_upd; ++k) SDO(sac_one_update(h, -1, h->stats_out + ((size_t)j * n_upd + k) * 8, timed && k == n_upd - 1 ? h->it_stamps + 2 + 2 * j : nullptr));
    }
    SDO(ssync(h));
    if (timed) {
        std::vector<unsigned long long> st((size_t)n_st);
        SHIP(h, hipMemcpy(st.data(), h->it_stamps, st.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long prev = st[0];
        for (int j = 0; j < count; ++j) {
            const unsigned long long ce = st[1 + 2 * j] ? st[1 + 2 * j] : prev, ue = st[2 + 2 * j] ? st[2 + 2 * j] : ce;   // (a phase that did not run left no stamp: zero length)
            const double c = 1e3 * (double)(ce - prev) / h->wall_hz, u = 1e3 * (double)(ue - ce) / h->wall_hz;             // ms
            if (h->cfg.profile_events) { h->collect_ms += c; h->collect_steps += tf; h->update_ms += u; h->updates += n_upd; }
            if (fps && j < fps_room) fps[j] = (double)tf * h->cfg.n_envs / (c > 0 ? 1e-3 * c : 1e-9);
            prev = ue;
        }
    }
    if (stats && n_upd > 0 && stats_room > 0) {
        std::vector<dril_sac_stats> tmp((size_t)count * n_upd);
        SDO(fetch_stats(h, count * n_upd, tmp.data()));
        for (int64_t k = 0; k < std::min<int64_t>(stats_room, (int64_t)tmp.size()); ++k) stats[k] = tmp[(size_t)k];
    }
    return DRIL_OK;
}
void clear_injected(dril_sac_handle* h) {
    if (h->inj_idx) hipFree(h->inj_idx); if (h->inj_ne) hipFree(h->inj_ne); if (h->inj_nn) hipFree(h->inj_nn); if (h->inj_np) hipFree(h->inj_np);
    h->inj_idx = nullptr; h->inj_ne = h->inj_nn = h->inj_np = nullptr; h->inj_updates = 0;
}
void reset_optimizer(dril_sac_handle* h) {
    h->bt_actor[0] = h->bt_critic[0] = h->bt_ent[0] = h->cfg.adam_beta1; h->bt_actor[1] = h->bt_critic[1] = h->bt_ent[1] = h->cfg.adam_beta2;
    h->grad_updates = 0;
}
// ---- host (ABI) layout <-> padded device layout ------------------------------------------------------------------------
struct Seg { int host, dev, len; };
void param_segs(const dril_sac_handle* h, Seg (&s)[4]) {
    s[0] = {0, 0, h->Pa}; s[1] = {h->Pa, h->q0.w1, h->Pq}; s[2] = {h->Pa + h->Pq, h->q0.w1 + h->Pqd, h->Pq}; s[3] = {h->Pa + 2 * h->Pq, h->log_std_off, h->A};
}
int params_to_device(dril_sac_handle* h, float* dev, const float* host) {
    Seg s[4]; param_segs(h, s);
    for (const Seg& g : s) SHIP(h, hipMemcpyAsync(dev + g.dev, host + g.host, (size_t)g.len * 4, hipMemcpyHostToDevice, h->stream));
    return ssync(h);
}
int params_from_device(dril_sac_handle* h, float* host, const float* dev) {
    SDO(ssync(h));
    Seg s[4]; param_segs(h, s);
    for (const Seg& g : s) SHIP(h, hipMemcpy(host + g.host, dev + g.dev, (size_t)g.len * 4, hipMemcpyDeviceToHost));
    return DRIL_OK;
}

}  // namespace

// =================================================================================================================
// exported entry points (include/dril_sac.h)
// =================================================================================================================
DRIL_EXPORT int32_t dril_sac_config_default(dril_sac_config* c, int32_t env_kind) {
    if (!c || (env_kind != DRIL_ENV_PENDULUM && env_kind != DRIL_ENV_PENDULUM_SCALED && env_kind != DRIL_ENV_MOUNTAINCAR_CONTINUOUS && env_kind != DRIL_ENV_MOUNTAINCAR_CONTINUOUS_SCALED && env_kind != DRIL_ENV_EXTERNAL)) return sfail(nullptr, DRIL_ERR_INVALID_ARG, "SAC needs a Box action space (sac.jl:74): env_kind must be DRIL_ENV_PENDULUM[_SCALED] or DRIL_ENV_EXTERNAL");
    memset(c, 0, sizeof(*c));
    c->abi_version = DRIL_SAC_ABI_VERSION; c->env_kind = env_kind; c->n_envs = 1; c->episode_len = (env_kind == DRIL_ENV_MOUNTAINCAR_CONTINUOUS || env_kind == DRIL_ENV_MOUNTAINCAR_CONTINUOUS_SCALED) ? 999 : 200;
    c->hidden1 = 512; c->hidden2 = 512; c->activation = 1;
    c->buffer_capacity = 1000000; c->start_steps = 100; c->batch_size = 256; c->tau = 0.005f; c->gamma = 0.99f;
    c->train_freq = 1; c->gradient_steps = 1; c->target_update_interval = 1;
    c->auto_ent_coef = 1; c->ent_coef_init = 1.0f; c->auto_target_entropy = 1; c->target_entropy = 0.0f;
    c->learning_rate = 3.0e-4f; c->adam_beta1 = 0.9f; c->adam_beta2 = 0.999f; c->adam_eps = 1.0e-8f;
    c->seed = 42;
    return DRIL_OK;
}

DRIL_EXPORT int32_t dril_sac_destroy(dril_sac_handle* h) {
    if (!h) return DRIL_OK;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) hipStreamSynchronize(h->stream);
    void* ptrs[] = {h->params, h->adam_m, h->adam_v, h->g_critic, h->g_actor, h->sc, h->sc_next, h->stats, h->stats_out, h->ssq_rows, h->counter, h->head_partials, h->head_counter,
                    h->state, h->step_count, h->episode, h->gstep, h->disc_returns, h->obs_cur, h->obs_nxt, h->e_rew, h->e_tobs, h->e_raw, h->e_envact, h->e_term, h->e_trunc,
                    h->rb_obs, h->rb_next, h->rb_act, h->rb_rew, h->rb_term, h->rb_trunc, h->xa, h->ah1, h->ah2, h->mu, h->xq, h->xq_pi,
                    h->qh1, h->qh2, h->q_cur, h->q_pi, h->dq, h->dz2, h->dz1, h->dxq, h->dmu, h->b_rew, h->b_ne, h->b_nn, h->b_np,
                    h->b_nlp, h->a_pi, h->g_pi, h->lp_pi, h->b_term, h->collect_noise, h->inj_idx, h->inj_ne, h->inj_nn, h->inj_np, h->s_in, h->s_act, h->s_noise, h->s_out, h->s_out2};
    for (void* p : ptrs) if (p) hipFree(p);
    if (h->ev_a) hipEventDestroy(h->ev_a); if (h->ev_b) hipEventDestroy(h->ev_b);
    for (hipEvent_t e : h->it_events) hipEventDestroy(e);
    if (h->it_stamps) hipFree(h->it_stamps);
    if (h->col_h1p) hipFree(h->col_h1p); if (h->col_w2p) hipFree(h->col_w2p); if (h->col_flags) hipFree(h->col_flags);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
    return DRIL_OK;
}

DRIL_EXPORT int32_t dril_sac_create(const dril_sac_config* cfg, dril_sac_handle** out) {
    if (!cfg || !out) return sfail(nullptr, DRIL_ERR_INVALID_ARG, "null config / out pointer");
    if (cfg->abi_version != DRIL_SAC_ABI_VERSION) return sfail(nullptr, DRIL_ERR_INVALID_ARG, "dril_sac_config.abi_version mismatch");
    const bool ext = cfg->env_kind == DRIL_ENV_EXTERNAL;
    if (!ext && cfg->env_kind != DRIL_ENV_PENDULUM && cfg->env_kind != DRIL_ENV_PENDULUM_SCALED && cfg->env_kind != DRIL_ENV_MOUNTAINCAR_CONTINUOUS && cfg->env_kind != DRIL_ENV_MOUNTAINCAR_CONTINUOUS_SCALED) return sfail(nullptr, DRIL_ERR_UNSUPPORTED, "SAC needs a Box action space (sac.jl:74): DRIL_ENV_PENDULUM[_SCALED] or DRIL_ENV_EXTERNAL");
    if (ext && (cfg->ext_obs_dim < 1 || cfg->ext_obs_dim > 1024 || cfg->ext_action_dim < 1 || cfg->ext_action_dim > kMaxA || !(cfg->ext_action_low < cfg->ext_action_high)))
        return sfail(nullptr, DRIL_ERR_INVALID_ARG, "DRIL_ENV_EXTERNAL: ext_obs_dim 1..1024, ext_action_dim 1..16, ext_action_low < ext_action_high");
    if (cfg->n_envs <= 0 || (!ext && cfg->episode_len <= 0) || cfg->batch_size <= 0 || cfg->buffer_capacity < cfg->n_envs || cfg->train_freq <= 0 || cfg->target_update_interval <= 0)
        return sfail(nullptr, DRIL_ERR_INVALID_ARG, "n_envs, episode_len, batch_size, train_freq, target_update_interval must be positive and buffer_capacity >= n_envs");
    if (cfg->hidden1 <= 0 || cfg->hidden2 <= 0 || cfg->hidden1 % 4 || cfg->hidden2 % 4) return sfail(nullptr, DRIL_ERR_UNSUPPORTED, "hidden dims must be positive multiples of 4");
    if (cfg->activation != 0 && cfg->activation != 1) return sfail(nullptr, DRIL_ERR_INVALID_ARG, "activation: 0 tanh, 1 relu");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return sfail(nullptr, DRIL_ERR_HIP, "no HIP device: libdril_hip has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return sfail(nullptr, DRIL_ERR_INVALID_ARG, "device ordinal out of range");
    dril_sac_handle* h = new dril_sac_handle(); h->cfg = *cfg;
#define CHK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { std::string m = std::string(#expr) + ": " + hipGetErrorString(_e); dril_sac_destroy(h); return sfail(nullptr, DRIL_ERR_HIP, m); } } while (0)
    CHK(hipSetDevice(cfg->device));
    CHK(hipStreamCreate(&h->stream));
    { int khz = 0; if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->cfg.device) == hipSuccess && khz > 0) h->wall_hz = 1e3 * (double)khz; }   // wall_clock64 ticks per second (phase_stamp)
    const int D = h->D = ext ? cfg->ext_obs_dim : ((cfg->env_kind == DRIL_ENV_MOUNTAINCAR_CONTINUOUS || cfg->env_kind == DRIL_ENV_MOUNTAINCAR_CONTINUOUS_SCALED) ? 2 : 3), A = h->A = ext ? cfg->ext_action_dim : 1, S = h->S = ext ? 0 : 2, H1 = h->H1 = cfg->hidden1, H2 = h->H2 = cfg->hidden2, E = cfg->n_envs, B = cfg->batch_size, W = D + A;
    h->Pa = D * H1 + H1 + H1 * H2 + H2 + H2 * A + A; h->Pq = W * H1 + H1 + H1 * H2 + H2 + H2 + 1; h->P = h->Pa + 2 * h->Pq + A;
    h->actor = net_off(0, D, H1, H2, A); h->Pqd = round4(h->Pq); h->q0 = net_off(round4(h->actor.end), W, H1, H2, 1);
    h->log_std_off = h->q0.w1 + 4 * h->Pqd; h->Pd = round4(h->log_std_off + A);      // device layout: actor | q1 | q2 | target q1 | target q2 | log_std
    h->nq = B; h->nmax = std::max(E, 2 * B);
    h->target_entropy = cfg->auto_target_entropy ? -(float)A : cfg->target_entropy;
    h->act_hi = (cfg->env_kind == DRIL_ENV_PENDULUM_SCALED || cfg->env_kind == DRIL_ENV_MOUNTAINCAR_CONTINUOUS || cfg->env_kind == DRIL_ENV_MOUNTAINCAR_CONTINUOUS_SCALED) ? 1.0f : 2.0f; h->act_lo = -h->act_hi; h->external = ext;
    if (ext) { h->act_lo = cfg->ext_action_low; h->act_hi = cfg->ext_action_high; }
    CHK(smalloc(&h->params, h->Pd)); CHK(smalloc(&h->adam_m, h->Pd)); CHK(smalloc(&h->adam_v, h->Pd));
    h->target = h->params + h->q0.w1 + 2 * h->Pqd;   // the targets sit right behind the critics so that one launch runs all four Q nets (blockIdx.z stride Pqd)
    CHK(smalloc(&h->g_critic, h->Pd)); CHK(smalloc(&h->g_actor, h->Pd)); CHK(smalloc(&h->sc, 1)); CHK(smalloc(&h->sc_next, 1)); CHK(smalloc(&h->stats, 8));
    h->adam_blocks_c = std::min(kSacAdamBlocks, (2 * h->Pqd + 255) / 256); h->end_blocks = std::min(kSacAdamBlocks, ((h->actor.end + 3) / 4 + 2 * h->Pqd / 4 + 255) / 256);   // elementwise optimiser kernels: up to 1 024 blocks (no grid-wide fold is left in them; 256 -> 1 024: update! 0.174 -> 0.168 ms)
    // narrow first layers: their gradient, step and (critics) re-evaluation inside the optimiser kernels (first_layer_opt_block; DRIL_SAC_NO_FUSED_DW1=1: launches of their own)
    h->fused_dw1 = std::getenv("DRIL_SAC_NO_FUSED_HEADS") == nullptr && std::getenv("DRIL_SAC_NO_FUSED_DW1") == nullptr && W <= kOptL1MaxIn && H1 % 4 == 0 && B <= kOptL1MaxB;
    if (h->fused_dw1) { h->fl_a = (H1 + kOptL1Units - 1) / kOptL1Units; h->fl_c = 2 * h->fl_a; h->adam_blocks_c += h->fl_c; h->end_blocks += h->fl_a; }
    CHK(smalloc(&h->counter, 1));
    CHK(smalloc(&h->head_partials, (size_t)(2 * kMaxA + 8) * ((B + kHeadSamplesPerBlock - 1) / kHeadSamplesPerBlock + 1))); CHK(smalloc(&h->head_counter, kMaxA + 1));   // doubles: [critic 2 | actor 2 | log_std kMaxA | entropy 2] x blocks
    h->fused_heads = std::getenv("DRIL_SAC_NO_FUSED_HEADS") == nullptr; h->fused_collect = std::getenv("DRIL_SAC_NO_FUSED_COLLECT") == nullptr; h->fused_fwd = std::getenv("DRIL_SAC_NO_FUSED_FWD") == nullptr; h->f16_fwd = std::getenv("DRIL_SAC_NO_F16_FWD") == nullptr; h->trace_enqueue = false;   // latched here: no getenv on the update path
    CHK(smalloc(&h->state, (size_t)E * S)); CHK(smalloc(&h->step_count, E)); CHK(smalloc(&h->episode, E)); CHK(smalloc(&h->gstep, E)); CHK(smalloc(&h->disc_returns, E));
    CHK(smalloc(&h->obs_cur, (size_t)E * D)); CHK(smalloc(&h->obs_nxt, (size_t)E * D)); CHK(smalloc(&h->e_rew, E)); CHK(smalloc(&h->e_tobs, (size_t)E * D));
    CHK(smalloc(&h->e_raw, (size_t)E * A)); CHK(smalloc(&h->e_envact, (size_t)E * A)); CHK(smalloc(&h->e_term, E)); CHK(smalloc(&h->e_trunc, E));
    h->cap = cfg->buffer_capacity;
    CHK(smalloc(&h->rb_obs, (size_t)h->cap * D)); CHK(smalloc(&h->rb_next, (size_t)h->cap * D)); CHK(smalloc(&h->rb_act, (size_t)h->cap * A));
    CHK(smalloc(&h->rb_rew, (size_t)h->cap)); CHK(smalloc(&h->rb_term, (size_t)h->cap)); CHK(smalloc(&h->rb_trunc, (size_t)h->cap));
    const size_t nm = h->nmax, nq = h->nq;
    if (H1 % 2 == 0) { CHK(smalloc(&h->col_h1p, (size_t)2 * nm * (H1 / 2))); CHK(smalloc(&h->col_w2p, (size_t)2 * H2 * (H1 / 2))); CHK(smalloc(&h->col_flags, 2)); }   // f16 planes of h1 [2][nm][H1/2] and of the actor's W2 [2][H2][H1/2]; their range flags
    CHK(smalloc(&h->xa, nm * D)); CHK(smalloc(&h->ah1, nm * H1)); CHK(smalloc(&h->ah2, nm * H2)); CHK(smalloc(&h->mu, nm * A));
    CHK(smalloc(&h->xq, 2 * nq * W)); h->xq_next = h->xq + nq * W; CHK(smalloc(&h->xq_pi, nq * W));       // [xq | xq_next]: one input buffer, batch index z / 2
    CHK(smalloc(&h->qh1, 4 * nq * H1)); CHK(smalloc(&h->qh2, 4 * nq * H2)); h->th1 = h->qh1 + 2 * nq * H1; h->th2 = h->qh2 + 2 * nq * H2;   // z = 0,1 critics (kept for the reverse pass), 2,3 targets
    CHK(smalloc(&h->q_cur, 4 * nq)); h->q_next = h->q_cur + 2 * nq; CHK(smalloc(&h->q_pi, 2 * nq)); CHK(smalloc(&h->dq, 2 * nq));
    CHK(smalloc(&h->dz2, 2 * nq * H2)); CHK(smalloc(&h->dz1, 2 * nq * H1)); CHK(smalloc(&h->dxq, 2 * nq * W)); CHK(smalloc(&h->dmu, nq * A));
    CHK(smalloc(&h->b_rew, nq)); CHK(smalloc(&h->b_ne, nq * A)); CHK(smalloc(&h->b_nn, nq * A)); CHK(smalloc(&h->b_np, nq * A)); CHK(smalloc(&h->b_nlp, nq));
    CHK(smalloc(&h->a_pi, nq * A)); CHK(smalloc(&h->g_pi, nq * A)); CHK(smalloc(&h->lp_pi, nq)); CHK(smalloc(&h->b_term, nq));
    CHK(smalloc(&h->s_in, nm * D)); CHK(smalloc(&h->s_act, nm * A)); CHK(smalloc(&h->s_noise, nm * A)); CHK(smalloc(&h->s_out, nm * A)); CHK(smalloc(&h->s_out2, nm * A));
    CHK(hipEventCreate(&h->ev_a)); CHK(hipEventCreate(&h->ev_b));
#undef CHK
    const SacScalars sc0{logf(cfg->ent_coef_init), 0.f, 0.f, cfg->ent_coef_init};                   // init_entropy_coefficient sac.jl:207-213
    if (hipMemcpy(h->sc, &sc0, sizeof(sc0), hipMemcpyHostToDevice) != hipSuccess) { dril_sac_destroy(h); return sfail(nullptr, DRIL_ERR_HIP, "hipMemcpy(log_ent_coef)"); }
    reset_optimizer(h);
    *out = h;
    return DRIL_OK;
}
DRIL_EXPORT const char* dril_sac_last_error(const dril_sac_handle* h) { return h ? h->err.c_str() : g_sac_create_error.c_str(); }
DRIL_EXPORT int32_t dril_sac_obs_dim(const dril_sac_handle* h) { return h ? h->D : 0; }
DRIL_EXPORT int32_t dril_sac_action_dim(const dril_sac_handle* h) { return h ? h->A : 0; }
DRIL_EXPORT int64_t dril_sac_param_count(const dril_sac_handle* h) { return h ? h->P : 0; }
DRIL_EXPORT int64_t dril_sac_q_param_count(const dril_sac_handle* h) { return h ? h->Pq : 0; }

DRIL_EXPORT int32_t dril_sac_set_params(dril_sac_handle* h, const float* flat, size_t n) {
    SNEED(h); if (!flat || n != (size_t)h->P) return sfail(h, DRIL_ERR_INVALID_ARG, "dril_sac_set_params: n must equal dril_sac_param_count");
    SDO(params_to_device(h, h->params, flat)); h->col_w2_dirty = true;
    SHIP(h, hipMemcpy(h->target, h->params + h->q0.w1, 2 * (size_t)h->Pqd * 4, hipMemcpyDeviceToDevice));   // copy_critic_parameters sac.jl:172,191-197
    return DRIL_OK;
}
DRIL_EXPORT int32_t dril_sac_get_params(dril_sac_handle* h, float* flat, size_t n) {
    SNEED(h); if (!flat || n != (size_t)h->P) return sfail(h, DRIL_ERR_INVALID_ARG, "dril_sac_get_params: n must equal dril_sac_param_count");
    return params_from_device(h, flat, h->params);
}
DRIL_EXPORT int32_t dril_sac_get_target_params(dril_sac_handle* h, float* flat, size_t n) {
    SNEED(h); if (!flat || n != 2 * (size_t)h->Pq) return sfail(h, DRIL_ERR_INVALID_ARG, "target parameters: n must equal 2 * dril_sac_q_param_count");
    SDO(ssync(h));
    for (int k = 0; k < 2; ++k) SHIP(h, hipMemcpy(flat + (size_t)k * h->Pq, h->target + (size_t)k * h->Pqd, (size_t)h->Pq * 4, hipMemcpyDeviceToHost));
    return DRIL_OK;
}
DRIL_EXPORT int32_t dril_sac_set_target_params(dril_sac_handle* h, const float* flat, size_t n) {
    SNEED(h); if (!flat || n != 2 * (size_t)h->Pq) return sfail(h, DRIL_ERR_INVALID_ARG, "target parameters: n must equal 2 * dril_sac_q_param_count");
    SDO(ssync(h));
    for (int k = 0; k < 2; ++k) SHIP(h, hipMemcpy(h->target + (size_t)k * h->Pqd, flat + (size_t)k * h->Pq, (size_t)h->Pq * 4, hipMemcpyHostToDevice));
    return DRIL_OK;
}
DRIL_EXPORT int32_t dril_sac_get_log_ent_coef(dril_sac_handle* h, float* v) {
    SNEED(h); if (!v) return sfail(h, DRIL_ERR_INVALID_ARG, "null out pointer");
    SDO(ssync(h)); SacScalars sc; SHIP(h, hipMemcpy(&sc, h->sc, sizeof(sc), hipMemcpyDeviceToHost)); *v = sc.log_ent; return DRIL_OK;
}
DRIL_EXPORT int32_t dril_sac_set_log_ent_coef(dril_sac_handle* h, float v) {
    SNEED(h); SDO(ssync(h)); SacScalars sc; SHIP(h, hipMemcpy(&sc, h->sc, sizeof(sc), hipMemcpyDeviceToHost));
    sc.log_ent = v; sc.alpha = expf(v); SHIP(h, hipMemcpy(h->sc, &sc, sizeof(sc), hipMemcpyHostToDevice)); return DRIL_OK;
}
DRIL_EXPORT int32_t dril_sac_reset_optimizer(dril_sac_handle* h) {
    SNEED(h); SDO(ssync(h));
    SHIP(h, hipMemset(h->adam_m, 0, (size_t)h->Pd * 4)); SHIP(h, hipMemset(h->adam_v, 0, (size_t)h->Pd * 4));
    SacScalars sc; SHIP(h, hipMemcpy(&sc, h->sc, sizeof(sc), hipMemcpyDeviceToHost)); sc.ent_m = sc.ent_v = 0.f;
    SHIP(h, hipMemcpy(h->sc, &sc, sizeof(sc), hipMemcpyHostToDevice));
    reset_optimizer(h); return DRIL_OK;
}

DRIL_EXPORT int32_t dril_sac_env_reset(dril_sac_handle* h, uint64_t seed) {
    SNEED(h); S_NOT_EXTERNAL(h, "dril_sac_env_reset");
    h->env_seed0 = seed;
    SHIP(h, launch_env_reset(h->cfg.env_kind, h->cfg.n_envs, seed, h->state, h->step_count, h->episode, h->gstep, h->disc_returns, h->stream));
    h->env_ready = true; h->obs_valid = false;
    return ssync(h);
}
DRIL_EXPORT int32_t dril_sac_env_observe(dril_sac_handle* h, float* host_obs) {
    SNEED(h); S_NOT_EXTERNAL(h, "dril_sac_env_observe"); if (!host_obs) return sfail(h, DRIL_ERR_INVALID_ARG, "null observation buffer");
    SDO(ensure_obs(h)); SDO(ssync(h));
    SHIP(h, hipMemcpy(host_obs, h->obs_cur, (size_t)h->cfg.n_envs * h->D * 4, hipMemcpyDeviceToHost));
    return DRIL_OK;
}

namespace {
// actor means of a host batch chunk already staged in h->xa, then the squashed sample / mode
int actor_chunk(dril_sac_handle* h, const float* obs, const float* noise, int n, int deterministic, int64_t chunk_id) {
    SHIP(h, hipMemcpyAsync(h->xa, obs, (size_t)n * h->D * 4, hipMemcpyHostToDevice, h->stream));
    if (!deterministic) {
        if (noise) SHIP(h, hipMemcpyAsync(h->s_noise, noise, (size_t)n * h->A * 4, hipMemcpyHostToDevice, h->stream));
        else {
            const SacRng rng{h->cfg.seed ^ 0x0b5e55edull, h->aux_counter++};
            hipLaunchKernelGGL(sac_noise_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, n, h->A, rng, h->s_noise);
        }
    }
    (void)chunk_id;
    return net_forward(h, h->params, h->actor, 0, h->D, h->A, h->xa, h->D, 0, n, actor_bufs(h), 1);
}
}  // namespace

DRIL_EXPORT int32_t dril_sac_action_log_prob(dril_sac_handle* h, const float* obs, int64_t batch, const float* noise, float* actions, float* logp) {
    SNEED(h); if (!obs || batch <= 0) return sfail(h, DRIL_ERR_INVALID_ARG, "dril_sac_action_log_prob: obs / batch");
    for (int64_t o = 0; o < batch; o += h->nmax) {
        const int n = (int)std::min<int64_t>(h->nmax, batch - o);
        SDO(actor_chunk(h, obs + o * h->D, noise ? noise + o * h->A : nullptr, n, 0, o));
        hipLaunchKernelGGL(sac_squash_eval_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, n, h->A, h->mu, h->params + h->log_std_off, h->s_noise, 0,
                           h->act_lo, h->act_hi, h->s_out, h->s_out2, (float*)nullptr);
        SDO(ssync(h));
        if (actions) SHIP(h, hipMemcpy(actions + o * h->A, h->s_out, (size_t)n * h->A * 4, hipMemcpyDeviceToHost));
        if (logp) SHIP(h, hipMemcpy(logp + o, h->s_out2, (size_t)n * 4, hipMemcpyDeviceToHost));
    }
    return DRIL_OK;
}
DRIL_EXPORT int32_t dril_sac_predict_actions(dril_sac_handle* h, const float* obs, int64_t batch, int32_t deterministic, const float* noise, float* raw, float* env) {
    SNEED(h); if (!obs || batch <= 0) return sfail(h, DRIL_ERR_INVALID_ARG, "dril_sac_predict_actions: obs / batch");
    for (int64_t o = 0; o < batch; o += h->nmax) {
        const int n = (int)std::min<int64_t>(h->nmax, batch - o);
        SDO(actor_chunk(h, obs + o * h->D, noise ? noise + o * h->A : nullptr, n, deterministic, o));
        hipLaunchKernelGGL(sac_squash_eval_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, n, h->A, h->mu, h->params + h->log_std_off, h->s_noise, deterministic,
                           h->act_lo, h->act_hi, h->s_out, (float*)nullptr, h->s_out2);
        SDO(ssync(h));
        if (raw) SHIP(h, hipMemcpy(raw + o * h->A, h->s_out, (size_t)n * h->A * 4, hipMemcpyDeviceToHost));
        if (env) SHIP(h, hipMemcpy(env + o * h->A, h->s_out2, (size_t)n * h->A * 4, hipMemcpyDeviceToHost));
    }
    return DRIL_OK;
}
DRIL_EXPORT int32_t dril_sac_predict_q(dril_sac_handle* h, const float* obs, const float* actions, int64_t batch, int32_t use_target, float* q) {
    SNEED(h); if (!obs || !actions || !q || batch <= 0) return sfail(h, DRIL_ERR_INVALID_ARG, "dril_sac_predict_q: null pointer / empty batch");
    const int W = h->D + h->A;
    std::vector<float> tmp(2 * (size_t)h->nq);
    for (int64_t o = 0; o < batch; o += h->nq) {
        const int n = (int)std::min<int64_t>(h->nq, batch - o);
        SHIP(h, hipMemcpyAsync(h->s_in, obs + o * h->D, (size_t)n * h->D * 4, hipMemcpyHostToDevice, h->stream));
        SHIP(h, hipMemcpyAsync(h->s_act, actions + o * h->A, (size_t)n * h->A * 4, hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(sac_concat_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, n, h->D, h->A, h->s_in, h->s_act, h->xq_next);
        if (use_target) SDO(net_forward(h, h->target, net_off(0, W, h->H1, h->H2, 1), h->Pqd, W, 1, h->xq_next, W, 0, n, q_bufs(h, h->th1, h->th2, h->q_next), 2));
        else SDO(net_forward(h, h->params, h->q0, h->Pqd, W, 1, h->xq_next, W, 0, n, q_bufs(h, h->th1, h->th2, h->q_next), 2));
        SDO(ssync(h));
        SHIP(h, hipMemcpy(tmp.data(), h->q_next, tmp.size() * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) { q[(o + i) * 2] = tmp[i]; q[(o + i) * 2 + 1] = tmp[(size_t)h->nq + i]; }       // vcat of the critics: (2 x B)
    }
    return DRIL_OK;
}

DRIL_EXPORT int32_t dril_sac_debug_set_collect_noise(dril_sac_handle* h, const float* noise, size_t count) {
    SNEED(h); SDO(ssync(h));
    if (h->collect_noise) { hipFree(h->collect_noise); h->collect_noise = nullptr; h->collect_noise_count = 0; }
    if (!noise || !count) return DRIL_OK;
    SHIP(h, smalloc(&h->collect_noise, count)); SHIP(h, hipMemcpy(h->collect_noise, noise, count * 4, hipMemcpyHostToDevice)); h->collect_noise_count = count;
    return DRIL_OK;
}
DRIL_EXPORT int32_t dril_sac_collect_rollout(dril_sac_handle* h, int32_t n_steps, int32_t use_random_actions, double* fps) {
    SNEED(h); S_NOT_EXTERNAL(h, "dril_sac_collect_rollout"); return collect(h, n_steps, use_random_actions != 0, fps);
}
// one env step of the caller's host envs into the ring: the same push kernel as the device collection (truncated envs store their terminal observation)
DRIL_EXPORT int32_t dril_sac_ext_push(dril_sac_handle* h, const float* obs, const float* stored_actions, const float* rewards, const uint8_t* terminated,
                                      const uint8_t* truncated, const float* next_obs, const float* terminal_obs) {
    SNEED(h);
    if (!h->external) return sfail(h, DRIL_ERR_UNSUPPORTED, "dril_sac_ext_push: the handle was not created with DRIL_ENV_EXTERNAL");
    if (!obs || !stored_actions || !rewards || !terminated || !truncated || !next_obs) return sfail(h, DRIL_ERR_INVALID_ARG, "dril_sac_ext_push: null pointer");
    const size_t E = h->cfg.n_envs, D = h->D, A = h->A;
    bool any_trunc = false; for (size_t e = 0; e < E; ++e) any_trunc = any_trunc || truncated[e] != 0;
    if (any_trunc && !terminal_obs) return sfail(h, DRIL_ERR_INVALID_ARG, "dril_sac_ext_push: truncated envs need terminal_obs");
    SHIP(h, hipMemcpyAsync(h->obs_cur, obs, E * D * 4, hipMemcpyHostToDevice, h->stream));
    SHIP(h, hipMemcpyAsync(h->obs_nxt, next_obs, E * D * 4, hipMemcpyHostToDevice, h->stream));
    SHIP(h, hipMemcpyAsync(h->e_raw, stored_actions, E * A * 4, hipMemcpyHostToDevice, h->stream));
    SHIP(h, hipMemcpyAsync(h->e_rew, rewards, E * 4, hipMemcpyHostToDevice, h->stream));
    SHIP(h, hipMemcpyAsync(h->e_term, terminated, E, hipMemcpyHostToDevice, h->stream));
    SHIP(h, hipMemcpyAsync(h->e_trunc, truncated, E, hipMemcpyHostToDevice, h->stream));
    if (any_trunc) SHIP(h, hipMemcpyAsync(h->e_tobs, terminal_obs, E * D * 4, hipMemcpyHostToDevice, h->stream));
    const long long tail = (h->head + h->size) % h->cap;
    PushArgs pa{(int)E, (int)D, (int)A, h->cap, tail, h->obs_cur, h->e_raw, h->e_rew, h->e_tobs, h->obs_nxt, h->e_term, h->e_trunc,
                h->rb_obs, h->rb_next, h->rb_act, h->rb_rew, h->rb_term, h->rb_trunc};
    hipLaunchKernelGGL(sac_push_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, h->stream, pa);
    SHIP(h, hipGetLastError());
    const long long over = h->size + (long long)E - h->cap;                                       // CircularBuffer: overwrite the oldest
    if (over > 0) { h->head = (h->head + over) % h->cap; h->size = h->cap; } else h->size += (long long)E;
    return ssync(h);                                                                              // the caller's arrays are pageable host memory: drain before returning
}

DRIL_EXPORT int64_t dril_sac_replay_size(const dril_sac_handle* h) { return h ? h->size : 0; }
DRIL_EXPORT int64_t dril_sac_replay_capacity(const dril_sac_handle* h) { return h ? h->cap : 0; }
DRIL_EXPORT int32_t dril_sac_replay_copy_out(dril_sac_handle* h, int32_t which, void* host, size_t bytes) {
    SNEED(h);
    size_t w; const char* src;
    switch (which) {
        case DRIL_RB_OBSERVATIONS: w = (size_t)h->D * 4; src = (const char*)h->rb_obs; break;
        case DRIL_RB_NEXT_OBSERVATIONS: w = (size_t)h->D * 4; src = (const char*)h->rb_next; break;
        case DRIL_RB_ACTIONS: w = (size_t)h->A * 4; src = (const char*)h->rb_act; break;
        case DRIL_RB_REWARDS: w = 4; src = (const char*)h->rb_rew; break;
        case DRIL_RB_TERMINATED: w = 1; src = (const char*)h->rb_term; break;
        case DRIL_RB_TRUNCATED: w = 1; src = (const char*)h->rb_trunc; break;
        default: return sfail(h, DRIL_ERR_INVALID_ARG, "unknown replay field");
    }
    if (!host || bytes != w * (size_t)h->size) return sfail(h, DRIL_ERR_INVALID_ARG, "dril_sac_replay_copy_out: bytes must equal replay_size * element size");
    SDO(ssync(h));
    const long long first = std::min(h->size, h->cap - h->head);                                  // logical 0.. = [head, cap) then [0, ...)
    if (first > 0) SHIP(h, hipMemcpy(host, src + (size_t)h->head * w, (size_t)first * w, hipMemcpyDeviceToHost));
    if (h->size > first) SHIP(h, hipMemcpy((char*)host + (size_t)first * w, src, (size_t)(h->size - first) * w, hipMemcpyDeviceToHost));
    return DRIL_OK;
}
DRIL_EXPORT int32_t dril_sac_replay_fill(dril_sac_handle* h, int64_t count, const float* obs, const float* actions, const float* rewards,
                                         const uint8_t* terminated, const uint8_t* truncated, const float* next_obs) {
    SNEED(h); if (count <= 0 || !obs || !actions || !rewards || !terminated || !next_obs) return sfail(h, DRIL_ERR_INVALID_ARG, "dril_sac_replay_fill: null pointer / empty");
    SDO(ssync(h));
    const int64_t skip = count > h->cap ? count - h->cap : 0, n = count - skip;                   // pushes beyond the capacity overwrite the oldest
    SHIP(h, hipMemcpy(h->rb_obs, obs + skip * h->D, (size_t)n * h->D * 4, hipMemcpyHostToDevice));
    SHIP(h, hipMemcpy(h->rb_next, next_obs + skip * h->D, (size_t)n * h->D * 4, hipMemcpyHostToDevice));
    SHIP(h, hipMemcpy(h->rb_act, actions + skip * h->A, (size_t)n * h->A * 4, hipMemcpyHostToDevice));
    SHIP(h, hipMemcpy(h->rb_rew, rewards + skip, (size_t)n * 4, hipMemcpyHostToDevice));
    SHIP(h, hipMemcpy(h->rb_term, terminated + skip, (size_t)n, hipMemcpyHostToDevice));
    if (truncated) SHIP(h, hipMemcpy(h->rb_trunc, truncated + skip, (size_t)n, hipMemcpyHostToDevice)); else SHIP(h, hipMemset(h->rb_trunc, 0, (size_t)n));
    h->head = 0; h->size = n;
    return DRIL_OK;
}

DRIL_EXPORT int32_t dril_sac_debug_set_batches(dril_sac_handle* h, int32_t n_updates, const int64_t* idx, const float* ne, const float* nn, const float* np) {
    SNEED(h); SDO(ssync(h));
    clear_injected(h);
    if (n_updates <= 0) return DRIL_OK;
    const size_t nb = (size_t)n_updates * h->cfg.batch_size, na = nb * h->A;
    if (idx) {
        for (size_t i = 0; i < nb; ++i) if (idx[i] < 0 || idx[i] >= h->size) return sfail(h, DRIL_ERR_INVALID_ARG, "injected replay index out of range");
        SHIP(h, smalloc(&h->inj_idx, nb)); SHIP(h, hipMemcpy(h->inj_idx, idx, nb * 8, hipMemcpyHostToDevice));
    }
    if (ne) { SHIP(h, smalloc(&h->inj_ne, na)); SHIP(h, hipMemcpy(h->inj_ne, ne, na * 4, hipMemcpyHostToDevice)); }
    if (nn) { SHIP(h, smalloc(&h->inj_nn, na)); SHIP(h, hipMemcpy(h->inj_nn, nn, na * 4, hipMemcpyHostToDevice)); }
    if (np) { SHIP(h, smalloc(&h->inj_np, na)); SHIP(h, hipMemcpy(h->inj_np, np, na * 4, hipMemcpyHostToDevice)); }
    h->inj_updates = n_updates;
    return DRIL_OK;
}
DRIL_EXPORT int32_t dril_sac_update(dril_sac_handle* h, int32_t n_updates, dril_sac_stats* out) {
    SNEED(h); if (n_updates <= 0) return sfail(h, DRIL_ERR_INVALID_ARG, "n_updates must be positive");
    if (h->inj_updates && h->inj_updates != n_updates) return sfail(h, DRIL_ERR_INVALID_ARG, "injected batches were set for a different n_updates");
    const int rc = run_updates(h, n_updates, h->inj_updates != 0, out);
    clear_injected(h);
    return rc;
}
DRIL_EXPORT int32_t dril_sac_get_last_grads(dril_sac_handle* h, float* gc, float* ga, size_t n) {
    SNEED(h); if (n != (size_t)h->P) return sfail(h, DRIL_ERR_INVALID_ARG, "dril_sac_get_last_grads: n must equal dril_sac_param_count");
    if (gc) { memset(gc, 0, n * 4); SDO(ssync(h));
        for (int k = 0; k < 2; ++k) SHIP(h, hipMemcpy(gc + h->Pa + (size_t)k * h->Pq, h->g_critic + h->q0.w1 + (size_t)k * h->Pqd, (size_t)h->Pq * 4, hipMemcpyDeviceToHost)); }
    if (ga) { memset(ga, 0, n * 4); SDO(ssync(h));
        SHIP(h, hipMemcpy(ga, h->g_actor, (size_t)h->Pa * 4, hipMemcpyDeviceToHost));
        SHIP(h, hipMemcpy(ga + h->Pa + 2 * (size_t)h->Pq, h->g_actor + h->log_std_off, (size_t)h->A * 4, hipMemcpyDeviceToHost)); }
    return DRIL_OK;
}

DRIL_EXPORT int32_t dril_sac_train(dril_sac_handle* h, int64_t max_steps, dril_sac_stats* stats, int64_t stats_capacity, int64_t* n_updates_done,
                                   double* fps, int64_t fps_capacity, int32_t* iterations_done, int64_t* total_steps) {
    SNEED(h); S_NOT_EXTERNAL(h, "dril_sac_train");
    const int64_t E = h->cfg.n_envs, tf = h->cfg.train_freq;
    const int64_t total_start = h->cfg.start_steps > 0 ? h->cfg.start_steps : tf * E;            // sac.jl:436
    const int64_t adjusted = std::max<int64_t>(1, total_start / E) * E;                           // :437
    int64_t n_steps = adjusted / E;                                                               // :438
    const int64_t iterations = (max_steps - adjusted) / (tf * E) + 1;                             // :443 (div truncates toward zero, as in Julia)
    const int64_t total = n_steps * E + tf * E * (iterations - 1);                                // :445
    const int64_t n_upd = h->cfg.gradient_steps == -1 ? tf * E : h->cfg.gradient_steps;           // get_gradient_steps :59-65
    int64_t done = 0; int64_t it = 0;
    if (iterations > 0) {                                                                         // the first iteration: the start_steps collection (random actions), then its gradient steps
        double f = 0;
        SDO(collect(h, (int)n_steps, h->cfg.start_steps > 0, &f));                                // :485-489
        if (fps && fps_capacity > 0) fps[0] = f;
        if (n_upd > 0) {
            const int64_t room = stats ? std::max<int64_t>(0, std::min<int64_t>(n_upd, stats_capacity)) : 0;
            std::vector<dril_sac_stats> tmp((size_t)n_upd);
            SDO(run_updates(h, (int)n_upd, false, tmp.data()));                                   // :514-531
            for (int64_t k = 0; k < room; ++k) stats[k] = tmp[(size_t)k];
            done += n_upd;
        }
        it = 1;
    }
    while (it < iterations) {                                                                     // every later iteration collects train_freq steps (:511): chunks without a host sync inside
        const int cnt = (int)std::min<int64_t>(std::min<int64_t>(h->iter_chunk, std::max<int64_t>(1, 4096 / std::max<int64_t>(1, n_upd))), iterations - it);   // at most 4 096 statistics rows per drain (gradient_steps = -1 means train_freq x n_envs updates per iteration)
        SDO(run_iterations(h, cnt, (int)tf, (int)n_upd, stats ? stats + std::min<int64_t>(done, stats_capacity) : nullptr, stats ? std::max<int64_t>(0, stats_capacity - done) : 0,
                           fps && it < fps_capacity ? fps + it : nullptr, fps ? std::max<int64_t>(0, fps_capacity - it) : 0));
        done += (int64_t)cnt * n_upd; it += cnt;
    }
    if (n_updates_done) *n_updates_done = done;
    if (iterations_done) *iterations_done = (int32_t)std::max<int64_t>(0, it);
    if (total_steps) *total_steps = it > 0 ? total : 0;
    return DRIL_OK;
}

DRIL_EXPORT int32_t dril_sac_iterate(dril_sac_handle* h, int32_t iterations, dril_sac_stats* stats, int64_t stats_capacity, double* fps, int64_t fps_capacity) {
    SNEED(h); S_NOT_EXTERNAL(h, "dril_sac_iterate");
    if (iterations <= 0) return sfail(h, DRIL_ERR_INVALID_ARG, "iterations must be positive");
    if (!h->env_ready) return sfail(h, DRIL_ERR_NOT_INITIALISED, "dril_sac_env_reset has not been called");
    const int64_t E = h->cfg.n_envs, tf = h->cfg.train_freq;
    const int64_t n_upd = h->cfg.gradient_steps == -1 ? tf * E : h->cfg.gradient_steps;
    int64_t done = 0;
    for (int64_t it = 0; it < iterations; ) {
        const int cnt = (int)std::min<int64_t>(std::min<int64_t>(h->iter_chunk, std::max<int64_t>(1, 4096 / std::max<int64_t>(1, n_upd))), iterations - it);   // at most 4 096 statistics rows per drain (gradient_steps = -1 means train_freq x n_envs updates per iteration)
        SDO(run_iterations(h, cnt, (int)tf, (int)n_upd, stats ? stats + std::min<int64_t>(done, stats_capacity) : nullptr, stats ? std::max<int64_t>(0, stats_capacity - done) : 0,
                           fps && it < fps_capacity ? fps + it : nullptr, fps ? std::max<int64_t>(0, fps_capacity - it) : 0));
        done += (int64_t)cnt * n_upd; it += cnt;
    }
    return DRIL_OK;
}
DRIL_EXPORT int32_t dril_sac_profile_get(dril_sac_handle* h, double* collect_ms, int64_t* collect_steps, double* update_ms, int64_t* updates) {
    SNEED(h);
    if (collect_ms) *collect_ms = h->collect_ms; if (collect_steps) *collect_steps = h->collect_steps;
    if (update_ms) *update_ms = h->update_ms; if (updates) *updates = h->updates;
    return DRIL_OK;
}
DRIL_EXPORT int32_t dril_sac_profile_reset(dril_sac_handle* h) { SNEED(h); h->collect_ms = h->update_ms = 0; h->collect_steps = h->updates = 0; return DRIL_OK; }
