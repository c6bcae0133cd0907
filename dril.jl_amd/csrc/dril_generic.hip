// dril_generic.hip — the on-policy path for ANY observation / action / hidden width (DRIL_ENV_EXTERNAL: the caller's own host envs).
//
// The fused kernels of dril_kernels.hip are built per env kind (obs dim <= 4, one or two outputs, hidden 64 / 128 / 256).  Everything else
// runs here, layer by layer, on the strided fp32-MFMA contraction of dril_gemm.hip:
//   generic_policy    layer(obs, ps, st) / evaluate_actions / predict_values  (layer_forward.jl:3-13,30-39; layer_methods.jl:28-61)
//   generic_ppo_grad  (alg::PPO)(layer, ps, st, batch) ppo.jl:365-407 and its reverse pass (Zygote in the reference, ppo.jl:207)
// generic_ppo_grad leaves its result in the SAME slab format as ppo_grad_kernel (one slab per row chunk: [net gradient | log_std gradient |
// 8 statistic sums]), so grad_reduce_kernel, the RCCL all-reduce, the global-norm clip, the target_kl check and Adam are shared with the
// fused path.  Per-sample head math uses accurate libm (expf / logf / tanhf), like the oracle.
#include <algorithm>

#include "dril_gemm.h"
#include "dril_internal.h"

namespace dril {

namespace {

// the generic path's contractions may use the bf16-split form of the LDS-tiled kernel (dril_gemm.hip)
GemmArgs gargs() { GemmArgs g = gemm_args(); g.allow_split = 1; return g; }

constexpr int kMaxOut = 64;
constexpr float kLog2PiG = 1.8378770664093453f;

hipError_t ws_reserve(GenericWs& ws, size_t floats) {
    if (floats <= ws.cap) return hipSuccess;
    if (ws.p) { hipError_t e = hipFree(ws.p); ws.p = nullptr; ws.cap = 0; if (e != hipSuccess) return e; }   // hipFree drains the device first
    const size_t want = floats + floats / 4;
    hipError_t e = hipMalloc((void**)&ws.p, want * sizeof(float));
    if (e != hipSuccess) return e;
    ws.cap = want;
    return hipSuccess;
}
struct Carver { float* p; size_t used = 0; float* take(size_t n) { float* r = p + used; used += (n + 3) & ~(size_t)3; return r; } };   // 16-byte aligned pieces

// one net's layout inside the flat parameter vector: layer l (0 .. nh) has W at w[l] (out x in, column-major) and b at b[l]
struct NetLay { int nl; int in[kMaxHidden + 1], out[kMaxHidden + 1], w[kMaxHidden + 1], b[kMaxHidden + 1]; int end; };
NetLay net_lay(const GenericDims& d, int base, int O) {
    NetLay n; n.nl = d.nh + 1; int off = base;
    for (int l = 0; l < n.nl; ++l) {
        n.in[l] = l == 0 ? d.D : d.H[l - 1]; n.out[l] = l == d.nh ? O : d.H[l];
        n.w[l] = off; off += n.in[l] * n.out[l]; n.b[l] = off; off += n.out[l];
    }
    n.end = off;
    return n;
}
int hidden_sum(const GenericDims& d) { int s = 0; for (int l = 0; l < d.nh; ++l) s += d.H[l]; return s; }
// hidden activations: one 16-byte-aligned block per layer (hb[l]); inside a block net z follows net z - 1 (n * H[l] floats apart), one row per sample —
// the layout mlp_forward_both's batched launches write and the reverse pass reads
int act_epi(const GenericDims& d) { return epi_of_activation(d.act); }
int mask_epi(const GenericDims& d) { return mask_epi_of_activation(d.act); }

// out[n][O] = net(X[n][D]) with every hidden activation kept (layer l in hb[l]): Dense(in => h, act) ... Dense(h_nh => out), layer_helpers.jl:27-57
hipError_t mlp_forward(const GenericDims& d, const float* P, const NetLay& L, const float* X, int n, float* const* hb, float* out, hipStream_t s) {
    const float* in = X;
    for (int l = 0; l < L.nl; ++l) {
        float* dst = l == d.nh ? out : hb[l];
        GemmArgs g = gargs();                                                       // y = act(W x + b), Lux.Dense
        g.A = P + L.w[l]; g.sAm = 1; g.sAk = L.out[l]; g.B = in; g.sBk = 1; g.sBn = L.in[l]; g.C = dst; g.sCm = 1; g.sCn = L.out[l]; g.bias = P + L.b[l];
        g.M = L.out[l]; g.N = n; g.K = L.in[l]; g.epi = l == d.nh ? EPI_NONE : act_epi(d);
        hipError_t e = launch_gemm(g, 1, s); if (e != hipSuccess) return e;
        in = dst;
    }
    return hipSuccess;
}

// both nets of the ActorCriticLayer on the same rows: their hidden layers have the same shapes, so each is ONE launch with blockIdx.z = net
// (layer l's block of hb holds the actor's activations followed by the critic's, n * H[l] floats apart); the output layers differ in width and share a
// launch through the pair kernel while the batch is small.  Halves the launch count of a rollout step / small minibatch (latency-bound there).
hipError_t mlp_forward_both(const GenericDims& d, const float* P, const NetLay& La, const NetLay& Lc, const float* X, int n, float* const* hb, float* out, float* v, hipStream_t s, float* const* zb = nullptr) {   // zb: pre-activations beside the activations (gelu / swish: the reverse pass needs them)
    const long long zP = (long long)Lc.w[0] - La.w[0];                              // same layout in both nets up to the output layer
    const float* in = X; long long zin = 0;
    for (int l = 0; l < d.nh; ++l) {
        float* dst = hb[l];
        GemmArgs g = gargs();
        g.A = P + La.w[l]; g.sAm = 1; g.sAk = La.out[l]; g.zA = zP; g.B = in; g.sBk = 1; g.sBn = La.in[l]; g.zB = zin; g.C = dst; g.sCm = 1; g.sCn = La.out[l]; g.zC = (long long)n * La.out[l];
        g.bias = P + La.b[l]; g.zBias = zP; g.M = La.out[l]; g.N = n; g.K = La.in[l]; g.epi = act_epi(d); g.zout = zb ? zb[l] : nullptr;
        hipError_t e = launch_gemm(g, 2, s); if (e != hipSuccess) return e;
        in = dst; zin = (long long)n * La.out[l];
    }
    const int l = d.nh, Hl = La.in[l];
    GemmArgs a = gargs(), c = gargs();
    a.A = P + La.w[l]; a.sAm = 1; a.sAk = La.out[l]; a.B = in; a.sBk = 1; a.sBn = Hl; a.C = out; a.sCm = 1; a.sCn = La.out[l]; a.bias = P + La.b[l]; a.M = La.out[l]; a.N = n; a.K = Hl;
    c.A = P + Lc.w[l]; c.sAm = 1; c.sAk = 1; c.B = in + zin; c.sBk = 1; c.sBn = Hl; c.C = v; c.sCm = 1; c.sCn = 1; c.bias = P + Lc.b[l]; c.M = 1; c.N = n; c.K = Hl;
    if (n <= 8192) return launch_gemm_pair(a, 1, c, 1, s);
    hipError_t e = launch_gemm(a, 1, s); if (e != hipSuccess) return e;
    return launch_gemm(c, 1, s);
}

// ---- per-sample distribution math (runtime action width) ------------------------------------------------------------------------
// Lux.softmax statistics of one logit row: max and sum(exp(z - max)); p_i = exp(z_i - m) / s (layer_forward.jl:141-149)
__device__ inline void softmax_stats(const float* z, int A, float& m, float& s) {
    m = z[0]; for (int i = 1; i < A; ++i) m = fmaxf(m, z[i]);
    s = 0.f; for (int i = 0; i < A; ++i) s += expf(z[i] - m);
}
__device__ inline float categorical_entropy_rt(const float* z, int A, float m, float s) {        // -sum(p log p), categorical.jl:38-40
    float e = 0.f; for (int i = 0; i < A; ++i) { const float p = expf(z[i] - m) / s; e += p * logf(p); }
    return -e;
}
__device__ inline float gauss_logpdf_rt(const float* x, const float* mu, const float* ls, int A) {   // diagGaussian.jl:25-36
    float lss = 0.f, dss = 0.f;
    for (int i = 0; i < A; ++i) { lss += ls[i]; const float d = x[i] - mu[i]; dss += d * d * expf(-2.0f * ls[i]); }
    return -0.5f * (2.0f * lss + dss + (float)A * kLog2PiG);
}
__device__ inline float gauss_entropy_rt(const float* ls, int A) {                                 // diagGaussian.jl:38-43
    float lss = 0.f; for (int i = 0; i < A; ++i) lss += ls[i];
    return 0.5f * (float)A * (1.0f + kLog2PiG) + lss;
}

struct PolicyHeadArgs {
    PolicyArgs a; int A, discrete; int64_t r0, n;   // rows [r0, r0 + n) of the call's batch; out is chunk-local
    const float* out;
};
// mode 0: sample (or mode(d)) + logprob; mode 1: logprob + entropy of the given actions
__global__ void generic_policy_head_kernel(PolicyHeadArgs g) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.n) return;
    const PolicyArgs& a = g.a;
    const int A = g.A; const int64_t b = g.r0 + i;
    const float* z = g.out + i * A;
    if (g.discrete) {
        float m, s; softmax_stats(z, A, m, s);
        int act;
        if (a.mode == 0) {
            if (a.deterministic) {                                                   // mode(d) = argmax(p), first maximum (categorical.jl:42-44)
                act = 0; float best = expf(z[0] - m) / s;
                for (int k = 1; k < A; ++k) { const float p = expf(z[k] - m) / s; if (p > best) { best = p; act = k; } }
            } else {
                double u;
                if (a.noise) u = ((const double*)a.noise)[b];
                else if (a.gstep) { const uint64_t k = a.env_seed0 + (uint64_t)b; uint32_t r[4]; philox4x32_10((uint32_t)k, (uint32_t)(k >> 32), a.gstep[b], 0, 1, 0, r); u = u01_f64(r[0], r[1]); }   // device envs: the env-keyed stream of rollout_kernel
                else { uint32_t r[4]; philox4x32_10((uint32_t)a.seed, (uint32_t)(a.seed >> 32), (uint32_t)b, (uint32_t)(b >> 32), 3, a.call_counter, r); u = u01_f64(r[0], r[1]); }
                float cs = 0.f; act = A - 1;                                          // findfirst(cumsum(p) .>= u), categorical.jl:47-52
                for (int k = 0; k < A; ++k) { cs += expf(z[k] - m) / s; if ((double)cs >= u) { act = k; break; } }
            }
            ((int32_t*)a.actions)[b] = act + a.action_start;
        } else act = ((const int32_t*)a.actions)[b] - a.action_start;
        act = act < 0 ? 0 : (act >= A ? A - 1 : act);
        a.logp[b] = logf(expf(z[act] - m) / s);
        if (a.mode == 1 && a.entropy) a.entropy[b] = categorical_entropy_rt(z, A, m, s);
    } else {
        const float* ls = a.params + a.log_std_off;
        float* x = (float*)a.actions + b * A;
        if (a.mode == 0) {
            for (int k = 0; k < A; ++k) {
                float n01;
                if (a.noise) n01 = ((const float*)a.noise)[b * A + k];
                else if (a.gstep) { const uint64_t kk = a.env_seed0 + (uint64_t)b; uint32_t r[4]; philox4x32_10((uint32_t)kk, (uint32_t)(kk >> 32), a.gstep[b], 0, 1, (uint32_t)(k / 2), r); n01 = (k & 1) ? randn_f32(r[2], r[3]) : randn_f32(r[0], r[1]); }
                else { uint32_t r[4]; philox4x32_10((uint32_t)a.seed, (uint32_t)(a.seed >> 32), (uint32_t)b, (uint32_t)(b >> 32), 3 + 16 * (uint32_t)k, a.call_counter, r); n01 = randn_f32(r[0], r[1]); }
                x[k] = a.deterministic ? z[k] : z[k] + expf(ls[k]) * n01;                // diagGaussian.jl:13-17, mode(d) = mean :45-47
            }
        }
        a.logp[b] = gauss_logpdf_rt(x, z, ls, A);
        if (a.mode == 1 && a.entropy) a.entropy[b] = gauss_entropy_rt(ls, A);
    }
}
__global__ void generic_select_kernel(int64_t n, const uint8_t* where, const float* src, float* dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && where[i]) dst[i] = src[i];
}
__global__ void generic_copy_kernel(const float* src, float* dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// ---- minibatch gather (DataLoader, ppo.jl:188-195): rows [row0, row0 + R) of the minibatch's epoch order into dense row-major pieces ----
struct GatherArgs {
    GradArgs a; int D, A, discrete; int64_t row0, R;
    float* X; float* act; float* adv; float* lpo; float* ret; float* vold; float* valid;
};
__device__ inline int64_t sample_index(const GradArgs& a, int64_t row, bool& valid) {
    valid = row < a.count;
    const int64_t p = a.pos0 + (valid ? row : 0);
    const int64_t gi = a.perm ? a.perm[p] : (a.perm_bits ? perm_index(p, a.N, a.perm_key, a.perm_bits) : p);
    const int64_t li = gi - a.idx_lo;
    valid = valid && li >= 0 && li < a.n_local;
    return valid ? li : 0;
}
__global__ void generic_gather_kernel(GatherArgs g) {
    const int W = g.D + 1;                                                           // D obs columns + one "scalars" column per row
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= g.R * W) return;
    const int64_t i = e / W; const int c = (int)(e - i * W);
    bool valid; const int64_t idx = sample_index(g.a, g.row0 + i, valid);
    if (c < g.D) { g.X[i * g.D + c] = valid ? g.a.obs[idx * g.D + c] : 0.f; return; }
    g.valid[i] = valid ? 1.f : 0.f;
    g.adv[i] = valid ? g.a.adv[idx] : 0.f; g.lpo[i] = valid ? g.a.logp_old[idx] : 0.f; g.ret[i] = valid ? g.a.ret[idx] : 0.f;
    g.vold[i] = (valid && g.a.has_clip_vf) ? g.a.val_old[idx] : 0.f;
    if (g.discrete) reinterpret_cast<int32_t*>(g.act)[i] = valid ? ((const int32_t*)g.a.actions)[idx] : g.a.action_start;
    else for (int k = 0; k < g.A; ++k) g.act[i * g.A + k] = valid ? ((const float*)g.a.actions)[idx * g.A + k] : 0.f;
}

// ---- loss head (ppo.jl:377-404): dLoss/d(actor out), dLoss/dV per row; statistic sums and the log_std gradient per row chunk ----
struct LossHeadArgs {
    GradArgs a; int A, discrete; int64_t R, Cr; int slab0;   // R rows = G chunks of Cr rows; chunk z owns slab slab0 + z
    const float* out; const float* v; const float* act; const float* adv; const float* lpo; const float* ret; const float* vold; const float* valid;
    float* dout; float* dv; float* dlp;   // dlp: dLoss/dlogp per row (feeds the log_std gradient)
};
__global__ __launch_bounds__(256) void generic_loss_head_kernel(LossHeadArgs g) {
    __shared__ double sh[256];
    const GradArgs& a = g.a;
    const int A = g.A, z = blockIdx.x, tid = threadIdx.x;
    float adv_mean = 0.f, adv_inv = 1.f;
    if (a.normalize_adv) {                                                            // normalize!, ppo.jl:350-356 (corrected std, eps on the std)
        const double s = a.adv_stats[0], q = a.adv_stats[1], n = a.adv_stats[2];
        const double mean = s / n; double var = (q - s * mean) / (n - 1.0); if (var < 0) var = 0;
        adv_mean = (float)mean; adv_inv = 1.0f / ((float)sqrt(var) + 1.0e-8f);
    }
    const float* ls = a.params + a.log_std_off;
    double st[6] = {0, 0, 0, 0, 0, 0};
    for (int64_t i = (int64_t)z * g.Cr + tid; i < (int64_t)(z + 1) * g.Cr; i += blockDim.x) {
        const bool valid = g.valid[i] != 0.f;
        const float* o = g.out + i * A; float* d_o = g.dout + i * A;
        // value head
        const float R = g.ret[i], ov = g.vold[i], val = g.v[i];
        const float dcl = val - ov;
        const bool vpass = !a.has_clip_vf || (dcl >= -a.clip_range_vf && dcl <= a.clip_range_vf);                  // clip_range, ppo.jl:344-346,378
        const float value = a.has_clip_vf ? ov + fminf(fmaxf(dcl, -a.clip_range_vf), a.clip_range_vf) : val;
        const float ve = value - R;
        g.dv[i] = (valid && vpass) ? a.invB * a.vf_coef * 2.0f * ve : 0.f;
        // policy head
        const float advn = (g.adv[i] - adv_mean) * adv_inv;
        float logp, ent, m = 0.f, s = 1.f; int act = 0;
        if (g.discrete) {
            softmax_stats(o, A, m, s);
            act = reinterpret_cast<const int32_t*>(g.act)[i] - a.action_start; act = act < 0 ? 0 : (act >= A ? A - 1 : act);
            logp = logf(expf(o[act] - m) / s); ent = categorical_entropy_rt(o, A, m, s);
        } else { logp = gauss_logpdf_rt(g.act + i * A, o, ls, A); ent = gauss_entropy_rt(ls, A); }
        const float lr = logp - g.lpo[i];
        const float r = expf(lr);                                                     // :380
        const float lo = 1.0f - a.clip_range, hi = 1.0f + a.clip_range;
        const float rc = fminf(fmaxf(r, lo), hi);                                     // :381
        const float t1 = r * advn, t2 = rc * advn;
        const float mn = t2 < t1 ? t2 : t1;                                           // :382
        const float dm_dr = (t2 < t1) ? ((r >= lo && r <= hi) ? advn : 0.f) : advn;
        const float dlogp = valid ? -a.invB * dm_dr * r : 0.f;
        const float dent = valid ? -a.invB * a.ent_coef : 0.f;                        // ent_loss = -mean(entropy), :383,:386
        g.dlp[i] = dlogp;
        if (g.discrete) {
            for (int k = 0; k < A; ++k) { const float p = expf(o[k] - m) / s; d_o[k] = dlogp * ((k == act ? 1.0f : 0.0f) - p) + dent * (-p * (logf(p) + ent)); }
        } else {
            for (int k = 0; k < A; ++k) { const float iv = expf(-2.0f * ls[k]), d = g.act[i * A + k] - o[k]; d_o[k] = dlogp * d * iv; }
        }
        if (valid) { st[0] += -mn; st[1] += ent; st[2] += (r != rc) ? 1.0 : 0.0; st[3] += (double)((r - 1.0f) - lr); st[4] += r; st[5] += (double)(ve * ve); }   // :382,:383,:390,:393,:402,:385
    }
    float* slab_a = a.slabs_actor + (size_t)(g.slab0 + z) * a.slab_a; float* slab_c = a.slabs_critic + (size_t)(g.slab0 + z) * a.slab_c;
    for (int k = 0; k < 6; ++k) {
        sh[tid] = st[k]; __syncthreads();
        for (int w = 128; w > 0; w >>= 1) { if (tid < w) sh[tid] += sh[tid + w]; __syncthreads(); }
        if (tid == 0) { if (k < 5) slab_a[a.slab_a - 8 + k] = (float)sh[0]; else slab_c[a.slab_c - 8] = (float)sh[0]; }
        __syncthreads();
    }
    if (tid < 3) slab_a[a.slab_a - 3 + tid] = 0.f;
    if (tid >= 1 && tid < 8) slab_c[a.slab_c - 8 + tid] = 0.f;
    if (!g.discrete) {                                                                // dLoss/dlog_std_k = sum_rows dlogp (d^2 exp(-2 ls) - 1) + dent, one dim at a time
        const int Pa = a.actor.end - a.actor.w1;   // the actor net's parameter count (any depth): log_std gradients sit right behind it in the slab
        for (int k = 0; k < A; ++k) {
            const float iv = expf(-2.0f * ls[k]);
            double acc = 0;
            for (int64_t i = (int64_t)z * g.Cr + tid; i < (int64_t)(z + 1) * g.Cr; i += blockDim.x) {
                if (g.valid[i] == 0.f) continue;
                const float d = g.act[i * A + k] - g.out[i * A + k];
                const float dlogp = g.dlp[i], dent = -a.invB * a.ent_coef;
                acc += (double)(dlogp * (d * d * iv - 1.0f) + dent);
            }
            sh[tid] = acc; __syncthreads();
            for (int w = 128; w > 0; w >>= 1) { if (tid < w) sh[tid] += sh[tid + w]; __syncthreads(); }
            if (tid == 0) slab_a[Pa + k] = (float)sh[0];
            __syncthreads();
        }
    }
}

// reverse pass of one net over R = G * Cr rows: data gradients over all rows at once, parameter gradients per row chunk straight into the slabs.
// The 2 (nh + 1) - 1 contractions of a net, in dependency order: [dW_L|db_L], dz_{L-1}, [dW_{L-1}|db_{L-1}], ..., dz_1, [dW_1|db_1]
constexpr int kMaxStages = 2 * (kMaxHidden + 1) - 1;
struct BackwardPlan { GemmArgs g[kMaxStages]; int Z[kMaxStages]; int n; };
// hb: this net's hidden activations, layer l at hb[l] (R rows x H[l]); dz: scratch of the same shapes; dOut: R x O
BackwardPlan plan_backward(const GenericDims& d, const float* P, const NetLay& L, const float* X, const float* const* hb, float* const* dz, const float* dOut,
                           int64_t R, int Cr, int G, float* slabs, int slab_stride, const float* const* zb = nullptr) {   // zb: pre-activations (the mask's operand for gelu / swish)
    const int base = L.w[0];                                                         // slab offsets are relative to the net's first parameter
    BackwardPlan p; p.n = 0;
    const float* up = dOut;                                                          // gradient w.r.t. the pre-activation of layer l
    for (int l = d.nh; l >= 0; --l) {
        const int O = L.out[l], I = L.in[l];
        const float* xin = l == 0 ? X : hb[l - 1];
        GemmArgs w = gargs();                                                        // [dW_l | db_l] = up . [x_in' | 1]   (b sits right behind the column-major W)
        w.A = up; w.sAm = 1; w.sAk = O; w.zA = (long long)Cr * O; w.B = xin; w.sBk = I; w.sBn = 1; w.zB = (long long)Cr * I; w.ones_n = 1;
        w.C = slabs + (L.w[l] - base); w.sCm = 1; w.sCn = O; w.zC = slab_stride; w.M = O; w.N = I + 1; w.K = Cr;
        p.g[p.n] = w; p.Z[p.n] = G; ++p.n;
        if (l == 0) break;
        GemmArgs g = gargs();                                                        // dz_{l-1} = (W_l' up) .* act'(h_{l-1})
        g.A = P + L.w[l]; g.sAm = O; g.sAk = 1; g.B = up; g.sBk = 1; g.sBn = O; g.C = dz[l - 1]; g.sCm = 1; g.sCn = I; g.aux = zb ? zb[l - 1] : hb[l - 1]; g.M = I; g.N = (int)R; g.K = O; g.epi = mask_epi(d);
        p.g[p.n] = g; p.Z[p.n] = 1; ++p.n;
        up = dz[l - 1];
    }
    return p;
}
long long tiles_of(const GemmArgs& g, int Z) { return (long long)((g.M + 31) / 32) * ((g.N + 31) / 32) * Z; }
// both nets: stage i of the actor and stage i of the critic are independent, so while both are in the split-K regime (few output tiles) they share a
// launch through the pair kernel: half the launches for a small minibatch
hipError_t run_backward_both(const BackwardPlan& a, const BackwardPlan& c, hipStream_t s) {
    for (int i = 0; i < a.n; ++i) {
        hipError_t e;
        if (tiles_of(a.g[i], a.Z[i]) + tiles_of(c.g[i], c.Z[i]) < 2048) e = launch_gemm_pair(a.g[i], a.Z[i], c.g[i], c.Z[i], s);
        else { e = launch_gemm(a.g[i], a.Z[i], s); if (e == hipSuccess) e = launch_gemm(c.g[i], c.Z[i], s); }
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace

hipError_t generic_select(int64_t n, const uint8_t* where, const float* src, float* dst, hipStream_t s) {
    generic_select_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(n, where, src, dst);
    return hipGetLastError();
}

void generic_ws_free(GenericWs& ws) { if (ws.p) (void)hipFree(ws.p); ws.p = nullptr; ws.cap = 0; }

int generic_net_size(const GenericDims& d, int out) { return net_lay(d, 0, out).end; }
int generic_slab_size(const GenericDims& d, bool actor) {
    return (generic_net_size(d, actor ? d.A : 1) + ((actor && !d.discrete) ? d.A : 0) + 8 + 3) / 4 * 4;
}

static size_t grad_floats_per_row(const GenericDims& d) { return (size_t)d.D + 3 * (size_t)d.A + (activation_needs_preactivation(d.act) ? 6 : 4) * (size_t)hidden_sum(d) + 40 + 8 * kMaxHidden; }
static int64_t grad_rows_max(const GenericDims& d) { return std::max<int64_t>((int64_t)(((size_t)1 << 29) / grad_floats_per_row(d)), 64); }   // <= 2 GiB of workspace per pass
int generic_pick_slabs(const GenericDims& d, int64_t count, int Gmax) {
    if (count < 1 || Gmax < 1) return -1;
    // >= 512 rows per slab, up to Gmax slabs.  Fewer slabs for wide nets (sized so that the largest weight-gradient contraction has just 2048 or 4096
    // output tiles) were measured slower (hidden 512: 38 / 49 vs 53 TFLOP/s) although grad_reduce_kernel then reads 8-16x fewer bytes: the contractions
    // want the parallelism
    int64_t G = std::min<int64_t>(Gmax, std::max<int64_t>(1, (count + 511) / 512));
    const int64_t need = (count + grad_rows_max(d) - 33) / (grad_rows_max(d) - 32);                   // a slab's rows (rounded up to 32) must fit one pass
    if (need > G) G = need;
    return G <= Gmax ? (int)G : -1;
}

hipError_t generic_policy(const GenericDims& d, const PolicyArgs& a, GenericWs& ws, hipStream_t s) {
    if (a.B <= 0) return hipSuccess;
    if (d.A > kMaxOut) return hipErrorInvalidValue;
    const size_t per_row = 2 * (size_t)hidden_sum(d) + d.A + 1 + 8 + 4 * kMaxHidden;
    const NetLay La = net_lay(d, a.actor.w1, d.A), Lc = net_lay(d, a.critic.w1, 1);
    int64_t Rmax = (int64_t)(((size_t)1 << 28) / per_row); Rmax = std::max<int64_t>(Rmax / 1024 * 1024, 1024);   // <= 1 GiB of activations per chunk
    const int64_t R = std::min<int64_t>(a.B, Rmax);
    hipError_t e = ws_reserve(ws, (size_t)R * per_row + 64); if (e != hipSuccess) return e;
    Carver c{ws.p};
    float* hb[kMaxHidden]; for (int l = 0; l < d.nh; ++l) hb[l] = c.take((size_t)2 * R * d.H[l]);
    float* out = c.take((size_t)R * d.A);
    for (int64_t r0 = 0; r0 < a.B; r0 += R) {
        const int64_t n = std::min<int64_t>(R, a.B - r0);
        const float* X = a.obs + r0 * d.D;
        if (a.obs_out) { const int64_t cnt = n * d.D; generic_copy_kernel<<<(unsigned)((cnt + 255) / 256), 256, 0, s>>>(X, a.obs_out + r0 * d.D, cnt); }
        if (a.values && a.mode != 2) { e = mlp_forward_both(d, a.params, La, Lc, X, (int)n, hb, out, a.values + r0, s); if (e != hipSuccess) return e; }
        else if (a.values) { e = mlp_forward(d, a.params, Lc, X, (int)n, hb, a.values + r0, s); if (e != hipSuccess) return e; }
        if (a.mode == 2) continue;
        if (!a.values) { e = mlp_forward(d, a.params, La, X, (int)n, hb, out, s); if (e != hipSuccess) return e; }
        PolicyHeadArgs hg{a, d.A, d.discrete, r0, n, out};
        generic_policy_head_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(hg);
        e = hipGetLastError(); if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// a.G slabs are written (a.G >= 1); rows of the minibatch are spread over them in order
hipError_t generic_ppo_grad(const GenericDims& d, const GradArgs& a, GenericWs& ws, hipStream_t s) {
    if (d.A > kMaxOut || a.G < 1 || a.count < 1) return hipErrorInvalidValue;
    const int G = a.G;
    const size_t per_row = grad_floats_per_row(d);
    const int64_t Rmax = grad_rows_max(d);
    int64_t Cr = (a.count + G - 1) / G; Cr = (Cr + 31) / 32 * 32;                                                // rows per slab (whole 32-deep contraction chunks)
    int Gp = (int)std::max<int64_t>(1, std::min<int64_t>(G, Rmax / Cr));                                       // slabs per pass
    if (Cr > Rmax) return hipErrorInvalidValue;                                                                 // the caller sizes G so that a chunk fits
    const int64_t R = (int64_t)Gp * Cr;
    hipError_t e = ws_reserve(ws, (size_t)R * per_row + 256); if (e != hipSuccess) return e;
    Carver c{ws.p};
    float* X = c.take((size_t)R * d.D); float* act = c.take((size_t)R * d.A); float* adv = c.take(R); float* lpo = c.take(R); float* ret = c.take(R);
    float* vold = c.take(R); float* valid = c.take(R);
    const NetLay La = net_lay(d, a.actor.w1, d.A), Lc = net_lay(d, a.critic.w1, 1);
    float* hb[kMaxHidden]; float* dzb[kMaxHidden]; float* zbb[kMaxHidden];                                      // per layer: the actor's rows, then the critic's
    const bool need_z = activation_needs_preactivation(d.act);
    for (int l = 0; l < d.nh; ++l) { hb[l] = c.take((size_t)2 * R * d.H[l]); dzb[l] = c.take((size_t)2 * R * d.H[l]); zbb[l] = need_z ? c.take((size_t)2 * R * d.H[l]) : nullptr; }
    float* outa = c.take((size_t)R * d.A); float* v = c.take(R);
    float* dout = c.take((size_t)R * d.A); float* dv = c.take(R); float* dlp = c.take(R);
    for (int slab0 = 0; slab0 < G; slab0 += Gp) {
        const int Gn = std::min(Gp, G - slab0); const int64_t Rn = (int64_t)Gn * Cr, row0 = (int64_t)slab0 * Cr;
        GatherArgs ga{a, d.D, d.A, d.discrete, row0, Rn, X, act, adv, lpo, ret, vold, valid};
        const int64_t ge = Rn * (d.D + 1);
        generic_gather_kernel<<<(unsigned)((ge + 255) / 256), 256, 0, s>>>(ga);
        e = hipGetLastError(); if (e != hipSuccess) return e;
        e = mlp_forward_both(d, a.params, La, Lc, X, (int)Rn, hb, outa, v, s, need_z ? zbb : nullptr); if (e != hipSuccess) return e;
        const float* ha[kMaxHidden]; const float* hc[kMaxHidden]; float* dza[kMaxHidden]; float* dzc[kMaxHidden]; const float* za[kMaxHidden]; const float* zc[kMaxHidden];
        for (int l = 0; l < d.nh; ++l) {                                               // layer l's block: the actor's rows, then the critic's (mlp_forward_both)
            ha[l] = hb[l]; hc[l] = ha[l] + (size_t)Rn * d.H[l];
            dza[l] = dzb[l]; dzc[l] = dza[l] + (size_t)Rn * d.H[l];
            za[l] = zbb[l]; zc[l] = need_z ? zbb[l] + (size_t)Rn * d.H[l] : nullptr;
        }
        LossHeadArgs lh{a, d.A, d.discrete, Rn, Cr, slab0, outa, v, act, adv, lpo, ret, vold, valid, dout, dv, dlp};
        generic_loss_head_kernel<<<Gn, 256, 0, s>>>(lh);
        e = hipGetLastError(); if (e != hipSuccess) return e;
        const BackwardPlan pa = plan_backward(d, a.params, La, X, ha, dza, dout, Rn, (int)Cr, Gn, a.slabs_actor + (size_t)slab0 * a.slab_a, a.slab_a, need_z ? za : nullptr);
        const BackwardPlan pc = plan_backward(d, a.params, Lc, X, hc, dzc, dv, Rn, (int)Cr, Gn, a.slabs_critic + (size_t)slab0 * a.slab_c, a.slab_c, need_z ? zc : nullptr);
        e = run_backward_both(pa, pc, s); if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace dril
