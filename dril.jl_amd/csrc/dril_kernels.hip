// dril_kernels.hip — every HIP kernel of libdril_hip.so (gfx950 / CDNA4 only) and their launchers.
//
// Kernel map (reference function each one replaces — paths relative to the reference root):
//   env_reset_kernel / env_observe_kernel / env_step_kernel   MultiThreadedParallelEnv reset!/observe/act!
//                                                            src/environment_wrappers/multithreadedParallelEnv.jl:12-74
//   policy_kernel        layer(obs,ps,st) / evaluate_actions / predict_values
//                        src/layers/layer_forward.jl:3-39, src/layers/layer_methods.jl:28-61
//   rollout_kernel       collect_trajectories, src/buffers/trajectory.jl:22-78 (persistent: one wave owns 32 envs for all T steps)
//   gae_kernel           compute_advantages! trajectory.jl:80-102 + returns rollout_buffer.jl:87
//   adv_moments_kernel   normalize! statistics, src/algorithms/ppo.jl:350-356
//   ppo_grad_kernel      (alg::PPO)(layer,ps,st,batch) ppo.jl:365-407 + its reverse pass (Zygote in the reference, ppo.jl:207)
//   grad_reduce_kernel / grad_norm_kernel / adam_kernel   nested_norm, nested_scale!, target_kl check, Adam — ppo.jl:213-239
//   explained_var_kernel ppo.jl:256
#include <utility>

#include "dril_internal.h"

namespace dril {

// =============================================================================================
// env verbs (step-granular path)
// =============================================================================================
template <int KIND>
__global__ void env_reset_kernel(int E, uint64_t seed0, float* state, int32_t* step_count, uint32_t* episode,
                                 uint32_t* gstep, float* disc_returns) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    float st[EnvSpec<KIND>::S];
    env_reset<KIND>(seed0 + (uint64_t)e, 0u, st);
#pragma unroll
    for (int i = 0; i < EnvSpec<KIND>::S; ++i) state[(size_t)e * EnvSpec<KIND>::S + i] = st[i];
    step_count[e] = 0; episode[e] = 0; gstep[e] = 0; disc_returns[e] = 0.f;
}

template <int KIND>
__global__ void env_observe_kernel(int E, const float* state, float* obs) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    float st[EnvSpec<KIND>::S], o[EnvSpec<KIND>::D];
#pragma unroll
    for (int i = 0; i < EnvSpec<KIND>::S; ++i) st[i] = state[(size_t)e * EnvSpec<KIND>::S + i];
    env_obs<KIND>(st, o);
#pragma unroll
    for (int i = 0; i < EnvSpec<KIND>::D; ++i) obs[(size_t)e * EnvSpec<KIND>::D + i] = o[i];
}

template <int KIND>
__global__ void env_step_kernel(int E, uint64_t seed0, int episode_len, int fixed_len, int action_start,
                                const void* actions, float* state, int32_t* step_count, uint32_t* episode,
                                uint32_t* gstep, float* rewards, uint8_t* term, uint8_t* trunc, float* terminal_obs, MonitorArgs mon) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    constexpr int S = EnvSpec<KIND>::S, D = EnvSpec<KIND>::D;
    float st[S];
#pragma unroll
    for (int i = 0; i < S; ++i) st[i] = state[(size_t)e * S + i];
    int ai = 0; float af = 0.f;
    if (EnvSpec<KIND>::discrete) ai = ((const int32_t*)actions)[e] - action_start; else af = ((const float*)actions)[e];
    bool t;
    const float r = env_step<KIND>(st, af, ai, fixed_len != 0, &t);
    const int sc = step_count[e] + 1;
    const bool tr = sc >= episode_len;
    rewards[e] = r; term[e] = t; trunc[e] = tr; gstep[e] += 1;
    if (mon.cur_ret) {                                             // MonitorWrapperEnv.act! (monitorWrapperEnv.jl:46-60), raw reward
        const float cr = mon.cur_ret[e] + r; const int cl = mon.cur_len[e] + 1;
        if (t || tr) { mon.ep_ret[e] = cr; mon.ep_len[e] = cl; mon.cur_ret[e] = 0.f; mon.cur_len[e] = 0; } else { mon.cur_ret[e] = cr; mon.cur_len[e] = cl; }
        mon.flags_out[e] = (uint8_t)((t ? 1 : 0) | (tr ? 2 : 0));
    }
    if (tr) { float o[D]; env_obs<KIND>(st, o);
#pragma unroll
        for (int i = 0; i < D; ++i) terminal_obs[(size_t)e * D + i] = o[i]; }
    if (t || tr) { const uint32_t ep = episode[e] + 1; episode[e] = ep; step_count[e] = 0; env_reset<KIND>(seed0 + (uint64_t)e, ep, st); }
    else step_count[e] = sc;
#pragma unroll
    for (int i = 0; i < S; ++i) state[(size_t)e * S + i] = st[i];
}

// =============================================================================================
// NormalizeWrapperEnv on device (src/environment_wrappers/normalizeWrapperEnv.jl): batch moments over the env axis,
// parallel-Welford merge (update_from_moments! :28-50), normalise + clip (:174-197).  Two launches per statistic:
// *_partials (per-block f64 sums) and *_apply (every block folds the partials in the same order and merges; block 0
// persists the new RunningMeanStd into the other half of the ping-pong state).
// =============================================================================================
__device__ __forceinline__ double block_sum_f64(double v, double* sh);

template <int KIND>
__global__ void obs_partials_kernel(int E, const float* __restrict__ state, float* __restrict__ raw, double* __restrict__ partials) {
    constexpr int S = EnvSpec<KIND>::S, D = EnvSpec<KIND>::D;
    __shared__ double sh[16];
    double s[D], q[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { s[d] = 0; q[d] = 0; }
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
        float st[S], o[D];
#pragma unroll
        for (int i = 0; i < S; ++i) st[i] = state[(size_t)e * S + i];
        env_obs<KIND>(st, o);
#pragma unroll
        for (int d = 0; d < D; ++d) { raw[(size_t)e * D + d] = o[d]; s[d] += o[d]; q[d] += (double)o[d] * o[d]; }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const double ss = block_sum_f64(s[d], sh), qq = block_sum_f64(q[d], sh);
        if (threadIdx.x == 0) { partials[(size_t)blockIdx.x * 16 + 2 * d] = ss; partials[(size_t)blockIdx.x * 16 + 2 * d + 1] = qq; }
    }
}

// update_from_moments! (normalizeWrapperEnv.jl:28-50) in the reference's f32 arithmetic
__device__ __forceinline__ void rms_merge(float& mean, float& var, long long count, float bmean, float bvar, long long bcount) {
    if (count == 0) { mean = bmean; var = bvar; }
    else {
        const long long tot = count + bcount;
        const float delta = bmean - mean;
        const float new_mean = mean + delta * (float)bcount / (float)tot;
        const float m_a = var * (float)count, m_b = bvar * (float)bcount;
        const float M2 = m_a + m_b + delta * delta * (float)count * (float)bcount / (float)tot;
        mean = new_mean; var = M2 / (float)tot;
    }
}

// column sums of the [nblocks][16] partial table with all 256 threads (16 columns x 16 block groups, independent loads in flight, fixed
// summation order): the serial `for b < nblocks` fold by D + 1 threads cost one dependent global round trip per block — 57 us per env step
__device__ __forceinline__ void fold_partials16(const double* __restrict__ partials, int nblocks, double (&s_part)[16][17], double (&s_col)[16]) {
    const int col = threadIdx.x & 15, seg = threadIdx.x >> 4;
    double t = 0;
#pragma unroll 4
    for (int b = seg; b < nblocks; b += 16) t += partials[(size_t)b * 16 + col];
    s_part[seg][col] = t;
    __syncthreads();
    if (threadIdx.x < 16) {
        double u = 0;
#pragma unroll
        for (int g = 0; g < 16; ++g) u += s_part[g][threadIdx.x];
        s_col[threadIdx.x] = u;
    }
    __syncthreads();
}

// data-parallel runs: the [nblocks][16] table of one rank's partial sums folded to ONE row, which RCCL then sums over ranks — the batch moments of
// NormalizeWrapperEnv cover every env of the job, as in the reference's single vector env (normalizeWrapperEnv.jl:21-26,139-171)
__global__ void fold_partials_kernel(const double* __restrict__ partials, int nblocks, double* __restrict__ out16) {
    __shared__ double s_part[16][17], s_col[16];
    fold_partials16(partials, nblocks, s_part, s_col);
    if (threadIdx.x < 16) out16[threadIdx.x] = s_col[threadIdx.x];
}

__global__ void norm_obs_apply_kernel(NormObsArgs a) {
    __shared__ float s_mean[8], s_var[8];
    __shared__ double s_part[16][17], s_col[16];
    if (a.update) fold_partials16(a.partials, a.nblocks, s_part, s_col);
    const long long nb_ = a.n_stats ? a.n_stats : a.E;          // envs behind the batch moments: all ranks' when the partials were all-reduced
    if (threadIdx.x < a.D) {
        const int d = threadIdx.x;
        float mean = a.in->mean[d], var = a.in->var[d];
        if (a.update) {
            const double s = s_col[2 * d], q = s_col[2 * d + 1];
            const double bm = s / nb_; double bv = q / nb_ - bm * bm; if (bv < 0) bv = 0;      // mean / var(corrected=false), :21-26
            rms_merge(mean, var, a.in->count, (float)bm, (float)bv, nb_);
        }
        s_mean[d] = mean; s_var[d] = var;
        if (blockIdx.x == 0) { a.out->mean[d] = mean; a.out->var[d] = var; if (d == 0) a.out->count = a.in->count + (a.update ? nb_ : 0); }
    }
    __syncthreads();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.E * a.D; i += gridDim.x * blockDim.x) {
        const int d = i % a.D;
        float v = a.raw[i];
        if (a.norm_obs) { v = (v - s_mean[d]) / sqrtf(s_var[d] + a.eps); v = fminf(fmaxf(v, -a.clip), a.clip); }   // normalize_obs! :174-179
        a.obs_n[i] = v;
    }
}

__global__ void rew_partials_kernel(int E, const float* __restrict__ rew_raw, float* __restrict__ disc, float gamma, int update,
                                    double* __restrict__ partials) {
    __shared__ double sh[16];
    double s = 0, q = 0;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
        float r = disc[e];
        if (update) { r = r * gamma + rew_raw[e]; disc[e] = r; }                        // update_reward_stats! :167-171
        s += r; q += (double)r * r;
    }
    s = block_sum_f64(s, sh); q = block_sum_f64(q, sh);
    if (threadIdx.x == 0) { partials[(size_t)blockIdx.x * 16] = s; partials[(size_t)blockIdx.x * 16 + 1] = q; }
}

__global__ void norm_rew_apply_kernel(NormRewArgs a) {
    __shared__ float s_var;
    __shared__ double s_part[16][17], s_col[16];
    if (a.update) fold_partials16(a.partials, a.nblocks, s_part, s_col);
    const long long nb_ = a.n_stats ? a.n_stats : a.E;
    if (threadIdx.x == 0) {
        float mean = a.in->mean[0], var = a.in->var[0];
        if (a.update) {
            const double s = s_col[0], q = s_col[1];
            const double bm = s / nb_; double bv = q / nb_ - bm * bm; if (bv < 0) bv = 0;
            rms_merge(mean, var, a.in->count, (float)bm, (float)bv, nb_);
        }
        s_var = var;
        if (blockIdx.x == 0) { a.out->mean[0] = mean; a.out->var[0] = var; a.out->count = a.in->count + (a.update ? nb_ : 0); }
    }
    __syncthreads();
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < a.E; e += gridDim.x * blockDim.x) {
        float r = a.rew_raw[e];
        if (a.norm_reward) { r = r / sqrtf(s_var + a.eps); r = fminf(fmaxf(r, -a.clip_reward), a.clip_reward); }   // normalize_rewards! :188-197 (no mean subtraction)
        a.rew_out[e] = r;
        if (a.flags_out) a.flags_out[e] = (uint8_t)((a.term[e] ? 1 : 0) | (a.trunc[e] ? 2 : 0));
        if (a.term[e] || a.trunc[e]) a.disc_returns[e] = 0.f;                                                    // :152-155
        if (a.norm_obs && a.trunc[e]) {                                                                          // terminal_observation, :157-163
            for (int d = 0; d < a.D; ++d) {
                float v = (a.tobs[(size_t)e * a.D + d] - a.obs_stats->mean[d]) / sqrtf(a.obs_stats->var[d] + a.eps);
                a.tobs[(size_t)e * a.D + d] = fminf(fmaxf(v, -a.clip_obs), a.clip_obs);
            }
        }
    }
}

// fused step of the normalised rollout: one launch does act! for every env (physics, flags, terminal_observation, auto-reset,
// discounted-return update) AND the per-block partial moments of the new observations and of the discounted returns;
// norm_apply_kernel then merges both RunningMeanStd states and normalises rewards, terminal observations (OLD obs statistics,
// normalizeWrapperEnv.jl:157-163) and the next observations (NEW statistics, :123-137).  7 launches per env step -> 3.
template <int KIND>
__global__ void norm_step_kernel(NormStepArgs a) {
    constexpr int S = EnvSpec<KIND>::S, D = EnvSpec<KIND>::D;
    __shared__ double sh[16];
    double acc[2 + 2 * D];
#pragma unroll
    for (int i = 0; i < 2 + 2 * D; ++i) acc[i] = 0;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < a.E; e += gridDim.x * blockDim.x) {
        float st[S];
#pragma unroll
        for (int i = 0; i < S; ++i) st[i] = a.state[(size_t)e * S + i];
        int ai = 0; float af = 0.f;
        if (EnvSpec<KIND>::discrete) ai = ((const int32_t*)a.actions)[e] - a.action_start; else af = ((const float*)a.actions)[e];
        bool t;
        const float r = env_step<KIND>(st, af, ai, a.fixed_len != 0, &t);
        const int sc = a.step_count[e] + 1;
        const bool tr = sc >= a.episode_len;
        a.rew_raw[e] = r; a.term[e] = t; a.trunc[e] = tr; a.gstep[e] += 1;
        if (a.flags_out) a.flags_out[e] = (uint8_t)((t ? 1 : 0) | (tr ? 2 : 0));
        if (a.mon_cur_ret) {                                       // MonitorWrapperEnv sits inside the normaliser: raw reward
            const float cr = a.mon_cur_ret[e] + r; const int cl = a.mon_cur_len[e] + 1;
            if (t || tr) { a.ep_ret[e] = cr; a.ep_len[e] = cl; a.mon_cur_ret[e] = 0.f; a.mon_cur_len[e] = 0; } else { a.mon_cur_ret[e] = cr; a.mon_cur_len[e] = cl; }
        }
        float disc = a.disc_returns[e];
        if (a.update_ret) { disc = disc * a.gamma + r; a.disc_returns[e] = disc; }       // update_reward_stats! :167-171 (reset of done envs happens in norm_apply_kernel)
        acc[0] += disc; acc[1] += (double)disc * disc;
        if (tr) { float o[D]; env_obs<KIND>(st, o);
#pragma unroll
            for (int i = 0; i < D; ++i) a.tobs_raw[(size_t)e * D + i] = o[i]; }
        if (t || tr) { const uint32_t ep = a.episode[e] + 1; a.episode[e] = ep; a.step_count[e] = 0; env_reset<KIND>(a.seed0 + (uint64_t)e, ep, st); }
        else a.step_count[e] = sc;
#pragma unroll
        for (int i = 0; i < S; ++i) a.state[(size_t)e * S + i] = st[i];
        float o[D]; env_obs<KIND>(st, o);
#pragma unroll
        for (int d = 0; d < D; ++d) { a.obs_raw[(size_t)e * D + d] = o[d]; acc[2 + 2 * d] += o[d]; acc[3 + 2 * d] += (double)o[d] * o[d]; }
    }
#pragma unroll
    for (int i = 0; i < 2 + 2 * D; ++i) {
        const double v = block_sum_f64(acc[i], sh);
        if (threadIdx.x == 0) a.partials[(size_t)blockIdx.x * 16 + i] = v;
    }
}

__global__ void norm_apply_kernel(NormApplyArgs a) {
    __shared__ float s_mean_new[8], s_var_new[8], s_mean_old[8], s_var_old[8], s_rvar;
    __shared__ double s_part[16][17], s_col[16];
    if (a.update_ret || a.update_obs) fold_partials16(a.partials, a.nblocks, s_part, s_col);
    const long long nb_ = a.n_stats ? a.n_stats : a.E;
    if (threadIdx.x <= a.D) {
        const int i = threadIdx.x;                       // 0: discounted returns, 1..D: observation dims
        const RmsState* in = i == 0 ? a.ret_in : a.obs_in;
        const int d = i == 0 ? 0 : i - 1;
        float mean = in->mean[d], var = in->var[d];
        const float mean_old = mean, var_old = var;
        const int upd = i == 0 ? a.update_ret : a.update_obs;
        if (upd) {
            const int col = i == 0 ? 0 : 2 * i;
            const double s = s_col[col], q = s_col[col + 1];
            const double bm = s / nb_; double bv = q / nb_ - bm * bm; if (bv < 0) bv = 0;
            rms_merge(mean, var, in->count, (float)bm, (float)bv, nb_);
        }
        if (i == 0) { s_rvar = var; if (blockIdx.x == 0) { a.ret_out->mean[0] = mean; a.ret_out->var[0] = var; a.ret_out->count = in->count + (upd ? nb_ : 0); } }
        else {
            s_mean_new[d] = mean; s_var_new[d] = var; s_mean_old[d] = mean_old; s_var_old[d] = var_old;
            if (blockIdx.x == 0) { a.obs_out->mean[d] = mean; a.obs_out->var[d] = var; if (d == 0) a.obs_out->count = in->count + (upd ? nb_ : 0); }
        }
    }
    __syncthreads();
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < a.E; e += gridDim.x * blockDim.x) {
        float r = a.rew_raw[e];
        if (a.norm_reward) { r = r / sqrtf(s_rvar + a.eps); r = fminf(fmaxf(r, -a.clip_reward), a.clip_reward); }
        a.rew_out[e] = r;
        const bool tr = a.trunc[e] != 0;
        if (a.term[e] || tr) a.disc_returns[e] = 0.f;
        for (int d = 0; d < a.D; ++d) {
            float v = a.obs_raw[(size_t)e * a.D + d];
            if (a.norm_obs) { v = (v - s_mean_new[d]) / sqrtf(s_var_new[d] + a.eps); v = fminf(fmaxf(v, -a.clip_obs), a.clip_obs); }
            a.obs_n[(size_t)e * a.D + d] = v;
            if (a.norm_obs && tr) {
                float tv = (a.tobs[(size_t)e * a.D + d] - s_mean_old[d]) / sqrtf(s_var_old[d] + a.eps);
                a.tobs[(size_t)e * a.D + d] = fminf(fmaxf(tv, -a.clip_obs), a.clip_obs);
            }
        }
    }
}

// MonitorWrapperEnv's CircularBuffer of the last W finished episodes (monitorWrapperEnv.jl:1-7,53-58), kept on device.
// Episodes finish in (step, env) order; after a rollout the collector pushes this rollout's events in that order, which
// only requires the LAST min(n_events, W) of them: count per step, suffix-sum to the first contributing step, then a
// block-wide prefix scan over the few rows that matter.
__global__ void monitor_count_kernel(const uint8_t* __restrict__ flags, int E, int T, int* __restrict__ cnt) {
    for (int t = blockIdx.x; t < T; t += gridDim.x) {
        int c = 0;
        for (int e = threadIdx.x; e < E; e += blockDim.x) c += flags[(size_t)t * E + e] != 0;
        __shared__ int sh[16];
        for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) { int s = 0; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += sh[w]; cnt[t] = s; }
        __syncthreads();
    }
}
__global__ void monitor_collect_kernel(const uint8_t* __restrict__ flags, const float* __restrict__ ep_ret, const int32_t* __restrict__ ep_len,
                                       int E, int T, int W, const int* __restrict__ cnt, float* ring_ret, int32_t* ring_len, int* meta) {
    __shared__ int s_t0, s_skip, s_total, s_base, wsum[16];
    if (threadIdx.x == 0) {
        int total = 0, t0 = T;
        for (int t = T - 1; t >= 0; --t) { if (total >= W) break; total += cnt[t]; t0 = t; }
        int all = 0; for (int t = 0; t < T; ++t) all += cnt[t];
        s_t0 = t0; s_total = all; s_skip = total > W ? total - W : 0; s_base = 0;
    }
    __syncthreads();
    if (s_total == 0) return;
    const int head0 = meta[1];
    for (int t = s_t0; t < T; ++t) {
        for (int e0 = 0; e0 < E; e0 += blockDim.x) {
            const int e = e0 + threadIdx.x;
            const int f = (e < E && flags[(size_t)t * E + e] != 0) ? 1 : 0;
            int incl = f;                                                            // inclusive scan within the wave
            const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
            for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d); if (lane >= d) incl += v; }
            if (lane == 63) wsum[wv] = incl;
            __syncthreads();
            int off = 0; for (int w = 0; w < wv; ++w) off += wsum[w];
            int chunk = 0; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) chunk += wsum[w];
            const int gi = s_base + off + incl - f;                                  // index of this event among the rows t0..T-1
            if (f && gi >= s_skip) { const int pos = (head0 + gi - s_skip) % W; ring_ret[pos] = ep_ret[(size_t)t * E + e]; ring_len[pos] = ep_len[(size_t)t * E + e]; }
            __syncthreads();
            if (threadIdx.x == 0) s_base += chunk;
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) {
        const int pushed = s_base - s_skip;
        meta[1] = (head0 + pushed) % W;
        const int c = meta[0] + pushed; meta[0] = c > W ? W : c;
    }
}

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

// first-layer B operand from an observation held in registers: xk[s] = obs[2s + h] (static register indices only)
template <int D> __device__ __forceinline__ void pair_obs(const float (&obs)[D], int h, float (&xk)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const float v0 = (2 * s < D) ? obs[(2 * s < D) ? 2 * s : 0] : 0.f;
        const float v1 = (2 * s + 1 < D) ? obs[(2 * s + 1 < D) ? 2 * s + 1 : 0] : 0.f;
        xk[s] = h ? v1 : v0;
    }
}

template <int D, int H, int O> struct NetLdsSplit {
    static_assert(H == 64, "the split kernel is laid out for hidden_dims [64,64]");
    static constexpr int DP = 4, OP = (O + 3) / 4 * 4;
    static constexpr int W1T = 0, B1 = W1T + DP * H, B2 = B1 + H, W3S = B2 + H, B3 = W3S + O * H, SMALL_END = (B3 + OP + 3) / 4 * 4;
    static constexpr int W2P = SMALL_END;                    // three pieces x [64][64] bf16 = 3 x 8192 bytes
    static constexpr int END = W2P + 3 * H * H / 2;          // in floats
};
__device__ __forceinline__ int w2img_gw(int r) { return (((r >> 1) & 1) << 3) | (((r >> 2) & 1) << 2) | (((r >> 3) & 1) << 1) | ((r >> 4) & 1); }
__device__ __forceinline__ int timg_gs(int r) { return (((r >> 1) & 1) << 3) | (((r >> 2) & 1) << 2) | (((r >> 3) & 1) << 1) | (r & 1); }

template <int D, int H, int O>
__device__ inline void stage_net_split(float* lds, const float* __restrict__ P, NetOff n, int tid, int nthreads) {
    using L = NetLdsSplit<D, H, O>;
    for (int i = tid; i < L::DP * H; i += nthreads) { const int o = i % H, k = i / H; lds[L::W1T + k * H + o] = k < D ? kTanhScale * P[n.w1 + o + k * H] : 0.0f; }
    for (int i = tid; i < H; i += nthreads) { lds[L::B1 + i] = kTanhScale * P[n.b1 + i]; lds[L::B2 + i] = kTanhScale * P[n.b2 + i]; }
    for (int i = tid; i < O * H; i += nthreads) { const int o = i % O, k = i / O; lds[L::W3S + o * H + k] = P[n.w3 + i]; }
    for (int i = tid; i < L::OP; i += nthreads) lds[L::B3 + i] = i < O ? P[n.b3 + i] : 0.0f;
    char* img = reinterpret_cast<char*>(lds + L::W2P);
    for (int i = tid; i < H * H / 2; i += nthreads) {         // pair (k, k+1) of row o: W2 is column-major (out x in), so consecutive threads read consecutive o
        const int o = i % H, kp = i / H;
        unsigned hi, mid, lo;
        split3_pair(kTanhScale * P[n.w2 + o + H * (2 * kp)], kTanhScale * P[n.w2 + o + H * (2 * kp + 1)], hi, mid, lo);
        const int byte = o * 128 + ((((kp >> 1) ^ w2img_gw(o)) & 15) << 3) + ((kp & 1) << 2);
        *reinterpret_cast<unsigned*>(img + byte) = hi; *reinterpret_cast<unsigned*>(img + 8192 + byte) = mid; *reinterpret_cast<unsigned*>(img + 16384 + byte) = lo;
    }
}

__device__ __forceinline__ bf16x8 chunk_frag(const unsigned (&pc)[3][4], int p) { return __builtin_bit_cast(bf16x8, u32x4{pc[p][0], pc[p][1], pc[p][2], pc[p][3]}); }
// ---- forward of one [64,64] net on the bf16 matrix cores (fp32-equivalent 3-piece split, as the gradient kernels): rollout_kernel / policy_kernel ----------------
// L1 on the f32 MFMA (K = 4), tanh and split of h1 a k16 step at a time, L2 as six v_mfma_f32_32x32x16_bf16 per step against the swizzled W2 piece image, tanh, L3 on
// the VALU.  Against the f32-MFMA forward (64 x 64 cycles on the VALU's lanes per net and tile) this is 48 x 32 cycles of matrix pipe + ~180 VALU instructions.
template <int D, int H, int O>
__device__ __forceinline__ void net_forward_split(const float* __restrict__ lds, const float (&xk)[2], float (&out)[O], int lane) {
    using L = NetLdsSplit<D, H, O>;
    constexpr int MT = H / 32;
    const int c = lane & 31, h = lane >> 5;
    const char* Wimg = reinterpret_cast<const char*>(lds + L::W2P);
    const int wf_base = c * 128 + (((h ^ w2img_gw(c)) & 15) << 3);
    f32x16 h1[MT], acc[MT];
    dense_first<H, MT>(lds + L::W1T, lds + L::B1, xk, h1, lane);
#pragma unroll
    for (int mo = 0; mo < MT; ++mo)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(lds + L::B2 + 32 * mo + 8 * q + 4 * h);
            acc[mo][4 * q + 0] = b[0]; acc[mo][4 * q + 1] = b[1]; acc[mo][4 * q + 2] = b[2]; acc[mo][4 * q + 3] = b[3];
        }
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float ex[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) ex[i] = __builtin_amdgcn_exp2f(h1[mi][8 * s + i]);
#pragma unroll
            for (int i = 0; i < 8; ++i) ex[i] = fmaf(-2.0f, __builtin_amdgcn_rcpf(ex[i] + 1.0f), 1.0f);
            unsigned pc[3][4];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) split3_pair(ex[2 * tt], ex[2 * tt + 1], pc[0][tt], pc[1][tt], pc[2][tt]);
#pragma unroll
            for (int mo = 0; mo < MT; ++mo) {
                const int a0 = (wf_base ^ (64 * mi + 32 * s)) + 4096 * mo;
                bf16x8 A[3];
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    A[p] = frag8(*reinterpret_cast<const u32x2*>(Wimg + 8192 * p + a0), *reinterpret_cast<const u32x2*>(Wimg + 8192 * p + (a0 ^ 16)));
                acc[mo] = mfma_split6(A[0], A[1], A[2], chunk_frag(pc, 0), chunk_frag(pc, 1), chunk_frag(pc, 2), acc[mo]);
            }
        }
#pragma unroll
    for (int mo = 0; mo < MT; ++mo) tanh16(acc[mo]);
    dense_out<MT, O, H>(lds + L::W3S, lds + L::B3, acc, out, lane);
}
#ifndef DRIL_FWD_SPLIT
#define DRIL_FWD_SPLIT 1     // 0: the f32-MFMA forward in rollout_kernel / policy_kernel (A/B)
#endif
// forward of one net for a 32-sample tile: LDS-resident weights (H = 64) or the wide path (W2 streamed from L2)
template <int D, int H, int O, bool WIDE>
__device__ __forceinline__ void eval_net(const float* __restrict__ lds, const float* __restrict__ w2a, const float (&xk)[2], float (&out)[O], int lane) {
    if constexpr (WIDE) net_forward_wide<D, H, O>(lds, w2a, xk, out, lane);
    else if constexpr (DRIL_FWD_SPLIT) net_forward_split<D, H, O>(lds, xk, out, lane);
    else { f32x16 h1[H / 32], h2[H / 32]; net_forward<D, H, H, O>(lds, xk, h1, h2, out, lane); }
}
template <int D, int H, int O, bool WIDE> struct FwdLds { static constexpr int SIZE = WIDE ? NetLdsSmall<D, H, O>::END : DRIL_FWD_SPLIT ? (NetLdsSplit<D, 64, O>::END + 3) / 4 * 4 : NetLds<D, H, H, O>::FWD_END; };
template <int D, int H, int O, bool WIDE>
__device__ __forceinline__ void stage_fwd(float* lds, const float* __restrict__ P, NetOff n, int tid, int nthreads) {
    if constexpr (WIDE) stage_net_small<D, H, O>(lds, P, n, tid, nthreads);
    else if constexpr (DRIL_FWD_SPLIT) stage_net_split<D, H, O>(lds, P, n, tid, nthreads);
    else stage_net<D, H, H, O, false>(lds, P, n, tid, nthreads);
}
// pre-tile W2 and W2' of one net for the wide path (see dril_device.h "wide nets")
__global__ void build_wimg_kernel(const float* __restrict__ P, NetOff off, int H, float* __restrict__ w2a, float* __restrict__ w2ta) {
    const int MT = H / 32;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < H * H; idx += gridDim.x * blockDim.x) {
        const int c = idx & 3, lane = (idx >> 2) & 63, q = (idx >> 8) & 3, mi = (idx >> 10) % MT, mo = (idx >> 10) / MT;
        const int o = 32 * mo + (lane & 31), k = 32 * mi + 8 * q + 4 * (lane >> 5) + c;
        w2a[idx] = kTanhScale * P[off.w2 + o + k * H];      // W2[o][k]  (column-major out x in), pre-scaled for tanh16
        w2ta[idx] = P[off.w2 + k + o * H];     // W2'[o][k] = W2[k][o]
    }
}

// =============================================================================================
// policy_kernel: host-batch / step-granular forward.  One wave = 32 samples.
// mode 0: sample + logprob + value; 1: evaluate given actions (+entropy); 2: critic only
// =============================================================================================
template <int KIND, int H, bool WIDE>
__global__ __launch_bounds__(256, WIDE ? 1 : 2) void policy_kernel(PolicyArgs a) {
    constexpr int D = EnvSpec<KIND>::D, A = EnvSpec<KIND>::A;
    constexpr bool DISC = EnvSpec<KIND>::discrete;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* la = smem; float* lc = smem + FwdLds<D, H, A, WIDE>::SIZE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.mode != 2) stage_fwd<D, H, A, WIDE>(la, a.params, a.actor, tid, blockDim.x);
    stage_fwd<D, H, 1, WIDE>(lc, a.params, a.critic, tid, blockDim.x);
    __syncthreads();
    const int64_t ntiles = (a.B + kTile - 1) / kTile;
    const int c = lane & 31, h = lane >> 5;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t b = tile * kTile + c;
        const bool valid = b < a.B;
        const int64_t bb = valid ? b : a.B - 1;
        if (a.only_where && !__any(valid && a.only_where[bb] != 0)) continue;     // wave-uniform skip
        if (a.boot_where) {                                                        // V(terminal_observation) of the previous env step
            const bool tr = valid && a.boot_where[bb] != 0;
            if (__any(tr)) {
                float tk[2];
#pragma unroll
                for (int s = 0; s < 2; ++s) { const int d = 2 * s + h; tk[s] = d < D ? a.boot_obs[bb * D + d] : 0.f; }
                float bv[1];
                eval_net<D, H, 1, WIDE>(lc, a.w2a_critic, tk, bv, lane);
                if (tr && h == 0) a.boot_out[b] = bv[0];
            }
        }
        float xk[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) { const int d = 2 * s + h; xk[s] = d < D ? a.obs[bb * D + d] : 0.f; }
        if (a.obs_out && valid) {
#pragma unroll
            for (int s = 0; s < 2; ++s) { const int d = 2 * s + h; if (d < D) a.obs_out[b * D + d] = xk[s]; }
        }
        float v[1];
        eval_net<D, H, 1, WIDE>(lc, a.w2a_critic, xk, v, lane);
        if (valid && h == 0 && a.values) a.values[b] = v[0];
        if (a.mode == 2) continue;
        float out[A];
        eval_net<D, H, A, WIDE>(la, a.w2a_actor, xk, out, lane);
        if (DISC) {
            float p[A]; softmax_n<A>(out, p);
            int act;
            if (a.mode == 0) {
                double u;
                if (a.noise) u = ((const double*)a.noise)[bb];
                else if (a.gstep) { const uint64_t k = a.env_seed0 + (uint64_t)bb; uint32_t r[4]; philox4x32_10((uint32_t)k, (uint32_t)(k >> 32), a.gstep[bb], 0, 1, 0, r); u = u01_f64(r[0], r[1]); }
                else { uint32_t r[4]; philox4x32_10((uint32_t)a.seed, (uint32_t)(a.seed >> 32), (uint32_t)bb, (uint32_t)(bb >> 32), 3, a.call_counter, r); u = u01_f64(r[0], r[1]); }
                act = categorical_sample<A>(p, u);
                if (a.deterministic) {                                   // mode(d) = argmax(p) (first maximum), categorical.jl:42-44
                    act = 0; float best = p[0];
#pragma unroll
                    for (int i = 1; i < A; ++i) if (p[i] > best) { best = p[i]; act = i; }
                }
                if (valid && h == 0) ((int32_t*)a.actions)[b] = act + a.action_start;
            } else act = ((const int32_t*)a.actions)[bb] - a.action_start;
            if (valid && h == 0) {
                a.logp[b] = flog(pick<A>(p, act));
                if (a.mode == 1 && a.entropy) a.entropy[b] = categorical_entropy<A>(p);
            }
        } else {
            const float* ls = a.params + a.log_std_off;
            float x[A];
            if (a.mode == 0) {
#pragma unroll
                for (int i = 0; i < A; ++i) {
                    float z;
                    if (a.noise) z = ((const float*)a.noise)[bb * A + i];
                    else if (a.gstep) { const uint64_t k = a.env_seed0 + (uint64_t)bb; uint32_t r[4]; philox4x32_10((uint32_t)k, (uint32_t)(k >> 32), a.gstep[bb], 0, 1, (uint32_t)(i / 2), r); z = (i & 1) ? randn_f32(r[2], r[3]) : randn_f32(r[0], r[1]); }
                    else { uint32_t r[4]; philox4x32_10((uint32_t)a.seed, (uint32_t)(a.seed >> 32), (uint32_t)bb, (uint32_t)(bb >> 32), 3 + 16 * (uint32_t)i, a.call_counter, r); z = randn_f32(r[0], r[1]); }
                    x[i] = a.deterministic ? out[i] : out[i] + fexp(ls[i]) * z;            // mode(d) = mean, diagGaussian.jl:45-47
                    if (valid && h == 0) ((float*)a.actions)[b * A + i] = x[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < A; ++i) x[i] = ((const float*)a.actions)[bb * A + i];
            }
            if (valid && h == 0) {
                a.logp[b] = gauss_logpdf<A>(x, out, ls);
                if (a.mode == 1 && a.entropy) a.entropy[b] = gauss_entropy<A>(ls);
            }
        }
    }
}

// =============================================================================================
// rollout_kernel: persistent collect_trajectories (trajectory.jl:22-78).  A wave owns 32 envs for all
// T steps: env state lives in registers, both nets' weights in LDS, only buffer writes touch HBM.
// Buffer layout: time-major, index k = t*E + e; every store of a wave is one full 128-byte line
// (or 512 B for the float4 observation rows).
// =============================================================================================
template <int KIND, int H, bool WIDE>
__global__ __launch_bounds__(256, WIDE ? 1 : 2) void rollout_kernel(RolloutArgs a) {
    constexpr int D = EnvSpec<KIND>::D, A = EnvSpec<KIND>::A, S = EnvSpec<KIND>::S;
    constexpr bool DISC = EnvSpec<KIND>::discrete;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* la = smem; float* lc = smem + FwdLds<D, H, A, WIDE>::SIZE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    stage_fwd<D, H, A, WIDE>(la, a.params, a.actor, tid, blockDim.x);
    stage_fwd<D, H, 1, WIDE>(lc, a.params, a.critic, tid, blockDim.x);
    __syncthreads();
    const int c = lane & 31, h = lane >> 5;
    const int e_raw = (blockIdx.x * 4 + wave) * kTile + c;
    if ((blockIdx.x * 4 + wave) * kTile >= a.E) return;  // whole wave out of range (wave-uniform)
    const bool valid = e_raw < a.E;
    const int e = valid ? e_raw : a.E - 1;
    const bool writer = valid && h == 0;
    const uint64_t env_seed = a.env_seed0 + (uint64_t)e;

    float st[S];
#pragma unroll
    for (int i = 0; i < S; ++i) st[i] = a.state[(size_t)e * S + i];
    int sc = a.step_count[e];
    uint32_t ep = a.episode[e], gs = a.gstep[e];
    float obs[D];
    env_obs<KIND>(st, obs);
    float lsr[4] = {0.f, 0.f, 0.f, 0.f};                               // log_std in registers for the whole rollout (the parameters are read-only during it)
    if (!DISC) {
#pragma unroll
        for (int i = 0; i < (A < 4 ? A : 4); ++i) lsr[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.params[a.log_std_off + i])));
    }
    const float* ls = lsr;
    float mon_ret = a.mon_cur_ret ? a.mon_cur_ret[e] : 0.f;           // MonitorWrapperEnv running episode return / length
    int mon_len = a.mon_cur_len ? a.mon_cur_len[e] : 0;

    for (int t = 0; t < a.T; ++t) {
        const size_t k = (size_t)t * a.E + e;
        // keep the weights in LDS: an opaque zero offset stops hipcc from hoisting ~150 loop-invariant
        // ds_reads into VGPRs (which spilled at the 256-register budget of 2 waves/SIMD)
        int zoff = 0; asm volatile("" : "+v"(zoff));
        const float* la_t = la + zoff; const float* lc_t = lc + zoff;
        float xk[2];
        pair_obs<D>(obs, h, xk);
        float v[1], out[A];
        eval_net<D, H, 1, WIDE>(lc_t, a.w2a_critic, xk, v, lane);
        __builtin_amdgcn_sched_barrier(0);   // do not interleave the two nets: that doubles the live weight fragments
        eval_net<D, H, A, WIDE>(la_t, a.w2a_actor, xk, out, lane);
        __builtin_amdgcn_sched_barrier(0);
        // ---- sample (layer_forward.jl:10-11 / :36-37) ----
        int act_env = 0; float actf_env = 0.f; float logp;
        if (DISC) {
            float p[A]; softmax_n<A>(out, p);
            double u;
            if (a.noise) u = ((const double*)a.noise)[k];
            else { uint32_t r[4]; philox4x32_10((uint32_t)env_seed, (uint32_t)(env_seed >> 32), gs, 0, 1, 0, r); u = u01_f64(r[0], r[1]); }
            const int act = categorical_sample<A>(p, u);
            logp = flog(pick<A>(p, act));
            act_env = act;                                                   // DiscreteAdapter: identity (default_adapters.jl:34-38)
            if (writer) ((int32_t*)a.act)[k] = act + a.action_start;         // raw action stored (trajectory.jl:48)
        } else {
            float x[A];
#pragma unroll
            for (int i = 0; i < A; ++i) {
                float z;
                if (a.noise) z = ((const float*)a.noise)[k * A + i];
                else { uint32_t r[4]; philox4x32_10((uint32_t)env_seed, (uint32_t)(env_seed >> 32), gs, 0, 1, (uint32_t)(i / 2), r); z = (i & 1) ? randn_f32(r[2], r[3]) : randn_f32(r[0], r[1]); }
                x[i] = out[i] + fexp(ls[i]) * z;
                if (writer) ((float*)a.act)[k * A + i] = x[i];
            }
            logp = gauss_logpdf<A>(x, out, ls);
            actf_env = fminf(fmaxf(x[0], -act_bound<KIND>()), act_bound<KIND>());   // ClampAdapter on action_space(env) (default_adapters.jl:4-11)
        }
        if (writer) {
            if (D == 4) *reinterpret_cast<float4*>(a.obs + k * 4) = make_float4(obs[0], obs[1], obs[2], obs[3]);
            else {
#pragma unroll
                for (int i = 0; i < D; ++i) a.obs[k * D + i] = obs[i];
            }
            a.val[k] = v[0]; a.logp[k] = logp;
        }
        // ---- act! with auto-reset (multithreadedParallelEnv.jl:56-71) ----
        bool term;
        const float rew = env_step<KIND>(st, actf_env, act_env, a.fixed_len != 0, &term);
        sc += 1; gs += 1;
        const bool trunc = sc >= a.episode_len;
        if (__any(trunc && valid)) {                                         // V(terminal_observation), trajectory.jl:57-61
            float tobs[D]; env_obs<KIND>(st, tobs);
            float tk[2];
            pair_obs<D>(tobs, h, tk);
            float bv[1];
            eval_net<D, H, 1, WIDE>(lc_t, a.w2a_critic, tk, bv, lane);
            if (writer && trunc) a.boot[k] = bv[0];
        }
        mon_ret += rew; mon_len += 1;
        if (term || trunc) {
            ep += 1; sc = 0; env_reset<KIND>(env_seed, ep, st);
            if (writer && a.ep_ret) { a.ep_ret[k] = mon_ret; a.ep_len[k] = mon_len; }     // finished episode (monitorWrapperEnv.jl:53-58)
            mon_ret = 0.f; mon_len = 0;
        }
        if (writer) { a.rew[k] = rew; a.flags[k] = (uint8_t)((term ? 1 : 0) | (trunc ? 2 : 0)); }
        env_obs<KIND>(st, obs);                                              // observe(env), trajectory.jl:45
    }
    {   // V(new_obs) for rollout-limited trajectories, trajectory.jl:65-70 (computed for every env; GAE uses it when needed)
        float xk[2];
        pair_obs<D>(obs, h, xk);
        float v[1];
        eval_net<D, H, 1, WIDE>(lc, a.w2a_critic, xk, v, lane);
        if (writer) a.last_values[e] = v[0];
    }
    if (writer) {
#pragma unroll
        for (int i = 0; i < S; ++i) a.state[(size_t)e * S + i] = st[i];
        a.step_count[e] = sc; a.episode[e] = ep; a.gstep[e] = gs;
        if (a.mon_cur_ret) { a.mon_cur_ret[e] = mon_ret; a.mon_cur_len[e] = mon_len; }
    }
}

// =============================================================================================
// gae_kernel: one thread per env, backward scan over the time-major buffer (coalesced 256 B rows)
// =============================================================================================
__global__ void gae_kernel(int E, int T, float gamma, float lam, const float* __restrict__ rew,
                           const float* __restrict__ val, const uint8_t* __restrict__ flags,
                           const float* __restrict__ boot, const float* __restrict__ last_values,
                           float* __restrict__ adv, float* __restrict__ ret) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    float next_adv = 0.f, next_val = 0.f;
    const float lv = last_values[e];
#pragma unroll 8
    for (int t = T - 1; t >= 0; --t) {
        const size_t k = (size_t)t * E + e;
        const float r = rew[k], v = val[k];
        const uint8_t f = flags[k];
        const bool term = f & 1, trunc = f & 2;
        float a;
        if (term || trunc || t == T - 1) {                    // last step of a trajectory, trajectory.jl:85-95
            float delta;
            if (term) delta = r - v;
            else if (trunc) delta = r + gamma * boot[k] - v;
            else delta = r + gamma * lv - v;
            a = delta;
        } else {                                              // trajectory.jl:96-99
            const float delta = r + gamma * next_val - v;
            a = delta + gamma * lam * next_adv;
        }
        adv[k] = a; ret[k] = a + v;                           // rollout_buffer.jl:87
        next_adv = a; next_val = v;
    }
}

// =============================================================================================
// advantage moments of one minibatch (normalize!, ppo.jl:350-356): per-block f64 partials, fixed-order finalize
// =============================================================================================
__device__ __forceinline__ double block_sum_f64(double v, double* sh) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double s = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += sh[w];
    return s;
}
__global__ void adv_moments_kernel(MomentsArgs a) {
    __shared__ double sh[16];
    if (*a.stop_flag) return;
    double s = 0, q = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.count; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = a.pos0 + i;
        const int64_t idx = a.perm ? a.perm[p] : (a.perm_bits ? perm_index(p, a.N, a.perm_key, a.perm_bits) : p);
        const int64_t li = idx - a.idx_lo;                    // this rank's shard of the buffer
        if (li >= 0 && li < a.n_local) { const double x = a.adv[li]; s += x; q += x * x; }
    }
    s = block_sum_f64(s, sh); q = block_sum_f64(q, sh);
    if (threadIdx.x == 0) { a.partials[2 * blockIdx.x] = s; a.partials[2 * blockIdx.x + 1] = q; }
}
// epoch_moments_kernel: advantage sums of ALL minibatches of one epoch in a single sequential pass over the buffer
// (v1-v4 gathered adv[perm(p)] per optimiser step: 32 x 85 us of random 4-byte reads per epoch).  Each index is mapped back
// to its position in the epoch order by the inverse bijection, binned by minibatch in LDS (ds_add_f64), one partial
// table per block; epoch_moments_finalize_kernel folds the blocks in a fixed order.
__global__ void epoch_moments_kernel(const float* __restrict__ adv, int64_t N, int64_t B, int nb, uint64_t key, int bits,
                                     double* __restrict__ block_tables, const int* stop_flag) {
    extern __shared__ double bins[];     // [nb][2]
    if (*stop_flag) return;
    for (int i = threadIdx.x; i < 2 * nb; i += blockDim.x) bins[i] = 0.0;
    __syncthreads();
    const int64_t per = (N + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < N ? lo + per : N;
    for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const double x = adv[i];
        const int64_t p = perm_position(i, N, key, bits);
        const int k = (int)((uint64_t)p / (uint64_t)B);
        atomicAdd(&bins[2 * k], x); atomicAdd(&bins[2 * k + 1], x * x);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * nb; i += blockDim.x) block_tables[(size_t)blockIdx.x * 2 * nb + i] = bins[i];
}
__global__ void epoch_moments_finalize_kernel(const double* __restrict__ block_tables, int nblocks, int nb, int64_t N, int64_t B,
                                              double* __restrict__ table3, const int* stop_flag) {
    if (*stop_flag) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb) return;
    double s = 0, q = 0;
#pragma unroll 8
    for (int b = 0; b < nblocks; ++b) { s += block_tables[(size_t)b * 2 * nb + 2 * i]; q += block_tables[(size_t)b * 2 * nb + 2 * i + 1]; }   // unrolled: 16 independent loads in flight
    const int64_t pos0 = (int64_t)i * B;
    table3[3 * i] = s; table3[3 * i + 1] = q; table3[3 * i + 2] = (double)(pos0 + B <= N ? B : N - pos0);
}

__global__ void moments_finalize_kernel(const double* partials, int nblocks, double* out3, double n_local, const int* stop_flag) {
    __shared__ double sh[16];
    if (*stop_flag) return;
    double s = 0, q = 0;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) { s += partials[2 * i]; q += partials[2 * i + 1]; }   // fixed order per thread
    s = block_sum_f64(s, sh); q = block_sum_f64(q, sh);
    if (threadIdx.x == 0) { out3[0] = s; out3[1] = q; out3[2] = n_local; }
}

// =============================================================================================
// ppo_grad_kernel — the dominant kernel.  Fused forward + loss + backward of ONE net per workgroup
// (even blocks: actor, odd blocks: critic — the two MLPs share no parameters, layer_helpers.jl:13-25,
// so their gradients decouple given the batch).  Per 32-sample tile and net: 228 v_mfma_f32_32x32x2_f32
//   fwd  L1 4 + L2 64                      (L3 and its transpose products run on the VALU, O <= 2)
//   bwd  dh1 = W2' dz2 64, dW2 += dz2 h1' 64, dW1|db1 += dz1 [x;1]' 32
// Weight gradients accumulate in registers over the workgroup's whole share of the minibatch and
// leave as ONE slab per workgroup (plain coalesced stores) — grad_reduce_kernel sums the slabs in a
// fixed order, so the result is bitwise reproducible and no float atomics are used.
// =============================================================================================
enum { HEAD_CATEGORICAL = 0, HEAD_GAUSSIAN = 1, HEAD_VALUE = 2 };
constexpr int kLsMax = 4;   // action dims whose log_std the kernels keep in registers

// diagnostic build only (-DDRIL_STAMPS): per-phase s_memtime shares of one tile; never used for timing claims
#ifdef DRIL_STAMPS
#define STAMP(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); \
                      __builtin_amdgcn_sched_barrier(0); stamp_acc[k] += _t - stamp_prev; stamp_prev = _t; } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

// pack_records_kernel: the six per-sample fields the loss reads (ppo.jl:366-371) as one 32-byte record, so that a
// minibatch gather is ONE 16-byte load per lane (the two half-waves of a sample fetch the two halves of its record = one
// 32-B sector) instead of five scattered 4-byte loads (v1-v4: FETCH 1.6-3.1 GB per launch vs 0.22 GB algorithmic).
template <int KIND>
__global__ void pack_records_kernel(int64_t N, const float* __restrict__ obs, const void* __restrict__ act, const float* __restrict__ adv,
                                    const float* __restrict__ logp, const float* __restrict__ ret, float4* __restrict__ rec) {
    constexpr int D = EnvSpec<KIND>::D;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < D; ++d) o[d] = obs[n * D + d];
        const float a = EnvSpec<KIND>::discrete ? __int_as_float(((const int32_t*)act)[n]) : ((const float*)act)[n];
        rec[2 * n] = make_float4(o[0], o[1], o[2], o[3]);
        rec[2 * n + 1] = make_float4(a, adv[n], logp[n], ret[n]);
    }
}

// one lane's share of a minibatch tile (DataLoader gather, ppo.jl:188-195).  Loaded one tile AHEAD of its use so the
// random-gather latency (~2 us under load, fully exposed in v1: 23 % of wave time in s_waitcnt) hides under the
// previous tile's MFMAs; the loop body issues no other vector-memory op, so the loads stay in flight until first use.
template <int O> struct TileIn { float xk[2]; float s0, s1; int act; float xa[O]; bool valid; float4 raw; };

template <int KIND, int O, int HEAD, bool REC>
__device__ __forceinline__ void load_tile(const GradArgs& a, int64_t tile, int64_t ntiles, int c, int h, TileIn<O>& t) {
    constexpr int D = EnvSpec<KIND>::D;
    const bool live = tile < ntiles;
    const int64_t i = (live ? tile : ntiles - 1) * kTile + c;
    const bool inb = live && i < a.count;
    const int64_t p = a.pos0 + (inb ? i : 0);
    const int64_t gidx = a.perm ? a.perm[p] : (a.perm_bits ? perm_index(p, a.N, a.perm_key, a.perm_bits) : p);   // bits 0 = identity order
    const int64_t li = gidx - a.idx_lo;
    t.valid = inb && li >= 0 && li < a.n_local;
    const int64_t idx = t.valid ? li : 0;
    t.act = 0; t.s0 = 0.f; t.s1 = 0.f;
    if (REC) {
        // lane (sample, h) loads half h of the record; t.raw is exchanged between the half-waves at first use (unpack_tile)
        t.raw = a.rec[2 * idx + h];
        if (HEAD == HEAD_VALUE && a.has_clip_vf) t.s1 = a.val_old[idx];
        return;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) { const int d = 2 * s + h; t.xk[s] = d < D ? a.obs[idx * D + d] : 0.f; }
    if (HEAD == HEAD_VALUE) { t.s0 = a.ret[idx]; t.s1 = a.has_clip_vf ? a.val_old[idx] : 0.f; }
    else {
        t.s0 = a.adv[idx]; t.s1 = a.logp_old[idx];
        if (HEAD == HEAD_CATEGORICAL) t.act = ((const int32_t*)a.actions)[idx] - a.action_start;
        else {
#pragma unroll
            for (int o = 0; o < O; ++o) t.xa[o] = ((const float*)a.actions)[idx * O + o];
        }
    }
}

// exchange the two record halves between the half-waves: v_permlane32_swap(a, b) swaps a[32..63] with b[0..31], so with
// a = b = v the results are {lo-half value in every lane, hi-half value in every lane}
template <int KIND, int O, int HEAD, bool REC>
__device__ __forceinline__ void unpack_tile(const GradArgs& a, int h, TileIn<O>& t) {
    if (!REC) return;
    float lo[4], hi[4];
    const float v[4] = {t.raw.x, t.raw.y, t.raw.z, t.raw.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned u = __float_as_uint(v[i]);
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        lo[i] = __uint_as_float(r[0]); hi[i] = __uint_as_float(r[1]);
    }
    t.xk[0] = h ? lo[1] : lo[0]; t.xk[1] = h ? lo[3] : lo[2];      // xk[s] = obs[2s + h]
    if (HEAD == HEAD_VALUE) t.s0 = hi[3];
    else {
        t.s0 = hi[1]; t.s1 = hi[2];
        if (HEAD == HEAD_CATEGORICAL) t.act = __float_as_int(hi[0]) - a.action_start; else t.xa[0] = hi[0];
    }
}

// (alg::PPO)(...) loss terms and dLoss/d(net output) for one sample per lane (ppo.jl:377-404); `tally` selects the lanes that
// add to the statistics / log_std sums (each sample is replicated in the two half-waves, and in every wave of a wide workgroup)
template <int O, int HEAD>
__device__ __forceinline__ void loss_head(const GradArgs& a, const TileIn<O>& cur, const float (&out)[O], bool valid, bool tally, const float* ls,
                                          float adv_mean, float adv_inv, float (&dz)[O], float (&st)[5], float (&dlsp)[O]) {
    // branch-free on purpose: a lane-dependent `if` here becomes an s_cbranch_execz in the middle of the tile loop and splits it into basic blocks
    // that the scheduler cannot move MFMAs / LDS reads across
    const bool count_it = valid && tally;
    if (HEAD == HEAD_VALUE) {
        const float R = cur.s0;
        const float ov = cur.s1, dcl = out[0] - ov;                        // clip_range, ppo.jl:344-346,378
        const bool inside = (dcl >= -a.clip_range_vf) & (dcl <= a.clip_range_vf);      // bitwise: && / ?: compile to branches
        const bool vpass = inside | (a.has_clip_vf == 0);
        const float vclip = ov + fminf(fmaxf(dcl, -a.clip_range_vf), a.clip_range_vf);
        const float value = a.has_clip_vf ? vclip : out[0];
        const float ve = value - R;
        dz[0] = (valid & vpass) ? a.invB * a.vf_coef * 2.0f * ve : 0.f;
        st[0] += count_it ? ve * ve : 0.f;                                 // value_loss numerator, ppo.jl:385
    } else {
        const float advn = (cur.s0 - adv_mean) * adv_inv;
        const float olp = cur.s1;
        float logp, ent;
        float p[O];
        int act = 0;
        float xa[O];
        if (HEAD == HEAD_CATEGORICAL) {
            softmax_n<O>(out, p);
            act = cur.act;
            logp = flog(pick<O>(p, act));
            ent = categorical_entropy<O>(p);
        } else {
#pragma unroll
            for (int o = 0; o < O; ++o) xa[o] = cur.xa[o];
            logp = gauss_logpdf<O>(xa, out, ls);
            ent = gauss_entropy<O>(ls);
        }
        const float lr = logp - olp;
        const float r = fexp(lr);                                          // ppo.jl:380
        const float lo = 1.0f - a.clip_range, hi = 1.0f + a.clip_range;
        const float rc = fminf(fmaxf(r, lo), hi);                          // :381
        const float t1 = r * advn, t2 = rc * advn;
        const float mn = t2 < t1 ? t2 : t1;                                // :382
        const float dm_dr = (t2 < t1) ? ((r >= lo && r <= hi) ? advn : 0.f) : advn;
        const float dlogp = valid ? -a.invB * dm_dr * r : 0.f;
        const float dent = valid ? -a.invB * a.ent_coef : 0.f;             // ent_loss = -mean(entropy), :383,:386
        if (HEAD == HEAD_CATEGORICAL) {
#pragma unroll
            for (int o = 0; o < O; ++o)
                dz[o] = dlogp * ((o == act ? 1.0f : 0.0f) - p[o]) + dent * (-p[o] * (flog(p[o]) + ent));
        } else {
#pragma unroll
            for (int o = 0; o < O; ++o) {
                const float iv = fexp(-2.0f * ls[o]), d = xa[o] - out[o];
                dz[o] = dlogp * d * iv;
                dlsp[o] += tally ? dlogp * (d * d * iv - 1.0f) + dent : 0.f;
            }
        }
        st[0] += count_it ? -mn : 0.f; st[1] += count_it ? ent : 0.f; st[2] += (count_it && r != rc) ? 1.0f : 0.0f;   // :382,:383,:390
        st[3] += count_it ? (r - 1.0f) - lr : 0.f; st[4] += count_it ? r : 0.f;                                       // :393,:402
    }
}

// per-wave LDS scratch of grad_body (floats): one [H][kTS] transpose image reused in turn for h2, h1, dz2, dz1,
// the [D+2][kTS] first-layer input image (rows 0..D-1 = x, row D = 1 for the bias column, row D+1 = 0) and the [O][kTS]
// dLoss/dout image.  2 workgroups (4 waves each) per CU => 2 waves per SIMD, so one wave's VALU/LDS phases overlap the
// other's MFMAs; that needs <= 256 registers and <= 80 KB LDS per workgroup.
template <int D, int H, int O> struct GradScratch {
    static constexpr int T = 0;
    static constexpr int XI = T + H * kTS;
    static constexpr int ZI = XI + (D + 2) * kTS;
    static constexpr int SIZE = ZI + O * kTS;
};

template <int KIND, int H, int O, int HEAD, bool REC>
__device__ __forceinline__ void grad_body(const GradArgs& a, float* smem) {
    constexpr int D = EnvSpec<KIND>::D, MT = H / 32;
    using L = NetLds<D, H, H, O>;
    using SC = GradScratch<D, H, O>;
    const int tid = threadIdx.x, lane = tid & 63;
    // threadIdx.x / 64 IS wave-uniform but hipcc cannot prove it: readfirstlane moves the wave id - and every tile index,
    // LDS base and loop bound derived from it - into SGPRs (v3 spilled those to scratch, and each scratch reload's
    // s_waitcnt vmcnt(0) drained the prefetched gathers: profiles/r01 stamps, "out+head" 7.0k cycles)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const NetOff off = HEAD == HEAD_VALUE ? a.critic : a.actor;
    float* wl = smem;
    float* T = smem + L::BWD_END + wave * SC::SIZE + SC::T;
    float* XI = smem + L::BWD_END + wave * SC::SIZE + SC::XI;
    float* ZI = smem + L::BWD_END + wave * SC::SIZE + SC::ZI;
    stage_net<D, H, H, O, true>(wl, a.params, off, tid, blockDim.x);
    for (int i = lane; i < (D + 2) * kTS; i += 64) XI[i] = (i / kTS == D) ? 1.0f : 0.0f;
    __syncthreads();

    // advantage normalisation constants (ppo.jl:350-356): mean, corrected std, eps added to the std
    float adv_mean = 0.f, adv_den = 1.f;
    if (HEAD != HEAD_VALUE && a.normalize_adv) {
        double s, q, n;
        if (a.inline_moments) {
            // small minibatch: sum A and A^2 of the whole minibatch here (same index map as load_tile), fixed-order tree => every workgroup gets the same bits
            double* shd = reinterpret_cast<double*>(smem + ((L::BWD_END + 1) & ~1));      // per-wave scratch, not yet in use
            double ls_ = 0, lq_ = 0;
            for (int64_t i2 = tid; i2 < a.count; i2 += blockDim.x) {
                const int64_t p2 = a.pos0 + i2;
                const int64_t gi = a.perm ? a.perm[p2] : (a.perm_bits ? perm_index(p2, a.N, a.perm_key, a.perm_bits) : p2);
                const int64_t li2 = gi - a.idx_lo;
                if (li2 >= 0 && li2 < a.n_local) { const float v = REC ? a.rec[2 * li2 + 1].y : a.adv[li2]; ls_ += v; lq_ += (double)v * v; }
            }
            shd[tid] = ls_; shd[256 + tid] = lq_;
            __syncthreads();
            for (int st_ = 128; st_ > 0; st_ >>= 1) { if (tid < st_) { shd[tid] += shd[tid + st_]; shd[256 + tid] += shd[256 + tid + st_]; } __syncthreads(); }
            s = shd[0]; q = shd[256]; n = (double)a.count;
            __syncthreads();
        } else { s = a.adv_stats[0]; q = a.adv_stats[1]; n = a.adv_stats[2]; }
        const double mean = s / n;
        double var = (q - s * mean) / (n - 1.0);
        if (var < 0) var = 0;
        adv_mean = (float)mean; adv_den = (float)sqrt(var) + 1.0e-8f;
    }
    adv_mean = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, adv_mean)));
    const float adv_inv = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, 1.0f / adv_den)));
    constexpr bool LS_GAUSS = HEAD == HEAD_GAUSSIAN; constexpr int LS_N = O;
    // log_std hoisted into scalar registers: a per-tile global load would sit in the in-order vmcnt queue between the prefetched
    // gathers and their first use and drain them every tile (Pendulum [64,64]: 91 -> TFLOP/s below)
    float lsr[kLsMax];
#pragma unroll
    for (int o = 0; o < kLsMax; ++o) lsr[o] = 0.f;
    if (LS_GAUSS) {
#pragma unroll
        for (int o = 0; o < LS_N; ++o) lsr[o] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.params[a.log_std_off + o])));
    }
    const float* ls = lsr;

    f32x16 dW2[MT][MT];
    f32x4 dW1[H / 16];                                             // 16x16x4 tiles: rows = hidden, cols = [x | 1 | 0...]
    float dW3a[O][MT], db2p[MT], db3p[O], dlsp[O], st[5];
#pragma unroll
    for (int i = 0; i < H / 16; ++i) dW1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        db2p[i] = 0.f;
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) dW2[i][j][r] = 0.f;
    }
#pragma unroll
    for (int o = 0; o < O; ++o) {
        db3p[o] = 0.f; dlsp[o] = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m) dW3a[o][m] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) st[i] = 0.f;

    const int g = a.layout ? (int)(blockIdx.x % a.G) : (int)(blockIdx.x >> 1);
    const int64_t ntiles_all = (a.count + kTile - 1) / kTile;
    // static priority experiment: co-resident workgroups (g of the actor, g of the critic) get opposite priorities;
    // the high-priority half of each net takes split_pct % of the tiles (deterministic partition)
    int64_t tile0 = 0, ntiles = ntiles_all, tstride = (int64_t)a.G * 4, first = (int64_t)g * 4 + wave;
    if (a.prio == 2) { if (HEAD == HEAD_VALUE) __builtin_amdgcn_s_setprio(1); }          // younger (second-dispatched) workgroups only
    else if (a.prio == 3) { if (HEAD != HEAD_VALUE) __builtin_amdgcn_s_setprio(1); }
    else if (a.prio && a.G >= 2 && (a.G & 1) == 0) {
        const bool hi = ((g & 1) == 0) == (HEAD != HEAD_VALUE);
        const int64_t nh = ntiles_all * a.split_pct / 100;
        tile0 = hi ? 0 : nh; ntiles = hi ? nh : ntiles_all;
        tstride = (int64_t)(a.G / 2) * 4; first = tile0 + (int64_t)(g >> 1) * 4 + wave;
        if (hi) __builtin_amdgcn_s_setprio(1);
    }
    TileIn<O> cur, nxt;
    int64_t tile = first;
    if (tile < ntiles) load_tile<KIND, O, HEAD, REC>(a, tile, ntiles, c, h, cur);
    if (HEAD == HEAD_VALUE && a.stagger > 0) {                      // start the critic workgroups out of phase with their co-resident actor workgroups
        for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(127);      // 127 * 64 clocks each
    }
#ifdef DRIL_STAMPS
    unsigned long long stamp_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev) :: "memory");
#endif
    for (; tile < ntiles; tile += tstride) {
        load_tile<KIND, O, HEAD, REC>(a, tile + tstride, ntiles, c, h, nxt);      // prefetch the next tile's gathers
        unpack_tile<KIND, O, HEAD, REC>(a, h, cur);
        const bool valid = cur.valid;
        float xk[2] = {cur.xk[0], cur.xk[1]};
        STAMP(0);
        // ---- forward ----
        f32x16 h1[MT], h2[MT];
        float out[O], dz[O];
        dense_first<H, MT>(wl + L::W1T, wl + L::B1, xk, h1, lane);
        tanh_tiles(h1);
        STAMP(1);
#pragma unroll
        for (int mo = 0; mo < MT; ++mo) {
            h2[mo] = dense_mfma_tile<MT, true>(wl + L::W2S, L::WS1, wl + L::B2, h1, mo, lane);
            tanh16(h2[mo]);
        }
        store_image<MT>(T, h2, lane);          // early: the LDS write -> read round trip hides under the head below
        STAMP(2);
        dense_out<MT, O, H>(wl + L::W3S, wl + L::B3, h2, out, lane);
        __builtin_amdgcn_sched_barrier(0);
        // ---- loss head (ppo.jl:377-404) and dLoss/dout ----
        loss_head<O, HEAD>(a, cur, out, valid, h == 0, ls, adv_mean, adv_inv, dz, st, dlsp);
        STAMP(3);
        __builtin_amdgcn_sched_barrier(0);
        // ---- output layer backward: dW3 += dz * h2' over samples (h2 read back transposed: hidden on the lane) ----
#pragma unroll
        for (int o = 0; o < O; ++o) { if (h == 0) { db3p[o] += dz[o]; ZI[o * kTS + c] = dz[o]; } }
        {
            f32x16 Bh2[MT];
#pragma unroll
            for (int mj = 0; mj < MT; ++mj) Bh2[mj] = load_operand(T, mj, lane);
#pragma unroll
            for (int o = 0; o < O; ++o) {
                float acc[MT];
#pragma unroll
                for (int mj = 0; mj < MT; ++mj) acc[mj] = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 z = *reinterpret_cast<const f32x4*>(ZI + o * kTS + 16 * h + 4 * q);   // broadcast within the half-wave
#pragma unroll
                    for (int mj = 0; mj < MT; ++mj) {
                        acc[mj] = fmaf(Bh2[mj][4 * q + 0], z[0], acc[mj]); acc[mj] = fmaf(Bh2[mj][4 * q + 1], z[1], acc[mj]);
                        acc[mj] = fmaf(Bh2[mj][4 * q + 2], z[2], acc[mj]); acc[mj] = fmaf(Bh2[mj][4 * q + 3], z[3], acc[mj]);
                    }
                }
#pragma unroll
                for (int mj = 0; mj < MT; ++mj) dW3a[o][mj] += acc[mj];
            }
        }
        STAMP(4);
        __builtin_amdgcn_sched_barrier(0);
        // ---- dz2 = (W3' dz) .* (1 - h2^2), in h2's registers ----
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float dh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int o = 0; o < O; ++o) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(wl + L::W3S + o * H + 32 * m + 8 * q + 4 * h);
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) dh[cc] = fmaf(w[cc], dz[o], dh[cc]);
                }
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) { const float hv = h2[m][4 * q + cc]; h2[m][4 * q + cc] = dh[cc] * (1.0f - hv * hv); }
            }
        STAMP(5);
        __builtin_amdgcn_sched_barrier(0);
        // ---- h1 image (the LDS unit executes a wave's accesses in order, so the Bh2 reads above precede these writes) ----
        store_image<MT>(T, h1, lane);
        // ---- dh1 = W2' dz2 ; dz1 = dh1 .* (1 - h1^2) ----
        f32x16 g1[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            g1[m] = dense_mfma_tile<MT, false>(wl + L::W2T, L::WS2, nullptr, h2, m, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) g1[m][r] = g1[m][r] * (1.0f - h1[m][r] * h1[m][r]);
        }
        STAMP(6);
        __builtin_amdgcn_sched_barrier(0);
        // ---- dW2 += dz2 * h1' ; db2 += rowsum(dz2) ----
        {
            f32x16 Bh[MT];
#pragma unroll
            for (int mj = 0; mj < MT; ++mj) Bh[mj] = load_operand(T, mj, lane);
            store_image<MT>(T, h2, lane);                                      // dz2 image
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                const f32x16 Az = load_operand(T, mi, lane);
                db2p[mi] += sum16(Az);
#pragma unroll
                for (int mj = 0; mj < MT; ++mj) dW2[mi][mj] = mfma_outer(Az, Bh[mj], dW2[mi][mj]);
            }
        }
        STAMP(7);
        __builtin_amdgcn_sched_barrier(0);
        // ---- dW1 | db1 += dz1 * [x; 1]' ----
        store_image<MT>(T, g1, lane);
#pragma unroll
        for (int s = 0; s < 2; ++s) { const int d = 2 * s + h; XI[(d < D ? d : D + 1) * kTS + c] = d < D ? xk[s] : 0.f; }   // branch-free: out-of-range components rewrite the zero row
        {   // v_mfma_f32_16x16x4_f32: M = 16 hidden rows, N = 16 columns [x_0..x_{D-1}, 1, 0...], K = 4 samples per step
            const int j = lane & 15;
            float bx[8];
            load_row8(XI, j <= D ? j : D + 1, lane, bx);
#pragma unroll
            for (int mt = 0; mt < H / 16; ++mt) {
                float az[8];
                load_row8(T, 16 * mt + j, lane, az);
#pragma unroll
                for (int k = 0; k < 8; ++k) dW1[mt] = mfma16(az[k], bx[k], dW1[mt]);
            }
        }
        STAMP(8);
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    }
#ifdef DRIL_STAMPS
    if (lane == 0 && a.dbg) {
        unsigned long long* o = a.dbg + ((size_t)blockIdx.x * 4 + wave) * 12;
        for (int k = 0; k < 10; ++k) o[k] = stamp_acc[k];
        o[10] = (unsigned long long)((ntiles - first + tstride - 1) / tstride); o[11] = HEAD;
    }
#endif

    // ---- epilogue: 4 waves -> one slab (fixed wave order => deterministic) ----
    __syncthreads();
    float* red = smem + L::BWD_END;
    const int SL = HEAD == HEAD_VALUE ? a.slab_c : a.slab_a;
    const int o_w1 = 0, o_b1 = H * D, o_w2 = o_b1 + H, o_b2 = o_w2 + H * H, o_w3 = o_b2 + H, o_b3 = o_w3 + O * H;
    const int o_ls = o_b3 + O, o_st = SL - 8;
    for (int i = tid; i < SL; i += blockDim.x) red[i] = 0.f;
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * mi + rowfn(r, h);
#pragma unroll
                    for (int mj = 0; mj < MT; ++mj) red[o_w2 + row + (32 * mj + c) * H] += dW2[mi][mj][r];
                }
                const float b2 = db2p[mi] + __shfl_xor(db2p[mi], 32);
                if (h == 0) red[o_b2 + 32 * mi + c] += b2;
            }
#pragma unroll
            for (int mt = 0; mt < H / 16; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * mt + 4 * (lane >> 4) + r, col = lane & 15;
                    if (col < D) red[o_w1 + row + col * H] += dW1[mt][r];
                    else if (col == D) red[o_b1 + row] += dW1[mt][r];
                }
#pragma unroll
            for (int o = 0; o < O; ++o) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const float v = dW3a[o][m] + __shfl_xor(dW3a[o][m], 32);      // the two halves hold different samples
                    if (h == 0) red[o_w3 + o + (32 * m + c) * O] += v;
                }
                const float b3 = half_sum(db3p[o]);
                if (lane == 0) red[o_b3 + o] += b3;
                if (HEAD == HEAD_GAUSSIAN) { const float l = half_sum(dlsp[o]); if (lane == 0) red[o_ls + o] += l; }
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) { const float v = half_sum(st[k]); if (lane == 0) red[o_st + k] += v; }
        }
        __syncthreads();
    }
    float* slab = (HEAD == HEAD_VALUE ? a.slabs_critic : a.slabs_actor) + (size_t)g * SL;
    for (int i = tid; i < SL; i += blockDim.x) slab[i] = red[i];
}

template <int KIND, int H, bool REC>
__global__ __launch_bounds__(256, 2) void ppo_grad_kernel(GradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (*a.stop_flag) return;
    constexpr int A = EnvSpec<KIND>::A;
    const bool actor = a.layout ? (blockIdx.x < (unsigned)a.G) : ((blockIdx.x & 1) == 0);
    if (actor) grad_body<KIND, H, A, EnvSpec<KIND>::discrete ? HEAD_CATEGORICAL : HEAD_GAUSSIAN, REC>(a, smem);
    else grad_body<KIND, H, 1, HEAD_VALUE, REC>(a, smem);
}

// =============================================================================================
// ppo_grad_split_kernel — the same fused forward + loss + backward with the three H x H contractions of a tile (L2 forward, dh1 = W2' dz2,
// dW2 += dz2 h1': 192 of the 212 MFMAs of ppo_grad_kernel) on the bf16 matrix cores with fp32-equivalent 3-piece operand splitting
// (dril_device.h: six v_mfma_f32_32x32x16_bf16 per k16 step, f32 accumulate).  Per tile and net 3 x 48 bf16 MFMAs x 32 cycles = 4.6 k cycles of
// matrix pipe instead of 12.3 k cycles of f32 MFMA on the VALU's lanes; the splits (h1 and dz2: 64 elements per lane) cost ~350 VALU instructions.
//
// LDS plan (H = 64; <= 80 KB per workgroup so that an actor and a critic workgroup still share a CU):
//   * ONE weight image serves both W2 (row reads, L2 forward) and W2' (transposed reads, dh1): three pieces of [64 rows = h2 unit][64 cols = h1 unit]
//     bf16, 128-byte rows, the 8-byte chunk ch of row r stored at chunk ch ^ gw(r), gw = bits (r1 r2 r3 r4) of r.  Row reads (ds_read_b64: lanes =
//     32 consecutive rows, one chunk) and transposed reads (ds_read_b64_tr_b16: 4 rows x 8 chunks per half-wave) are both conflict-free on it
//     (tools/lds_layout_check.py).  Staged pre-scaled by kTanhScale like the f32 images; dh1 folds the 1 / kTanhScale into its tanh' mask.
//   * activations enter a product that sums over HIDDEN UNITS straight from registers: the packed pieces of an accumulator tile (registers 8s..8s+7
//     of k-step s) ARE the B operand — element j of lane half h is unit 16s + 8(j>>2) + 4h + (j&3), and the A operand's two 8-byte chunk reads
//     follow that order.
//   * products that sum over SAMPLES (dW2) need the transpose: every lane stores its packed pieces (registers 4g..4g+3 = 8 bytes) at
//     [its sample][unit 8g + 4h] of a per-wave [32 samples][64 units] bf16 image (chunk ^ swap-bits-1,3(sample)), read back with
//     ds_read_b64_tr_b16: 12 KB per wave for three pieces, used in turn for h2 (f32, output-layer gradient), h1', dz2', dz1 (f32, first-layer gradient).
//   * db2 = rowsum(dz2) costs no pass of its own: dz2 = (W3' dz) .* (1 - h2^2), so db2[u] = sum_o W3[o][u] * S[o][u] with
//     S[o][u] = sum_n dz[o][n] (1 - h2[u][n]^2), accumulated beside dW3 (same operands, already in registers) and multiplied by W3 once, in the epilogue.
// =============================================================================================
// a lane constant the optimiser cannot see through: image addresses derived from it are rebuilt per tile (2-3 VALU) instead of being hoisted out of the tile loop as
// loop invariants, where they occupy registers for the whole kernel (ppo_grad_wide_split_kernel: 92 -> 12 spilled registers)
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
template <int D, int H, int O> struct GradScratchSplit {
    static constexpr int T = 0;                              // 12288 bytes: the h1' piece images (three [32 samples][64 units] bf16)
    static constexpr int T3 = T + 3 * 32 * H / 2;            // 12288 bytes: the dz2' piece images; after dW2 has consumed them the dz1 f32 image [H][kTS] (9216 bytes) for dW1
    static constexpr int XI = T3 + 3 * 32 * H / 2;
    static constexpr int SIZE = XI + (D + 2) * kTS;
    static_assert(H * kTS <= 3 * 32 * H / 2, "the f32 image must fit the piece images' space");
};
// MFMA operand (A or B) of unit tile m, k16 step s of a contraction over samples: lane (unit 32m + (lane&31), half h) gets samples 16s + 8h + j
__device__ __forceinline__ int timg_read_base(int lane) {
    const int h = lane >> 5, gm = (lane >> 4) & 1, e = lane & 15, q = e >> 2, p = e & 3;
    return (8 * h + q) * 128 + ((((4 * gm + p) ^ (8 * (q >> 1) + 2 * h + (q & 1))) & 15) << 3);
}
__device__ __forceinline__ bf16x8 load_frag_T(const char* T, int rbase, int piece, int m, int s) {
    const int a = (rbase ^ (64 * m)) + 2048 * s + 4096 * piece;
    return frag8(lds_read_tr16(T, a), lds_read_tr16(T, a ^ (512 | 32)));
}

// ---- the tile loop as six stages --------------------------------------------------------------------------------------------------------------------------
// S1  unpack, prefetch of the next tile, L1 (f32 MFMA, 4)                         S4  per k16 step: dz2 (8 registers), split, piece image, dh1 MFMAs (48); mask -> dz1
// S2  per k16 step: tanh + split of 8 registers of h1, piece image, L2 MFMAs      S5  dW2 (48 bf16 MFMA): both operands as transposed fragments of the piece images
//     (48 bf16 MFMA, both output m-tiles); tanh -> h2                             S6  dz1 f32 image; dW1 as 32 rank-1 updates (v_mfma_f32_4x4x1_16B_f32: lane = hidden unit,
// S3  L3, loss head, per-lane dW3 accumulation                                         4 input components per instruction), db1 = row sum on the VALU
// One wave per SIMD (<= 512 registers).  What the profile of this kernel says (profiles/r02_split_kernel.md): with ONE wave per SIMD the matrix pipe and the VALU do not
// run beside each other — an interleaved instruction stream is issued in order and v_fma / v_exp stall behind the wave's own MFMA (microbenchmark mfma_bf16_valu:
// 16 v_fma after every bf16 MFMA cost 69 instead of 34 + 35 cycles; two waves per SIMD hide them completely) — so a tile costs MFMA + VALU, and the stages below are
// written to execute FEWER VALU / LDS instructions rather than to overlap them: activation pieces are produced a k16 step at a time and go straight to their image
// (12 registers live instead of 96), the output-layer gradient and the bias gradients are per-lane accumulations reduced once in the epilogue (no f32 h2 image,
// no second pass over it), and dW1 runs on the 4x4x1 MFMA (256 instead of 1024 cycles on the VALU's lanes).
template <int MT, int O> struct TileCtx {
    TileIn<O> cur, nxt;
    float xk[2]; bool valid;
    f32x16 h1[MT], h2[MT], g1[MT];
    float dz[O];
    char* Tb; char* T3b; float* T3; float* XI;
};
template <int H, int O, int D> struct SplitAcc {
    static constexpr int MT = H / 32, ND = (D + 3) / 4;
    f32x16 dW2[MT][MT];
    f32x16 dW3[O][MT];        // per lane: sum over this lane's samples of dz[o] * h2[unit]; summed over the 32 lanes of a half in the epilogue
    f32x16 db2[MT];           // per lane: sum of dz2[unit]
    f32x4 dW1[ND][2];         // lane 4b + j, register i: dW1[unit 4b + i][component 4nd + j] (two accumulators: even / odd samples)
    float db1;                // lane = unit
    float db3p[O], dlsp[O], st[5];
};
struct SplitEnv { float* wl; const char* Wimg; int lane, c, h, wf_base, wt_base, tr_base, ts_base; const float* ls; float adv_mean, adv_inv; };

// transposed piece images [32 samples][64 units]: lane (sample c, half h) stores registers 4g..4g+3 of m-tile m (units 32m + 8g + 4h ..+3) as one 8-byte chunk;
// ts_base = c * 128 + (((h ^ gs(c)) & 15) << 3); pc[piece][t] = packed registers (8s + 2t, 8s + 2t + 1) of the k16 step s
__device__ __forceinline__ void store_piece_chunk(char* T, int ts_base, int m, int s, const unsigned (&pc)[3][4]) {
#pragma unroll
    for (int gg = 0; gg < 2; ++gg) {
        const int a = ts_base ^ (64 * m + 16 * (2 * s + gg));
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x2*>(T + 4096 * p + a) = u32x2{pc[p][2 * gg], pc[p][2 * gg + 1]};
    }
}
__device__ __forceinline__ u32x2 lds_read_u32x2(const char* p) { return *reinterpret_cast<const u32x2*>(p); }
// acc += a . b with the accumulator pinned to AGPRs and updated in place
__device__ __forceinline__ void mfma_acc_agpr(f32x16& acc, bf16x8 a, bf16x8 b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// v_mfma_f32_4x4x1_16B_f32: 16 independent 4x4 blocks; lane 4b + i gives A[i] and B[i] of block b, register r of lane 4b + j receives D[r][j]
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }

template <int KIND, int H, int O, int HEAD, bool REC>
struct SplitStages {
    static constexpr int D = EnvSpec<KIND>::D, MT = H / 32, ND = (D + 3) / 4;
    using L = NetLdsSplit<D, H, O>;
    using Ctx = TileCtx<MT, O>;
    using Acc = SplitAcc<H, O, D>;

    static __device__ __forceinline__ void s1(const GradArgs& a, const SplitEnv& e, Ctx& t, int64_t next_tile, int64_t ntiles) {
        t.cur = t.nxt;
        unpack_tile<KIND, O, HEAD, REC>(a, e.h, t.cur);
        load_tile<KIND, O, HEAD, REC>(a, next_tile, ntiles, e.c, e.h, t.nxt);      // prefetch the next tile: consumed one whole tile from now
        t.valid = t.cur.valid; t.xk[0] = t.cur.xk[0]; t.xk[1] = t.cur.xk[1];
        dense_first<H, MT>(e.wl + L::W1T, e.wl + L::B1, t.xk, t.h1, e.lane);        // pre-activations (scaled by kTanhScale)
    }
    static __device__ __forceinline__ void s2(const SplitEnv& e, Ctx& t) {
        f32x16 acc[MT];
#pragma unroll
        for (int mo = 0; mo < MT; ++mo)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(e.wl + L::B2 + 32 * mo + 8 * q + 4 * e.h);
                acc[mo][4 * q + 0] = b[0]; acc[mo][4 * q + 1] = b[1]; acc[mo][4 * q + 2] = b[2]; acc[mo][4 * q + 3] = b[3];
            }
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 Aw[MT][3];                                                   // weight fragments first: their LDS latency runs under the chunk's VALU work
#pragma unroll
                for (int mo = 0; mo < MT; ++mo) {
                    const int a0 = (e.wf_base ^ (64 * mi + 32 * s)) + 4096 * mo;
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        Aw[mo][p] = frag8(lds_read_u32x2(e.Wimg + 8192 * p + a0), lds_read_u32x2(e.Wimg + 8192 * p + (a0 ^ 16)));
                }
                __builtin_amdgcn_sched_barrier(0);                                  // pin the requests here (the scheduler otherwise sinks them to their first use)
                float ex[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) ex[i] = __builtin_amdgcn_exp2f(t.h1[mi][8 * s + i]);
#pragma unroll
                for (int i = 0; i < 8; ++i) ex[i] = __builtin_amdgcn_rcpf(ex[i] + 1.0f);
#pragma unroll
                for (int i = 0; i < 8; ++i) t.h1[mi][8 * s + i] = fmaf(-2.0f, ex[i], 1.0f);
                unsigned pc[3][4];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) split3_pair(t.h1[mi][8 * s + 2 * tt], t.h1[mi][8 * s + 2 * tt + 1], pc[0][tt], pc[1][tt], pc[2][tt]);
                store_piece_chunk(t.Tb, e.ts_base, mi, s, pc);                      // h1' images for dW2
#pragma unroll
                for (int mo = 0; mo < MT; ++mo) acc[mo] = mfma_split6(Aw[mo][0], Aw[mo][1], Aw[mo][2], chunk_frag(pc, 0), chunk_frag(pc, 1), chunk_frag(pc, 2), acc[mo]);
            }
#pragma unroll
        for (int mo = 0; mo < MT; ++mo) { tanh16(acc[mo]); t.h2[mo] = acc[mo]; }
    }
    static __device__ __forceinline__ void s3(const GradArgs& a, const SplitEnv& e, Ctx& t, Acc& A) {
        float out[O];
        dense_out<MT, O, H>(e.wl + L::W3S, e.wl + L::B3, t.h2, out, e.lane);
        loss_head<O, HEAD>(a, t.cur, out, t.valid, e.h == 0, e.ls, e.adv_mean, e.adv_inv, t.dz, A.st, A.dlsp);   // ppo.jl:377-404 and dLoss/dout
#pragma unroll
        for (int o = 0; o < O; ++o) {
            if (e.h == 0) A.db3p[o] += t.dz[o];
#pragma unroll
            for (int m = 0; m < MT; ++m) A.dW3[o][m] += t.dz[o] * t.h2[m];          // dW3[o][unit] += dz[o][sample] h2[unit][sample]: the lane IS the sample
        }
    }
    static __device__ __forceinline__ void s4(const SplitEnv& e, Ctx& t, Acc& A) {  // dz2 = (W3' dz) .* (1 - h2^2); dh1 = W2' dz2 (A: transposed reads of the weight image); dz1 = dh1 .* (1 - h1^2)
        constexpr float kInvTanhScale = 1.0f / kTanhScale;
        f32x16 acc[MT];
#pragma unroll
        for (int mk = 0; mk < MT; ++mk)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mk][r] = 0.f;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 Aw[MT][3];
#pragma unroll
                for (int mk = 0; mk < MT; ++mk) {
                    const int a0 = (e.wt_base ^ (64 * mk + 2048 * s + 8 * s)) + 4096 * mi;
#pragma unroll
                    for (int p = 0; p < 3; ++p) Aw[mk][p] = frag8(lds_read_tr16(e.Wimg, 8192 * p + a0), lds_read_tr16(e.Wimg, 8192 * p + (a0 ^ (1024 | 16))));
                }
                __builtin_amdgcn_sched_barrier(0);
                float z[8];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float dh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int o = 0; o < O; ++o) {
                        const f32x4 w = *reinterpret_cast<const f32x4*>(e.wl + L::W3S + o * H + 32 * mi + 8 * (2 * s + q) + 4 * e.h);
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) dh[cc] = fmaf(w[cc], t.dz[o], dh[cc]);
                    }
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) { const float hv = t.h2[mi][8 * s + 4 * q + cc]; z[4 * q + cc] = dh[cc] * fmaf(-hv, hv, 1.0f); }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) A.db2[mi][8 * s + i] += z[i];
                unsigned pc[3][4];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) split3_pair(z[2 * tt], z[2 * tt + 1], pc[0][tt], pc[1][tt], pc[2][tt]);
                store_piece_chunk(t.T3b, e.ts_base, mi, s, pc);                     // dz2' images for dW2
#pragma unroll
                for (int mk = 0; mk < MT; ++mk) acc[mk] = mfma_split6(Aw[mk][0], Aw[mk][1], Aw[mk][2], chunk_frag(pc, 0), chunk_frag(pc, 1), chunk_frag(pc, 2), acc[mk]);
            }
#pragma unroll
        for (int mk = 0; mk < MT; ++mk)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float t2 = t.h1[mk][r] * t.h1[mk][r]; t.g1[mk][r] = acc[mk][r] * fmaf(-t2, kInvTanhScale, kInvTanhScale); }
    }
    static __device__ __forceinline__ void s5(const SplitEnv& e, Ctx& t, Acc& A) {   // dW2 += dz2 h1' over the 32 samples
        bf16x8 Bf[2][MT][3], Af[2][MT][3];                                             // all 24 fragments are requested before the first MFMA (96 registers; nothing else is live here)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int p = 0; p < 3; ++p) { Bf[s][m][p] = load_frag_T(t.Tb, e.tr_base, p, m, s); Af[s][m][p] = load_frag_T(t.T3b, e.tr_base, p, m, s); }
        __builtin_amdgcn_sched_barrier(0);
        // The dW2 accumulators are touched by nothing but these MFMAs and the epilogue: they live in AGPRs, in place (mfma_acc_agpr).  The rest of the file is compiled
        // with MFMA results in VGPRs (Makefile: -amdgpu-mfma-vgpr-form) because the VALU consumes them; for THESE 64 registers that would mean parking them in AGPRs
        // between tiles and moving them in and out around every S5 (128 v_accvgpr moves per tile).  Round robin over the four accumulators: a dependent MFMA is
        // issued three MFMAs after its predecessor, so no software wait state is needed between them.
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int term = 0; term < 6; ++term) {
                // small terms first, as in mfma_split6: (A piece, B piece) = (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi)
                const int pa = term == 0 ? 2 : (term == 2 || term == 3) ? 1 : 0, pb = term == 1 ? 2 : (term == 2 || term == 4) ? 1 : 0;
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int mj = 0; mj < MT; ++mj) mfma_acc_agpr(A.dW2[mi][mj], Af[s][mi][pa], Bf[s][mj][pb]);
            }
    }
    // dW1 += dz1 x' as rank-1 updates per sample; db1 += row sum of dz1.  In two parts: the images are written right after dW2 (s6a), and read back one stage later, after
    // the NEXT tile's S1 (s6b) — the LDS write -> read round trip runs under that stage instead of stalling the wave
    static __device__ __forceinline__ void s6a(const SplitEnv& e, Ctx& t) {
        store_image<MT>(t.T3, t.g1, e.lane);                                          // the dz1 f32 image takes the dz2' images' place (dW2 has consumed them: same wave, LDS in order)
#pragma unroll
        for (int s = 0; s < 2; ++s) { const int d = 2 * s + e.h; t.XI[(d < D ? d : D + 1) * kTS + e.c] = d < D ? t.xk[s] : 0.f; }   // branch-free: out-of-range components rewrite the zero row
    }
    struct DW1In { float az[32]; f32x4 x[ND][8]; };
    static __device__ __forceinline__ void s6b_load(const SplitEnv& e, Ctx& t, DW1In& in) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(t.T3 + e.lane * kTS + 4 * q);
            in.az[4 * q] = v[0]; in.az[4 * q + 1] = v[1]; in.az[4 * q + 2] = v[2]; in.az[4 * q + 3] = v[3];
        }
#pragma unroll
        for (int nd = 0; nd < ND; ++nd) {
            const int d = 4 * nd + (e.lane & 3);
            const float* xr = t.XI + (d < D ? d : D + 1) * kTS;
#pragma unroll
            for (int q = 0; q < 8; ++q) in.x[nd][q] = *reinterpret_cast<const f32x4*>(xr + 4 * q);
        }
    }
    static __device__ __forceinline__ void s6b_math(const DW1In& in, Acc& A) {
#pragma unroll
        for (int nd = 0; nd < ND; ++nd)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                A.dW1[nd][0] = mfma4(in.az[4 * q], in.x[nd][q][0], A.dW1[nd][0]); A.dW1[nd][1] = mfma4(in.az[4 * q + 1], in.x[nd][q][1], A.dW1[nd][1]);
                A.dW1[nd][0] = mfma4(in.az[4 * q + 2], in.x[nd][q][2], A.dW1[nd][0]); A.dW1[nd][1] = mfma4(in.az[4 * q + 3], in.x[nd][q][3], A.dW1[nd][1]);
            }
        float s8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) s8[i] = (in.az[i] + in.az[8 + i]) + (in.az[16 + i] + in.az[24 + i]);
        A.db1 += ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    }
};

template <int KIND, int H, int O, int HEAD, bool REC>
__device__ __forceinline__ void grad_body_split(const GradArgs& a, float* smem) {
    constexpr int D = EnvSpec<KIND>::D, MT = H / 32, NCTX = 1;
    using L = NetLdsSplit<D, H, O>;
    using SC = GradScratchSplit<D, H, O>;
    using ST = SplitStages<KIND, H, O, HEAD, REC>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const NetOff off = HEAD == HEAD_VALUE ? a.critic : a.actor;
    float* wl = smem;
    float* scratch = smem + L::END + wave * (NCTX * SC::SIZE);
    stage_net_split<D, H, O>(wl, a.params, off, tid, blockDim.x);
    for (int i = lane; i < (D + 2) * kTS; i += 64) scratch[SC::XI + i] = (i / kTS == D) ? 1.0f : 0.0f;
    for (int i = lane; i < H * kTS; i += 64) scratch[SC::T3 + i] = 0.0f;     // the first tile's S6b reads the (empty) dz1 image of "the tile before"
    __syncthreads();

    // advantage normalisation constants (ppo.jl:350-356): mean, corrected std, eps added to the std
    float adv_mean = 0.f, adv_den = 1.f;
    if (HEAD != HEAD_VALUE && a.normalize_adv) {
        double s, q, n;
        if (a.inline_moments) {
            double* shd = reinterpret_cast<double*>(smem + ((L::END + 1) & ~1));      // the first 4 KB of wave 0's T image, not yet in use
            double ls_ = 0, lq_ = 0;
            for (int64_t i2 = tid; i2 < a.count; i2 += blockDim.x) {
                const int64_t p2 = a.pos0 + i2;
                const int64_t gi = a.perm ? a.perm[p2] : (a.perm_bits ? perm_index(p2, a.N, a.perm_key, a.perm_bits) : p2);
                const int64_t li2 = gi - a.idx_lo;
                if (li2 >= 0 && li2 < a.n_local) { const float v = REC ? a.rec[2 * li2 + 1].y : a.adv[li2]; ls_ += v; lq_ += (double)v * v; }
            }
            shd[tid] = ls_; shd[256 + tid] = lq_;
            __syncthreads();
            for (int st_ = 128; st_ > 0; st_ >>= 1) { if (tid < st_) { shd[tid] += shd[tid + st_]; shd[256 + tid] += shd[256 + tid + st_]; } __syncthreads(); }
            s = shd[0]; q = shd[256]; n = (double)a.count;
            __syncthreads();
        } else { s = a.adv_stats[0]; q = a.adv_stats[1]; n = a.adv_stats[2]; }
        const double mean = s / n;
        double var = (q - s * mean) / (n - 1.0);
        if (var < 0) var = 0;
        adv_mean = (float)mean; adv_den = (float)sqrt(var) + 1.0e-8f;
    }
    float lsr[kLsMax];
#pragma unroll
    for (int o = 0; o < kLsMax; ++o) lsr[o] = 0.f;
    if (HEAD == HEAD_GAUSSIAN) {
#pragma unroll
        for (int o = 0; o < O; ++o) lsr[o] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.params[a.log_std_off + o])));
    }
    SplitEnv e;
    e.wl = wl; e.Wimg = reinterpret_cast<const char*>(smem + L::W2P); e.lane = lane; e.c = c; e.h = h; e.ls = lsr;
    e.adv_mean = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, adv_mean)));
    e.adv_inv = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, 1.0f / adv_den)));
    // lane constants of the LDS images
    e.wf_base = c * 128 + (((h ^ w2img_gw(c)) & 15) << 3);                                  // W2 row read: row 32 mo + c, chunk (8 mi + 4 s + 2 rho + h) ^ gw
    { const int e16 = lane & 15, tq = e16 >> 2, tp = e16 & 3, tg = (lane >> 4) & 1;
      e.wt_base = (4 * h + tq) * 128 + ((((4 * tg + tp) ^ (8 * (tq >> 1) + 4 * h)) & 15) << 3); }   // W2' transposed read: rows 32 mi + 16 s + 8 rho + 4 h + q, chunk 8 mk + 4 g + p
    e.tr_base = timg_read_base(lane);
    e.ts_base = c * 128 + (((h ^ timg_gs(c)) & 15) << 3);

    SplitAcc<H, O, D> A;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) A.db2[i][r] = 0.f;
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) A.dW2[i][j][r] = 0.f;
    }
#pragma unroll
    for (int nd = 0; nd < (D + 3) / 4; ++nd) { A.dW1[nd][0] = f32x4{0.f, 0.f, 0.f, 0.f}; A.dW1[nd][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    A.db1 = 0.f;
#pragma unroll
    for (int o = 0; o < O; ++o) {
        A.db3p[o] = 0.f; A.dlsp[o] = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) A.dW3[o][m][r] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) A.st[i] = 0.f;

    const int g = HEAD == HEAD_VALUE ? (int)blockIdx.x - a.G : (int)blockIdx.x;            // the first G workgroups run the actor, the next Gc the critic
    const int64_t ntiles = (a.count + kTile - 1) / kTile;
    const int64_t tstride = (int64_t)(HEAD == HEAD_VALUE ? a.Gc : a.G) * 4, first = (int64_t)g * 4 + wave;
    typename ST::Ctx ta;
    ta.Tb = reinterpret_cast<char*>(scratch + SC::T); ta.T3 = scratch + SC::T3; ta.T3b = reinterpret_cast<char*>(ta.T3); ta.XI = scratch + SC::XI;
    {
        int64_t tile = first;
        load_tile<KIND, O, HEAD, REC>(a, tile, ntiles, c, h, ta.nxt);       // a tile index past the end loads an all-invalid tile
#ifdef DRIL_STAMPS
        unsigned long long stamp_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev) :: "memory");
#endif
        for (; tile < ntiles; tile += tstride) {
            typename ST::DW1In w1;
            ST::s6b_load(e, ta, w1); __builtin_amdgcn_sched_barrier(0);               // dW1 of the previous tile (zeros before the first): operands requested before S1, consumed after it
            ST::s1(a, e, ta, tile + tstride, ntiles); __builtin_amdgcn_sched_barrier(0); STAMP(0);
            ST::s6b_math(w1, A); __builtin_amdgcn_sched_barrier(0); STAMP(5);
            ST::s2(e, ta); __builtin_amdgcn_sched_barrier(0); STAMP(1);
            ST::s3(a, e, ta, A); __builtin_amdgcn_sched_barrier(0); STAMP(2);
            ST::s4(e, ta, A); __builtin_amdgcn_sched_barrier(0); STAMP(3);
            ST::s5(e, ta, A); __builtin_amdgcn_sched_barrier(0); STAMP(4);
            ST::s6a(e, ta); __builtin_amdgcn_sched_barrier(0);
        }
        if (first < ntiles) { typename ST::DW1In w1; ST::s6b_load(e, ta, w1); ST::s6b_math(w1, A); }   // the last tile's dW1
#ifdef DRIL_STAMPS
        if (lane == 0 && a.dbg) {
            unsigned long long* o = a.dbg + ((size_t)blockIdx.x * 4 + wave) * 12;
            for (int k = 0; k < 10; ++k) o[k] = stamp_acc[k];
            o[10] = (unsigned long long)((ntiles - first + tstride - 1) / tstride); o[11] = HEAD;
        }
#endif
    }

    // ---- epilogue: 4 waves -> one slab (fixed wave order => deterministic) ----
    __syncthreads();
    float* red = smem + L::END;
    const int SL = HEAD == HEAD_VALUE ? a.slab_c : a.slab_a;
    const int o_w1 = 0, o_b1 = H * D, o_w2 = o_b1 + H, o_b2 = o_w2 + H * H, o_w3 = o_b2 + H, o_b3 = o_w3 + O * H;
    const int o_ls = o_b3 + O, o_st = SL - 8;
    for (int i = tid; i < SL; i += blockDim.x) red[i] = 0.f;
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * mi + rowfn(r, h);
#pragma unroll
                    for (int mj = 0; mj < MT; ++mj) red[o_w2 + row + (32 * mj + c) * H] += A.dW2[mi][mj][r];
                }
#pragma unroll
            for (int nd = 0; nd < (D + 3) / 4; ++nd)
#pragma unroll
                for (int i = 0; i < 4; ++i) {                                          // 4x4x1 blocks: lane 4b + j, register i = dW1[unit 4b + i][component 4 nd + j]
                    const int row = 4 * (lane >> 2) + i, col = 4 * nd + (lane & 3);
                    if (col < D) red[o_w1 + row + col * H] += A.dW1[nd][0][i] + A.dW1[nd][1][i];
                }
            red[o_b1 + lane] += A.db1;                                                 // lane = unit
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {                                         // per-lane sums over samples -> sum over the 32 lanes of each half (the halves hold different units)
                    const int unit = 32 * m + rowfn(r, h);
                    const float b2 = half_sum(A.db2[m][r]);
                    if (c == 0) red[o_b2 + unit] += b2;
#pragma unroll
                    for (int o = 0; o < O; ++o) { const float v = half_sum(A.dW3[o][m][r]); if (c == 0) red[o_w3 + o + unit * O] += v; }
                }
#pragma unroll
            for (int o = 0; o < O; ++o) {
                const float b3 = half_sum(A.db3p[o]);
                if (lane == 0) red[o_b3 + o] += b3;
                if (HEAD == HEAD_GAUSSIAN) { const float l = half_sum(A.dlsp[o]); if (lane == 0) red[o_ls + o] += l; }
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) { const float v = half_sum(A.st[k]); if (lane == 0) red[o_st + k] += v; }
        }
        __syncthreads();
    }
    float* slab = (HEAD == HEAD_VALUE ? a.slabs_critic : a.slabs_actor) + (size_t)g * SL;
    for (int i = tid; i < SL; i += blockDim.x) slab[i] = red[i];
}

// one 4-wave workgroup per CU (<= 512 registers per wave)
template <int KIND, int H, bool REC>
__global__ __launch_bounds__(256, 1) void ppo_grad_split_kernel(GradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (*a.stop_flag) return;
    constexpr int A = EnvSpec<KIND>::A;
    const bool actor = blockIdx.x < (unsigned)a.G;
    if (actor) grad_body_split<KIND, H, A, EnvSpec<KIND>::discrete ? HEAD_CATEGORICAL : HEAD_GAUSSIAN, REC>(a, smem);
    else grad_body_split<KIND, H, 1, HEAD_VALUE, REC>(a, smem);
}

// =============================================================================================
// ppo_grad_wide_kernel — the same fused forward + loss + backward for hidden widths that do not fit one wave
// (H = 256: W2 is 256 KB, dW2 is 64 K accumulators).  A workgroup of H/32 waves processes a 32-sample tile TOGETHER:
// wave w owns m-tile w (hidden rows 32w..32w+31) of every layer and the 32 x H slice of dW2 in registers (H/2 VGPRs);
// activations are exchanged through LDS (B-operand images [m][lane][16]), the two H x H operand streams W2 / W2' come
// pre-tiled from L2 (dril_device.h "wide nets").  Four workgroup barriers per tile.  Per tile and wave:
//   2 (L1) + 4*MT (L2) + 4*MT (dh1) + 4*MT... in 32x32x2 units: L2 16*MT, dh1 16*MT, dW2 16*MT, dW1 1.
// Every wave owns distinct rows of every gradient, so the slab is written straight from registers (no cross-wave sum).
// =============================================================================================
template <int D, int H, int O> struct WideScratch {
    static constexpr int MT = H / 32;
    static constexpr int SMALL = NetLdsSmall<D, H, O>::END;
    static constexpr int XA = (SMALL + 3) / 4 * 4;          // h1 as B-operand image [MT][64][16]
    static constexpr int XB = XA + MT * 1024;               // dz2 as B-operand image
    static constexpr int TA = XB + MT * 1024;               // h1 transposed [H][kTS]
    static constexpr int TB = TA + H * kTS;                 // per-wave rows: h2', then dz2', then dz1'
    static constexpr int XI = TB + H * kTS;                 // [D+2][kTS]
    static constexpr int ZI = XI + (D + 2) * kTS;           // [MT waves][O][kTS]
    static constexpr int PO = ZI + MT * O * kTS;            // [MT waves][O][32] output-layer partial sums
    static constexpr int SIZE = PO + MT * O * 32;
};

__device__ __forceinline__ void store_breg(float* img, int m, const f32x16& x, int lane) {
    float* p = img + ((size_t)m * 64 + lane) * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(p + 4 * q) = f32x4{x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]};
}
__device__ __forceinline__ f32x16 load_breg(const float* img, int m, int lane) {
    const float* p = img + ((size_t)m * 64 + lane) * 16;
    f32x16 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * q); v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3]; }
    return v;
}
__device__ __forceinline__ void store_image_tile(float* img, int m, const f32x16& x, int lane) {
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) img[(32 * m + rowfn(r, h)) * kTS + c] = x[r];
}
// Y tile mo = W * X with W from the pre-tiled global image and X read tile by tile from an LDS B-operand image
__device__ __forceinline__ void wide_preload(const float* __restrict__ wimg, int MTv, int mo, int lane, f32x4 (&af)[4]) {
    const float* base = wimg + ((size_t)mo * MTv * 4 * 64 + lane) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) af[q] = *reinterpret_cast<const f32x4*>(base + (size_t)q * 256);
}
template <int MT, bool BIAS>
__device__ __forceinline__ f32x16 dense_tile_global_ldsB(const float* __restrict__ wimg, const float* __restrict__ bias, const float* __restrict__ ximg, int mo, int lane, f32x4 (&af)[4]) {
    const int h = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if (BIAS) b = *reinterpret_cast<const f32x4*>(bias + 32 * mo + 8 * q + 4 * h);
        acc[4 * q + 0] = b[0]; acc[4 * q + 1] = b[1]; acc[4 * q + 2] = b[2]; acc[4 * q + 3] = b[3];
    }
    const float* base = wimg + ((size_t)mo * MT * 4 * 64 + lane) * 4;
    // The A fragments stream from L2 (pre-tiled image, 1 KiB contiguous per wave-instruction).  Each of the four fragment registers is refilled
    // with the NEXT m-tile's fragment right after the four MFMAs that consumed it were issued, so every load has 12-16 MFMAs (~1 k cycles) of
    // cover with only 16 registers of buffering (the loop stays rolled: fully unrolled, hipcc hoists all 32 loads and spills 330 VGPRs)
    // af[] arrives preloaded with the first m-tile's fragments (wide_preload, issued before the workgroup barrier that precedes this chain)
#pragma unroll 1
    for (int mi = 0; mi < MT; ++mi) {
        const f32x16 X = load_breg(ximg, mi, lane);
        const float* nextp = base + (size_t)((mi + 1 < MT ? mi + 1 : mi) * 4) * 256;   // last iteration re-reads its own fragments (in bounds, unused)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc = mfma32(af[q][0], X[4 * q + 0], acc); acc = mfma32(af[q][1], X[4 * q + 1], acc);
            acc = mfma32(af[q][2], X[4 * q + 2], acc); acc = mfma32(af[q][3], X[4 * q + 3], acc);
            af[q] = *reinterpret_cast<const f32x4*>(nextp + (size_t)q * 256);
        }
    }
    return acc;
}

template <int KIND, int H, int O, int HEAD, bool REC>
__device__ __forceinline__ void grad_body_wide(const GradArgs& a, float* smem) {
    constexpr int D = EnvSpec<KIND>::D, MT = H / 32;
    using L = NetLdsSmall<D, H, O>;
    using SC = WideScratch<D, H, O>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);          // this wave's m-tile
    const int c = lane & 31, h = lane >> 5;
    const NetOff off = HEAD == HEAD_VALUE ? a.critic : a.actor;
    const float* w2a = HEAD == HEAD_VALUE ? a.w2a_critic : a.w2a_actor;
    const float* w2ta = HEAD == HEAD_VALUE ? a.w2ta_critic : a.w2ta_actor;
    float* wl = smem;
    float* XA = smem + SC::XA; float* XB = smem + SC::XB; float* TA = smem + SC::TA; float* TB = smem + SC::TB;
    float* XI = smem + SC::XI; float* ZI = smem + SC::ZI + w * O * kTS; float* PO = smem + SC::PO;
    stage_net_small<D, H, O>(wl, a.params, off, tid, blockDim.x);
    for (int i = tid; i < (D + 2) * kTS; i += blockDim.x) XI[i] = (i / kTS == D) ? 1.0f : 0.0f;
    __syncthreads();

    float adv_mean = 0.f, adv_den = 1.f;
    if (HEAD != HEAD_VALUE && a.normalize_adv) {
        const double s = a.adv_stats[0], q = a.adv_stats[1], n = a.adv_stats[2];
        const double mean = s / n;
        double var = (q - s * mean) / (n - 1.0);
        if (var < 0) var = 0;
        adv_mean = (float)mean; adv_den = (float)sqrt(var) + 1.0e-8f;
    }
    adv_mean = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, adv_mean)));
    const float adv_inv = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, 1.0f / adv_den)));
    constexpr bool LS_GAUSS = HEAD == HEAD_GAUSSIAN; constexpr int LS_N = O;
    // log_std hoisted into scalar registers: a per-tile global load would sit in the in-order vmcnt queue between the prefetched
    // gathers and their first use and drain them every tile (Pendulum [64,64]: 91 -> TFLOP/s below)
    float lsr[kLsMax];
#pragma unroll
    for (int o = 0; o < kLsMax; ++o) lsr[o] = 0.f;
    if (LS_GAUSS) {
#pragma unroll
        for (int o = 0; o < LS_N; ++o) lsr[o] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.params[a.log_std_off + o])));
    }
    const float* ls = lsr;

    f32x16 dW2[MT];                                                  // rows 32w.., all H columns
    f32x4 dW1[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    float dW3a[O], db2p = 0.f, db3p[O], dlsp[O], st[5];
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) dW2[j][r] = 0.f;
#pragma unroll
    for (int o = 0; o < O; ++o) { dW3a[o] = 0.f; db3p[o] = 0.f; dlsp[o] = 0.f; }
#pragma unroll
    for (int i = 0; i < 5; ++i) st[i] = 0.f;

    const int g = a.layout ? (int)(blockIdx.x % a.G) : (int)(blockIdx.x >> 1);
    const int64_t ntiles = (a.count + kTile - 1) / kTile;
    TileIn<O> cur, nxt;
    int64_t tile = g;
    if (tile < ntiles) load_tile<KIND, O, HEAD, REC>(a, tile, ntiles, c, h, cur);
#ifdef DRIL_STAMPS
    unsigned long long stamp_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev) :: "memory");
#endif
    for (; tile < ntiles; tile += a.G) {
        unpack_tile<KIND, O, HEAD, REC>(a, h, cur);
        const bool valid = cur.valid;
        const float xk[2] = {cur.xk[0], cur.xk[1]};
        // ---- S2: h1 tile w ----
        f32x16 h1w;
        {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(wl + L::B1 + 32 * w + 8 * q + 4 * h);
                h1w[4 * q + 0] = b[0]; h1w[4 * q + 1] = b[1]; h1w[4 * q + 2] = b[2]; h1w[4 * q + 3] = b[3];
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) h1w = mfma32(wl[L::W1T + (2 * s + h) * H + 32 * w + c], xk[s], h1w);
            tanh16(h1w);
        }
        store_breg(XA, w, h1w, lane);
        store_image_tile(TA, w, h1w, lane);
        if (w == 0) {
#pragma unroll
            for (int s = 0; s < 2; ++s) { const int d = 2 * s + h; XI[(d < D ? d : D + 1) * kTS + c] = d < D ? xk[s] : 0.f; }   // branch-free: out-of-range components rewrite the zero row
        }
        STAMP(0);
        f32x4 afw[4];
        wide_preload(w2a, MT, w, lane, afw);                                          // first W2 fragments in flight across the barrier
        __syncthreads();                                                              // B1: XA, TA, XI complete
        STAMP(1);
        // prefetch the next tile's record only now: issued before unpack_tile(cur) it sat behind cur's loads in the in-order vmcnt queue and the
        // spill reloads' s_waitcnt vmcnt(0) made every tile wait for a full gather latency (stamps: 10 k cycles in this phase)
        load_tile<KIND, O, HEAD, REC>(a, tile + a.G, ntiles, c, h, nxt);
        // ---- S3: h2 tile w ----
        f32x16 h2w = dense_tile_global_ldsB<MT, true>(w2a, wl + L::B2, XA, w, lane, afw);
        tanh16(h2w);
        STAMP(2);
        // ---- S4: output layer: partial over this wave's rows, summed across waves through LDS ----
        float out[O], dz[O];
#pragma unroll
        for (int o = 0; o < O; ++o) {
            float p = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + L::W3S + o * H + 32 * w + 8 * q + 4 * h);
                p = fmaf(wv[0], h2w[4 * q + 0], p); p = fmaf(wv[1], h2w[4 * q + 1], p);
                p = fmaf(wv[2], h2w[4 * q + 2], p); p = fmaf(wv[3], h2w[4 * q + 3], p);
            }
            p += __shfl_xor(p, 32);
            if (h == 0) PO[(w * O + o) * 32 + c] = p;
        }
        store_image_tile(TB, w, h2w, lane);                                            // h2' (own rows; only this wave reads them)
        __syncthreads();                                                              // B2: PO complete
        STAMP(3);
#pragma unroll
        for (int o = 0; o < O; ++o) {
            float v = wl[L::B3 + o];
#pragma unroll
            for (int ww = 0; ww < MT; ++ww) v += PO[(ww * O + o) * 32 + c];            // fixed order: every wave gets the same bits
            out[o] = v;
        }
        loss_head<O, HEAD>(a, cur, out, valid, h == 0 && w == 0, ls, adv_mean, adv_inv, dz, st, dlsp);
        // ---- dW3 (own rows) ----
#pragma unroll
        for (int o = 0; o < O; ++o) { if (h == 0) { if (w == 0) db3p[o] += dz[o]; ZI[o * kTS + c] = dz[o]; } }
        {
            const f32x16 Bh2 = load_operand(TB, w, lane);
#pragma unroll
            for (int o = 0; o < O; ++o) {
                float acc = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 z = *reinterpret_cast<const f32x4*>(ZI + o * kTS + 16 * h + 4 * q);
                    acc = fmaf(Bh2[4 * q + 0], z[0], acc); acc = fmaf(Bh2[4 * q + 1], z[1], acc);
                    acc = fmaf(Bh2[4 * q + 2], z[2], acc); acc = fmaf(Bh2[4 * q + 3], z[3], acc);
                }
                dW3a[o] += acc;
            }
        }
        // ---- dz2 tile w (in h2w's registers) ----
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float dh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int o = 0; o < O; ++o) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + L::W3S + o * H + 32 * w + 8 * q + 4 * h);
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) dh[cc] = fmaf(wv[cc], dz[o], dh[cc]);
            }
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) { const float hv = h2w[4 * q + cc]; h2w[4 * q + cc] = dh[cc] * (1.0f - hv * hv); }
        }
        store_breg(XB, w, h2w, lane);
        store_image_tile(TB, w, h2w, lane);                                            // dz2' (after the Bh2 read: same wave, LDS in order)
        wide_preload(w2ta, MT, w, lane, afw);                                         // first W2' fragments in flight across the barrier
        STAMP(4);
        __syncthreads();                                                              // B3: XB complete
        STAMP(5);
        // ---- S6: dh1 tile w = W2' dz2 ; dz1 ----
        f32x16 g1 = dense_tile_global_ldsB<MT, false>(w2ta, nullptr, XB, w, lane, afw);
        {
            const f32x16 h1r = load_breg(XA, w, lane);                                 // h1 tile w re-read from its LDS image: 16 registers less across the two MFMA chains
#pragma unroll
            for (int r = 0; r < 16; ++r) g1[r] = g1[r] * (1.0f - h1r[r] * h1r[r]);
        }
        STAMP(6);
        // ---- S7: dW2[rows of w][:] += dz2 h1' ----
        {
            const f32x16 Az = load_operand(TB, w, lane);
            db2p += sum16(Az);
#pragma unroll
            for (int mj = 0; mj < MT; ++mj) {
                const f32x16 Bh = load_operand(TA, mj, lane);
                dW2[mj] = mfma_outer(Az, Bh, dW2[mj]);
            }
        }
        STAMP(7);
        // ---- S8: dW1 | db1 (own rows) ----
        store_image_tile(TB, w, g1, lane);
        {
            const int j = lane & 15;
            float bx[8];
            load_row8(XI, j <= D ? j : D + 1, lane, bx);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float az[8];
                load_row8(TB, 32 * w + 16 * t + j, lane, az);
#pragma unroll
                for (int k = 0; k < 8; ++k) dW1[t] = mfma16(az[k], bx[k], dW1[t]);
            }
        }
        __syncthreads();                                                              // B4: XA/TA/XB/PO/XI free for the next tile
        STAMP(8);
        cur = nxt;
    }
#ifdef DRIL_STAMPS
    if (lane == 0 && a.dbg) {
        unsigned long long* o_ = a.dbg + ((size_t)(blockIdx.x % (2 * a.G)) * 4 + (w & 3)) * 12;
        if (w < 4) { for (int k = 0; k < 10; ++k) o_[k] = stamp_acc[k]; o_[10] = (unsigned long long)((ntiles - g + a.G - 1) / a.G); o_[11] = HEAD; }
    }
#endif

    // ---- epilogue: every wave owns distinct gradient rows -> straight to the workgroup's slab ----
    const int SL = HEAD == HEAD_VALUE ? a.slab_c : a.slab_a;
    const int o_w1 = 0, o_b1 = H * D, o_w2 = o_b1 + H, o_b2 = o_w2 + H * H, o_w3 = o_b2 + H, o_b3 = o_w3 + O * H;
    const int o_ls = o_b3 + O, o_st = SL - 8;
    float* slab = (HEAD == HEAD_VALUE ? a.slabs_critic : a.slabs_actor) + (size_t)g * SL;
#pragma unroll
    for (int mj = 0; mj < MT; ++mj)
#pragma unroll
        for (int r = 0; r < 16; ++r) slab[o_w2 + 32 * w + rowfn(r, h) + (32 * mj + c) * H] = dW2[mj][r];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 32 * w + 16 * t + 4 * (lane >> 4) + r, col = lane & 15;
            if (col < D) slab[o_w1 + row + col * H] = dW1[t][r];
            else if (col == D) slab[o_b1 + row] = dW1[t][r];
        }
    { const float b2 = db2p + __shfl_xor(db2p, 32); if (h == 0) slab[o_b2 + 32 * w + c] = b2; }
#pragma unroll
    for (int o = 0; o < O; ++o) {
        const float v = dW3a[o] + __shfl_xor(dW3a[o], 32);
        if (h == 0) slab[o_w3 + o + (32 * w + c) * O] = v;
        const float b3 = half_sum(db3p[o]);
        if (w == 0 && lane == 0) slab[o_b3 + o] = b3;
        if (HEAD == HEAD_GAUSSIAN) { const float l = half_sum(dlsp[o]); if (w == 0 && lane == 0) slab[o_ls + o] = l; }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) { const float v = half_sum(st[k]); if (w == 0 && lane == 0) slab[o_st + k] = v; }
    if (w == 0 && lane < 3) slab[o_st + 5 + lane] = 0.f;
    for (int i = (HEAD == HEAD_GAUSSIAN ? o_ls + O : o_ls) + tid; i < o_st; i += blockDim.x) slab[i] = 0.f;   // padding
}

template <int KIND, int H, bool REC>
__global__ __launch_bounds__(H * 2, 2) void ppo_grad_wide_kernel(GradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (*a.stop_flag) return;
    constexpr int A = EnvSpec<KIND>::A;
    const bool actor = a.layout ? (blockIdx.x < (unsigned)a.G) : ((blockIdx.x & 1) == 0);
    if (actor) grad_body_wide<KIND, H, A, EnvSpec<KIND>::discrete ? HEAD_CATEGORICAL : HEAD_GAUSSIAN, REC>(a, smem);
    else grad_body_wide<KIND, H, 1, HEAD_VALUE, REC>(a, smem);
}

// =============================================================================================
// ppo_grad_wide_split_kernel — ppo_grad_wide_kernel with its three H x H contractions on the bf16 matrix cores (fp32-equivalent 3-piece operand
// splitting, dril_device.h).  Same decomposition (a workgroup of H/32 waves owns a 32-sample tile, wave w the m-tile w of every layer and the
// 32 x H slice of dW2), same four workgroup barriers per tile; what changes is the operand plumbing:
//   * W2 / W2' stream from L2 as PRE-SPLIT bf16 fragments (build_wimg_split_kernel, once per optimiser step): [(mo*MT + mi)*2 + s][piece][lane][8 bf16],
//     one 16-byte load per lane, piece and k16 step; 1.5 x the bytes of the f32 stream for a third of the matrix-pipe time.
//   * activations: every wave splits its own 16 registers once and writes the packed pieces into ONE workgroup image per activation set,
//     [piece][32 samples][H units] bf16, 16-byte chunk ch of row n stored at ch ^ g(n), g(n) = ((n & 3) << 2) | ((n >> 2) & 3).  The same image gives the
//     B operand of a product that sums over units (ds_read_b128 along the row: 8 consecutive units of one sample) and both operands of the product
//     that sums over samples (ds_read_b64_tr_b16: 4 samples x 16 units per 16-lane group); row reads, transposed reads and the 8-byte stores are all
//     bank-conflict-free under that swizzle (the 4 rows of a transposed read land in the 4 different 64-byte windows, 16 consecutive rows in 16 different chunks).
//     Two images (h1, dz2) of 192 H bytes replace the four f32 images XA, XB, TA and half of TB.
//   * no AGPRs: at two waves per SIMD the allocator gives a function that uses ANY AGPR only 128 VGPRs; the 128 dW2 accumulators are VGPR-form MFMA results like the rest.
// =============================================================================================
template <int D, int H, int O> struct WideSplitScratch {
    static constexpr int MT = H / 32;
    static constexpr int SMALL = NetLdsSmall<D, H, O>::END;
    static constexpr int P1 = (SMALL + 3) / 4 * 4;          // h1 pieces: 3 x 32 x H bf16 = 48 H floats
    static constexpr int P2 = P1 + 48 * H;                  // dz2 pieces
    static constexpr int TB = P2 + 48 * H;                  // per-wave rows [H][kTS] f32: h2', then dz2' (bias gradient), then dz1'
    static constexpr int XI = TB + H * kTS;                 // [D+2][kTS]
    static constexpr int ZI = XI + (D + 2) * kTS;           // [MT waves][O][kTS]
    static constexpr int PO = ZI + MT * O * kTS;            // [MT waves][O][32] output-layer partial sums
    static constexpr int SIZE = PO + MT * O * 32;
};
// chunk swizzle of the piece images.  Rows of >= 256 bytes (H >= 128) alias in every bank: 16 consecutive rows must land in 16 different 16-byte chunks and the 4 rows of a
// transposed read in the 4 different 64-byte windows.  128-byte rows (H = 64): rows n and n + 1 already sit in different halves of the 256-byte bank window, so 3 bits suffice
// (and only 8 chunks exist): bit 2 (the 64-byte window) from n bit 1, bits 0-1 from n bits 2-3.
template <int H> __device__ __forceinline__ int wimg_g(int n) { return H >= 128 ? (((n & 3) << 2) | ((n >> 2) & 3)) : ((((n >> 1) & 1) << 2) | ((n >> 2) & 3)); }

// pre-split fragment streams of one net: forward A[i][k] = kTanhScale W2[32mo + i][k], reverse A[i][k] = W2[k][32mo + i]; k = 32mi + 16s + 8(lane>>5) + j
__global__ void build_wimg_split_kernel(const float* __restrict__ P, NetOff off, int H, u32x4* __restrict__ w2p, u32x4* __restrict__ w2tp) {
    const int MT = H / 32, total = MT * MT * 2 * 64;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int lane = idx & 63, s = (idx >> 6) & 1, mi = (idx >> 7) % MT, mo = (idx >> 7) / MT;
        const int i = 32 * mo + (lane & 31), k0 = 32 * mi + 16 * s + 8 * (lane >> 5);
        unsigned f[3][4], b[3][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = k0 + 2 * t;
            split3_pair(kTanhScale * P[off.w2 + i + (size_t)k * H], kTanhScale * P[off.w2 + i + (size_t)(k + 1) * H], f[0][t], f[1][t], f[2][t]);   // W2[o][k] (column-major out x in)
            split3_pair(P[off.w2 + k + (size_t)i * H], P[off.w2 + k + 1 + (size_t)i * H], b[0][t], b[1][t], b[2][t]);                               // W2'[i][k] = W2[k][i]
        }
        const size_t base = ((size_t)((mo * MT + mi) * 2 + s) * 3) * 64 + lane;
#pragma unroll
        for (int p = 0; p < 3; ++p) { w2p[base + (size_t)p * 64] = u32x4{f[p][0], f[p][1], f[p][2], f[p][3]}; w2tp[base + (size_t)p * 64] = u32x4{b[p][0], b[p][1], b[p][2], b[p][3]}; }
    }
}

// split the 16 registers of m-tile w (accumulator layout) and store the packed pieces: registers 4g..4g+3 = units 32w + 8g + 4h .. +3 of sample c = one 8-byte chunk
template <int H>
__device__ __forceinline__ void store_tile_pieces(char* pimg, int w, const f32x16& x, int lane) {
    constexpr int RB = 2 * H, PS = 32 * RB;
    const int c = lane & 31, h = lane >> 5, rowb = c * RB + 8 * h, gsw = wimg_g<H>(c);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        unsigned hi[2], mid[2], lo[2];
        split3_pair(x[4 * g], x[4 * g + 1], hi[0], mid[0], lo[0]); split3_pair(x[4 * g + 2], x[4 * g + 3], hi[1], mid[1], lo[1]);
        const int a = rowb + (((4 * w + g) ^ gsw) << 4);
        *reinterpret_cast<u32x2*>(pimg + a) = u32x2{hi[0], hi[1]}; *reinterpret_cast<u32x2*>(pimg + PS + a) = u32x2{mid[0], mid[1]}; *reinterpret_cast<u32x2*>(pimg + 2 * PS + a) = u32x2{lo[0], lo[1]};
    }
}
// the inverse of store_tile_pieces for the lane's own chunks: x = hi + mid + lo (exact)
template <int H>
__device__ __forceinline__ void load_tile_pieces(const char* pimg, int w, f32x16& x, int lane) {
    constexpr int RB = 2 * H, PS = 32 * RB;
    const int c = lane & 31, h = lane >> 5, rowb = c * RB + 8 * h, gsw = wimg_g<H>(c);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int a = rowb + (((4 * w + g) ^ gsw) << 4);
        const u32x2 hi = *reinterpret_cast<const u32x2*>(pimg + a), mid = *reinterpret_cast<const u32x2*>(pimg + PS + a), lo = *reinterpret_cast<const u32x2*>(pimg + 2 * PS + a);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            x[4 * g + 2 * t] = (__uint_as_float(hi[t] << 16) + __uint_as_float(mid[t] << 16)) + __uint_as_float(lo[t] << 16);
            x[4 * g + 2 * t + 1] = (__uint_as_float(hi[t] & 0xffff0000u) + __uint_as_float(mid[t] & 0xffff0000u)) + __uint_as_float(lo[t] & 0xffff0000u);
        }
    }
}
// operand of a product that sums over SAMPLES: lane (unit 32m + (lane & 31), half kh) gets samples 16s + 8kh + j of its unit; tbase from wide_tr_base
template <int H>
__device__ __forceinline__ int wide_tr_base(int lane) {
    constexpr int RB = 2 * H;
    const int kh = lane >> 5, gm = (lane >> 4) & 1, e = lane & 15, q = e >> 2, p = e & 3, n = 8 * kh + q;
    return n * RB + ((((2 * gm + (p >> 1)) ^ wimg_g<H>(n)) & 15) << 4) + 8 * (p & 1);
}
template <int H>
__device__ __forceinline__ bf16x8 load_frag_wide_T(const char* pimg, int tbase, int piece, int m, int s) {
    constexpr int RB = 2 * H, PS = 32 * RB;
    const int a = (tbase ^ (64 * m)) + 16 * s * RB + piece * PS;
    return frag8(lds_read_tr16(pimg, a), lds_read_tr16(pimg, (a ^ 16) + 4 * RB));     // samples +0..3, +4..7: the row's chunk swizzle flips bit 0 with (n >> 2) & 1
}
__device__ __forceinline__ void wide_split_preload(const u32x4* __restrict__ wimg, int MTv, int mo, int lane, u32x4 (&af)[2][3]) {
    const u32x4* base = wimg + ((size_t)mo * MTv * 6) * 64 + lane;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int p = 0; p < 3; ++p) af[s][p] = base[(size_t)(s * 3 + p) * 64];
}
// output m-tile mo of Y = W X: W as pre-split fragments from L2 (af arrives preloaded with m-tile 0's, each refilled in place right after its MFMAs), X from the piece image
template <int H, bool BIAS>
__device__ __forceinline__ f32x16 dense_tile_split(const u32x4* __restrict__ wimg, const float* __restrict__ bias, const char* pimg, int mo, int lane, u32x4 (&af)[2][3]) {
    constexpr int MT = H / 32, RB = 2 * H, PS = 32 * RB;
    const int c = lane & 31, h = lane >> 5, rowb = c * RB, gsw = wimg_g<H>(c);
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if (BIAS) b = *reinterpret_cast<const f32x4*>(bias + 32 * mo + 8 * q + 4 * h);
        acc[4 * q + 0] = b[0]; acc[4 * q + 1] = b[1]; acc[4 * q + 2] = b[2]; acc[4 * q + 3] = b[3];
    }
    const u32x4* base = wimg + ((size_t)mo * MT * 6) * 64 + lane;
#pragma unroll 1
    for (int mi = 0; mi < MT; ++mi) {
        const u32x4* nextp = base + (size_t)((mi + 1 < MT ? mi + 1 : mi) * 6) * 64;   // the last iteration re-reads its own fragments (in bounds, unused)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int a = rowb + (((4 * mi + 2 * s + h) ^ gsw) << 4);
            bf16x8 B[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) B[p] = *reinterpret_cast<const bf16x8*>(pimg + p * PS + a);
            acc = mfma_split6(__builtin_bit_cast(bf16x8, af[s][0]), __builtin_bit_cast(bf16x8, af[s][1]), __builtin_bit_cast(bf16x8, af[s][2]), B[0], B[1], B[2], acc);
#pragma unroll
            for (int p = 0; p < 3; ++p) af[s][p] = nextp[(size_t)(s * 3 + p) * 64];
        }
    }
    return acc;
}

template <int KIND, int H, int O, int HEAD>
__device__ __forceinline__ void grad_body_wide_split(const GradArgs& a, float* smem) {
    constexpr int D = EnvSpec<KIND>::D, MT = H / 32;
    constexpr bool REC = true;
    using L = NetLdsSmall<D, H, O>;
    using SC = WideSplitScratch<D, H, O>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);          // this wave's m-tile
    const int c = lane & 31, h = lane >> 5;
    const NetOff off = HEAD == HEAD_VALUE ? a.critic : a.actor;
    const u32x4* w2p = HEAD == HEAD_VALUE ? a.w2p_critic : a.w2p_actor;
    const u32x4* w2tp = HEAD == HEAD_VALUE ? a.w2tp_critic : a.w2tp_actor;
    float* wl = smem;
    char* P1 = reinterpret_cast<char*>(smem + SC::P1); char* P2 = reinterpret_cast<char*>(smem + SC::P2);
    float* TB = smem + SC::TB; float* XI = smem + SC::XI; float* ZI = smem + SC::ZI + w * O * kTS; float* PO = smem + SC::PO;
    stage_net_small<D, H, O>(wl, a.params, off, tid, blockDim.x);
    for (int i = tid; i < (D + 2) * kTS; i += blockDim.x) XI[i] = (i / kTS == D) ? 1.0f : 0.0f;
    __syncthreads();

    float adv_mean = 0.f, adv_den = 1.f;
    if (HEAD != HEAD_VALUE && a.normalize_adv) {
        const double s = a.adv_stats[0], q = a.adv_stats[1], n = a.adv_stats[2];
        const double mean = s / n;
        double var = (q - s * mean) / (n - 1.0);
        if (var < 0) var = 0;
        adv_mean = (float)mean; adv_den = (float)sqrt(var) + 1.0e-8f;
    }
    adv_mean = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, adv_mean)));
    const float adv_inv = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, 1.0f / adv_den)));
    float lsr[kLsMax];
#pragma unroll
    for (int o = 0; o < kLsMax; ++o) lsr[o] = 0.f;
    if (HEAD == HEAD_GAUSSIAN) {
#pragma unroll
        for (int o = 0; o < O; ++o) lsr[o] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.params[a.log_std_off + o])));
    }
    const float* ls = lsr;
    const int tbase = wide_tr_base<H>(lane);

    f32x16 dW2[MT];                                                  // rows 32w.., all H columns
    f32x4 dW1[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    float dW3a[O], db2p = 0.f, db3p[O], dlsp[O], st[5];
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) dW2[j][r] = 0.f;
#pragma unroll
    for (int o = 0; o < O; ++o) { dW3a[o] = 0.f; db3p[o] = 0.f; dlsp[o] = 0.f; }
#pragma unroll
    for (int i = 0; i < 5; ++i) st[i] = 0.f;

    const int g = a.layout ? (int)(blockIdx.x % a.G) : (int)(blockIdx.x >> 1);
    const int64_t ntiles = (a.count + kTile - 1) / kTile;
    TileIn<O> cur, nxt;
    int64_t tile = g;
    if (tile < ntiles) load_tile<KIND, O, HEAD, REC>(a, tile, ntiles, c, h, cur);
#ifdef DRIL_STAMPS
    unsigned long long stamp_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev) :: "memory");
#endif
    for (; tile < ntiles; tile += a.G) {
        unpack_tile<KIND, O, HEAD, REC>(a, h, cur);
        const bool valid = cur.valid;
        const float xk[2] = {cur.xk[0], cur.xk[1]};
        // ---- h1 tile w; its pieces into the workgroup image ----
        f32x16 h1w;
        {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(wl + L::B1 + 32 * w + 8 * q + 4 * h);
                h1w[4 * q + 0] = b[0]; h1w[4 * q + 1] = b[1]; h1w[4 * q + 2] = b[2]; h1w[4 * q + 3] = b[3];
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) h1w = mfma32(wl[L::W1T + (2 * s + h) * H + 32 * w + c], xk[s], h1w);
            tanh16(h1w);
        }
        // `opaque(lane)`: the image addresses are lane constants, and hoisted out of the tile loop as loop invariants they hold ~60 registers for the whole kernel (they cost 2-3 VALU to rebuild)
        store_tile_pieces<H>(P1, w, h1w, opaque(lane));
        if (w == 0) {
#pragma unroll
            for (int s = 0; s < 2; ++s) { const int d = 2 * s + h; XI[(d < D ? d : D + 1) * kTS + c] = d < D ? xk[s] : 0.f; }   // branch-free: out-of-range components rewrite the zero row
        }
        STAMP(0);
        u32x4 afw[2][3];
        wide_split_preload(w2p, MT, w, lane, afw);                                    // first W2 fragments in flight across the barrier
        __syncthreads();                                                              // B1: P1, XI complete
        STAMP(1);
        load_tile<KIND, O, HEAD, REC>(a, tile + a.G, ntiles, c, h, nxt);              // after the barrier (see ppo_grad_wide_kernel)
        // ---- h2 tile w ----
        f32x16 h2w = dense_tile_split<H, true>(w2p, wl + L::B2, P1, w, opaque(lane), afw);
        tanh16(h2w);
        STAMP(2);
        // ---- output layer: partial over this wave's rows, summed across waves through LDS ----
        float out[O], dz[O];
#pragma unroll
        for (int o = 0; o < O; ++o) {
            float p = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + L::W3S + o * H + 32 * w + 8 * q + 4 * h);
                p = fmaf(wv[0], h2w[4 * q + 0], p); p = fmaf(wv[1], h2w[4 * q + 1], p);
                p = fmaf(wv[2], h2w[4 * q + 2], p); p = fmaf(wv[3], h2w[4 * q + 3], p);
            }
            p += __shfl_xor(p, 32);
            if (h == 0) PO[(w * O + o) * 32 + c] = p;
        }
        store_image_tile(TB, w, h2w, lane);                                            // h2' (own rows; only this wave reads them)
        __syncthreads();                                                              // B2: PO complete
        STAMP(3);
#pragma unroll
        for (int o = 0; o < O; ++o) {
            float v = wl[L::B3 + o];
#pragma unroll
            for (int ww = 0; ww < MT; ++ww) v += PO[(ww * O + o) * 32 + c];            // fixed order: every wave gets the same bits
            out[o] = v;
        }
        loss_head<O, HEAD>(a, cur, out, valid, h == 0 && w == 0, ls, adv_mean, adv_inv, dz, st, dlsp);
        // ---- dW3 (own rows) ----
#pragma unroll
        for (int o = 0; o < O; ++o) { if (h == 0) { if (w == 0) db3p[o] += dz[o]; ZI[o * kTS + c] = dz[o]; } }
        {
            const f32x16 Bh2 = load_operand(TB, w, lane);
#pragma unroll
            for (int o = 0; o < O; ++o) {
                float acc = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 z = *reinterpret_cast<const f32x4*>(ZI + o * kTS + 16 * h + 4 * q);
                    acc = fmaf(Bh2[4 * q + 0], z[0], acc); acc = fmaf(Bh2[4 * q + 1], z[1], acc);
                    acc = fmaf(Bh2[4 * q + 2], z[2], acc); acc = fmaf(Bh2[4 * q + 3], z[3], acc);
                }
                dW3a[o] += acc;
            }
        }
        // ---- dz2 tile w (in h2w's registers); its pieces into the workgroup image; the f32 transposed copy (own rows) gives db2 ----
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float dh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int o = 0; o < O; ++o) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + L::W3S + o * H + 32 * w + 8 * q + 4 * h);
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) dh[cc] = fmaf(wv[cc], dz[o], dh[cc]);
            }
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) { const float hv = h2w[4 * q + cc]; h2w[4 * q + cc] = dh[cc] * (1.0f - hv * hv); }
        }
        store_tile_pieces<H>(P2, w, h2w, opaque(lane));
        store_image_tile(TB, w, h2w, lane);                                            // dz2' (after the Bh2 read: same wave, LDS in order)
        wide_split_preload(w2tp, MT, w, lane, afw);                                   // first W2' fragments in flight across the barrier
        STAMP(4);
        __syncthreads();                                                              // B3: P2 complete
        STAMP(5);
        // ---- dh1 tile w = W2' dz2 ; dz1 ----
        f32x16 g1 = dense_tile_split<H, false>(w2tp, nullptr, P2, w, opaque(lane), afw);
        {
            f32x16 h1r;
            load_tile_pieces<H>(P1, w, h1r, opaque(lane));                                     // h1 tile w rebuilt from its own pieces (hi + mid + lo is exact): 16 registers less across both MFMA chains
#pragma unroll
            for (int r = 0; r < 16; ++r) g1[r] = g1[r] * (1.0f - h1r[r] * h1r[r]);
        }
        STAMP(6);
        // ---- db2 from the f32 transposed copy; then dW1 | db1 (own rows) BEFORE dW2, so that dz1 is dead while the 128 accumulators are being updated ----
        {
            const f32x16 Az32 = load_operand(TB, w, lane);
            db2p += sum16(Az32);
        }
        store_image_tile(TB, w, g1, lane);
        {
            const int j = lane & 15;
            float bx[8];
            load_row8(XI, j <= D ? j : D + 1, lane, bx);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float az[8];
                load_row8(TB, 32 * w + 16 * t + j, lane, az);
#pragma unroll
                for (int k = 0; k < 8; ++k) dW1[t] = mfma16(az[k], bx[k], dW1[t]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        STAMP(7);
        // ---- dW2[rows of w][:] += dz2 h1' (both operands as transposed fragments of the piece images) ----
        {
            const int tb = opaque(tbase);
            bf16x8 Az[2][3];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int p = 0; p < 3; ++p) Az[s][p] = load_frag_wide_T<H>(P2, tb, p, w, s);
#pragma unroll
            for (int mj = 0; mj < MT; ++mj) {
                bf16x8 Bh[2][3];
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int p = 0; p < 3; ++p) Bh[s][p] = load_frag_wide_T<H>(P1, tb, p, mj, s);
#pragma unroll
                for (int s = 0; s < 2; ++s) dW2[mj] = mfma_split6(Az[s][0], Az[s][1], Az[s][2], Bh[s][0], Bh[s][1], Bh[s][2], dW2[mj]);
                __builtin_amdgcn_sched_barrier(0);                                    // keep the next m-tile's fragment requests behind these MFMAs (hoisted, they spill)
            }
        }
        __syncthreads();                                                              // B4: P1 / P2 / PO / XI free for the next tile
        STAMP(8);
        cur = nxt;
    }
#ifdef DRIL_STAMPS
    if (lane == 0 && a.dbg) {
        unsigned long long* o_ = a.dbg + ((size_t)(blockIdx.x % (2 * a.G)) * 4 + (w & 3)) * 12;
        if (w < 4) { for (int k = 0; k < 10; ++k) o_[k] = stamp_acc[k]; o_[10] = (unsigned long long)((ntiles - g + a.G - 1) / a.G); o_[11] = HEAD; }
    }
#endif

    // ---- epilogue: every wave owns distinct gradient rows -> straight to the workgroup's slab ----
    const int SL = HEAD == HEAD_VALUE ? a.slab_c : a.slab_a;
    const int o_w1 = 0, o_b1 = H * D, o_w2 = o_b1 + H, o_b2 = o_w2 + H * H, o_w3 = o_b2 + H, o_b3 = o_w3 + O * H;
    const int o_ls = o_b3 + O, o_st = SL - 8;
    float* slab = (HEAD == HEAD_VALUE ? a.slabs_critic : a.slabs_actor) + (size_t)g * SL;
#pragma unroll
    for (int mj = 0; mj < MT; ++mj)
#pragma unroll
        for (int r = 0; r < 16; ++r) slab[o_w2 + 32 * w + rowfn(r, h) + (32 * mj + c) * H] = dW2[mj][r];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 32 * w + 16 * t + 4 * (lane >> 4) + r, col = lane & 15;
            if (col < D) slab[o_w1 + row + col * H] = dW1[t][r];
            else if (col == D) slab[o_b1 + row] = dW1[t][r];
        }
    { const float b2 = db2p + __shfl_xor(db2p, 32); if (h == 0) slab[o_b2 + 32 * w + c] = b2; }
#pragma unroll
    for (int o = 0; o < O; ++o) {
        const float v = dW3a[o] + __shfl_xor(dW3a[o], 32);
        if (h == 0) slab[o_w3 + o + (32 * w + c) * O] = v;
        const float b3 = half_sum(db3p[o]);
        if (w == 0 && lane == 0) slab[o_b3 + o] = b3;
        if (HEAD == HEAD_GAUSSIAN) { const float l = half_sum(dlsp[o]); if (w == 0 && lane == 0) slab[o_ls + o] = l; }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) { const float v = half_sum(st[k]); if (w == 0 && lane == 0) slab[o_st + k] = v; }
    if (w == 0 && lane < 3) slab[o_st + 5 + lane] = 0.f;
    for (int i = (HEAD == HEAD_GAUSSIAN ? o_ls + O : o_ls) + tid; i < o_st; i += blockDim.x) slab[i] = 0.f;   // padding
}

template <int KIND, int H>
__global__ __launch_bounds__(H * 2, 2) void ppo_grad_wide_split_kernel(GradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (*a.stop_flag) return;
    constexpr int A = EnvSpec<KIND>::A;
    const bool actor = a.layout ? (blockIdx.x < (unsigned)a.G) : ((blockIdx.x & 1) == 0);
    if (actor) grad_body_wide_split<KIND, H, A, EnvSpec<KIND>::discrete ? HEAD_CATEGORICAL : HEAD_GAUSSIAN>(a, smem);
    else grad_body_wide_split<KIND, H, 1, HEAD_VALUE>(a, smem);
}

// =============================================================================================
// ppo_grad_pair_kernel — hidden [64,64], large minibatches: TWO waves own one 32-sample tile (wave p the m-tile p of every layer and the 32 x 64 slice p of dW2: the
// decomposition of ppo_grad_wide_split_kernel at MT = 2), two pairs per workgroup, two workgroups per CU = TWO waves per SIMD at <= 256 registers.  That is the one
// arrangement in which the matrix pipe and the VALU run beside each other on this part (profiles/r02_split_kernel.md: a lone wave's VALU work does not run under its own
// MFMAs; a second wave's does, completely).  Same arithmetic as ppo_grad_split_kernel (bf16 matrix cores, fp32-equivalent 3-piece operand splitting).
//   * W2 lives in LDS once per workgroup as three bf16 pieces in the piece-image layout of the wide split kernel (128-byte rows, 16-byte chunk ch of row r at ch ^ f(r)):
//     row reads (ds_read_b128) give the A operand of L2, ds_read_b64_tr_b16 the A operand of dh1 (W2'); pre-scaled by kTanhScale, dh1 folds 1 / kTanhScale into its mask.
//   * each pair has two 12 KB piece images (h1, dz2): every wave writes its own 32 columns once; row reads give the B operand of L2 / dh1 (both m-tiles), transposed
//     reads both operands of dW2.
//   * no f32 image at all: dW3, db2, dW1 and db1 are per-lane accumulations (the lane is the sample), reduced over the 32 lanes of a half once, in the epilogue.
//   * four workgroup barriers per tile; both pairs of a workgroup run the same number of tiles (the second pair's last tile may be an all-invalid one).
// Every wave owns distinct rows of every gradient: one slab per PAIR, written straight from registers.  a.G / a.Gc = pairs of the actor / the critic (even);
// grid = (a.G + a.Gc) / 2 workgroups, the first a.G / 2 run the actor.
// =============================================================================================
template <int D, int O> struct PairLds {
    static constexpr int H = 64, DP = 4, OP = (O + 3) / 4 * 4;
    static constexpr int W1T = 0, B1 = W1T + DP * H, B2 = B1 + H, W3S = B2 + H, B3 = W3S + O * H, SMALL_END = (B3 + OP + 3) / 4 * 4;
    static constexpr int WIMG = SMALL_END;                    // three pieces x [64 out][64 in] bf16 = 3 x 8192 bytes
    static constexpr int PAIR0 = WIMG + 3 * 2048;
    static constexpr int P1 = 0, P2 = P1 + 3 * 1024, PO = P2 + 3 * 1024, PAIR_SIZE = (PO + 2 * O * 32 + 3) / 4 * 4;   // per pair: two 12 KB piece images, [2 waves][O][32] partial sums
    static constexpr int END = PAIR0 + 2 * PAIR_SIZE;
};
// A operand of dh1 = W2': lane (in-unit 32mk + (lane & 31), half kh) gets out-units 32mi + 16s + 8kh + j of the weight image (64 rows, piece stride 8192); tbase = wide_tr_base<64>
__device__ __forceinline__ bf16x8 load_frag_W_T(const char* wimg, int tbase, int piece, int mk, int mi, int s) {
    const int a = (tbase ^ (64 * mk)) + (32 * mi + 16 * s) * 128 + piece * 8192;
    return frag8(lds_read_tr16(wimg, a), lds_read_tr16(wimg, (a ^ 16) + 4 * 128));
}

template <int KIND, int O, int HEAD>
__device__ __forceinline__ void grad_body_pair(const GradArgs& a, float* smem) {
    constexpr int D = EnvSpec<KIND>::D, H = 64, MT = 2;
    constexpr bool REC = true, kKeepH1 = HEAD == HEAD_VALUE;
    constexpr float kInvTanhScale = 1.0f / kTanhScale;
    using L = PairLds<D, O>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = wave & 1, pr = wave >> 1;                                           // this wave's m-tile; this wave's pair
    const int c = lane & 31, h = lane >> 5;
    const NetOff off = HEAD == HEAD_VALUE ? a.critic : a.actor;
    float* wl = smem;
    char* Wimg = reinterpret_cast<char*>(smem + L::WIMG);
    float* pb = smem + L::PAIR0 + pr * L::PAIR_SIZE;
    char* P1 = reinterpret_cast<char*>(pb + L::P1); char* P2 = reinterpret_cast<char*>(pb + L::P2); float* PO = pb + L::PO;
    {   // stage the small parts (as stage_net_split) and the W2 piece image
        const float* __restrict__ P = a.params;
        for (int i = tid; i < L::DP * H; i += blockDim.x) { const int o = i % H, k = i / H; wl[L::W1T + k * H + o] = k < D ? kTanhScale * P[off.w1 + o + k * H] : 0.0f; }
        for (int i = tid; i < H; i += blockDim.x) { wl[L::B1 + i] = kTanhScale * P[off.b1 + i]; wl[L::B2 + i] = kTanhScale * P[off.b2 + i]; }
        for (int i = tid; i < O * H; i += blockDim.x) { const int o = i % O, k = i / O; wl[L::W3S + o * H + k] = P[off.w3 + i]; }
        for (int i = tid; i < L::OP; i += blockDim.x) wl[L::B3 + i] = i < O ? P[off.b3 + i] : 0.0f;
        for (int i = tid; i < H * H / 2; i += blockDim.x) {       // pair (k, k+1) of row o: W2 is column-major (out x in), consecutive threads read consecutive o
            const int o = i % H, kp = i / H;
            unsigned hi, mid, lo;
            split3_pair(kTanhScale * P[off.w2 + o + H * (2 * kp)], kTanhScale * P[off.w2 + o + H * (2 * kp + 1)], hi, mid, lo);
            const int byte = o * 128 + ((((kp >> 2) ^ wimg_g<64>(o)) & 7) << 4) + ((kp & 3) << 2);
            *reinterpret_cast<unsigned*>(Wimg + byte) = hi; *reinterpret_cast<unsigned*>(Wimg + 8192 + byte) = mid; *reinterpret_cast<unsigned*>(Wimg + 16384 + byte) = lo;
        }
    }
    __syncthreads();

    float adv_mean = 0.f, adv_den = 1.f;
    if (HEAD != HEAD_VALUE && a.normalize_adv) {
        const double s = a.adv_stats[0], q = a.adv_stats[1], n = a.adv_stats[2];
        const double mean = s / n;
        double var = (q - s * mean) / (n - 1.0);
        if (var < 0) var = 0;
        adv_mean = (float)mean; adv_den = (float)sqrt(var) + 1.0e-8f;
    }
    adv_mean = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, adv_mean)));
    const float adv_inv = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, 1.0f / adv_den)));
    float lsr[kLsMax];
#pragma unroll
    for (int o = 0; o < kLsMax; ++o) lsr[o] = 0.f;
    if (HEAD == HEAD_GAUSSIAN) {
#pragma unroll
        for (int o = 0; o < O; ++o) lsr[o] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.params[a.log_std_off + o])));
    }
    const float* ls = lsr;
    const int tbase = wide_tr_base<64>(lane);

    f32x16 dW2[MT], dW3acc[O], db2acc, dW1acc[D], db1acc;            // dW2: rows 32w.., all 64 columns; the others: per-lane sums over this lane's samples (units rowfn(r, h) of m-tile w)
    float db3p[O], dlsp[O], st[5];
#pragma unroll
    for (int r = 0; r < 16; ++r) { db2acc[r] = 0.f; db1acc[r] = 0.f; }
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) dW2[j][r] = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dW1acc[d][r] = 0.f;
#pragma unroll
    for (int o = 0; o < O; ++o) {
        db3p[o] = 0.f; dlsp[o] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) dW3acc[o][r] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) st[i] = 0.f;

    const int nb = HEAD == HEAD_VALUE ? (int)blockIdx.x - a.G / 2 : (int)blockIdx.x;  // workgroup within its net
    const int GP = HEAD == HEAD_VALUE ? a.Gc : a.G;                                   // pairs of this net
    const int g = 2 * nb + pr;                                                        // pair within its net = slab index
    const int64_t ntiles = (a.count + kTile - 1) / kTile;
    const int64_t g0 = 2 * nb;
    const int64_t trips = g0 < ntiles ? (ntiles - g0 + GP - 1) / GP : 0;           // the same for both pairs of the workgroup (barriers inside the loop)
    TileIn<O> cur, nxt;
    int64_t tile = g;
    load_tile<KIND, O, HEAD, REC>(a, tile, ntiles, c, h, cur);                        // a tile index past the end loads an all-invalid tile
#ifdef DRIL_STAMPS
    unsigned long long stamp_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev) :: "memory");
#endif
    for (int64_t it = 0; it < trips; ++it, tile += GP) {
        unpack_tile<KIND, O, HEAD, REC>(a, h, cur);
        const bool valid = cur.valid;
        const float xk[2] = {cur.xk[0], cur.xk[1]};
        // ---- h1 tile w; its pieces into the pair's image ----
        f32x16 h1k;                                                                   // kept across the tile where the registers allow it (the critic), rebuilt from the pieces elsewhere
        {
            f32x16 h1w;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(wl + L::B1 + 32 * w + 8 * q + 4 * h);
                h1w[4 * q + 0] = b[0]; h1w[4 * q + 1] = b[1]; h1w[4 * q + 2] = b[2]; h1w[4 * q + 3] = b[3];
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) h1w = mfma32(wl[L::W1T + (2 * s + h) * H + 32 * w + c], xk[s], h1w);
            tanh16(h1w);
            store_tile_pieces<64>(P1, w, h1w, opaque(lane));
            if (kKeepH1) h1k = h1w;
        }
        STAMP(0);
        __syncthreads();                                                              // B1: the pair's h1 image complete
        STAMP(1);
        load_tile<KIND, O, HEAD, REC>(a, tile + GP, ntiles, c, h, nxt);
        // ---- h2 tile w = tanh(W2[rows of w] h1 + b2): A from the weight image, B from the pair's h1 image (both row reads with the same chunk index) ----
        f32x16 h2w;
        {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(wl + L::B2 + 32 * w + 8 * q + 4 * h);
                h2w[4 * q + 0] = b[0]; h2w[4 * q + 1] = b[1]; h2w[4 * q + 2] = b[2]; h2w[4 * q + 3] = b[3];
            }
            const int lo_ = opaque(lane), cc = lo_ & 31, hh = lo_ >> 5, gsw = wimg_g<64>(cc);
            const char* arow = Wimg + (32 * w + cc) * 128; const char* brow = P1 + cc * 128;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {                                          // ks = 2 mi + s
                const int ch = ((2 * ks + hh) ^ gsw) << 4;
                bf16x8 A[3], B[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) { A[p] = *reinterpret_cast<const bf16x8*>(arow + p * 8192 + ch); B[p] = *reinterpret_cast<const bf16x8*>(brow + p * 4096 + ch); }
                h2w = mfma_split6(A[0], A[1], A[2], B[0], B[1], B[2], h2w);
            }
            tanh16(h2w);
        }
        STAMP(2);
        // ---- output layer: partial over this wave's 32 units, summed across the pair through LDS ----
        float out[O], dz[O];
#pragma unroll
        for (int o = 0; o < O; ++o) {
            float p = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + L::W3S + o * H + 32 * w + 8 * q + 4 * h);
                p = fmaf(wv[0], h2w[4 * q + 0], p); p = fmaf(wv[1], h2w[4 * q + 1], p);
                p = fmaf(wv[2], h2w[4 * q + 2], p); p = fmaf(wv[3], h2w[4 * q + 3], p);
            }
            p += __shfl_xor(p, 32);
            if (h == 0) PO[(w * O + o) * 32 + c] = p;
        }
        __syncthreads();                                                              // B2: both partial sums
        STAMP(3);
#pragma unroll
        for (int o = 0; o < O; ++o) out[o] = (wl[L::B3 + o] + PO[o * 32 + c]) + PO[(O + o) * 32 + c];   // fixed order: both waves get the same bits
        loss_head<O, HEAD>(a, cur, out, valid, h == 0 && w == 0, ls, adv_mean, adv_inv, dz, st, dlsp);
#pragma unroll
        for (int o = 0; o < O; ++o) {
            if (h == 0 && w == 0) db3p[o] += dz[o];
            dW3acc[o] += dz[o] * h2w;                                                  // dW3[o][unit] += dz[o][sample] h2[unit][sample]: the lane IS the sample
        }
        // ---- dz2 tile w (in h2w's registers); db2; its pieces into the pair's image ----
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float dh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int o = 0; o < O; ++o) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + L::W3S + o * H + 32 * w + 8 * q + 4 * h);
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) dh[cc] = fmaf(wv[cc], dz[o], dh[cc]);
            }
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) { const float hv = h2w[4 * q + cc]; h2w[4 * q + cc] = dh[cc] * fmaf(-hv, hv, 1.0f); }
        }
        db2acc += h2w;
        store_tile_pieces<64>(P2, w, h2w, opaque(lane));
        STAMP(4);
        __syncthreads();                                                              // B3: the pair's dz2 image complete
        STAMP(5);
        // ---- dz1 tile w = (W2'[rows of w] dz2) .* (1 - h1^2): A = transposed reads of the weight image, B = row reads of the dz2 image ----
        f32x16 g1;
        {
#pragma unroll
            for (int r = 0; r < 16; ++r) g1[r] = 0.f;
            const int lo_ = opaque(lane), cc = lo_ & 31, hh = lo_ >> 5, gsw = wimg_g<64>(cc), tb = opaque(tbase);
            const char* brow = P2 + cc * 128;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int ch = ((2 * ks + hh) ^ gsw) << 4;
                bf16x8 A[3], B[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) { A[p] = load_frag_W_T(Wimg, tb, p, w, ks >> 1, ks & 1); B[p] = *reinterpret_cast<const bf16x8*>(brow + p * 4096 + ch); }
                g1 = mfma_split6(A[0], A[1], A[2], B[0], B[1], B[2], g1);
            }
            f32x16 h1r;
            if (kKeepH1) h1r = h1k; else load_tile_pieces<64>(P1, w, h1r, lo_);         // h1 tile w rebuilt from its own pieces (exact)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float t2 = h1r[r] * h1r[r]; g1[r] = g1[r] * fmaf(-t2, kInvTanhScale, kInvTanhScale); }
        }
        STAMP(6);
        // ---- dW1 | db1: per-lane accumulation, dW1[unit][d] += dz1[unit][sample] x[sample][d] ----
        {
            const float xo0 = __shfl_xor(xk[0], 32), xo1 = __shfl_xor(xk[1], 32);      // the other half holds x[2s + 1 - h]
            const float x4[4] = {h ? xo0 : xk[0], h ? xk[0] : xo0, h ? xo1 : xk[1], h ? xk[1] : xo1};
#pragma unroll
            for (int d = 0; d < D; ++d) dW1acc[d] += x4[d] * g1;
            db1acc += g1;
        }
        // ---- dW2[rows of w][:] += dz2 h1' (both operands as transposed fragments of the pair's images) ----
        {
            const int tb = opaque(tbase);
            bf16x8 Az[2][3];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int p = 0; p < 3; ++p) Az[s][p] = load_frag_wide_T<64>(P2, tb, p, w, s);
#pragma unroll
            for (int mj = 0; mj < MT; ++mj) {
                bf16x8 Bh[2][3];
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int p = 0; p < 3; ++p) Bh[s][p] = load_frag_wide_T<64>(P1, tb, p, mj, s);
#pragma unroll
                for (int s = 0; s < 2; ++s) dW2[mj] = mfma_split6(Az[s][0], Az[s][1], Az[s][2], Bh[s][0], Bh[s][1], Bh[s][2], dW2[mj]);
            }
        }
        STAMP(7);
        __syncthreads();                                                              // B4: the pair's images and partial sums free for the next tile
        STAMP(8);
        cur = nxt;
    }
#ifdef DRIL_STAMPS
    if (lane == 0 && a.dbg) {
        unsigned long long* o_ = a.dbg + ((size_t)blockIdx.x * 4 + wave) * 12;
        for (int k = 0; k < 10; ++k) o_[k] = stamp_acc[k];
        o_[10] = (unsigned long long)trips; o_[11] = HEAD;
    }
#endif

    // ---- epilogue: every wave owns distinct gradient rows -> straight to the pair's slab ----
    const int SL = HEAD == HEAD_VALUE ? a.slab_c : a.slab_a;
    const int o_w1 = 0, o_b1 = H * D, o_w2 = o_b1 + H, o_b2 = o_w2 + H * H, o_w3 = o_b2 + H, o_b3 = o_w3 + O * H;
    const int o_ls = o_b3 + O, o_st = SL - 8;
    float* slab = (HEAD == HEAD_VALUE ? a.slabs_critic : a.slabs_actor) + (size_t)g * SL;
#pragma unroll
    for (int mj = 0; mj < MT; ++mj)
#pragma unroll
        for (int r = 0; r < 16; ++r) slab[o_w2 + 32 * w + rowfn(r, h) + (32 * mj + c) * H] = dW2[mj][r];
#pragma unroll
    for (int r = 0; r < 16; ++r) {                                                    // per-lane sums over samples -> sum over the 32 lanes of each half (the halves hold different units)
        const int unit = 32 * w + rowfn(r, h);
        const float b2 = half_sum(db2acc[r]), b1 = half_sum(db1acc[r]);
        if (c == 0) { slab[o_b2 + unit] = b2; slab[o_b1 + unit] = b1; }
#pragma unroll
        for (int d = 0; d < D; ++d) { const float v = half_sum(dW1acc[d][r]); if (c == 0) slab[o_w1 + unit + d * H] = v; }
#pragma unroll
        for (int o = 0; o < O; ++o) { const float v = half_sum(dW3acc[o][r]); if (c == 0) slab[o_w3 + o + unit * O] = v; }
    }
#pragma unroll
    for (int o = 0; o < O; ++o) {
        const float b3 = half_sum(db3p[o]);
        if (w == 0 && lane == 0) slab[o_b3 + o] = b3;
        if (HEAD == HEAD_GAUSSIAN) { const float l = half_sum(dlsp[o]); if (w == 0 && lane == 0) slab[o_ls + o] = l; }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) { const float v = half_sum(st[k]); if (w == 0 && lane == 0) slab[o_st + k] = v; }
    if (w == 0 && lane < 3) slab[o_st + 5 + lane] = 0.f;
    for (int i = (HEAD == HEAD_GAUSSIAN ? o_ls + O : o_ls) + w * 64 + lane; i < o_st; i += 128) slab[i] = 0.f;   // padding
}

template <int KIND>
__global__ __launch_bounds__(256, 2) void ppo_grad_pair_kernel(GradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (*a.stop_flag) return;
    constexpr int A = EnvSpec<KIND>::A;
    const bool actor = blockIdx.x < (unsigned)(a.G / 2);
    if (actor) grad_body_pair<KIND, A, EnvSpec<KIND>::discrete ? HEAD_CATEGORICAL : HEAD_GAUSSIAN>(a, smem);
    else grad_body_pair<KIND, 1, HEAD_VALUE>(a, smem);
}

// =============================================================================================
// slab reduction -> flat [grads | stats] buffer; norm; clip + KL check + Adam
// flat layout: params order (actor net, critic net, log_std) then 8 stats:
//   0 sum(-min term)  1 sum(entropy)  2 sum(clipped)  3 sum(kl)  4 sum(ratio)  5 sum((V-R)^2)  6 n_samples  7 unused
// =============================================================================================
__global__ __launch_bounds__(1024) void grad_reduce_kernel(ReduceArgs a) {
    // block = 32 parameters x 32 slab groups: every thread has all its (<= 8) slab loads in flight at once; the slabs were
    // written by other XCDs, so each load is a full memory round trip and serial loops cost ~0.5 us per iteration (v2: 65 us)
    __shared__ double sh[16];
    __shared__ float part[32][33];
    if (*a.stop_flag) return;
    const int pl = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int p = blockIdx.x * 32 + pl;
    float acc = 0.f;
    if (p < a.P) {
        const float* base; int stride, offp;
        int Gn = a.G;
        if (p < a.Pa) { base = a.slabs_actor; stride = a.slab_a; offp = p; }
        else if (p < a.Pa + a.Pc) { base = a.slabs_critic; stride = a.slab_c; offp = p - a.Pa; Gn = a.Gc; }
        else { base = a.slabs_actor; stride = a.slab_a; offp = a.Pa + (p - a.Pa - a.Pc); }   // log_std grads sit after the actor net
#pragma unroll 8
        for (int g = grp; g < Gn; g += 32) acc += base[(size_t)g * stride + offp];
    }
    part[grp][pl] = acc;
    __syncthreads();
    float gsum = 0.f;
    if (grp == 0) {
#pragma unroll
        for (int k = 0; k < 32; ++k) gsum += part[k][pl];                      // fixed order => deterministic
        if (p < a.P) a.flat[p] = gsum;
    }
    const double q = block_sum_f64(grp == 0 ? (double)gsum * (double)gsum : 0.0, sh);
    if (threadIdx.x == 0) a.norm_partials[blockIdx.x] = q;
    if (blockIdx.x == 0) {                                                     // loss/statistics sums: 8 stats x 32 slab groups
        __syncthreads();
        const int k = threadIdx.x & 7, sg = (threadIdx.x >> 3) & 31;
        float sacc = 0.f;
        if (threadIdx.x < 256) {
            if (k < 5) { for (int g = sg; g < a.G; g += 32) sacc += a.slabs_actor[(size_t)g * a.slab_a + a.slab_a - 8 + k]; }
            else if (k == 5) { for (int g = sg; g < a.Gc; g += 32) sacc += a.slabs_critic[(size_t)g * a.slab_c + a.slab_c - 8]; }
            part[sg][k] = sacc;
        }
        __syncthreads();
        if (threadIdx.x < 8) {
            double t = 0;
            for (int j = 0; j < 32; ++j) t += (double)part[j][threadIdx.x];
            if (threadIdx.x == 6) t = a.n_samples_local;
            a.flat[a.P + threadIdx.x] = (float)t;
        }
    }
}
__global__ void grad_norm_kernel(const float* flat, int P, double* norm_partials, const int* stop_flag) {
    __shared__ double sh[16];
    if (*stop_flag) return;
    const int p = blockIdx.x * 32 + (threadIdx.x & 31);          // same 32-parameter blocking as grad_reduce_kernel
    const float g = (p < P && threadIdx.x < 32) ? flat[p] : 0.f;
    const double q = block_sum_f64((double)g * (double)g, sh);
    if (threadIdx.x == 0) norm_partials[blockIdx.x] = q;
}
__global__ void adam_kernel(AdamArgs a) {
    if (*a.stop_flag) return;
    __shared__ double shn[16];
    double ps = 0;
    for (int i = threadIdx.x; i < a.n_partials; i += blockDim.x) ps += a.norm_partials[i];    // same order in every block
    const float norm = sqrtf((float)block_sum_f64(ps, shn));
    const float* stf = a.flat + a.P;
    const float n = a.use_stats ? stf[6] : 1.f;
    const float kl = a.use_stats ? stf[3] / n : 0.f;
    const bool bad = !(norm == norm) || isinf(norm);                              // NaN/Inf anywhere poisons the norm (ppo.jl:213-214)
    const bool kl_stop = a.use_stats && a.has_target_kl && kl > 1.5f * a.target_kl;   // ppo.jl:235-238: skip this apply, stop
    const float* bt_in = a.bt + 2 * (a.step_parity & 1);
    float* bt_out = a.bt + 2 * ((a.step_parity + 1) & 1);
    const float bt1 = bt_in[0], bt2 = bt_in[1];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (a.step_stats) {
            float* o = a.step_stats;
            const float pl = stf[0] / n, ent = stf[1] / n, vl = stf[5] / n;
            o[0] = pl; o[1] = vl; o[2] = -ent; o[3] = stf[2] / n; o[4] = kl; o[5] = ent; o[6] = stf[4] / n;
            o[7] = pl + a.ent_coef * (-ent) + a.vf_coef * vl;                       // loss, ppo.jl:386
            o[8] = norm; o[9] = (bad || kl_stop) ? 0.f : 1.f; o[10] = bad ? 1.f : 0.f; o[11] = kl_stop ? 1.f : 0.f;
        }
        if (a.norm_out) *a.norm_out = norm;
        if (bad) { *a.nan_flag = 1; *a.stop_flag_w = 1; }
        if (kl_stop) *a.stop_flag_w = 1;
        if (bad || kl_stop) { bt_out[0] = bt1; bt_out[1] = bt2; } else { bt_out[0] = bt1 * a.beta1; bt_out[1] = bt2 * a.beta2; }
    }
    if (bad || kl_stop) return;
    const float scale = (a.has_max_grad_norm && norm > a.max_grad_norm) ? a.max_grad_norm / norm : 1.0f;   // optimization_utils.jl:98-107
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < a.P) {
        float g = a.flat[p];
        if (scale != 1.0f) g = g * scale;
        const float m = a.beta1 * a.m[p] + (1.0f - a.beta1) * g;
        const float v = a.beta2 * a.v[p] + (1.0f - a.beta2) * g * g;
        a.m[p] = m; a.v[p] = v;
        a.params[p] -= m / (1.0f - bt1) / (sqrtf(v / (1.0f - bt2)) + a.eps) * a.lr;  // Optimisers.Adam, eps=1e-5 (ppo.jl:64-66)
    }
}

// Small minibatches (few slabs): grad_reduce_kernel + norm + adam_kernel in ONE workgroup — at B = 64 the optimiser step is bound by its
// dependent launches (~6 x 5-8 us), not by work.  Same arithmetic as the two kernels; the squared norm is summed in a different (fixed) order.
__global__ __launch_bounds__(1024) void ppo_finish_small_kernel(ReduceArgs r, AdamArgs a) {
    __shared__ double sh[16];
    __shared__ float stf[8];
    if (*a.stop_flag) return;
    constexpr int KMAX = 16;                                                            // P <= 16 384 (checked by the host)
    const int tid = threadIdx.x;
    float gacc[KMAX], mo[KMAX], vo[KMAX], po[KMAX];
    double ss = 0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int p = tid + 1024 * k;
        gacc[k] = 0.f; mo[k] = 0.f; vo[k] = 0.f; po[k] = 0.f;
        if (p < r.P) {
            const float* base; int stride, offp;
            int Gn = r.G;
            if (p < r.Pa) { base = r.slabs_actor; stride = r.slab_a; offp = p; }
            else if (p < r.Pa + r.Pc) { base = r.slabs_critic; stride = r.slab_c; offp = p - r.Pa; Gn = r.Gc; }
            else { base = r.slabs_actor; stride = r.slab_a; offp = r.Pa + (p - r.Pa - r.Pc); }
            float acc = 0.f;
            for (int g = 0; g < Gn; ++g) acc += base[(size_t)g * stride + offp];      // fixed order
            gacc[k] = acc; r.flat[p] = acc; ss += (double)acc * (double)acc;
            mo[k] = a.m[p]; vo[k] = a.v[p]; po[k] = a.params[p];                        // optimiser state in flight under the norm reduction
        }
    }
    if (tid < 8) {
        double t = 0;
        if (tid < 5) { for (int g = 0; g < r.G; ++g) t += (double)r.slabs_actor[(size_t)g * r.slab_a + r.slab_a - 8 + tid]; }
        else if (tid == 5) { for (int g = 0; g < r.Gc; ++g) t += (double)r.slabs_critic[(size_t)g * r.slab_c + r.slab_c - 8]; }
        else if (tid == 6) t = r.n_samples_local;
        stf[tid] = (float)t; r.flat[r.P + tid] = (float)t;
    }
    const float norm = sqrtf((float)block_sum_f64(ss, sh));
    __syncthreads();
    const float n = a.use_stats ? stf[6] : 1.f;
    const float kl = a.use_stats ? stf[3] / n : 0.f;
    const bool bad = !(norm == norm) || isinf(norm);
    const bool kl_stop = a.use_stats && a.has_target_kl && kl > 1.5f * a.target_kl;
    const float* bt_in = a.bt + 2 * (a.step_parity & 1);
    float* bt_out = a.bt + 2 * ((a.step_parity + 1) & 1);
    const float bt1 = bt_in[0], bt2 = bt_in[1];
    if (tid == 0) {
        if (a.step_stats) {
            float* o = a.step_stats;
            const float pl = stf[0] / n, ent = stf[1] / n, vl = stf[5] / n;
            o[0] = pl; o[1] = vl; o[2] = -ent; o[3] = stf[2] / n; o[4] = kl; o[5] = ent; o[6] = stf[4] / n;
            o[7] = pl + a.ent_coef * (-ent) + a.vf_coef * vl;
            o[8] = norm; o[9] = (bad || kl_stop) ? 0.f : 1.f; o[10] = bad ? 1.f : 0.f; o[11] = kl_stop ? 1.f : 0.f;
        }
        if (a.norm_out) *a.norm_out = norm;
        if (bad) { *a.nan_flag = 1; *a.stop_flag_w = 1; }
        if (kl_stop) *a.stop_flag_w = 1;
        if (bad || kl_stop) { bt_out[0] = bt1; bt_out[1] = bt2; } else { bt_out[0] = bt1 * a.beta1; bt_out[1] = bt2 * a.beta2; }
    }
    if (bad || kl_stop) return;
    const float scale = (a.has_max_grad_norm && norm > a.max_grad_norm) ? a.max_grad_norm / norm : 1.0f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int p = tid + 1024 * k;
        if (p < r.P) {
            float g = gacc[k];
            if (scale != 1.0f) g = g * scale;
            const float m = a.beta1 * mo[k] + (1.0f - a.beta1) * g;
            const float v = a.beta2 * vo[k] + (1.0f - a.beta2) * g * g;
            a.m[p] = m; a.v[p] = v;
            a.params[p] = po[k] - m / (1.0f - bt1) / (sqrtf(v / (1.0f - bt2)) + a.eps) * a.lr;
        }
    }
}

// explained_variance sums over the whole buffer (ppo.jl:256): partials of (v-r), (v-r)^2, r, r^2
__global__ void explained_var_kernel(const float* val, const float* ret, int64_t N, double* partials) {
    __shared__ double sh[16];
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const double d = (double)val[i] - (double)ret[i], r = ret[i];
        s0 += d; s1 += d * d; s2 += r; s3 += r * r;
    }
    s0 = block_sum_f64(s0, sh); s1 = block_sum_f64(s1, sh); s2 = block_sum_f64(s2, sh); s3 = block_sum_f64(s3, sh);
    if (threadIdx.x == 0) { double* o = partials + 4 * blockIdx.x; o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3; }
}

// =============================================================================================
// launchers
// =============================================================================================
template <int KIND, int H, bool WIDE> static size_t fwd_lds_bytes() {
    return sizeof(float) * (FwdLds<EnvSpec<KIND>::D, H, EnvSpec<KIND>::A, WIDE>::SIZE + FwdLds<EnvSpec<KIND>::D, H, 1, WIDE>::SIZE);
}
template <int KIND, int H> static size_t grad_lds_bytes() {
    constexpr int D = EnvSpec<KIND>::D, A = EnvSpec<KIND>::A;
    constexpr int wa = NetLds<D, H, H, A>::BWD_END + 4 * GradScratch<D, H, A>::SIZE;
    constexpr int wc = NetLds<D, H, H, 1>::BWD_END + 4 * GradScratch<D, H, 1>::SIZE;
    return sizeof(float) * (wa > wc ? wa : wc);
}

// kind 2 (ScalingWrapperEnv(Pendulum)) shares every kernel that never touches the simulator with kind 1
#define DRIL_DISPATCH(kind, hidden, CALL)                                            \
    do {                                                                             \
        if ((kind) == 0 && (hidden) == 64) { CALL(0, 64); }                          \
        else if (((kind) == 1 || (kind) == 2) && (hidden) == 64) { CALL(1, 64); }    \
        else if ((kind) == 3 && (hidden) == 64) { CALL(3, 64); }                     \
        else if ((kind) == 4 && (hidden) == 64) { CALL(4, 64); }                     \
        else return hipErrorInvalidValue;                                            \
    } while (0)

// every kernel below is built for hidden widths 64 (one wave per net), 128 and 256 (wide path)
#define DRIL_DISPATCH_H(K, hidden, CALL)                                             \
    { if ((hidden) == 64) { CALL(K, 64); } else if ((hidden) == 128) { CALL(K, 128); } else if ((hidden) == 256) { CALL(K, 256); } else return hipErrorInvalidValue; }
#define DRIL_DISPATCH_FWD(kind, hidden, CALL)                                        \
    do {                                                                             \
        if ((kind) == 0) DRIL_DISPATCH_H(0, hidden, CALL)                            \
        else if ((kind) == 1 || (kind) == 2) DRIL_DISPATCH_H(1, hidden, CALL)        \
        else if ((kind) == 3) DRIL_DISPATCH_H(3, hidden, CALL)                       \
        else if ((kind) == 4) DRIL_DISPATCH_H(4, hidden, CALL)                       \
        else return hipErrorInvalidValue;                                            \
    } while (0)

// kernels that step / observe the simulator: one instantiation per env kind
#define DRIL_DISPATCH_ENV(kind, hidden, CALL)                                        \
    do {                                                                             \
        if ((kind) == 0) DRIL_DISPATCH_H(0, hidden, CALL)                            \
        else if ((kind) == 1) DRIL_DISPATCH_H(1, hidden, CALL)                       \
        else if ((kind) == 2) DRIL_DISPATCH_H(2, hidden, CALL)                       \
        else if ((kind) == 3) DRIL_DISPATCH_H(3, hidden, CALL)                       \
        else if ((kind) == 4) DRIL_DISPATCH_H(4, hidden, CALL)                       \
        else return hipErrorInvalidValue;                                            \
    } while (0)

hipError_t launch_fold_partials(const double* partials, int nblocks, double* out16, hipStream_t s) {
    fold_partials_kernel<<<1, 256, 0, s>>>(partials, nblocks, out16);
    return hipGetLastError();
}
hipError_t launch_build_wimg_split(const float* params, NetOff off, int H, void* w2p, void* w2tp, hipStream_t s) {
    const int total = (H / 32) * (H / 32) * 2 * 64;
    build_wimg_split_kernel<<<(total + 255) / 256, 256, 0, s>>>(params, off, H, (u32x4*)w2p, (u32x4*)w2tp);
    return hipGetLastError();
}
hipError_t launch_build_wimg(const float* params, NetOff off, int H, float* w2a, float* w2ta, hipStream_t s) {
    build_wimg_kernel<<<(H * H + 255) / 256, 256, 0, s>>>(params, off, H, w2a, w2ta);
    return hipGetLastError();
}

hipError_t launch_env_reset(int kind, int E, uint64_t seed0, float* state, int32_t* sc, uint32_t* ep, uint32_t* gs, float* dr, hipStream_t s) {
    const int blocks = (E + 255) / 256;
    if (kind == 0) env_reset_kernel<0><<<blocks, 256, 0, s>>>(E, seed0, state, sc, ep, gs, dr);
    else if (kind == 3 || kind == 4) env_reset_kernel<3><<<blocks, 256, 0, s>>>(E, seed0, state, sc, ep, gs, dr);
    else if (kind == 6) env_reset_kernel<6><<<blocks, 256, 0, s>>>(E, seed0, state, sc, ep, gs, dr);
    else env_reset_kernel<1><<<blocks, 256, 0, s>>>(E, seed0, state, sc, ep, gs, dr);
    return hipGetLastError();
}
hipError_t launch_env_observe(int kind, int E, const float* state, float* obs, hipStream_t s) {
    const int blocks = (E + 255) / 256;
    if (kind == 0) env_observe_kernel<0><<<blocks, 256, 0, s>>>(E, state, obs);
    else if (kind == 1) env_observe_kernel<1><<<blocks, 256, 0, s>>>(E, state, obs);
    else if (kind == 2) env_observe_kernel<2><<<blocks, 256, 0, s>>>(E, state, obs);
    else if (kind == 6) env_observe_kernel<6><<<blocks, 256, 0, s>>>(E, state, obs);
    else env_observe_kernel<3><<<blocks, 256, 0, s>>>(E, state, obs);
    return hipGetLastError();
}
hipError_t launch_env_step(int kind, int E, uint64_t seed0, int episode_len, int fixed_len, int action_start, const void* actions,
                           float* state, int32_t* sc, uint32_t* ep, uint32_t* gs, float* rew, uint8_t* term, uint8_t* trunc,
                           float* tobs, MonitorArgs mon, hipStream_t s) {
    const int blocks = (E + 255) / 256;
    if (kind == 0) env_step_kernel<0><<<blocks, 256, 0, s>>>(E, seed0, episode_len, fixed_len, action_start, actions, state, sc, ep, gs, rew, term, trunc, tobs, mon);
    else if (kind == 1) env_step_kernel<1><<<blocks, 256, 0, s>>>(E, seed0, episode_len, fixed_len, action_start, actions, state, sc, ep, gs, rew, term, trunc, tobs, mon);
    else if (kind == 2) env_step_kernel<2><<<blocks, 256, 0, s>>>(E, seed0, episode_len, fixed_len, action_start, actions, state, sc, ep, gs, rew, term, trunc, tobs, mon);
    else if (kind == 3) env_step_kernel<3><<<blocks, 256, 0, s>>>(E, seed0, episode_len, fixed_len, action_start, actions, state, sc, ep, gs, rew, term, trunc, tobs, mon);
    else if (kind == 6) env_step_kernel<6><<<blocks, 256, 0, s>>>(E, seed0, episode_len, fixed_len, action_start, actions, state, sc, ep, gs, rew, term, trunc, tobs, mon);
    else env_step_kernel<4><<<blocks, 256, 0, s>>>(E, seed0, episode_len, fixed_len, action_start, actions, state, sc, ep, gs, rew, term, trunc, tobs, mon);
    return hipGetLastError();
}
hipError_t launch_monitor_collect(const uint8_t* flags, const float* ep_ret, const int32_t* ep_len, int E, int T, int W, int* cnt,
                                  float* ring_ret, int32_t* ring_len, int* meta, hipStream_t s) {
    monitor_count_kernel<<<T < 1024 ? T : 1024, 256, 0, s>>>(flags, E, T, cnt);
    monitor_collect_kernel<<<1, 1024, 0, s>>>(flags, ep_ret, ep_len, E, T, W, cnt, ring_ret, ring_len, meta);
    return hipGetLastError();
}

hipError_t launch_norm_step(int kind, const NormStepArgs& a, int nblocks, hipStream_t s) {
    if (kind == 0) norm_step_kernel<0><<<nblocks, 256, 0, s>>>(a); else if (kind == 1) norm_step_kernel<1><<<nblocks, 256, 0, s>>>(a); else if (kind == 2) norm_step_kernel<2><<<nblocks, 256, 0, s>>>(a);
    else if (kind == 3) norm_step_kernel<3><<<nblocks, 256, 0, s>>>(a); else if (kind == 6) norm_step_kernel<6><<<nblocks, 256, 0, s>>>(a); else norm_step_kernel<4><<<nblocks, 256, 0, s>>>(a);
    return hipGetLastError();
}
hipError_t launch_norm_apply(const NormApplyArgs& a, hipStream_t s) {
    int blocks = (a.E + 255) / 256; if (blocks > 1024) blocks = 1024;
    norm_apply_kernel<<<blocks, 256, 0, s>>>(a);
    return hipGetLastError();
}
hipError_t launch_obs_partials(int kind, int E, const float* state, float* raw, double* partials, int nblocks, hipStream_t s) {
    if (kind == 0) obs_partials_kernel<0><<<nblocks, 256, 0, s>>>(E, state, raw, partials);
    else if (kind == 1) obs_partials_kernel<1><<<nblocks, 256, 0, s>>>(E, state, raw, partials);
    else if (kind == 2) obs_partials_kernel<2><<<nblocks, 256, 0, s>>>(E, state, raw, partials);
    else if (kind == 6) obs_partials_kernel<6><<<nblocks, 256, 0, s>>>(E, state, raw, partials);
    else obs_partials_kernel<3><<<nblocks, 256, 0, s>>>(E, state, raw, partials);
    return hipGetLastError();
}
hipError_t launch_norm_obs_apply(const NormObsArgs& a, hipStream_t s) {
    int blocks = (a.E * a.D + 255) / 256; if (blocks > 1024) blocks = 1024;
    norm_obs_apply_kernel<<<blocks, 256, 0, s>>>(a);
    return hipGetLastError();
}
hipError_t launch_rew_partials(int E, const float* rew_raw, float* disc_returns, float gamma, int update, double* partials, int nblocks, hipStream_t s) {
    rew_partials_kernel<<<nblocks, 256, 0, s>>>(E, rew_raw, disc_returns, gamma, update, partials);
    return hipGetLastError();
}
hipError_t launch_norm_rew_apply(const NormRewArgs& a, hipStream_t s) {
    int blocks = (a.E + 255) / 256; if (blocks > 1024) blocks = 1024;
    norm_rew_apply_kernel<<<blocks, 256, 0, s>>>(a);
    return hipGetLastError();
}

hipError_t launch_policy(int kind, int hidden, const PolicyArgs& a, int max_blocks, hipStream_t s) {
    const int64_t ntiles = (a.B + kTile - 1) / kTile;
    int blocks = (int)((ntiles + 3) / 4);
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks < 1) blocks = 1;
#define CALL(K, HH)                                                                                           \
    {                                                                                                         \
        const size_t lds = fwd_lds_bytes<K, HH, (HH > 64)>();                                                 \
        static bool attr_set = false;                                                                         \
        if (!attr_set) { hipError_t e = hipFuncSetAttribute((const void*)policy_kernel<K, HH, (HH > 64)>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return e; attr_set = true; }                                                 \
        policy_kernel<K, HH, (HH > 64)><<<blocks, 256, lds, s>>>(a);                                          \
    }
    DRIL_DISPATCH_FWD(kind, hidden, CALL);
#undef CALL
    return hipGetLastError();
}

hipError_t launch_rollout(int kind, int hidden, const RolloutArgs& a, hipStream_t s) {
    const int blocks = (a.E + 4 * kTile - 1) / (4 * kTile);
#define CALL(K, HH)                                                                                           \
    {                                                                                                         \
        const size_t lds = fwd_lds_bytes<K, HH, (HH > 64)>();                                                 \
        static bool attr_set = false;                                                                         \
        if (!attr_set) { hipError_t e = hipFuncSetAttribute((const void*)rollout_kernel<K, HH, (HH > 64)>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return e; attr_set = true; }                                                 \
        rollout_kernel<K, HH, (HH > 64)><<<blocks, 256, lds, s>>>(a);                                         \
    }
    DRIL_DISPATCH_ENV(kind, hidden, CALL);
#undef CALL
    return hipGetLastError();
}

hipError_t launch_gae(int E, int T, float gamma, float lam, const float* rew, const float* val, const uint8_t* flags,
                      const float* boot, const float* last_values, float* adv, float* ret, hipStream_t s) {
    gae_kernel<<<(E + 63) / 64, 64, 0, s>>>(E, T, gamma, lam, rew, val, flags, boot, last_values, adv, ret);
    return hipGetLastError();
}

hipError_t launch_adv_moments(const MomentsArgs& a, int nblocks, hipStream_t s) {
    adv_moments_kernel<<<nblocks, 256, 0, s>>>(a);
    return hipGetLastError();
}
hipError_t launch_epoch_moments(const float* adv, int64_t N, int64_t B, int nb, uint64_t key, int bits, double* block_tables, int nblocks,
                                double* table3, const int* stop_flag, hipStream_t s) {
    epoch_moments_kernel<<<nblocks, 256, (size_t)2 * nb * sizeof(double), s>>>(adv, N, B, nb, key, bits, block_tables, stop_flag);
    epoch_moments_finalize_kernel<<<(nb + 63) / 64, 64, 0, s>>>(block_tables, nblocks, nb, N, B, table3, stop_flag);
    return hipGetLastError();
}
hipError_t launch_moments_finalize(const double* partials, int nblocks, double* out3, double n_local, const int* stop_flag, hipStream_t s) {
    moments_finalize_kernel<<<1, 256, 0, s>>>(partials, nblocks, out3, n_local, stop_flag);
    return hipGetLastError();
}

template <int KIND, int H> static size_t grad_split_lds_bytes() {
    constexpr int D = EnvSpec<KIND>::D, A = EnvSpec<KIND>::A, N = 4;
    constexpr int wa = NetLdsSplit<D, H, A>::END + N * GradScratchSplit<D, H, A>::SIZE;
    constexpr int wc = NetLdsSplit<D, H, 1>::END + N * GradScratchSplit<D, H, 1>::SIZE;
    return sizeof(float) * (wa > wc ? wa : wc);
}
template <int KIND, int H> static size_t grad_wide_lds_bytes() {
    constexpr int D = EnvSpec<KIND>::D, A = EnvSpec<KIND>::A;
    constexpr int wa = WideScratch<D, H, A>::SIZE, wc = WideScratch<D, H, 1>::SIZE;
    return sizeof(float) * (wa > wc ? wa : wc);
}
template <int KIND> static size_t grad_pair_lds_bytes() {
    constexpr int D = EnvSpec<KIND>::D, A = EnvSpec<KIND>::A;
    constexpr int wa = PairLds<D, A>::END, wc = PairLds<D, 1>::END;
    return sizeof(float) * (wa > wc ? wa : wc);
}
template <int KIND, int H> static size_t grad_wide_split_lds_bytes() {
    constexpr int D = EnvSpec<KIND>::D, A = EnvSpec<KIND>::A;
    constexpr int wa = WideSplitScratch<D, H, A>::SIZE, wc = WideSplitScratch<D, H, 1>::SIZE;
    return sizeof(float) * (wa > wc ? wa : wc);
}
hipError_t launch_ppo_grad(int kind, int hidden, const GradArgs& a, hipStream_t s) {
#define CALLR(K, HH, R)                                                                                       \
    {                                                                                                         \
        const size_t lds = grad_lds_bytes<K, HH>();                                                           \
        static bool attr_set = false;                                                                         \
        if (!attr_set) { hipError_t e = hipFuncSetAttribute((const void*)ppo_grad_kernel<K, HH, R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return e; attr_set = true; }                                                 \
        ppo_grad_kernel<K, HH, R><<<2 * a.G, 256, lds, s>>>(a);                                               \
    }
#define CALLW(K, HH, R)                                                                                       \
    {                                                                                                         \
        const size_t lds = grad_wide_lds_bytes<K, HH>();                                                      \
        static bool attr_set = false;                                                                         \
        if (!attr_set) { hipError_t e = hipFuncSetAttribute((const void*)ppo_grad_wide_kernel<K, HH, R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return e; attr_set = true; }                                                 \
        ppo_grad_wide_kernel<K, HH, R><<<2 * a.G, HH * 2, lds, s>>>(a);                                       \
    }
    if (hidden == 64 && a.variant == 2 && a.rec) {   // two waves per tile, two waves per SIMD
#define CALLP(K) { const size_t lds = grad_pair_lds_bytes<K>(); static bool attr_set = false; \
        if (!attr_set) { hipError_t e = hipFuncSetAttribute((const void*)ppo_grad_pair_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (e != hipSuccess) return e; attr_set = true; } \
        ppo_grad_pair_kernel<K><<<(a.G + a.Gc) / 2, 256, lds, s>>>(a); }
        if (kind == 0) CALLP(0) else if (kind == 3) CALLP(3) else if (kind == 4) CALLP(4) else CALLP(1)
#undef CALLP
        return hipGetLastError();
    }
    if (hidden > 64 && a.variant && a.rec) {      // wide nets on the bf16 matrix cores
#define CALLWS(K, HH)                                                                                         \
    {                                                                                                         \
        const size_t lds = grad_wide_split_lds_bytes<K, HH>();                                                \
        static bool attr_set = false;                                                                         \
        if (!attr_set) { hipError_t e = hipFuncSetAttribute((const void*)ppo_grad_wide_split_kernel<K, HH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return e; attr_set = true; }                                                 \
        ppo_grad_wide_split_kernel<K, HH><<<2 * a.G, HH * 2, lds, s>>>(a);                                    \
    }
#define CALLWSH(K) { if (hidden == 256) CALLWS(K, 256) else if (hidden == 128) CALLWS(K, 128) else return hipErrorInvalidValue; }
        if (kind == 0) CALLWSH(0) else if (kind == 3) CALLWSH(3) else if (kind == 4) CALLWSH(4) else CALLWSH(1)
#undef CALLWSH
#undef CALLWS
        return hipGetLastError();
    }
    if (hidden > 64) {
#define CALLWK(K, HH) { if (a.rec) CALLW(K, HH, true) else CALLW(K, HH, false) }
#define CALLWH(K) { if (hidden == 256) CALLWK(K, 256) else if (hidden == 128) CALLWK(K, 128) else return hipErrorInvalidValue; }
        if (kind == 0) CALLWH(0) else if (kind == 3) CALLWH(3) else if (kind == 4) CALLWH(4) else CALLWH(1)
#undef CALLWH
#undef CALLWK
        return hipGetLastError();
    }
    if (a.variant && hidden == 64) {              // bf16 matrix cores, fp32-equivalent operand splitting
#define CALLS(K, R)                                                                                           \
    {                                                                                                         \
        const size_t lds = grad_split_lds_bytes<K, 64>();                                                     \
        static bool attr_set = false;                                                                         \
        if (!attr_set) { hipError_t e = hipFuncSetAttribute((const void*)ppo_grad_split_kernel<K, 64, R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return e; attr_set = true; }                                                 \
        ppo_grad_split_kernel<K, 64, R><<<a.G + a.Gc, 256, lds, s>>>(a);                                         \
    }
#define CALLSK(K) { if (a.rec) CALLS(K, true) else CALLS(K, false) }
        if (kind == 0) CALLSK(0) else if (kind == 3) CALLSK(3) else if (kind == 4) CALLSK(4) else CALLSK(1)
#undef CALLSK
#undef CALLS
        return hipGetLastError();
    }
#define CALL(K, HH) { if (a.rec) CALLR(K, HH, true) else CALLR(K, HH, false) }
    DRIL_DISPATCH(kind, hidden, CALL);
#undef CALL
#undef CALLR
#undef CALLW
    return hipGetLastError();
}

hipError_t launch_pack_records(int kind, int64_t N, const float* obs, const void* act, const float* adv, const float* logp, const float* ret, float4* rec, hipStream_t s) {
    int blocks = (int)((N + 255) / 256); if (blocks > 8192) blocks = 8192;
    if (kind == 0) pack_records_kernel<0><<<blocks, 256, 0, s>>>(N, obs, act, adv, logp, ret, rec);
    else if (kind == 3) pack_records_kernel<3><<<blocks, 256, 0, s>>>(N, obs, act, adv, logp, ret, rec);
    else if (kind == 4) pack_records_kernel<4><<<blocks, 256, 0, s>>>(N, obs, act, adv, logp, ret, rec);
    else pack_records_kernel<1><<<blocks, 256, 0, s>>>(N, obs, act, adv, logp, ret, rec);
    return hipGetLastError();
}
hipError_t launch_grad_reduce(const ReduceArgs& a, hipStream_t s) {
    grad_reduce_kernel<<<(a.P + 31) / 32, 1024, 0, s>>>(a);
    return hipGetLastError();
}
hipError_t launch_finish_small(const ReduceArgs& r, const AdamArgs& a, hipStream_t s) {
    ppo_finish_small_kernel<<<1, 1024, 0, s>>>(r, a);
    return hipGetLastError();
}
hipError_t launch_grad_norm(const float* flat, int P, double* norm_partials, const int* stop_flag, hipStream_t s) {
    grad_norm_kernel<<<(P + 31) / 32, 256, 0, s>>>(flat, P, norm_partials, stop_flag);
    return hipGetLastError();
}
hipError_t launch_adam(const AdamArgs& a, hipStream_t s) {
    adam_kernel<<<(a.P + 255) / 256, 256, 0, s>>>(a);
    return hipGetLastError();
}
hipError_t launch_explained_var(const float* val, const float* ret, int64_t N, double* partials, int nblocks, hipStream_t s) {
    explained_var_kernel<<<nblocks, 256, 0, s>>>(val, ret, N, partials);
    return hipGetLastError();
}

static void kind_dims(int kind, int& D, int& A, bool& disc) {
    switch (kind) {
        case 0: D = 4; A = 2; disc = true; break;
        case 3: D = 2; A = 3; disc = true; break;
        case 4: D = 2; A = 1; disc = false; break;
        default: D = 3; A = 1; disc = false; break;
    }
}
int slab_size_actor(int kind, int hidden) {
    int D, A; bool disc; kind_dims(kind, D, A, disc);
    const NetOff n = net_off(0, D, hidden, hidden, A);
    return (n.end + (disc ? 0 : A) + 8 + 3) / 4 * 4 + 0;
}
int slab_size_critic(int kind, int hidden) {
    int D, A_; bool disc_; kind_dims(kind, D, A_, disc_);
    const NetOff n = net_off(0, D, hidden, hidden, 1);
    return (n.end + 8 + 3) / 4 * 4;
}

}  // namespace dril
