// dril_internal.h — kernel argument blocks and launcher prototypes shared by dril_kernels.hip and dril_api.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>

#include "dril_device.h"

// The forward of rollout_kernel / rollout_duo_kernel / policy_kernel puts ONE operand on f16 pieces: kTanhScale kWScale W2 (h1 = tanh is bounded, L1 and L3 are f32).
// A W2 entry at or beyond this magnitude would overflow its hi piece (2.885 x 64 x 354.7 = 65 504), so the host tracks max |W2| of both nets (dril_set_params, and with
// every update's statistics) and launches the f32-MFMA instantiations of the same kernels instead (exact_f32 below) — the reference puts no range limit on parameters.
constexpr float kFwdSplitMaxW = 350.0f;
namespace dril {

struct PolicyArgs {
    const float* params; const float* obs; int64_t B; const void* noise;
    void* actions; float* values; float* logp; float* entropy;
    int mode; int action_start; int log_std_off; uint64_t seed; uint32_t call_counter;
    const uint32_t* gstep; uint64_t env_seed0;   // step-granular rollout: draw from the env-keyed Philox stream (same as rollout_kernel)
    float* obs_out;                              // optional copy of the consumed observations (rollout buffer slice)
    const uint8_t* only_where;                   // optional: waves with no flagged sample skip (bootstrap critic on truncated envs)
    const float* w2a_actor; const float* w2a_critic;   // wide nets: pre-tiled W2 images in global memory
    int deterministic;                           // predict_actions(...; deterministic = true): mode(d) instead of rand(d)
    const float* boot_obs; const uint8_t* boot_where; float* boot_out;   // fused V(terminal_observation) of the PREVIOUS env step (trajectory.jl:57-61)
    int exact_f32;                               // 1: the f32-MFMA forward (w2a_* = the f32 W2 images), 0: the f16 two-piece forward (w2a_* = the pre-split fragment streams)
    NetOff actor, critic;
};

// RunningMeanStd of NormalizeWrapperEnv (normalizeWrapperEnv.jl:8-19); kept ping-pong so that every block of the
// kernel that merges new batch moments reads the old statistics while block 0 writes the new ones
struct RmsState { float mean[8]; float var[8]; long long count; };

struct NormObsArgs {
    int E, D, update, nblocks; const float* raw; const double* partials; const RmsState* in; RmsState* out;
    float* obs_n; float clip, eps; int norm_obs;
    long long n_stats;   // envs behind the partial sums (0 = E: this rank only; world * E after the all-reduce of the folded row)
};
struct NormRewArgs {
    int E, D, update, nblocks, norm_obs, norm_reward; const float* rew_raw; const double* partials; const RmsState* in; RmsState* out;
    const RmsState* obs_stats; float* rew_out; float* disc_returns; const uint8_t* term; const uint8_t* trunc;
    float* tobs; float clip_obs, clip_reward, eps; uint8_t* flags_out;
    long long n_stats;
};

struct RolloutArgs {
    const float* params;
    float* state; int32_t* step_count; uint32_t* episode; uint32_t* gstep;
    float* obs; void* act; float* rew; float* logp; float* val; float* boot; uint8_t* flags; float* last_values;
    const void* noise;
    int E, T, episode_len, fixed_len, action_start, log_std_off;
    uint64_t env_seed0;
    const float* w2a_actor; const float* w2a_critic; int exact_f32;   // as in PolicyArgs
    float* mon_cur_ret; int32_t* mon_cur_len; float* ep_ret; int32_t* ep_len;   // MonitorWrapperEnv (null = off)
    NetOff actor, critic;
};

struct MomentsArgs {
    const float* adv; const int64_t* perm; int64_t pos0, count, N, idx_lo, n_local; uint64_t perm_key; int perm_bits;
    double* partials; const int* stop_flag;
    const int32_t* perm32;    // the epoch's order as 32-bit indices (epoch_index_kernel, N < 2^31): takes precedence over perm
};

struct GradArgs {
    const float* params; const float* obs; const void* actions; const float* adv; const float* ret;
    const float* logp_old; const float* val_old;
    const int64_t* perm; int64_t pos0, count, N, idx_lo, n_local; uint64_t perm_key; int perm_bits;
    const int32_t* perm32;    // the epoch's DataLoader order written out as 32-bit indices (epoch_index_kernel; N < 2^31): takes precedence over perm / the keyed bijection
    const float* w2a_actor; const float* w2ta_actor; const float* w2a_critic; const float* w2ta_critic;   // wide nets: pre-tiled W2 / W2' images
    const u32x4* w2p_actor; const u32x4* w2tp_actor; const u32x4* w2p_critic; const u32x4* w2tp_critic;   // wide nets, bf16-split form: pre-split fragment streams
    const float4* rec;   // packed minibatch records [N][2] x float4: {obs0..3} {action bits, adv, logp_old, ret}; null = gather from the SoA buffers
    const double* adv_stats;
    float invB, clip_range, ent_coef, vf_coef, clip_range_vf;
    int has_clip_vf, normalize_adv, action_start, log_std_off;
    float* slabs_actor; float* slabs_critic; int slab_a, slab_c, G;
    int Gc;       // critic workgroups (== G except in ppo_grad_pair_kernel, where G / Gc count the PAIRS of each net: the chip-filling grid is divided between the nets by their measured cost per tile)
    unsigned long long* dbg;   // -DDRIL_STAMPS diagnostic buffer (12 x u64 per wave), else unused
    int inline_moments;   // small minibatches: every actor workgroup computes the advantage moments itself (adv_stats == nullptr), saving two launches per optimiser step
    int variant;  // hidden 64: 0 = ppo_grad_kernel (f32 MFMA), 2 = ppo_grad_pair_kernel (bf16 x 3 operand split); wide nets: 0 = ppo_grad_wide_kernel, 1 = ppo_grad_wide_split_kernel
    const int* stop_flag;
    NetOff actor, critic;
};

// ppo_update_small_kernel (dril_update_small.hip): a run of optimiser steps [step0, step0 + nsteps) of the epochs x minibatches sequence in one launch of two persistent workgroups (actor, critic)
struct SmallUpdateArgs {
    float* params; float* adam_m; float* adam_v; float* bt; int step_parity;
    const float4* rec; const float* val_old;
    const int64_t* perm;          // injected DataLoader order [epochs][N], or null: the keyed bijection with keys[epoch]
    const uint64_t* keys; int perm_bits;
    int64_t N, B; int nb, step0, nsteps;
    float* step_stats; float* norm_out; int* nan_flag; int* stop_flag;
    float lr, beta1, beta2, eps, max_grad_norm, target_kl, ent_coef, vf_coef, clip_range, clip_range_vf;
    int has_max_grad_norm, has_target_kl, has_clip_vf, normalize_adv, action_start;
    int P, Pa, Pc;
    unsigned long long* dbg;      // -DDRIL_STAMPS diagnostic buffer (16 x u64 per wave), else unused
    int debug_solo;               // DRIL_SMALL_DEBUG_SOLO (tests): launch the actor's workgroup alone — it must give up waiting and report, not hang
    unsigned long long* xchg;     // kSmallXchgWords words {sequence, value}: the messages between the actor's and the critic's workgroup (zeroed by the launcher)
};
constexpr int kSmallXchgWords = 4 * 16;
hipError_t launch_ppo_update_small(int kind, const SmallUpdateArgs& a, hipStream_t s);
hipError_t launch_epoch_index(int64_t N, uint64_t key, int bits, int32_t* out, hipStream_t s);   // N < 2^31

struct ReduceArgs {
    const float* slabs_actor; const float* slabs_critic; int slab_a, slab_c, G, Gc;   // G actor slabs, Gc critic slabs
    int P, Pa, Pc; float* flat; double* norm_partials; double n_samples_local; const int* stop_flag;
};

struct AdamArgs {
    float* params; float* m; float* v; const float* flat; int P;
    const double* norm_partials; int n_partials;
    int norm_from_flat;                            // data-parallel steps: |g|^2 summed here from the all-reduced flat gradient (every block, same order) instead of a grad_norm_kernel launch in between
    float* bt; int step_parity;
    float beta1, beta2, eps, lr, max_grad_norm, target_kl, ent_coef, vf_coef;
    int has_max_grad_norm, has_target_kl, use_stats;
    float* step_stats; float* norm_out; int* nan_flag; const int* stop_flag; int* stop_flag_w;
};

// raise a kernel's dynamic-LDS limit once per (kernel, device): a process may hold handles on several devices, and the attribute is per device context
// (dril_kernels.hip; the launchers of every kernel that takes more than 64 KB call it before each launch — a vector scan under a mutex after the first call)
hipError_t set_max_dynamic_lds(const void* fn, size_t bytes);
// n-tiles of 32 samples that one workgroup of ppo_grad_wide_split_kernel takes through its stages together (dril_grad_wide.hip); ppo_step sizes the grid by it
constexpr int kWideSplitNT = 2;

hipError_t launch_env_reset(int kind, int E, uint64_t seed0, float* state, int32_t* sc, uint32_t* ep, uint32_t* gs, float* dr, hipStream_t s);
hipError_t launch_env_observe(int kind, int E, const float* state, float* obs, hipStream_t s);
struct MonitorArgs { float* cur_ret; int32_t* cur_len; float* ep_ret; int32_t* ep_len; uint8_t* flags_out; };
hipError_t launch_env_step(int kind, int E, uint64_t seed0, int episode_len, int fixed_len, int action_start, const void* actions,
                           float* state, int32_t* sc, uint32_t* ep, uint32_t* gs, float* rew, uint8_t* term, uint8_t* trunc,
                           float* tobs, MonitorArgs mon, hipStream_t s);
hipError_t launch_monitor_collect(const uint8_t* flags, const float* ep_ret, const int32_t* ep_len, int E, int T, int W, int* cnt,
                                  float* ring_ret, int32_t* ring_len, int* meta, hipStream_t s);
hipError_t launch_policy(int kind, int hidden, const PolicyArgs& a, int max_blocks, hipStream_t s);
struct NormStepArgs {   // fused act! of NormalizeWrapperEnv over MultiThreadedParallelEnv: physics + auto-reset + all partial moments
    int E, episode_len, fixed_len, action_start; uint64_t seed0; float gamma; int update_ret;
    const void* actions; float* state; int32_t* step_count; uint32_t* episode; uint32_t* gstep; float* disc_returns;
    float* rew_raw; uint8_t* term; uint8_t* trunc; uint8_t* flags_out; float* tobs_raw; float* obs_raw; double* partials;
    float* mon_cur_ret; int32_t* mon_cur_len; float* ep_ret; int32_t* ep_len;   // MonitorWrapperEnv (null = off)
};
struct NormApplyArgs {
    int E, D, nblocks, update_obs, update_ret, norm_obs, norm_reward;
    const double* partials; const RmsState* obs_in; RmsState* obs_out; const RmsState* ret_in; RmsState* ret_out;
    const float* rew_raw; float* rew_out; float* disc_returns; const uint8_t* term; const uint8_t* trunc; float* tobs; const float* obs_raw; float* obs_n;
    float clip_obs, clip_reward, eps;
    long long n_stats;
};
hipError_t launch_fold_partials(const double* partials, int nblocks, double* out16, hipStream_t s);
hipError_t launch_norm_step(int kind, const NormStepArgs& a, int nblocks, hipStream_t s);
hipError_t launch_norm_apply(const NormApplyArgs& a, hipStream_t s);
hipError_t launch_obs_partials(int kind, int E, const float* state, float* raw, double* partials, int nblocks, hipStream_t s);
hipError_t launch_norm_obs_apply(const NormObsArgs& a, hipStream_t s);
hipError_t launch_rew_partials(int E, const float* rew_raw, float* disc_returns, float gamma, int update, double* partials, int nblocks, hipStream_t s);
hipError_t launch_norm_rew_apply(const NormRewArgs& a, hipStream_t s);
hipError_t launch_rollout(int kind, int hidden, const RolloutArgs& a, hipStream_t s);
// carry: gae_chunks(T) x E 64-bit words {tag | f32}, zero at allocation; tag: a non-zero number no earlier launch on this carry buffer used; err: device int, set to 1 if a
// workgroup gave up waiting for its predecessor (the advantages it wrote are NaN)
int gae_chunks(int T);
hipError_t launch_gae(int E, int T, float gamma, float lam, const float* rew, const float* val, const uint8_t* flags,
                      const float* boot, const float* last_values, float* adv, float* ret, unsigned long long* carry, unsigned tag, int* err, hipStream_t s);
hipError_t launch_adv_moments(const MomentsArgs& a, int nblocks, hipStream_t s);
hipError_t launch_epoch_moments(const float* adv, int64_t N, int64_t B, int nb, uint64_t key, int bits, double* block_tables, int nblocks,
                                double* table3, const int* stop_flag, hipStream_t s);
hipError_t launch_moments_finalize(const double* partials, int nblocks, double* out3, double n_local, const int* stop_flag, hipStream_t s);
hipError_t launch_ppo_grad(int kind, int hidden, const GradArgs& a, hipStream_t s);
// per translation unit (dril_grad_f32.hip / dril_grad_pair.hip / dril_grad_wide.hip); launch_ppo_grad picks one
hipError_t launch_ppo_grad_f32(int kind, int hidden, const GradArgs& a, hipStream_t s);
hipError_t launch_ppo_grad_pair(int kind, const GradArgs& a, hipStream_t s);
hipError_t launch_ppo_grad_wide(int kind, int hidden, const GradArgs& a, hipStream_t s);
hipError_t launch_pack_records(int kind, int64_t N, const float* obs, const void* act, const float* adv, const float* logp, const float* ret, float4* rec, hipStream_t s);
hipError_t launch_grad_reduce(const ReduceArgs& a, hipStream_t s);
hipError_t launch_grad_norm(const float* flat, int P, double* norm_partials, const int* stop_flag, hipStream_t s);
hipError_t launch_adam(const AdamArgs& a, hipStream_t s);
hipError_t launch_finish_small(const ReduceArgs& r, const AdamArgs& a, hipStream_t s);   // grad_reduce + norm + Adam in one workgroup (few slabs)
hipError_t launch_explained_var(const float* val, const float* ret, int64_t N, double* partials, int nblocks, hipStream_t s);
hipError_t launch_build_wimg(const float* params, NetOff off, int H, float* w2a, float* w2ta, hipStream_t s);
hipError_t launch_w2_absmax(const float* params, NetOff actor, NetOff critic, int H, unsigned* out_bits, hipStream_t s);   // max |W2| of both nets, as float bits
hipError_t launch_build_wimg_split(const float* params, NetOff off, int H, void* w2p, void* w2tp, void* w2pf, hipStream_t s);
int slab_size_actor(int kind, int hidden);
int slab_size_critic(int kind, int hidden);

// generic (any obs / action / hidden width) on-policy path, dril_generic.hip; same argument blocks and slab format as the fused kernels
constexpr int kMaxHidden = 4;
// any-depth MLP of the generic path: nh hidden layers of widths H[0..nh-1], activation act (0 tanh, 1 relu); layer l (0-based, nh + 1 layers) maps
// dim(l) -> dim(l + 1) with dim(0) = D, dim(l) = H[l-1], dim(nh + 1) = out
struct GenericDims { int D, A, nh, H[kMaxHidden], discrete, act; };
int generic_net_size(const GenericDims& d, int out);   // parameters of one net {W_1 b_1 ... W_{nh+1} b_{nh+1}}
struct GenericWs { float* p = nullptr; size_t cap = 0; };          // grow-only device workspace (floats), owned by the handle
hipError_t generic_policy(const GenericDims& d, const PolicyArgs& a, GenericWs& ws, hipStream_t s);
hipError_t generic_ppo_grad(const GenericDims& d, const GradArgs& a, GenericWs& ws, hipStream_t s);
int generic_slab_size(const GenericDims& d, bool actor);
int generic_pick_slabs(const GenericDims& d, int64_t count, int Gmax);   // slabs (row chunks) for a minibatch of `count` rows; < 1: does not fit
void generic_ws_free(GenericWs& ws);
hipError_t generic_select(int64_t n, const uint8_t* where, const float* src, float* dst, hipStream_t s);   // dst[i] = src[i] where where[i] != 0

}  // namespace dril
