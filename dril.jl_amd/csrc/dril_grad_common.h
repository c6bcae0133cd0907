// dril_grad_common.h — shared by the PPO update kernels (dril_grad_f32.hip, dril_grad_pair.hip, dril_grad_wide.hip): minibatch tile gather, loss head (ppo.jl:365-407 per sample), diagnostic stamps
#pragma once
#include "dril_internal.h"
#include "dril_heads.h"

namespace dril {

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

// a lane constant the optimiser cannot see through: image addresses derived from it are rebuilt per tile (2-3 VALU) instead of being hoisted out of the tile loop as
// loop invariants, where they occupy registers for the whole kernel (ppo_grad_wide_split_kernel: 92 -> 12 spilled registers)
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }

// sum of one double per thread over the workgroup (fixed order: wave butterflies, then the waves in index order); sh = 16 doubles of LDS; two barriers inside
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

// one lane's share of a minibatch tile (DataLoader gather, ppo.jl:188-195).  Loaded one tile AHEAD of its use so the
// random-gather latency (~2 us under load, fully exposed in v1: 23 % of wave time in s_waitcnt) hides under the
// previous tile's MFMAs; the loop body issues no other vector-memory op, so the loads stay in flight until first use.
// packed minibatch records: RS float4 per sample — {obs0..3}{action bits, adv, logp_old, ret} for D <= 4, {obs0..3}{obs4..7}{scalars} for D <= 8
template <int D> struct RecLayout { static constexpr int RS = D <= 4 ? 2 : 3; };
template <int O, int KS = 2> struct TileIn { float xk[KS]; float s0, s1; int act; float xa[O]; bool valid; float4 raw; float4 raw2; };   // raw2: the scalar quad of a 3-quad record (KS = 4) only

// the two halves of load_tile, so that a kernel can request the INDEX of a tile one pass before its records (ppo_grad_wide_split_kernel: perm32[p] -> wait -> rec[idx] is a
// dependent pair of memory round trips; issued back to back after a barrier they were ~2 k cycles of every pass): tile_index issues the load of the epoch order's entry
// (or evaluates the keyed bijection), load_tile_at the record / field loads for a known index
// (g32: the epoch order's 32-bit entry exactly as loaded — widening it at the load would put the `s_waitcnt vmcnt(0)` of its first use right behind the request: a kernel
// that requests an entry one pass ahead, behind an LDS-DMA, then waits for both on the spot; tile_gidx() widens it where the index is consumed)
struct TileIdx { int64_t gidx; int32_t g32; bool is32; bool inb; };
__device__ __forceinline__ int64_t tile_gidx(const TileIdx& t) { return t.is32 ? (int64_t)t.g32 : t.gidx; }
__device__ __forceinline__ TileIdx tile_index(const GradArgs& a, int64_t tile, int64_t ntiles, int c) {
    const bool live = tile < ntiles;
    const int64_t i = (live ? tile : ntiles - 1) * kTile + c;
    TileIdx r; r.inb = live && i < a.count; r.gidx = 0; r.g32 = 0; r.is32 = a.perm32 != nullptr;
    const int64_t p = a.pos0 + (r.inb ? i : 0);
    if (a.perm32) r.g32 = a.perm32[p];
    else if (a.perm) r.gidx = a.perm[p];
    else {
        // (the keyed bijection's masks and shift counts are loop invariants of the caller's tile loop: hidden from the optimiser here, they are rebuilt per call on this
        // cold path instead of living in ~50 scalar registers — and their spills — across the whole loop of a kernel that is at its register limit)
        uint64_t key = a.perm_key; int bits = a.perm_bits; int64_t n = a.N;
        asm volatile("" : "+s"(key), "+s"(bits), "+s"(n));
        r.gidx = bits ? perm_index(p, n, key, bits) : p;                      // bits 0 = identity order
    }
    return r;
}
template <int KIND, int O, int HEAD, bool REC>
__device__ __forceinline__ void load_tile_at(const GradArgs& a, const TileIdx& ti, int h, TileIn<O, FirstLayer<EnvSpec<KIND>::D>::KS>& t) {
    constexpr int D = EnvSpec<KIND>::D, KS = FirstLayer<D>::KS, RS = RecLayout<D>::RS;
    const int64_t li = tile_gidx(ti) - a.idx_lo;
    t.valid = ti.inb && li >= 0 && li < a.n_local;
    const int64_t idx = t.valid ? li : 0;
    t.act = 0; t.s0 = 0.f; t.s1 = 0.f;
    if (REC) {
        // lane (sample, h) loads quad h of the record; t.raw is exchanged between the half-waves at first use (unpack_tile).  Three-quad records (D > 4): the two
        // half-waves load the two observation quads and every lane the scalar quad
        t.raw = a.rec[RS * idx + h];
        if (RS == 3) t.raw2 = a.rec[RS * idx + 2];
        if (HEAD == HEAD_VALUE && a.has_clip_vf) t.s1 = a.val_old[idx];
        return;
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) { const int d = 2 * s + h; t.xk[s] = d < D ? a.obs[idx * D + d] : 0.f; }
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
template <int KIND, int O, int HEAD, bool REC>
__device__ __forceinline__ void load_tile(const GradArgs& a, int64_t tile, int64_t ntiles, int c, int h, TileIn<O, FirstLayer<EnvSpec<KIND>::D>::KS>& t) {
    load_tile_at<KIND, O, HEAD, REC>(a, tile_index(a, tile, ntiles, c), h, t);
}

// LDS-DMA (global_load_lds_dwordx4 / _dword): lane l's 16 / 4 bytes land at the wave-uniform LDS base + l x size; no destination registers, counted by vmcnt
__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void glds4(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}
// the minibatch records of ONE sample tile straight into the workgroup's LDS, by one wave: lane (c, h) fetches quad h of sample c's record -> rec_t[h][c] (a three-quad
// record's scalar quad with a second instruction -> rec_t[2][c]), the old value -> vo_t[c], and writes the validity word (ppo_grad_wide_split_kernel; ppo_grad_pair_kernel with -DDRIL_PAIR_LDS_REC).  Until round 5 every one of the H/32 waves
// gathered the same records into its own registers (8 - 12 of them, live across the whole pass) in front of a streaming chain, whose first fragment wait then sat out the gather
template <int KIND, int HEAD>
__device__ __forceinline__ void request_records_lds(const GradArgs& a, const TileIdx& ti, int lane, float* rec_t, float* vo_t, int* val_t) {
    constexpr int RS = RecLayout<EnvSpec<KIND>::D>::RS;
    const int64_t li = tile_gidx(ti) - a.idx_lo;
    const bool valid = ti.inb && li >= 0 && li < a.n_local;
    const int64_t idx = valid ? li : 0;
    glds16(a.rec + RS * idx + (lane >> 5), rec_t);
    if (RS == 3) glds16(a.rec + RS * idx + 2, rec_t + 64 * 4);
    if (HEAD == HEAD_VALUE && a.has_clip_vf) glds4(a.val_old + idx, vo_t);
    val_t[lane] = valid ? 1 : 0;
}
// exchange the two record halves between the half-waves: v_permlane32_swap(a, b) swaps a[32..63] with b[0..31], so with
// a = b = v the results are {lo-half value in every lane, hi-half value in every lane}
template <int KIND, int O, int HEAD, bool REC>
__device__ __forceinline__ void unpack_tile(const GradArgs& a, int h, TileIn<O, FirstLayer<EnvSpec<KIND>::D>::KS>& t) {
    if (!REC) return;
    constexpr int KS = FirstLayer<EnvSpec<KIND>::D>::KS;
    float lo[4], hi[4];
    const float v[4] = {t.raw.x, t.raw.y, t.raw.z, t.raw.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned u = __float_as_uint(v[i]);
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        lo[i] = __uint_as_float(r[0]); hi[i] = __uint_as_float(r[1]);
    }
    t.xk[0] = h ? lo[1] : lo[0]; t.xk[1] = h ? lo[3] : lo[2];      // xk[s] = obs[2s + h]
    float sc[4] = {hi[0], hi[1], hi[2], hi[3]};                    // {action bits, adv, logp_old, ret}
    if (KS == 4) {                                                 // three-quad record: hi = obs4..7, the scalars came with raw2
        t.xk[KS - 2] = h ? hi[1] : hi[0]; t.xk[KS - 1] = h ? hi[3] : hi[2];
        sc[0] = t.raw2.x; sc[1] = t.raw2.y; sc[2] = t.raw2.z; sc[3] = t.raw2.w;
    }
    if (HEAD == HEAD_VALUE) t.s0 = sc[3];
    else {
        t.s0 = sc[1]; t.s1 = sc[2];
        if (HEAD == HEAD_CATEGORICAL) t.act = __float_as_int(sc[0]) - a.action_start; else t.xa[0] = sc[0];
    }
}

// (alg::PPO)(...) loss terms and dLoss/d(net output) for one sample per lane (ppo.jl:377-404); `tally` selects the lanes that
// add to the statistics / log_std sums (each sample is replicated in the two half-waves, and in every wave of a wide workgroup)
template <int O, int HEAD, int KS>
__device__ __forceinline__ void loss_head(const GradArgs& a, const TileIn<O, KS>& cur, const float (&out)[O], bool valid, bool tally, const float* ls,
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

}  // namespace dril
