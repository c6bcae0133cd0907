// dril_update_small.hip — ppo_update_small_kernel: the epoch x minibatch loop of train! (ppo.jl:205-239) for the reference's default PPO() (batch_size = 64) as ONE
// persistent workgroup.  At B = 64 an optimiser step is 3.4 MFLOP: the two-launch small path (ppo_grad_kernel + ppo_finish_small_kernel) spends its 40 us on
// staging both weight images, writing / re-reading slabs through L2 and two dependent launches, 1 280 times per iteration of the README quick-start.  Here the
// whole sequence of optimiser steps runs inside one launch: the parameters and both Adam moments live in REGISTERS of the thread that owns them, the weights'
// LDS images are rebuilt from LDS after every step, gradients are exchanged through LDS only, and the per-step statistics row is the only global store.
//
//   workgroup = 8 waves: waves 0-3 the actor (two PAIRS of waves, pair p owns samples 32p .. 32p+31 of the minibatch), waves 4-7 the critic; a pair runs the
//   tile code of ppo_grad_pair_kernel (wave w = m-tile w of every layer; bf16 matrix cores, fp32-equivalent 3-piece operand split, dril_device.h).
//   per step:  gather (prefetched one step ahead)  ->  L1, h1 pieces | B | L2, output partials | B | loss head, dz2 pieces | B | dh1, dW1, dW2 | B |
//              the pair's gradient slab into its own (now dead) image area | B | all 512 threads: g = slab0 + slab1, |g|^2 -> norm, KL / NaN flags, Adam on the
//              registers, new parameter back into slab0 | B | weight images rebuilt from slab0 | B.
//   Same order of operations per optimiser step as ppo.jl:207-239: gradient -> NaN / Inf check -> norm -> clip -> KL check (skip this apply, stop) -> Adam.
#include <utility>

#include "dril_grad_common.h"
#include "dril_split_pieces.h"

namespace dril {

template <int D, int O> struct SmallNet {                     // one net's weights in LDS (floats), rebuilt from the parameters every optimiser step
    static constexpr int H = 64, DP = 4, OP = (O + 3) / 4 * 4;
    static constexpr int W1T = 0, B1 = W1T + DP * H, B2 = B1 + H, W3S = B2 + H, B3 = W3S + O * H, SMALL_END = (B3 + OP + 3) / 4 * 4;
    static constexpr int LS = SMALL_END, WIMG = LS + 4, END = WIMG + 3 * 2048;   // log_std copy (actor, continuous heads); three pieces x [64 out][64 in] bf16
};
template <int D, int OMAX> struct SmallPair {                 // one pair's area (floats)
    static constexpr int P1 = 0, P2 = P1 + 3 * 1024, PO = P2 + 3 * 1024;                 // two 12 KB piece images, [2 waves][O][32] output partial sums
    // small gradients of the pair's tile in parameter order {W1 (o + 64 k) | b1 | b2 | W3 (o + O k) | b3 | log_std | 8 statistics}; never overlaid by the images.
    // After the optimiser phase the same words of pair 0 hold the NEW parameters, which is where the staging reads them.
    static constexpr int G_W1 = PO + 2 * OMAX * 32, G_B1 = G_W1 + 64 * D, G_B2 = G_B1 + 64, G_W3 = G_B2 + 64, G_B3 = G_W3 + 64 * OMAX, G_LS = G_B3 + 4, G_ST = G_LS + 4;
    static constexpr int SIZE = (G_ST + 8 + 3) / 4 * 4;
    // dW2 overlays the two piece images between the barrier after the last image read and the next step's first image store: rows padded to 65 floats so that the
    // column-wise stores of the accumulator layout and the row-wise reads of the optimiser phase are both conflict-free
    static constexpr int S_W2 = 0;
    static_assert(S_W2 + 64 * 65 <= PO, "the dW2 overlay must stay inside the two piece images");
};
// A operand of dh1 = W2' (transposed reads of the weight image): as load_frag_W_T of dril_grad_pair.hip (tmk = tbase ^ (64 mk), tmk16 = tmk ^ 16)
__device__ __forceinline__ bf16x8 small_frag_W_T(const char* wimg, int tmk, int tmk16, int piece, int mi, int s) {
    const int off = (32 * mi + 16 * s) * 128 + piece * 8192;
    return frag8(lds_read_tr16(wimg, tmk + off), lds_read_tr16(wimg, tmk16 + off + 4 * 128));
}

// one 32-sample tile of one net on a pair of waves: forward, loss head, reverse pass; the pair's gradient goes into its slab overlay.  Barriers are workgroup-wide
// (all eight waves execute the same sequence).
template <int KIND, int O, int HEAD, int OMAX>
#ifdef DRIL_STAMPS
#define SMALL_STAMP_PARAMS , unsigned long long (&stamp_acc)[16], unsigned long long& stamp_prev
#define SMALL_STAMP_ARGS , stamp_acc, stamp_prev
#else
#define SMALL_STAMP_PARAMS
#define SMALL_STAMP_ARGS
#endif
__device__ __forceinline__ void small_tile(const GradArgs& ga, float* wl, float* pb, TileIn<O>& cur, const float* mom, int normalize_adv, const float* ls, int lane, int w SMALL_STAMP_PARAMS) {
    constexpr int D = EnvSpec<KIND>::D, H = 64, MT = 2;
    constexpr float kInvTanhScale = 1.0f / kTanhScale;
    using L = SmallNet<D, O>; using S = SmallPair<D, OMAX>;
    const int c = lane & 31, h = lane >> 5;
    char* Wimg = reinterpret_cast<char*>(wl + L::WIMG);
    char* P1 = reinterpret_cast<char*>(pb + S::P1); char* P2 = reinterpret_cast<char*>(pb + S::P2); float* PO = pb + S::PO;
    const int tbase = wide_tr_base<64>(lane);
    unpack_tile<KIND, O, HEAD, true>(ga, h, cur);
    const bool valid = cur.valid;
    const float xk[2] = {cur.xk[0], cur.xk[1]};
    // ---- h1 tile w; its pieces into the pair's image ----
    f32x16 h1k;
    {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(wl + L::B1 + 32 * w + 8 * q + 4 * h);
            h1k[4 * q + 0] = b[0]; h1k[4 * q + 1] = b[1]; h1k[4 * q + 2] = b[2]; h1k[4 * q + 3] = b[3];
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) h1k = mfma32(wl[L::W1T + (2 * s + h) * H + 32 * w + c], xk[s], h1k);
        tanh16(h1k);
        store_tile_pieces<64>(P1, w, h1k, opaque(lane));
    }
    STAMP(1);
    lds_barrier();                                                                    // B1: the pair's h1 image complete
    STAMP(2);
    // ---- h2 tile w ----
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
        for (int ks = 0; ks < 4; ++ks) {
            const int ch = ((2 * ks + hh) ^ gsw) << 4;
            bf16x8 A[3], B[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) { A[p] = *reinterpret_cast<const bf16x8*>(arow + p * 8192 + ch); B[p] = *reinterpret_cast<const bf16x8*>(brow + p * 4096 + ch); }
            h2w = mfma_split6(A[0], A[1], A[2], B[0], B[1], B[2], h2w);
        }
        tanh16(h2w);
    }
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
    STAMP(3);
    lds_barrier();                                                                    // B2: both partial sums
    STAMP(4);
#pragma unroll
    for (int o = 0; o < O; ++o) out[o] = (wl[L::B3 + o] + PO[o * 32 + c]) + PO[(O + o) * 32 + c];   // fixed order: both waves get the same bits
    float st[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, dlsp[O];
#pragma unroll
    for (int o = 0; o < O; ++o) dlsp[o] = 0.f;
    const float adv_mean = (HEAD != HEAD_VALUE && normalize_adv) ? mom[0] : 0.f, adv_inv = (HEAD != HEAD_VALUE && normalize_adv) ? mom[1] : 1.f;   // written by the critic's first wave before B1
    loss_head<O, HEAD>(ga, cur, out, valid, h == 0 && w == 0, ls, adv_mean, adv_inv, dz, st, dlsp);
    float* gq = pb;                                                                   // the pair's small-gradient words
    const int u4 = 32 * w + 4 * h + (c & 3);                                          // unit of register 4 i + (c & 3): u4 + 8 i (rowfn)
#pragma unroll
    for (int o = 0; o < O; ++o) {                                                     // dW3[o][unit] = sum over samples (lanes) of dz[o] h2[unit]
        float r4[4]; half_reduce16(dz[o] * h2w, lane, r4);
        if (c < 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) gq[S::G_W3 + o + (u4 + 8 * i) * O] = r4[i];
        }
    }
    {   // b3 / log_std gradients and the five statistics: one register each of a sixteen-register reduction (lane c & 3 = k receives scalar 4 i + k)
        f32x16 sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.f;
        const bool tal = h == 0 && w == 0;                                            // each sample once: the first half-wave of the pair's first wave
#pragma unroll
        for (int o = 0; o < O; ++o) { sc[o] = tal ? dz[o] : 0.f; if (HEAD == HEAD_GAUSSIAN) sc[4 + o] = dlsp[o]; }
#pragma unroll
        for (int k = 0; k < 5; ++k) sc[8 + k] = st[k];
        float r4[4]; half_reduce16(sc, lane, r4);
        if (w == 0 && h == 0 && c < 4) {
            if (c < O) gq[S::G_B3 + c] = r4[0];
            if (HEAD == HEAD_GAUSSIAN && c < O) gq[S::G_LS + c] = r4[1];
            gq[S::G_ST + c] = r4[2]; if (c == 0) gq[S::G_ST + 4] = r4[3];
        }
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
    {
        float r4[4]; half_reduce16(h2w, lane, r4);
        if (c < 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) gq[S::G_B2 + u4 + 8 * i] = r4[i];
        }
    }
    store_tile_pieces<64>(P2, w, h2w, opaque(lane));
    STAMP(5);
    lds_barrier();                                                                    // B3: the pair's dz2 image complete
    STAMP(6);
    // ---- dz1 tile w = (W2'[rows of w] dz2) .* (1 - h1^2) ----
    f32x16 g1;
    {
#pragma unroll
        for (int r = 0; r < 16; ++r) g1[r] = 0.f;
        const int lo_ = opaque(lane), cc = lo_ & 31, hh = lo_ >> 5, gsw = wimg_g<64>(cc), tbw = opaque(tbase) ^ (64 * w), tbw16 = tbw ^ 16;
        const char* brow = P2 + cc * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int ch = ((2 * ks + hh) ^ gsw) << 4;
            bf16x8 A[3], B[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) { A[p] = small_frag_W_T(Wimg, tbw, tbw16, p, ks >> 1, ks & 1); B[p] = *reinterpret_cast<const bf16x8*>(brow + p * 4096 + ch); }
            g1 = mfma_split6(A[0], A[1], A[2], B[0], B[1], B[2], g1);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float t2 = h1k[r] * h1k[r]; g1[r] = g1[r] * fmaf(-t2, kInvTanhScale, kInvTanhScale); }
    }
    {   // db1 and dW1: per-lane products summed over the samples
        const float xo0 = __shfl_xor(xk[0], 32), xo1 = __shfl_xor(xk[1], 32);        // the other half holds x[2s + 1 - h]
        const float x4[4] = {h ? xo0 : xk[0], h ? xk[0] : xo0, h ? xo1 : xk[1], h ? xk[1] : xo1};
        float r4[4]; half_reduce16(g1, lane, r4);
        if (c < 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) gq[S::G_B1 + u4 + 8 * i] = r4[i];
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            half_reduce16(x4[d] * g1, lane, r4);
            if (c < 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) gq[S::G_W1 + u4 + 8 * i + d * H] = r4[i];
            }
        }
    }
    // ---- dW2[rows of w][:] = dz2 h1' (both operands as transposed fragments of the pair's images) ----
    f32x16 dW2[MT];
    {
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) dW2[j][r] = 0.f;
        const int tb = opaque(tbase), tbw = tb ^ (64 * w), tbw16 = tbw ^ 16;
        bf16x8 Az[2][3];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int p = 0; p < 3; ++p) Az[s][p] = load_frag_wide_T<64>(P2, tbw, tbw16, p, s);
#pragma unroll
        for (int mj = 0; mj < MT; ++mj) {
            bf16x8 Bh[2][3];
            const int tbj = tb ^ (64 * mj), tbj16 = tbj ^ 16;
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int p = 0; p < 3; ++p) Bh[s][p] = load_frag_wide_T<64>(P1, tbj, tbj16, p, s);
#pragma unroll
            for (int s = 0; s < 2; ++s) dW2[mj] = mfma_split6(Az[s][0], Az[s][1], Az[s][2], Bh[s][0], Bh[s][1], Bh[s][2], dW2[mj]);
        }
    }
    STAMP(7);
    lds_barrier();                                                                    // B4: nobody reads the pair's images any more: dW2 goes over them
    STAMP(8);
#pragma unroll
    for (int mj = 0; mj < MT; ++mj)
#pragma unroll
        for (int r = 0; r < 16; ++r) pb[S::S_W2 + (32 * w + rowfn(r, h)) * 65 + 32 * mj + c] = dW2[mj][r];
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
    return v;
}
// sum over the 64 lanes on the VALU (DPP) + two LDS-crossbar permutes; every lane gets the total
__device__ __forceinline__ float wave_sum_f32(float v) {
    v += dpp_mov<0xB1>(0.f, v); v += dpp_mov<0x4E>(0.f, v);
    float t = dpp_mov<0x104, 0x5>(0.f, v); t = dpp_mov<0x114, 0xa>(t, v); v += t;
    v += dpp_mov<0x128>(0.f, v);
    v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
    return v;
}

// small parameters of one net (everything but W2) in the order {W1 | b1 | b2 | W3 | b3}: flat parameter index, gradient word in the pair area, staged word in the
// net's LDS block and the scale of the staged copy (the forward images are pre-scaled by kTanhScale)
template <int D, int O, int OMAX>
__device__ __forceinline__ void small_param_map(int r, int& flat, int& goff, int& dst, float& scale) {
    using L = SmallNet<D, O>; using S = SmallPair<D, OMAX>;
    constexpr int H = 64, n_w1 = H * D, n_b1 = n_w1 + H, n_w2 = n_b1 + H * H, n_b2 = n_w2 + H, n_w3 = n_b2 + O * H;
    if (r < n_w1) { flat = r; goff = S::G_W1 + r; dst = L::W1T + r; scale = kTanhScale; return; }
    r -= n_w1;
    if (r < H) { flat = n_w1 + r; goff = S::G_B1 + r; dst = L::B1 + r; scale = kTanhScale; return; }
    r -= H;
    if (r < H) { flat = n_w2 + r; goff = S::G_B2 + r; dst = L::B2 + r; scale = kTanhScale; return; }
    r -= H;
    if (r < O * H) { flat = n_b2 + r; goff = S::G_W3 + r; dst = L::W3S + (r % O) * H + r / O; scale = 1.0f; return; }   // W3[o + O k] -> row o of the staged copy
    r -= O * H;
    flat = n_w3 + r; goff = S::G_B3 + r; dst = L::B3 + r; scale = 1.0f;
}

template <int KIND>
__global__ __launch_bounds__(512, 1) void ppo_update_small_kernel(SmallUpdateArgs a) {
    constexpr int D = EnvSpec<KIND>::D, A = EnvSpec<KIND>::A, H = 64, OMAX = A;
    constexpr bool DISC = EnvSpec<KIND>::discrete;
    constexpr int AHEAD = DISC ? HEAD_CATEGORICAL : HEAD_GAUSSIAN;
    using LA = SmallNet<D, A>; using LC = SmallNet<D, 1>; using S = SmallPair<D, OMAX>;
    constexpr int NSA = H * (D + 2 + A) + A, NSC = H * (D + 2 + 1) + 1, NLS = DISC ? 0 : A;    // small parameters of the actor / the critic, log_std
    static_assert(NSA + NSC + NLS <= 1024, "two small parameters per thread");
    constexpr int WOFF = H * D + H;                                                    // W2 inside a net's parameters
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float shf[16];
    __shared__ float stf[8];
    __shared__ float mom[2];
    if (*a.stop_flag) return;
    float* wl_a = smem; float* wl_c = smem + LA::END;
    float* pairs = smem + LA::END + LC::END;                                          // [net][pair] areas
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int net = wave >> 2, pr = (wave >> 1) & 1, w = wave & 1;
    const int c = lane & 31, h = lane >> 5;
    float* pb = pairs + (net * 2 + pr) * S::SIZE;
    float* sl_a0 = pairs, *sl_a1 = pairs + S::SIZE, *sl_c0 = pairs + 2 * S::SIZE, *sl_c1 = pairs + 3 * S::SIZE;

    // ---- ownership: the thread keeps, for the whole launch and in registers, the parameters and Adam moments of
    //        W2 pairs (o, 2 kp), (o, 2 kp + 1) with o = tid & 63, kp = (tid >> 6) + 8 j, j = 0..3, of BOTH nets (16 + 16 parameters: all 8 192 of them), and
    //        two small parameters (index tid and tid + 512 of {actor small | critic small | log_std}).
    //      After Adam the owner writes the staged form itself: the three bf16 pieces of a W2 pair into the net's weight image, a small parameter (scaled) into its
    //      staged word — no f32 copy of the parameters goes through LDS and no separate staging pass exists.
    const int wo = tid & 63, wk = tid >> 6;
    float wp[2][4][2], wm[2][4][2], wv[2][4][2];                                       // [net][j][element of the pair]
    float sp[2], sm[2], sv[2], sscale[2]; int sflat[2], sgoff[2], sdst[2];             // small parameters (sflat < 0: none)
    auto publish_pair = [&](int n, int j) {
        char* Wimg = reinterpret_cast<char*>((n ? wl_c + LC::WIMG : wl_a + LA::WIMG));
        const int kp = wk + 8 * j;
        unsigned hi, mid, lo;
        split3_pair(kTanhScale * wp[n][j][0], kTanhScale * wp[n][j][1], hi, mid, lo);
        const int byte = wo * 128 + ((((kp >> 2) ^ wimg_g<64>(wo)) & 7) << 4) + ((kp & 3) << 2);
        *reinterpret_cast<unsigned*>(Wimg + byte) = hi; *reinterpret_cast<unsigned*>(Wimg + 8192 + byte) = mid; *reinterpret_cast<unsigned*>(Wimg + 16384 + byte) = lo;
    };
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int p = (n ? a.Pa : 0) + WOFF + wo + 64 * (2 * (wk + 8 * j) + e);
                wp[n][j][e] = a.params[p]; wm[n][j][e] = a.adam_m[p]; wv[n][j][e] = a.adam_v[p];
            }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int si = tid + 512 * q;
        sflat[q] = -1; sgoff[q] = 0; sdst[q] = 0; sscale[q] = 1.0f; sp[q] = 0.f; sm[q] = 0.f; sv[q] = 0.f;
        if (si < NSA) { int f; small_param_map<D, A, OMAX>(si, f, sgoff[q], sdst[q], sscale[q]); sflat[q] = f; }
        else if (si < NSA + NSC) { int f; small_param_map<D, 1, OMAX>(si - NSA, f, sgoff[q], sdst[q], sscale[q]); sflat[q] = a.Pa + f; sgoff[q] += 2 * S::SIZE; sdst[q] += LA::END; }
        else if (si < NSA + NSC + NLS) { const int i = si - NSA - NSC; sflat[q] = a.Pa + a.Pc + i; sgoff[q] = S::G_LS + i; sdst[q] = LA::LS + i; }
        if (sflat[q] >= 0) { sp[q] = a.params[sflat[q]]; sm[q] = a.adam_m[sflat[q]]; sv[q] = a.adam_v[sflat[q]]; }
    }
    for (int i = tid; i < LA::DP * H; i += 512) { wl_a[LA::W1T + i] = 0.f; wl_c[LC::W1T + i] = 0.f; }   // rows k >= D of the first-layer images stay zero
    for (int i = tid; i < LA::OP; i += 512) wl_a[LA::B3 + i] = 0.f;
    for (int i = tid; i < LC::OP; i += 512) wl_c[LC::B3 + i] = 0.f;
    lds_barrier();
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int j = 0; j < 4; ++j) publish_pair(n, j);
#pragma unroll
    for (int q = 0; q < 2; ++q) if (sflat[q] >= 0) smem[sdst[q]] = sscale[q] * sp[q];
    const float* bt_in = a.bt + 2 * (a.step_parity & 1);
    float bt1 = bt_in[0], bt2 = bt_in[1];

    GradArgs ga{};                                                                    // what unpack_tile / loss_head read
    ga.clip_range = a.clip_range; ga.ent_coef = a.ent_coef; ga.vf_coef = a.vf_coef; ga.clip_range_vf = a.clip_range_vf; ga.has_clip_vf = a.has_clip_vf;
    ga.normalize_adv = a.normalize_adv; ga.action_start = a.action_start;

    // gather of step s: this lane's half record of sample 32 pr + c, and (critic waves, which have the lighter tile) the advantage of sample `lane` for the
    // minibatch moments
    auto sample_index = [&](int ep, int64_t pos) -> int64_t {
        return a.perm ? a.perm[(int64_t)ep * a.N + pos] : perm_index(pos, a.N, a.keys[ep], a.perm_bits);
    };
    float4 raw_n = make_float4(0.f, 0.f, 0.f, 0.f); float vold_n = 0.f, adv_n = 0.f; bool valid_n = false;
    auto gather = [&](int s) {
        const int ep = s / a.nb, k = s - ep * a.nb;
        const int64_t pos0 = (int64_t)k * a.B, count = (pos0 + a.B <= a.N) ? a.B : a.N - pos0;
        const int i = 32 * pr + c;
        valid_n = i < count;
        const int64_t idx = sample_index(ep, pos0 + (valid_n ? i : 0));
        raw_n = a.rec[2 * idx + h];
        vold_n = (net == 1 && a.has_clip_vf) ? a.val_old[idx] : 0.f;
        adv_n = 0.f;
        if (net == 1 && a.normalize_adv && lane < count) adv_n = a.rec[2 * sample_index(ep, pos0 + lane) + 1].y;
    };
    const int s_end = a.step0 + a.nsteps;
    if (a.step0 < s_end) gather(a.step0);
#ifdef DRIL_STAMPS
    unsigned long long stamp_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev) :: "memory");
#endif
    for (int s = a.step0; s < s_end; ++s) {
        const int ep = s / a.nb, kb = s - ep * a.nb;
        const int64_t pos0 = (int64_t)kb * a.B, count = (pos0 + a.B <= a.N) ? a.B : a.N - pos0;
        ga.invB = 1.0f / (float)count;
        // ---- this step's inputs out of the prefetch registers; the next step's gathers go out now and land under this step's arithmetic ----
        const float4 raw = raw_n; const float vold = vold_n, advl = adv_n; const bool valid = valid_n;
        if (s + 1 < s_end) gather(s + 1);
        if (net == 1 && a.normalize_adv) {                                            // normalize!(advantages) per minibatch, ppo.jl:350-356 (corrected std + 1e-8)
            const double sm_ = wave_sum_f64((double)advl), sq = wave_sum_f64((double)advl * (double)advl), n = (double)count;
            const double mean = sm_ / n;
            double var = (sq - sm_ * mean) / (n - 1.0);
            if (var < 0) var = 0;
            if (tid == 256) { mom[0] = (float)mean; mom[1] = 1.0f / ((float)sqrt(var) + 1.0e-8f); }   // read by the actor after the second barrier of the tile
        }
        float lsr[kLsMax];
#pragma unroll
        for (int o = 0; o < kLsMax; ++o) lsr[o] = 0.f;
        lds_barrier();                                                                // (weight images and staged parameters of the previous step complete; the dW2 overlay becomes images again)
        if (!DISC) {
#pragma unroll
            for (int o = 0; o < A; ++o) lsr[o] = wl_a[LA::LS + o];
        }
        STAMP(0);
        if (net == 0) {
            TileIn<A> cur; cur.raw = raw; cur.valid = valid; cur.s0 = 0.f; cur.s1 = 0.f; cur.act = 0;
            small_tile<KIND, A, AHEAD, OMAX>(ga, wl_a, pb, cur, mom, a.normalize_adv, lsr, lane, w SMALL_STAMP_ARGS);
        } else {
            TileIn<1> cur; cur.raw = raw; cur.valid = valid; cur.s0 = 0.f; cur.s1 = vold; cur.act = 0;
            small_tile<KIND, 1, HEAD_VALUE, OMAX>(ga, wl_c, pb, cur, mom, 0, lsr, lane, w SMALL_STAMP_ARGS);
        }
        STAMP(9);
        lds_barrier();                                                                // B5: all four pairs' gradients complete
        STAMP(10);
        // ---- optimiser phase ----
        float gw[2][4][2], gs[2]; double ss = 0;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int off = 2 * n * S::SIZE + S::S_W2 + wo * 65 + 2 * (wk + 8 * j) + e;
                    const float g = pairs[off] + pairs[off + S::SIZE];               // pair 0 + pair 1 of the net, fixed order
                    gw[n][j][e] = g; ss += (double)g * (double)g;
                }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float g = sflat[q] >= 0 ? pairs[sgoff[q]] + pairs[sgoff[q] + S::SIZE] : 0.f;
            gs[q] = g; ss += (double)g * (double)g;
        }
        if (tid >= 448 && tid < 456) {                                                // the statistics sums (the last critic wave has the least to do)
            const int k = tid - 448;
            float t = 0.f;
            if (k < 5) t = sl_a0[S::G_ST + k] + sl_a1[S::G_ST + k];
            else if (k == 5) t = sl_c0[S::G_ST] + sl_c1[S::G_ST];
            else if (k == 6) t = (float)count;
            stf[k] = t;
        }
        {   // |g|^2: per thread in f64, per wave on the VALU, the eight waves in index order
            const float wsum = wave_sum_f32((float)ss);
            if (lane == 0) shf[wave] = wsum;
        }
        lds_barrier();
        double tot = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) tot += (double)shf[k];
        const float norm = sqrtf((float)tot);
        STAMP(11);
        const float n = stf[6], kl = stf[3] / n;
        const bool bad = !(norm == norm) || isinf(norm);                              // NaN / Inf anywhere poisons the norm (ppo.jl:213-214)
        const bool kl_stop = a.has_target_kl && kl > 1.5f * a.target_kl;              // ppo.jl:235-238: skip this apply, stop
        if (tid == 448) {
            float* o = a.step_stats + (size_t)s * 16;
            const float pl = stf[0] / n, ent = stf[1] / n, vl = stf[5] / n;
            o[0] = pl; o[1] = vl; o[2] = -ent; o[3] = stf[2] / n; o[4] = kl; o[5] = ent; o[6] = stf[4] / n;
            o[7] = pl + a.ent_coef * (-ent) + a.vf_coef * vl;                          // loss, ppo.jl:386
            o[8] = norm; o[9] = (bad || kl_stop) ? 0.f : 1.f; o[10] = bad ? 1.f : 0.f; o[11] = kl_stop ? 1.f : 0.f;
            if (a.norm_out) *a.norm_out = norm;
            if (bad) { *a.nan_flag = 1; *a.stop_flag = 1; }
            if (kl_stop) *a.stop_flag = 1;
        }
        if (bad || kl_stop) break;                                                    // uniform: every thread computed the same norm / kl
        const float scale = (a.has_max_grad_norm && norm > a.max_grad_norm) ? a.max_grad_norm / norm : 1.0f;   // optimization_utils.jl:98-107
        // bias corrections once per step (every thread the same value); per parameter one hardware reciprocal and one hardware square root (1 ulp each: the update is
        // lr x O(1), so their error is ~1e-11 absolute — far below one ulp of a parameter) instead of three IEEE divisions and an IEEE square root (~40 instructions)
        const float ic1 = 1.0f / (1.0f - bt1), ic2 = 1.0f / (1.0f - bt2), omb1 = 1.0f - a.beta1, omb2 = 1.0f - a.beta2;
        auto adam = [&](float g, float& p, float& m, float& v) {                       // Optimisers.Adam, eps = 1e-5 (ppo.jl:64-66)
            g = g * scale;
            m = a.beta1 * m + omb1 * g; v = a.beta2 * v + omb2 * g * g;
            p = p - (m * ic1) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v * ic2) + a.eps) * a.lr;
        };
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                adam(gw[n][j][0], wp[n][j][0], wm[n][j][0], wv[n][j][0]); adam(gw[n][j][1], wp[n][j][1], wm[n][j][1], wv[n][j][1]);
                publish_pair(n, j);
            }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            adam(gs[q], sp[q], sm[q], sv[q]);
            if (sflat[q] >= 0) smem[sdst[q]] = sscale[q] * sp[q];
        }
        bt1 *= a.beta1; bt2 *= a.beta2;
        STAMP(12);
        // (the barrier at the top of the next step separates these image stores from the next reads of the weight images and from the next image stores)
    }
#ifdef DRIL_STAMPS
    if (lane == 0 && a.dbg) { for (int k = 0; k < 16; ++k) a.dbg[wave * 16 + k] = stamp_acc[k]; }
#endif
    // ---- state back to global memory ----
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int p = (n ? a.Pa : 0) + WOFF + wo + 64 * (2 * (wk + 8 * j) + e);
                a.params[p] = wp[n][j][e]; a.adam_m[p] = wm[n][j][e]; a.adam_v[p] = wv[n][j][e];
            }
#pragma unroll
    for (int q = 0; q < 2; ++q) if (sflat[q] >= 0) { a.params[sflat[q]] = sp[q]; a.adam_m[sflat[q]] = sm[q]; a.adam_v[sflat[q]] = sv[q]; }
    if (tid == 0) { a.bt[0] = bt1; a.bt[1] = bt2; a.bt[2] = bt1; a.bt[3] = bt2; }     // both ping-pong slots: the host's step parity no longer matters
}

template <int KIND> static size_t update_small_lds_bytes() {
    constexpr int D = EnvSpec<KIND>::D, A = EnvSpec<KIND>::A;
    return sizeof(float) * (SmallNet<D, A>::END + SmallNet<D, 1>::END + 4 * SmallPair<D, A>::SIZE);
}

hipError_t launch_ppo_update_small(int kind, const SmallUpdateArgs& a, hipStream_t s) {
#define CALLU(K) { const size_t lds = update_small_lds_bytes<K>(); static bool attr_set = false; \
        if (!attr_set) { hipError_t e = hipFuncSetAttribute((const void*)ppo_update_small_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (e != hipSuccess) return e; attr_set = true; } \
        ppo_update_small_kernel<K><<<1, 512, lds, s>>>(a); }
    if (kind == 0) CALLU(0) else if (kind == 3) CALLU(3) else if (kind == 4) CALLU(4) else CALLU(1)
#undef CALLU
    return hipGetLastError();
}

}  // namespace dril
