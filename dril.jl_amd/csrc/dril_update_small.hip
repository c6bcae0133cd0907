// dril_update_small.hip — ppo_update_small_kernel: the epoch x minibatch loop of train! (ppo.jl:205-239) for the reference's default PPO() (batch_size = 64) as TWO
// persistent workgroups, one per net, on two CUs of one XCD.  At B = 64 an optimiser step is 3.4 MFLOP: the two-launch small path (ppo_grad_kernel +
// ppo_finish_small_kernel) spends its 40 us on staging both weight images, writing / re-reading slabs through L2 and two dependent launches, 1 280 times per
// iteration of the README quick-start.  Here the whole sequence of optimiser steps runs inside one launch: a net's parameters and both Adam moments live in REGISTERS
// of the thread that owns them, its weight images in LDS are rewritten by the owners after every step, gradients are exchanged through LDS only; what crosses between
// the two workgroups is one 16-float message per step (partial |g|^2, the statistics sums, the next minibatch's advantage moments) through L2.
//
//   workgroup = 4 waves = two PAIRS of waves, pair p owns samples 32p .. 32p+31 of the minibatch; a pair runs the tile code of ppo_grad_pair_kernel (wave w = m-tile w
//   of every layer; bf16 matrix cores, fp32-equivalent 3-piece operand split, dril_device.h).  One wave per SIMD (round 3, first form: both nets in ONE workgroup of
//   eight waves, 12 us per step — two waves per SIMD on one CU; two CUs: see docs/kernels/ppo_update_small_kernel.md).
//   per step:  gather (prefetched one step ahead)  ->  L1, h1 pieces | B | L2, output partials | B | loss head, dz2 pieces | B | dh1, dW1, dW2 | B |
//              the pair's gradient slab into its own (now dead) image area | B | all 256 threads: g = slab0 + slab1, partial |g|^2 | B | message out, message in | B |
//              norm, KL / NaN flags, Adam on the registers, new weight images.
//   Same order of operations per optimiser step as ppo.jl:207-239: gradient -> NaN / Inf check -> norm -> clip -> KL check (skip this apply, stop) -> Adam.
//   The launch has 9 workgroups: workgroups are dealt to the XCDs round-robin by id, so ids 0 and 8 share an XCD (one L2); ids 1-7 exit at once.
#include <utility>

#include "dril_grad_common.h"
#include "dril_split_pieces.h"

namespace dril {

// one net's weights in LDS (floats), rewritten from the owners' registers every optimiser step; the weight image starts at a multiple of 512 bytes (XOR addressing)
template <int D, int O> struct SmallNet {
    static constexpr int H = 64, DP = FirstLayer<D>::DP, OP = (O + 3) / 4 * 4;   // observations of 5 .. 8 components (Acrobot): four first-layer k-steps, eight rows of W1'
    static constexpr int W1T = 0, B1 = W1T + DP * H, B2 = B1 + H, W3S = B2 + H, W3B = W3S + O * H, B3 = W3B + O * H, LS = (B3 + OP + 3) / 4 * 4;   // W3S = W3 / kActScale (forward), W3B = W3 / kActScale^2 (dh); LS: log_std copy (actor, continuous heads)
    static constexpr int WIMG = (LS + 4 + 127) / 128 * 128, END = WIMG + 2 * 2048;                                                // two f16 pieces x [64 out][64 in]
};
template <int D, int O> struct SmallPair {                    // one pair's area (floats), a multiple of 512 bytes
    static constexpr int P1 = 0, P2 = P1 + 2 * 1024, PO = 4224;                          // two 8 KB piece images (f16 x 2 pieces; the dW2 overlay needs 64 x 65 floats: a gap follows P2), [2 waves][O][32] output partial sums
    // small gradients of the pair's tile in parameter order {W1 (o + 64 k) | b1 | b2 | W3 (o + O k) | b3 | log_std | 8 statistics}; never overlaid by the images
    static constexpr int G_W1 = PO + 2 * O * 32, G_B1 = G_W1 + 64 * D, G_B2 = G_B1 + 64, G_W3 = G_B2 + 64, G_B3 = G_W3 + 64 * O, G_LS = G_B3 + 4, G_ST = G_LS + 4;
    static constexpr int SIZE = (G_ST + 8 + 127) / 128 * 128;
    // dW2 overlays the two piece images between the barrier after the last image read and the next step's first image store: rows padded to 65 floats so that the
    // column-wise stores of the accumulator layout and the row-wise reads of the optimiser phase are both conflict-free
    static constexpr int S_W2 = 0;
    static_assert(S_W2 + 64 * 65 <= PO, "the dW2 overlay must stay inside the two piece images");
};
// A operand of dh1 = W2' (transposed reads of the weight image): as load_frag_W_T of dril_grad_pair.hip (tmk = tbase ^ (64 mk), tmk16 = tmk ^ 16)
__device__ __forceinline__ f16x8 small_frag_W_T(const char* wimg, int tmk, int tmk16, int piece, int mi, int s) {
    const int off = (32 * mi + 16 * s) * 128 + piece * 8192;
    return __builtin_bit_cast(f16x8, frag8(lds_read_tr16(wimg, tmk + off), lds_read_tr16(wimg, tmk16 + off + 4 * 128)));
}

// one 32-sample tile of one net on a pair of waves: forward, loss head, reverse pass; the pair's gradient goes into its slab overlay.  Barriers are workgroup-wide
// (all four waves execute the same sequence).  LDS addresses of the images in the XOR form of dril_split_pieces.h: pairB = byte offset of the pair's area.
#ifdef DRIL_STAMPS
#define SMALL_STAMP_PARAMS , unsigned long long (&stamp_acc)[16], unsigned long long& stamp_prev
#define SMALL_STAMP_ARGS , stamp_acc, stamp_prev
#else
#define SMALL_STAMP_PARAMS
#define SMALL_STAMP_ARGS
#endif
template <int KIND, int O, int HEAD>
__device__ __forceinline__ void small_tile(const GradArgs& ga, float* smem, float* pb, int pairB, TileIn<O, FirstLayer<EnvSpec<KIND>::D>::KS>& cur, const float* mom, int normalize_adv, const float* ls, int lane, int w, float inv_sg SMALL_STAMP_PARAMS) {   // ga.invB carries the gradient scale SG (dril_device.h: f16 pieces), inv_sg = 1 / SG
    constexpr int D = EnvSpec<KIND>::D, H = 64, MT = 2, KS = FirstLayer<D>::KS;
    constexpr float kInvTanhScale = 1.0f / kTanhScale;
    using L = SmallNet<D, O>; using S = SmallPair<D, O>;
    constexpr int kWimgB = 4 * L::WIMG, kP1B = 4 * S::P1, kP2B = 4 * S::P2;
    const int c = lane & 31, h = lane >> 5;
    float* wl = smem;
    lds_char* lds = (lds_char*)smem;
    char* Wimg = reinterpret_cast<char*>(wl + L::WIMG);
    char* P1 = reinterpret_cast<char*>(pb + S::P1); char* P2 = reinterpret_cast<char*>(pb + S::P2); float* PO = pb + S::PO;
    const int tbase = wide_tr_base<64>(lane);
    const int rowc = c * 128 + ((h ^ wimg_g<64>(c)) << 4);                               // row c, chunk h of the row (row reads step the chunk by 2 ks)
    const int rowP = pairB + rowc, rowW = rowc + 4096 * w;
    const int ownT = pairB + c * 128 + 8 * h + (((4 * w) ^ wimg_g<64>(c)) << 4);         // the lane's own chunk 4 w + g of row c
    const bool writer = (lane & 16) == 0;                                                // half_reduce16_lane: lanes l and l ^ 16 hold the same sum
    const int runit = 32 * w + rowfn(lane & 15, h);                                      // ... of the unit that register (lane & 15) of this half belongs to
    unpack_tile<KIND, O, HEAD, true>(ga, h, cur);
    const bool valid = cur.valid;
    float xk[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) xk[s] = cur.xk[s];
    const float inv_sa = inv_sg * (1.0f / kActScale);                                    // products with an activation operand carry SG kActScale, the others SG
    // ---- h1 tile w; its pieces into the pair's image ----
    f32x16 h1k;
    {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(wl + L::B1 + 32 * w + 8 * q + 4 * h);
            h1k[4 * q + 0] = b[0]; h1k[4 * q + 1] = b[1]; h1k[4 * q + 2] = b[2]; h1k[4 * q + 3] = b[3];
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) h1k = mfma32(wl[L::W1T + (2 * s + h) * H + 32 * w + c], xk[s], h1k);
        tanh16_scaled<false>(h1k, 1.0f);                                              // kActScale h1
        pair_store_pieces2<kP1B>(lds, ownT, h1k);
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
        // one wave per SIMD: nobody else covers an LDS round trip, so the fragments of k16 step ks + 1 are requested before the six MFMAs of step ks (pinned)
        f16x8 A[2][2], B[2][2];
        auto fetch = [&](int ks) {                                                    // chunk 2 ks + h of the row
            const int ak = rowW ^ (ks << 5), bk = rowP ^ (ks << 5);
#pragma unroll
            for (int p = 0; p < 2; ++p) { A[ks & 1][p] = pl_read<f16x8>(lds, ak + kWimgB + p * 8192); B[ks & 1][p] = pl_read<f16x8>(lds, bk + kP1B + p * 4096); }
        };
        fetch(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < 3) fetch(ks + 1);
            __builtin_amdgcn_sched_barrier(0);
            h2w = mfma_split3(A[ks & 1][0], A[ks & 1][1], B[ks & 1][0], B[ks & 1][1], h2w);
            __builtin_amdgcn_sched_barrier(0);
        }
        tanh16_scaled<true>(h2w, 1.0f / (kWScale * kActScale));                        // kActScale h2
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
        PO[(w * O + o) * 32 + c] = both_halves_sum(p);                                // both half-waves write the same bits to the same word
    }
    STAMP(3);
    lds_barrier();                                                                    // B2: both partial sums
    STAMP(4);
#pragma unroll
    for (int o = 0; o < O; ++o) out[o] = (wl[L::B3 + o] + PO[o * 32 + c]) + PO[(O + o) * 32 + c];   // fixed order: both waves get the same bits
    float st[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, dlsp[O];
#pragma unroll
    for (int o = 0; o < O; ++o) dlsp[o] = 0.f;
    const float adv_mean = (HEAD != HEAD_VALUE && normalize_adv) ? mom[0] : 0.f, adv_inv = (HEAD != HEAD_VALUE && normalize_adv) ? mom[1] : 1.f;   // this minibatch's moments: arrived with the critic's last message
    loss_head<O, HEAD>(ga, cur, out, valid, h == 0 && w == 0, ls, adv_mean, adv_inv, dz, st, dlsp);
    float* gq = pb;                                                                   // the pair's small-gradient words
#pragma unroll
    for (int o = 0; o < O; ++o) {                                                     // dW3[o][unit] = sum over samples (lanes) of dz[o] h2[unit]
        const float v = half_reduce16_lane(dz[o] * h2w, lane) * inv_sa;               // (SG dz)(kActScale h2)
        if (writer) gq[S::G_W3 + o + runit * O] = v;
    }
    {   // b3 / log_std gradients and the five statistics: one register each of a sixteen-register reduction (lane l receives scalar l & 15)
        f32x16 sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.f;
        const bool tal = h == 0 && w == 0;                                            // each sample once: the first half-wave of the pair's first wave
#pragma unroll
        for (int o = 0; o < O; ++o) { sc[o] = tal ? dz[o] * inv_sg : 0.f; if (HEAD == HEAD_GAUSSIAN) sc[4 + o] = dlsp[o] * inv_sg; }
#pragma unroll
        for (int k = 0; k < 5; ++k) sc[8 + k] = st[k];
        const float v = half_reduce16_lane(sc, lane);
        const int r = lane & 15;
        if (tal && writer) {
            if (r < O) gq[S::G_B3 + r] = v;
            if (HEAD == HEAD_GAUSSIAN && r >= 4 && r < 4 + O) gq[S::G_LS + r - 4] = v;
            if (r >= 8 && r < 13) gq[S::G_ST + r - 8] = v;
        }
    }
    // ---- dz2 tile w (in h2w's registers); db2; its pieces into the pair's image ----
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float dh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < O; ++o) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + L::W3B + o * H + 32 * w + 8 * q + 4 * h);
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) dh[cc] = fmaf(wv[cc], dz[o], dh[cc]);
        }
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) { const float hv = h2w[4 * q + cc]; h2w[4 * q + cc] = dh[cc] * fmaf(-hv, hv, kActScale * kActScale); }   // = SG dz2
    }
    {
        const float v = half_reduce16_lane(h2w, lane) * inv_sg;
        if (writer) gq[S::G_B2 + runit] = v;
    }
    pair_store_pieces2<kP2B>(lds, ownT, h2w);
    STAMP(5);
    lds_barrier();                                                                    // B3: the pair's dz2 image complete
    STAMP(6);
    // ---- dz1 tile w = (W2'[rows of w] dz2) .* (1 - h1^2) ----
    f32x16 g1;
    {
#pragma unroll
        for (int r = 0; r < 16; ++r) g1[r] = 0.f;
        const int tbw = tbase ^ (64 * w), tbw16 = tbw ^ 16;
        f16x8 A[2][2], B[2][2];
        auto fetch = [&](int ks) {
            const int bk = rowP ^ (ks << 5);
#pragma unroll
            for (int p = 0; p < 2; ++p) { A[ks & 1][p] = small_frag_W_T(Wimg, tbw, tbw16, p, ks >> 1, ks & 1); B[ks & 1][p] = pl_read<f16x8>(lds, bk + kP2B + p * 4096); }
        };
        fetch(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < 3) fetch(ks + 1);
            __builtin_amdgcn_sched_barrier(0);
            g1 = mfma_split3(A[ks & 1][0], A[ks & 1][1], B[ks & 1][0], B[ks & 1][1], g1);
            __builtin_amdgcn_sched_barrier(0);
        }
        constexpr float c0 = kInvTanhScale / kWScale, c1 = c0 / (kActScale * kActScale);   // g1 = (kTanhScale kWScale W2' . SG dz2) (1 - h1^2) / (kTanhScale kWScale) = SG dz1
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float t2 = h1k[r] * h1k[r]; g1[r] = g1[r] * fmaf(-t2, c1, c0); }
    }
    {   // db1 and dW1: per-lane products summed over the samples
        float x4[2 * KS];                                                             // xk[s] = x[2 s + h]: the lower half's value is x[2 s], the upper half's x[2 s + 1]
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const unsigned u = __float_as_uint(xk[s]);
            const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            x4[2 * s] = __uint_as_float(r[0]); x4[2 * s + 1] = __uint_as_float(r[1]);
        }
        const float v1 = half_reduce16_lane(g1, lane) * inv_sg;
        if (writer) gq[S::G_B1 + runit] = v1;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float v = half_reduce16_lane(x4[d] * g1, lane) * inv_sg;
            if (writer) gq[S::G_W1 + runit + d * H] = v;
        }
    }
    // ---- dW2[rows of w][:] = dz2 h1' (both operands as transposed fragments of the pair's images) ----
    f32x16 dW2[MT];
    {
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) dW2[j][r] = 0.f;
        const int tb = tbase, tbw = tb ^ (64 * w), tbw16 = tbw ^ 16;
        f16x8 Az[2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int p = 0; p < 2; ++p) Az[s][p] = __builtin_bit_cast(f16x8, load_frag_wide_T<64>(P2, tbw, tbw16, p, s));
#pragma unroll
        for (int mj = 0; mj < MT; ++mj) {
            f16x8 Bh[2][2];
            const int tbj = tb ^ (64 * mj), tbj16 = tbj ^ 16;
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int p = 0; p < 2; ++p) Bh[s][p] = __builtin_bit_cast(f16x8, load_frag_wide_T<64>(P1, tbj, tbj16, p, s));
#pragma unroll
            for (int s = 0; s < 2; ++s) dW2[mj] = mfma_split3(Az[s][0], Az[s][1], Bh[s][0], Bh[s][1], dW2[mj]);   // (SG dz2)(kActScale h1)'
        }
    }
    STAMP(7);
    lds_barrier();                                                                    // B4: nobody reads the pair's images any more: dW2 goes over them
    STAMP(8);
#pragma unroll
    for (int mj = 0; mj < MT; ++mj)
#pragma unroll
        for (int r = 0; r < 16; ++r) pb[S::S_W2 + (32 * w + rowfn(r, h)) * 65 + 32 * mj + c] = dW2[mj][r] * inv_sa;
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
    return v;
}
// sum over the 64 lanes on the VALU (DPP) + lane-permute swaps; every lane gets the total
__device__ __forceinline__ float wave_sum_f32(float v) {
    v += dpp_mov<0xB1>(0.f, v); v += dpp_mov<0x4E>(0.f, v);
    float t = dpp_mov<0x104, 0x5>(0.f, v); t = dpp_mov<0x114, 0xa>(t, v); v += t;
    v += dpp_mov<0x128>(0.f, v);
    { const unsigned q = __float_as_uint(v); const auto r = __builtin_amdgcn_permlane16_swap(q, q, false, false); v = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
    return both_halves_sum(v);
}

// small parameters of one net (everything but W2) in the order {W1 | b1 | b2 | W3 | b3}: index within the net's parameters, gradient word in the pair area, staged
// word in the net's LDS block and the scale of the staged copy (the forward images are pre-scaled by kTanhScale)
template <int D, int O>
__device__ __forceinline__ void small_param_map(int r, int& flat, int& goff, int& dst, float& scale) {
    using L = SmallNet<D, O>; using S = SmallPair<D, O>;
    constexpr int H = 64, n_w1 = H * D, n_b1 = n_w1 + H, n_w2 = n_b1 + H * H, n_b2 = n_w2 + H, n_w3 = n_b2 + O * H;
    if (r < n_w1) { flat = r; goff = S::G_W1 + r; dst = L::W1T + r; scale = kTanhScale; return; }
    r -= n_w1;
    if (r < H) { flat = n_w1 + r; goff = S::G_B1 + r; dst = L::B1 + r; scale = kTanhScale; return; }
    r -= H;
    if (r < H) { flat = n_w2 + r; goff = S::G_B2 + r; dst = L::B2 + r; scale = kTanhScale * kWScale * kActScale; return; }   // b2 starts the SCALED accumulator of L2
    r -= H;
    if (r < O * H) { flat = n_b2 + r; goff = S::G_W3 + r; dst = L::W3S + (r % O) * H + r / O; scale = 1.0f / kActScale; return; }   // W3[o + O k] -> row o of the staged copies (W3S; W3B = the same word + L::W3B - L::W3S, / kActScale again)
    r -= O * H;
    flat = n_w3 + r; goff = S::G_B3 + r; dst = L::B3 + r; scale = 1.0f;
}

// ---- the message between the two workgroups (through L2; agent-scope accesses) ----
// a.xchg: [role][slot = exchange index & 1][16] 64-bit words {sequence number, f32 value}.  A word is valid when its sequence number is that of the exchange: value and
// validity travel in ONE 8-byte store per lane and one 8-byte load per poll — no flag word, no fence, one L2 round trip per direction (a flag + fence protocol
// measured 1.3 us per step here).  Message of exchange k >= 1 (optimiser step step0 + k - 1):   words 0-3 the sender's per-wave partial |g|^2, then
//   actor:  4-8 the five statistics sums of the minibatch (policy loss, entropy, clip fraction, approx KL, ratio)
//   critic: 4 the value-loss sum, 5 / 6 mean and 1 / (std + 1e-8) of the NEXT minibatch's advantages (the critic's waves have the lighter tile and gather them)
// Exchange 0 (before the first step) carries the first minibatch's moments.  Two slots suffice: a role publishes message k + 2 only after it has seen message k + 1 of
// the other role, which that role published after reading message k.  The launcher zeroes the buffer (sequence 0 = nothing yet).
constexpr int kMsgWords = 16;
constexpr unsigned kSpinLimit = 1u << 20;      // ~ a second (a step is 8 us); reached only if the partner workgroup never runs (then: stop flag, nan flag = 2, both workgroups leave)
// wave 0 of a workgroup: publish shx[0..15], wait for the partner's message k, fetch it into xin[0..15]; false = gave up waiting
__device__ __forceinline__ bool small_exchange(unsigned long long* xchg, int role, unsigned k, const float* shx, float* xin, int lane) {
    unsigned long long* mine = xchg + (role * 2 + (k & 1)) * kMsgWords;
    unsigned long long* theirs = xchg + ((1 - role) * 2 + (k & 1)) * kMsgWords;
    const unsigned long long seq = (unsigned long long)(k + 1) << 32;
    if (lane < kMsgWords) __hip_atomic_store(mine + lane, seq | (unsigned long long)__float_as_uint(shx[lane]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool ok = true; unsigned spins = 0; unsigned long long v = seq;
    for (;;) {
        if (lane < kMsgWords) v = __hip_atomic_load(theirs + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all((v >> 32) == (seq >> 32))) break;                                   // wave-uniform
        if (++spins > kSpinLimit) { ok = false; break; }
        __builtin_amdgcn_s_sleep(1);
    }
    if (lane < kMsgWords) xin[lane] = ok ? __uint_as_float((unsigned)v) : 0.f;
    return ok;
}

// one workgroup = one net: ROLE 0 the actor (O = A outputs, Categorical / DiagGaussian head), ROLE 1 the critic
template <int KIND, int O, int HEAD, int ROLE>
__device__ __forceinline__ void small_net_loop(const SmallUpdateArgs& a, float* smem, float* shx, float* xin, float* mom) {
    constexpr int D = EnvSpec<KIND>::D, H = 64;
    using L = SmallNet<D, O>; using S = SmallPair<D, O>;
    constexpr int NS = H * (D + 2 + O) + O, NLS = HEAD == HEAD_GAUSSIAN ? O : 0;       // small parameters of this net (+ log_std, which lives behind both nets)
    static_assert(NS + NLS <= 768, "three small parameters per thread");
    constexpr int WOFF = H * D + H;                                                    // W2 inside a net's parameters
    float* wl = smem; float* pairs = smem + L::END;                                    // [pair] areas
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pr = wave >> 1, w = wave & 1;
    const int c = lane & 31, h = lane >> 5;
    float* pb = pairs + pr * S::SIZE;
    const int pairB = 4 * (L::END + pr * S::SIZE);
    const int Poff = ROLE ? a.Pa : 0;                                                  // this net's parameters inside the flat vector

    // ---- ownership: the thread keeps, for the whole launch and in registers, the parameters and Adam moments of
    //        W2 pairs (o, 2 kp), (o, 2 kp + 1) with o = tid & 63, kp = (tid >> 6) + 4 j, j = 0..7 (16 parameters: all 4 096 of the net), and
    //        up to three small parameters (index tid + 256 q of {small | log_std}).
    //      After Adam the owner writes the staged form itself: the three bf16 pieces of a W2 pair into the weight image, a small parameter (scaled) into its
    //      staged word — no f32 copy of the parameters goes through LDS and no separate staging pass exists.
    const int wo = tid & 63, wk = tid >> 6;
    float wp[8][2], wm[8][2], wv[8][2];                                                // [j][element of the pair]
    float sp[3], sm[3], sv[3], sscale[3]; int sflat[3], sgoff[3], sdst[3];             // small parameters (sflat < 0: none)
    lds_char* lds = (lds_char*)smem;
    const int wbyte = 4 * L::WIMG + wo * 128 + (wimg_g<64>(wo) << 4) + 4 * wk;          // pair kp = wk + 4 j: chunk j ^ g(o) of row o, word wk — the XOR form again
    auto publish_pair = [&](int j) {
        unsigned hi, lo;
        split2_pair((kTanhScale * kWScale) * wp[j][0], (kTanhScale * kWScale) * wp[j][1], hi, lo);
        const int byte = wbyte ^ (j << 4);
        pl_write(lds, byte, hi); pl_write(lds, byte + 8192, lo);
    };
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int p = Poff + WOFF + wo + 64 * (2 * (wk + 4 * j) + e);
            wp[j][e] = a.params[p]; wm[j][e] = a.adam_m[p]; wv[j][e] = a.adam_v[p];
        }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int si = tid + 256 * q;
        sflat[q] = -1; sgoff[q] = 0; sdst[q] = 0; sscale[q] = 1.0f; sp[q] = 0.f; sm[q] = 0.f; sv[q] = 0.f;
        if (si < NS) { int f; small_param_map<D, O>(si, f, sgoff[q], sdst[q], sscale[q]); sflat[q] = Poff + f; }
        else if (si < NS + NLS) { const int i = si - NS; sflat[q] = a.Pa + a.Pc + i; sgoff[q] = S::G_LS + i; sdst[q] = L::LS + i; }
        if (sflat[q] >= 0) { sp[q] = a.params[sflat[q]]; sm[q] = a.adam_m[sflat[q]]; sv[q] = a.adam_v[sflat[q]]; }
    }
    for (int i = tid; i < L::DP * H; i += 256) wl[L::W1T + i] = 0.f;                     // rows k >= D of the first-layer image stay zero
    for (int i = tid; i < L::OP; i += 256) wl[L::B3 + i] = 0.f;
    lds_barrier();
#pragma unroll
    for (int j = 0; j < 8; ++j) publish_pair(j);
    auto publish_small = [&](int q) {
        if (sflat[q] < 0) return;
        smem[sdst[q]] = sscale[q] * sp[q];
        if (sdst[q] >= L::W3S && sdst[q] < L::W3B) smem[sdst[q] + (L::W3B - L::W3S)] = sp[q] * (1.0f / (kActScale * kActScale));   // the dh copy of W3
    };
#pragma unroll
    for (int q = 0; q < 3; ++q) publish_small(q);
    const float* bt_in = a.bt + 2 * (a.step_parity & 1);
    float bt1 = bt_in[0], bt2 = bt_in[1];

    GradArgs ga{};                                                                    // what unpack_tile / loss_head read
    ga.clip_range = a.clip_range; ga.ent_coef = a.ent_coef; ga.vf_coef = a.vf_coef; ga.clip_range_vf = a.clip_range_vf; ga.has_clip_vf = a.has_clip_vf;
    ga.normalize_adv = a.normalize_adv; ga.action_start = a.action_start;

    // gather of step s: this lane's half record of sample 32 pr + c, and (critic) the advantage of sample `lane` for the minibatch moments
    auto sample_index = [&](int ep, int64_t pos) -> int64_t {
        return a.perm ? a.perm[(int64_t)ep * a.N + pos] : perm_index(pos, a.N, a.keys[ep], a.perm_bits);
    };
    auto count_of = [&](int s) -> int64_t { const int ep = s / a.nb, k = s - ep * a.nb; const int64_t pos0 = (int64_t)k * a.B; return (pos0 + a.B <= a.N) ? a.B : a.N - pos0; };
    constexpr int RS = RecLayout<D>::RS;
    float4 raw_n = make_float4(0.f, 0.f, 0.f, 0.f), raw2_n = raw_n; float vold_n = 0.f, adv_n = 0.f; bool valid_n = false;
    auto gather = [&](int s) {
        const int ep = s / a.nb, k = s - ep * a.nb;
        const int64_t pos0 = (int64_t)k * a.B, count = (pos0 + a.B <= a.N) ? a.B : a.N - pos0;
        const int i = 32 * pr + c;
        valid_n = i < count;
        const int64_t idx = sample_index(ep, pos0 + (valid_n ? i : 0));
        raw_n = a.rec[RS * idx + h];
        if (RS == 3) raw2_n = a.rec[RS * idx + 2];                                      // three-quad records (D > 4): the scalar quad, every lane
        vold_n = (ROLE == 1 && a.has_clip_vf) ? a.val_old[idx] : 0.f;
        adv_n = 0.f;
        if (ROLE == 1 && a.normalize_adv && lane < count) adv_n = a.rec[RS * sample_index(ep, pos0 + lane) + RS - 1].y;
    };
    // normalize!(advantages) per minibatch, ppo.jl:350-356 (corrected std + 1e-8): the critic's first wave, from the advantages it gathered for minibatch s
    auto moments_into = [&](int s, float* dst) {
        const double sm_ = wave_sum_f64((double)adv_n), sq = wave_sum_f64((double)adv_n * (double)adv_n), n = (double)count_of(s);
        const double mean = sm_ / n;
        double var = (sq - sm_ * mean) / (n - 1.0);
        if (var < 0) var = 0;
        if (lane == 0) { dst[0] = (float)mean; dst[1] = 1.0f / ((float)sqrt(var) + 1.0e-8f); }
    };
    const int s_end = a.step0 + a.nsteps;
    if (a.step0 < s_end) gather(a.step0);
    unsigned xk_ = 0;                                                                  // exchange index
    bool alive = true;
    {   // exchange 0: the first minibatch's advantage moments
        if (tid < kMsgWords) shx[tid] = 0.f;
        lds_barrier();
        if (ROLE == 1 && a.normalize_adv && wave == 0 && a.step0 < s_end) moments_into(a.step0, shx + 5);
        lds_barrier();
        if (wave == 0) { const bool ok = small_exchange(a.xchg, ROLE, xk_, shx, xin, lane); if (lane == 0) xin[kMsgWords] = ok ? 1.f : 0.f; }
        lds_barrier();
        alive = xin[kMsgWords] != 0.f;
        if (!alive && tid == 0) { *a.nan_flag = 2; *a.stop_flag = 1; }
        if (ROLE == 0 && tid == 0) { mom[0] = xin[5]; mom[1] = xin[6]; }
        ++xk_;
    }
#ifdef DRIL_STAMPS
    unsigned long long stamp_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev) :: "memory");
#endif
    for (int s = a.step0; s < s_end && alive; ++s) {
        const int64_t count = count_of(s);
        const float invB = 1.0f / (float)count;
        const float sg = __uint_as_float((((__float_as_uint((float)count) >> 23) & 0xffu) + 3u) << 23), inv_sg = 1.0f / sg;   // gradient tiles are split as SG dz2, SG = 4 ... 8 x count, a power of two
        ga.invB = invB * sg;                                                          // what loss_head multiplies dLoss/dout with
        // ---- this step's inputs out of the prefetch registers; the next step's gathers go out now and land under this step's arithmetic ----
        const float4 raw = raw_n, raw2 = raw2_n; const float vold = vold_n; const bool valid = valid_n;
        if (s + 1 < s_end) gather(s + 1);
        float lsr[kLsMax];
#pragma unroll
        for (int o = 0; o < kLsMax; ++o) lsr[o] = 0.f;
        lds_barrier();                                                                // (weight images and staged parameters of the previous step complete; the dW2 overlay becomes images again; mom[] in place)
        if (HEAD == HEAD_GAUSSIAN) {
#pragma unroll
            for (int o = 0; o < O; ++o) lsr[o] = wl[L::LS + o];
        }
        STAMP(0);
        {
            TileIn<O, FirstLayer<D>::KS> cur; cur.raw = raw; cur.raw2 = raw2; cur.valid = valid; cur.s0 = 0.f; cur.s1 = ROLE == 1 ? vold : 0.f; cur.act = 0;
            small_tile<KIND, O, HEAD>(ga, smem, pb, pairB, cur, mom, ROLE == 0 ? a.normalize_adv : 0, lsr, lane, w, inv_sg SMALL_STAMP_ARGS);
        }
        STAMP(9);
        lds_barrier();                                                                // B5: both pairs' gradients complete
        STAMP(10);
        // ---- optimiser phase, first part: this net's gradient and its share of |g|^2 ----
        float gw[8][2], gs[3]; float ss = 0.f;                                         // 19 squares per thread in f32 (fixed order), 64 threads by DPP, 8 waves in index order
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int off = S::S_W2 + wo * 65 + 2 * (wk + 4 * j) + e;
                const float g = pairs[off] + pairs[off + S::SIZE];                   // pair 0 + pair 1, fixed order
                gw[j][e] = g; ss = fmaf(g, g, ss);
            }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float g = sflat[q] >= 0 ? pairs[sgoff[q]] + pairs[sgoff[q] + S::SIZE] : 0.f;
            gs[q] = g; ss = fmaf(g, g, ss);
        }
        {   // |g|^2: per thread and per wave on the VALU, the waves of both workgroups in index order after the exchange
            const float wsum = wave_sum_f32(ss);
            if (lane == 0) shx[wave] = wsum;
        }
        if (wave == 3 && lane < 8) {                                                  // the statistics sums of the minibatch (the last wave has the least to do here)
            float t = 0.f;
            if (ROLE == 0) { if (lane < 5) t = pairs[S::G_ST + lane] + pairs[S::SIZE + S::G_ST + lane]; shx[4 + lane] = t; }
            else if (lane == 0) shx[4] = pairs[S::G_ST] + pairs[S::SIZE + S::G_ST];
        }
        if (ROLE == 1 && a.normalize_adv && wave == 0 && s + 1 < s_end) moments_into(s + 1, shx + 5);   // adv_n: the advantages gathered at the top of this step
        lds_barrier();
        if (wave == 0) { const bool ok = small_exchange(a.xchg, ROLE, xk_, shx, xin, lane); if (lane == 0) xin[kMsgWords] = ok ? 1.f : 0.f; }
        ++xk_;
        lds_barrier();
        STAMP(11);
        if (xin[kMsgWords] == 0.f) {                                                   // the partner never answered: leave (uniform), the host reports a non-finite update
            if (tid == 0) { *a.nan_flag = 2; *a.stop_flag = 1; }
            alive = false; break;
        }
        const float* xa = ROLE == 0 ? shx : xin; const float* xc = ROLE == 0 ? xin : shx;   // the actor's and the critic's message
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) tot += xa[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) tot += xc[k];
        const float norm = __builtin_amdgcn_sqrtf(tot);                                 // (1 ulp; NaN / Inf pass through to the test below)
        const float n = (float)count, inv_n = 1.0f / n, kl = xa[4 + 3] * inv_n;
        const bool bad = !(norm == norm) || isinf(norm);                              // NaN / Inf anywhere poisons the norm (ppo.jl:213-214)
        const bool kl_stop = a.has_target_kl && kl > 1.5f * a.target_kl;              // ppo.jl:235-238: skip this apply, stop
        if (ROLE == 0 && tid == 0) {
            float* o = a.step_stats + (size_t)s * 16;
            const float pl = xa[4] * inv_n, ent = xa[5] * inv_n, vl = xc[4] * inv_n;
            o[0] = pl; o[1] = vl; o[2] = -ent; o[3] = xa[6] * inv_n; o[4] = kl; o[5] = ent; o[6] = xa[8] * inv_n;
            o[7] = pl + a.ent_coef * (-ent) + a.vf_coef * vl;                          // loss, ppo.jl:386
            o[8] = norm; o[9] = (bad || kl_stop) ? 0.f : 1.f; o[10] = bad ? 1.f : 0.f; o[11] = kl_stop ? 1.f : 0.f;
            if (a.norm_out) *a.norm_out = norm;
            if (bad) { *a.nan_flag = 1; *a.stop_flag = 1; }
            if (kl_stop) *a.stop_flag = 1;
        }
        if (ROLE == 0 && tid == 0) { mom[0] = xc[5]; mom[1] = xc[6]; }                  // the next minibatch's advantage moments (read after the next top barrier)
        if (bad || kl_stop) break;                                                    // uniform in both workgroups: every thread computed the same norm / kl
        const float scale = (a.has_max_grad_norm && norm > a.max_grad_norm) ? a.max_grad_norm * __builtin_amdgcn_rcpf(norm) : 1.0f;   // optimization_utils.jl:98-107
        // bias corrections once per step (every thread the same value); per parameter one hardware reciprocal and one hardware square root (1 ulp each: the update is
        // lr x O(1), so their error is ~1e-11 absolute — far below one ulp of a parameter) instead of three IEEE divisions and an IEEE square root (~40 instructions)
        const float ic1 = __builtin_amdgcn_rcpf(1.0f - bt1), ic2 = __builtin_amdgcn_rcpf(1.0f - bt2), omb1 = 1.0f - a.beta1, omb2 = 1.0f - a.beta2;
        auto adam = [&](float g, float& p, float& m, float& v) {                       // Optimisers.Adam, eps = 1e-5 (ppo.jl:64-66)
            g = g * scale;
            m = a.beta1 * m + omb1 * g; v = a.beta2 * v + omb2 * g * g;
            p = p - (m * ic1) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v * ic2) + a.eps) * a.lr;
        };
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            adam(gw[j][0], wp[j][0], wm[j][0], wv[j][0]); adam(gw[j][1], wp[j][1], wm[j][1], wv[j][1]);
            publish_pair(j);
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            adam(gs[q], sp[q], sm[q], sv[q]);
            publish_small(q);
        }
        bt1 *= a.beta1; bt2 *= a.beta2;
        STAMP(12);
        // (the barrier at the top of the next step separates these image stores from the next reads of the weight images and from the next image stores)
    }
#ifdef DRIL_STAMPS
    if (lane == 0 && a.dbg) { for (int k = 0; k < 16; ++k) a.dbg[(ROLE * 4 + wave) * 16 + k] = stamp_acc[k]; }
#endif
    // ---- state back to global memory ----
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int p = Poff + WOFF + wo + 64 * (2 * (wk + 4 * j) + e);
            a.params[p] = wp[j][e]; a.adam_m[p] = wm[j][e]; a.adam_v[p] = wv[j][e];
        }
#pragma unroll
    for (int q = 0; q < 3; ++q) if (sflat[q] >= 0) { a.params[sflat[q]] = sp[q]; a.adam_m[sflat[q]] = sm[q]; a.adam_v[sflat[q]] = sv[q]; }
    if (ROLE == 0 && tid == 0) { a.bt[0] = bt1; a.bt[1] = bt2; a.bt[2] = bt1; a.bt[3] = bt2; }     // both ping-pong slots: the host's step parity no longer matters
}

constexpr int kSmallGrid = 9, kSmallCriticBlock = 8;       // workgroups 0 and 8 land on the same XCD (ids are dealt round-robin over the eight XCDs)
template <int KIND>
__global__ __launch_bounds__(256, 1) void ppo_update_small_kernel(SmallUpdateArgs a) {
    constexpr int A = EnvSpec<KIND>::A;
    constexpr int AHEAD = EnvSpec<KIND>::discrete ? HEAD_CATEGORICAL : HEAD_GAUSSIAN;
    extern __shared__ __attribute__((aligned(1024))) float smem[];          // XOR addressing needs 512-byte image bases
    __shared__ float shx[kMsgWords];
    __shared__ float xin[kMsgWords + 1];
    __shared__ float mom[2];
    if (blockIdx.x != 0 && blockIdx.x != kSmallCriticBlock) return;
    if (*a.stop_flag) return;
    if (blockIdx.x == 0) small_net_loop<KIND, A, AHEAD, 0>(a, smem, shx, xin, mom);
    else small_net_loop<KIND, 1, HEAD_VALUE, 1>(a, smem, shx, xin, mom);
}

template <int KIND> static size_t update_small_lds_bytes() {
    constexpr int D = EnvSpec<KIND>::D, A = EnvSpec<KIND>::A;
    constexpr int wa = SmallNet<D, A>::END + 2 * SmallPair<D, A>::SIZE, wc = SmallNet<D, 1>::END + 2 * SmallPair<D, 1>::SIZE;
    return sizeof(float) * (wa > wc ? wa : wc);
}

hipError_t launch_ppo_update_small(int kind, const SmallUpdateArgs& a, hipStream_t s) {
    if (kind == 7) kind = 4;                  // ScalingWrapperEnv(MountainCarContinuous): the update never touches the simulator
    if (!a.xchg) return hipErrorInvalidValue;
    { hipError_t e = hipMemsetAsync(a.xchg, 0, sizeof(unsigned long long) * kSmallXchgWords, s); if (e != hipSuccess) return e; }   // sequence number 0 = no message yet
#define CALLU(K) { const size_t lds = update_small_lds_bytes<K>(); \
        { hipError_t e = set_max_dynamic_lds((const void*)ppo_update_small_kernel<K>, lds); if (e != hipSuccess) return e; } \
        ppo_update_small_kernel<K><<<a.debug_solo ? 1 : kSmallGrid, 256, lds, s>>>(a); }
    if (kind == 0) CALLU(0) else if (kind == 3) CALLU(3) else if (kind == 4) CALLU(4) else if (kind == 6) CALLU(6) else CALLU(1)
#undef CALLU
    return hipGetLastError();
}

}  // namespace dril
