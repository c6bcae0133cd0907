// dril_grad_wide_teams.h — ppo_grad_wide_teams_kernel: ppo_grad_wide_split_kernel's arithmetic (dril_grad_wide_split.h) with the workgroup's waves in two TEAMS that run half
// a pass apart, so that on every SIMD one wave is in a matrix stage while the other one is in a vector stage.  Included by dril_grad_wide.hip.
//   * team q (waves q TW .. q TW + TW - 1, TW = H / 64) owns the q-th sample tile of every pass; its wave j runs the chains (h2, dh1') and the vector stages for the m-tiles
//     2j, 2j + 1 of that tile (dense_tile_split<H, 1, 2>: every W2 / W2' fragment feeds ONE sample tile — twice the fragment stream per sample of the split kernel);
//   * dW2 | db2 stay where they were: wave w accumulates the rows of ONE m-tile (2j + q) in 128 registers, over BOTH teams' sample tiles — the other team's tile in its first
//     half, its own in its second half, each time from the piece images of that tile;
//   * LDS as the split kernel (two tiles' piece images, 143 - 161 KB at H = 256): slot q of every image and buffer belongs to team q.
// Reference: the same lines as the split kernel (ppo.jl:365-407 loss, :207 gradient); parity tests: tests/test_gpu_wide*.py run both kernels.
#pragma once
#include "dril_grad_wide_split.h"

namespace dril {

template <int KIND, int H, int O, int HEAD>
__device__ __forceinline__ void grad_body_wide_teams(const GradArgs& a, float* smem) {
    constexpr int D = EnvSpec<KIND>::D, MT = H / 32, NT = 2, NW = MT, TW = NW / 2, MW = 2;   // two teams of TW waves; a team's wave owns MW = 2 m-tiles of the chains of its team's sample tile
    static_assert(MT % 4 == 0 && kWideSplitNT == 2, "two teams, two sample tiles per pass");
    constexpr int RB = 2 * H, PS = 32 * RB, NTS = 2 * PS;            // bytes of an image row, of one piece of a sample tile, of a sample tile's image
    constexpr bool REC = true;
    using L = NetLdsSmall<D, H, O>;
    using SC = WideSplitScratch<D, H, O, NT>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = w / TW, j = w - q * TW;                            // team (waves w and w + TW share a SIMD at H = 256), wave within the team
    const int mw0 = MW * j;                                          // the chains' m-tiles: MW j, MW j + 1 (of the team's sample tile)
    const int mdw = MW * j + q;                                      // the m-tile of dW2 | db2 this wave accumulates (over BOTH teams' sample tiles)
    const int c = lane & 31, h = lane >> 5;
    const NetOff off = HEAD == HEAD_VALUE ? a.critic : a.actor;
    const u32x4* w2p = HEAD == HEAD_VALUE ? a.w2p_critic : a.w2p_actor;
    const u32x4* w2tp = HEAD == HEAD_VALUE ? a.w2tp_critic : a.w2tp_actor;
    float* wl = smem;
    char* P1 = reinterpret_cast<char*>(smem + SC::P1); char* P2 = reinterpret_cast<char*>(smem + SC::P2);
    float* XI = smem + SC::XI; float* PO = smem + SC::PO;
    float* RECS = smem + SC::REC; float* VO = smem + SC::VO; int* VAL = reinterpret_cast<int*>(smem + SC::VAL);
    constexpr int RS = RecLayout<D>::RS, RECT = SC::RQ * 32 * 4;    // record quads per sample; floats of one sample tile's record block
    // staged small parameters in the scales of the f16-piece arithmetic (dril_device.h): b2 starts the SCALED accumulator of L2, W3S = W3 / kActScale for the output layer
    // (its operand is kActScale h2), W3B = W3 / kActScale^2 for dh
    {
        const float* __restrict__ P = a.params;
        for (int i = tid; i < L::DP * H; i += blockDim.x) { const int o = i % H, k = i / H; wl[L::W1T + k * H + o] = k < D ? kTanhScale * P[off.w1 + o + k * H] : 0.0f; }
        for (int i = tid; i < H; i += blockDim.x) { wl[L::B1 + i] = kTanhScale * P[off.b1 + i]; wl[L::B2 + i] = (kTanhScale * kWScale * kActScale) * P[off.b2 + i]; }
        for (int i = tid; i < O * H; i += blockDim.x) { const int o = i % O, k = i / O; const float w3 = P[off.w3 + i]; wl[L::W3S + o * H + k] = w3 * (1.0f / kActScale); smem[SC::W3B + o * H + k] = w3 * (1.0f / (kActScale * kActScale)); }
        for (int i = tid; i < L::OP; i += blockDim.x) wl[L::B3 + i] = i < O ? P[off.b3 + i] : 0.0f;
    }
    for (int i = tid; i < NT * (D + 2) * kTS; i += blockDim.x) XI[i] = 0.0f;
    for (int i = tid; i < 2 * NT * 32 * H; i += blockDim.x) smem[SC::P1 + i] = 0.0f;   // the piece images start as zeros: a dW2 product over a sample tile that does not exist yet (or never will) adds nothing
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
    const int tbase = wide_tr_base<H>(lane), tmbase = wide_trm_base<H>(lane);
    // gradient tiles are split as dz2 SG with SG = 2^(exponent of 1 / invB + 3): 4 ... 8 / invB, a power of two; every scale is undone exactly in the epilogue
    const float sg = __uint_as_float((((__float_as_uint(1.0f / a.invB) >> 23) & 0xffu) + 3u) << 23);
    const float inv_sg = 1.0f / sg, inv_sa = inv_sg * (1.0f / kActScale);
    GradArgs as = a; as.invB = a.invB * sg;                                            // what loss_head multiplies dLoss/dout with
    const float* W3B = smem + SC::W3B;

    f32x16 dW2[MT];                                                  // rows of m-tile mdw, all H columns
    float dW1a[MW][D], db1a[MW], dW3a[MW][O], db2a = 0.f, db3p[O], dlsp[O], st[5];   // per-lane partial sums over the TEAM's samples: dW1a / db1a for unit 32 (mw0 + m) + (lane & 31), dW3a for unit 32 (mw0 + m) + rowfn(lane & 15, h); db2a for unit 32 mdw + (lane & 31) over all samples
#pragma unroll
    for (int mj = 0; mj < MT; ++mj)
#pragma unroll
        for (int r = 0; r < 16; ++r) dW2[mj][r] = 0.f;
#pragma unroll
    for (int m = 0; m < MW; ++m) {
#pragma unroll
        for (int d = 0; d < D; ++d) dW1a[m][d] = 0.f;
#pragma unroll
        for (int o = 0; o < O; ++o) dW3a[m][o] = 0.f;
        db1a[m] = 0.f;
    }
#pragma unroll
    for (int o = 0; o < O; ++o) { db3p[o] = 0.f; dlsp[o] = 0.f; }
#pragma unroll
    for (int i = 0; i < 5; ++i) st[i] = 0.f;

    const int g = (int)(blockIdx.x % a.G);
    const int64_t ntiles = (a.count + kTile - 1) / kTile;            // sample tiles of 32; a pass of the workgroup = two consecutive ones, team q's is the q-th (a missing one is all-invalid)
    constexpr int KS = FirstLayer<D>::KS;
    const int64_t tile0 = (int64_t)g * NT, stride = (int64_t)a.G * NT;
    const int P = tile0 < ntiles ? (int)((ntiles - tile0 + stride - 1) / stride) : 0;   // passes of this workgroup
    // this team's slots of the workgroup images and buffers
    char* P1q = P1 + q * NTS; char* P2q = P2 + q * NTS;
    float* XIq = XI + q * (D + 2) * kTS; float* POq = PO + q * TW * O * 32;
    float* RECq = RECS + q * RECT; float* VOq = VO + q * 64; int* VALq = VAL + q * 64;
    // wave 0 of a team is its loader: the records of the team's NEXT sample tile by LDS-DMA, requested behind the dh1 chain (the stages that follow issue no vector-memory
    // instruction); the barrier that ends the half-pass drains the DMA
    TileIdx nidx; nidx.gidx = 0; nidx.g32 = 0; nidx.is32 = false; nidx.inb = false;
    if (j == 0 && P > 0) {
        request_records_lds<KIND, HEAD>(a, tile_index(a, tile0 + q, ntiles, c), lane, RECq, VOq, VALq);
        nidx = tile_index(a, tile0 + stride + q, ntiles, c);
    }
    __syncthreads();
#ifdef DRIL_STAMPS
    unsigned long long stamp_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev) :: "memory");
#endif
    // Half-passes.  A team alternates a FIRST half (slot a: h1 + pieces | slot b: the other team's tile into dW2, then the h2 chain + tanh + output partials | slot c: head,
    // dW3, dz2 + pieces) and a SECOND half (slot a: dh1 chain | slot b: dz1, dW1 | slot c: its own tile into dW2) on its sample tile, team 1 half a pass behind team 0: in
    // every slot one wave of a SIMD is in a matrix stage and the other one in a vector stage (tools/micro/mfma_valu_asm.hip: the two run side by side across waves).
    // Three workgroup barriers per half-pass (A, B, C), crossed by all waves.  The dW2 product is ONE copy of code between the two arms — inside both arms of a branch its
    // 128 accumulators meet in phi nodes and the allocator spills hundreds of registers (profiles/r05_wide_split.md §7).
    for (int hp = 0; hp <= 2 * P; ++hp) {
        f32x16 g1[MW][1];                                            // dh1' of the team's tile: from the second half's slot a to its slot b (declared per iteration: not live around the loop)
        const bool FH = ((hp + q) & 1) == 0;
        const int pq = (hp - q) >> 1;                                // the team's pass (first half: hp - q = 2 pq, second half: 2 pq + 1)
        const bool active = hp >= q && pq < P;
        const int64_t tile = tile0 + q + (int64_t)pq * stride;
        // ================= slot a =================
        if (FH) {
            if (active) {
                // ---- h1 m-tiles mw0, mw0 + 1 of the team's sample tile; its pieces into the team's image ----
                const int ln_ = opaque(lane), c = ln_ & 31, h = ln_ >> 5;
                float xk[KS];
#pragma unroll
                for (int s = 0; s < KS; ++s) xk[s] = RECq[((((2 * s) >> 2) * 32 + c) << 2) + ((2 * s) & 3) + h];
#pragma unroll
                for (int m = 0; m < MW; ++m) {
                    const int mt = mw0 + m;
                    f32x16 h1w;
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const f32x4 b = *reinterpret_cast<const f32x4*>(wl + L::B1 + 32 * mt + 8 * qq + 4 * h);
                        h1w[4 * qq + 0] = b[0]; h1w[4 * qq + 1] = b[1]; h1w[4 * qq + 2] = b[2]; h1w[4 * qq + 3] = b[3];
                    }
#pragma unroll
                    for (int s = 0; s < KS; ++s) h1w = mfma32(wl[L::W1T + (2 * s + h) * H + 32 * mt + c], xk[s], h1w);
                    tanh16_scaled<false>(h1w, 1.0f);                                  // kActScale h1
                    store_tile_pieces2<H>(P1q, mt, h1w, ln_);
                }
                if (j == 0) {                                                         // the team's wave 0 keeps the tile's observations for the dW1 sums
#pragma unroll
                    for (int s = 0; s < KS; ++s) { const int d = 2 * s + h; XIq[(d < D ? d : D + 1) * kTS + c] = d < D ? xk[s] : 0.f; }
                }
            }
            STAMP(0);
        } else {
            if (active) {
                // ---- dh1' m-tiles mw0, mw0 + 1 = (W2' dz2)', transposed: lane = unit 32 (mw0 + m) + (lane & 31), register r = sample rowfn(r, h) ----
                u32x4 afw[MW][2][2];
#pragma unroll
                for (int m = 0; m < MW; ++m) wide_split_preload(w2tp, MT, mw0 + m, lane, afw[m]);
                dense_tile_split<H, 1, MW, false, true>(w2tp, nullptr, P2q, mw0, opaque(lane), afw, g1);
                if (j == 0) {                                                         // the team's next tile's records; the epoch-order entry of the one after
                    const int ln_ = opaque(lane);
                    request_records_lds<KIND, HEAD>(a, nidx, ln_, RECq, VOq, VALq);
                    nidx = tile_index(a, tile + 2 * stride, ntiles, ln_ & 31);
                }
            }
            STAMP(1);
        }
        lds_barrier();                                                                // A: the team's h1 image | nothing for the second half
        STAMP(2);
        // ================= slot b =================
        if (!FH) {
            if (active) {
                // ---- dz1', then dW1 | db1 as per-lane sums over the lane's samples: four groups (register group Q = samples 8Q + 4h + {0..3} of the lane's unit), the reads of
                // group Q + 1 requested before group Q is computed ----
                constexpr float c0 = 1.0f / kWScale, c1 = c0 / (kActScale * kActScale);         // g1 = (SG dz2 . kWScale W2) (1 - h1^2) / kWScale = SG dz1
                const int ln_ = opaque(lane), h = ln_ >> 5;
                const int tm0 = opaque(tmbase) ^ (64 * mw0);
                const float* xrow = XIq + 4 * h;
                u32x2 hp_[2][MW][2]; f32x4 xq[2][D];
#define S3_LOAD(B, Q) { _Pragma("unroll") for (int m = 0; m < MW; ++m) { const int a_ = (tm0 ^ (64 * m) ^ (((Q) & 1) ? 32 : 0)) + 8 * (Q) * RB; \
                               hp_[B][m][0] = __builtin_bit_cast(u32x2, lds_read_tr16(P1q, a_)); hp_[B][m][1] = __builtin_bit_cast(u32x2, lds_read_tr16(P1q, a_ + PS)); } \
                           _Pragma("unroll") for (int d = 0; d < D; ++d) xq[B][d] = *reinterpret_cast<const f32x4*>(xrow + d * kTS + 8 * (Q)); }
#define S3_COMP(B, Q) { _Pragma("unroll") for (int m = 0; m < MW; ++m) { float hv[4]; pieces_sum2(hp_[B][m][0].x, hp_[B][m][1].x, hv[0], hv[1]); pieces_sum2(hp_[B][m][0].y, hp_[B][m][1].y, hv[2], hv[3]); \
                           _Pragma("unroll") for (int i = 0; i < 4; ++i) { const float t2 = hv[i] * hv[i]; const float gz = g1[m][0][4 * (Q) + i] * fmaf(-t2, c1, c0); db1a[m] += gz; \
                               _Pragma("unroll") for (int d = 0; d < D; ++d) dW1a[m][d] = fmaf(gz, xq[B][d][i], dW1a[m][d]); } } }
                S3_LOAD(0, 0)
                S3_LOAD(1, 1)
                __builtin_amdgcn_sched_barrier(0);
                S3_COMP(0, 0)
                __builtin_amdgcn_sched_barrier(0);
                S3_LOAD(0, 2)
                __builtin_amdgcn_sched_barrier(0);
                S3_COMP(1, 1)
                __builtin_amdgcn_sched_barrier(0);
                S3_LOAD(1, 3)
                __builtin_amdgcn_sched_barrier(0);
                S3_COMP(0, 2)
                __builtin_amdgcn_sched_barrier(0);
                S3_COMP(1, 3)
#undef S3_LOAD
#undef S3_COMP
            }
            STAMP(3);
            lds_barrier();                                                            // B (second half: the first half's output partials are what it orders)
            STAMP(4);
        }
        // ---- dW2[rows of mdw][:] += dz2 h1' of ONE sample tile (first half: the other team's, complete since that team's first half; second half: the own one); db2 from the
        // dz2 fragments.  Both operands as transposed fragments of the piece images; the h1 fragments of m-tile mj + 1 are requested before the MFMAs of m-tile mj ----
        {
            const int sel = FH ? 1 - q : q;
            const char* P1s = P1 + sel * NTS; const char* P2s = P2 + sel * NTS;
            const int tb = opaque(tbase), tbw = tb ^ (64 * mdw), tbw16 = tbw ^ 16;
            f16x8 Az[2][2];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int p = 0; p < 2; ++p) { Az[s][p] = __builtin_bit_cast(f16x8, load_frag_wide_T<H>(P2s, tbw, tbw16, p, s)); db2a = frag_sum8(Az[s][p], db2a); }
            f16x8 BhA[2][2], BhB[2][2];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int p = 0; p < 2; ++p) BhA[s][p] = __builtin_bit_cast(f16x8, load_frag_wide_T<H>(P1s, tb, tb ^ 16, p, s));
#pragma unroll
            for (int mj = 0; mj < MT; mj += 2) {
                const int tb1 = tb ^ (64 * (mj + 1)), mn = mj + 2 < MT ? mj + 2 : mj, tbn = tb ^ (64 * mn);
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int p = 0; p < 2; ++p) BhB[s][p] = __builtin_bit_cast(f16x8, load_frag_wide_T<H>(P1s, tb1, tb1 ^ 16, p, s));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 2; ++s) dW2[mj] = mfma_split3(Az[s][0], Az[s][1], BhA[s][0], BhA[s][1], dW2[mj]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int p = 0; p < 2; ++p) BhA[s][p] = __builtin_bit_cast(f16x8, load_frag_wide_T<H>(P1s, tbn, tbn ^ 16, p, s));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 2; ++s) dW2[mj + 1] = mfma_split3(Az[s][0], Az[s][1], BhB[s][0], BhB[s][1], dW2[mj + 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        STAMP(5);
        if (FH) {
            f32x16 h2w[MW][1];
            if (active) {
                // ---- h2 m-tiles mw0, mw0 + 1; output layer: partial over this wave's rows, summed across the team through LDS ----
                u32x4 afw[MW][2][2];
#pragma unroll
                for (int m = 0; m < MW; ++m) wide_split_preload(w2p, MT, mw0 + m, lane, afw[m]);
                dense_tile_split<H, 1, MW, true, false>(w2p, wl + L::B2, P1q, mw0, opaque(lane), afw, h2w);
#pragma unroll
                for (int m = 0; m < MW; ++m) tanh16_scaled<true>(h2w[m][0], 1.0f / (kWScale * kActScale));   // kActScale h2
#pragma unroll
                for (int o = 0; o < O; ++o) {
                    const int ln_ = opaque(lane), c = ln_ & 31, h = ln_ >> 5;
                    float p = 0.f;
#pragma unroll
                    for (int m = 0; m < MW; ++m)
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) {
                            const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + L::W3S + o * H + 32 * (mw0 + m) + 8 * qq + 4 * h);
                            p = fmaf(wv[0], h2w[m][0][4 * qq + 0], p); p = fmaf(wv[1], h2w[m][0][4 * qq + 1], p);
                            p = fmaf(wv[2], h2w[m][0][4 * qq + 2], p); p = fmaf(wv[3], h2w[m][0][4 * qq + 3], p);
                        }
                    p += __shfl_xor(p, 32);
                    if (h == 0) POq[(j * O + o) * 32 + c] = p;
                }
            }
            STAMP(6);
            lds_barrier();                                                            // B: the team's output partials complete
            STAMP(7);
            if (active) {
                const int ln_ = opaque(lane), c = ln_ & 31, h = ln_ >> 5;
                float out[O], dz[O];
#pragma unroll
                for (int o = 0; o < O; ++o) {
                    float v = wl[L::B3 + o];
#pragma unroll
                    for (int ww = 0; ww < TW; ++ww) v += POq[(ww * O + o) * 32 + c];            // fixed order: every wave of the team gets the same bits
                    out[o] = v;
                }
                const f32x4 sc = *reinterpret_cast<const f32x4*>(RECq + (((RS - 1) * 32 + c) << 2));   // {action bits, adv, logp_old, ret} of this lane's sample
                TileIn<O, KS> cur; cur.act = 0; cur.s0 = 0.f; cur.s1 = 0.f;
                const bool valid = VALq[c] != 0;
                if (HEAD == HEAD_VALUE) { cur.s0 = sc[3]; cur.s1 = a.has_clip_vf ? VOq[c] : 0.f; }
                else {
                    cur.s0 = sc[1]; cur.s1 = sc[2];
                    if (HEAD == HEAD_CATEGORICAL) cur.act = __float_as_int(sc[0]) - a.action_start; else cur.xa[0] = sc[0];
                }
                loss_head<O, HEAD>(as, cur, out, valid, h == 0 && j == 0, ls, adv_mean, adv_inv, dz, st, dlsp);     // dz = SG dLoss/dout
#pragma unroll
                for (int o = 0; o < O; ++o) db3p[o] += (h == 0 && j == 0) ? dz[o] : 0.f;
                // ---- dW3 (own rows): over the lanes (= samples) of each half-wave; lane l ends with unit 32 (mw0 + m) + rowfn(l & 15, h) ----
#pragma unroll
                for (int m = 0; m < MW; ++m)
#pragma unroll
                    for (int o = 0; o < O; ++o) {
                        f32x16 v;
#pragma unroll
                        for (int r = 0; r < 16; ++r) v[r] = h2w[m][0][r] * dz[o];
                        dW3a[m][o] += half_reduce16_lane(v, opaque(lane));
                    }
                // ---- dz2 (in h2w's registers); its pieces into the team's image ----
#pragma unroll
                for (int m = 0; m < MW; ++m) {
                    const int mt = mw0 + m;
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        float dh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int o = 0; o < O; ++o) {
                            const f32x4 wv = *reinterpret_cast<const f32x4*>(W3B + o * H + 32 * mt + 8 * qq + 4 * h);
#pragma unroll
                            for (int cc = 0; cc < 4; ++cc) dh[cc] = fmaf(wv[cc], dz[o], dh[cc]);
                        }
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) { const float hv = h2w[m][0][4 * qq + cc]; h2w[m][0][4 * qq + cc] = dh[cc] * fmaf(-hv, hv, kActScale * kActScale); }   // = SG dz2
                    }
                    store_tile_pieces2<H>(P2q, mt, h2w[m][0], ln_);
                }
            }
            STAMP(8);
        }
        __syncthreads();                                                              // C: the images of both teams as the next half-pass reads them; the loaders' DMA has landed
        STAMP(9);
    }
#ifdef DRIL_STAMPS
    if (lane == 0 && a.dbg) {
        unsigned long long* o_ = a.dbg + ((size_t)(blockIdx.x % (2 * a.G)) * 4 + (w & 3)) * 12;
#ifdef DRIL_STAMPS_HI
        if (w >= NW / 2 && w < NW / 2 + 4)
#else
        if (w < 4)
#endif
        { for (int k = 0; k < 10; ++k) o_[k] = stamp_acc[k]; o_[10] = (unsigned long long)P; o_[11] = HEAD; }
    }
#endif

    // ---- epilogue.  dW2 | db2 rows (m-tile mdw) belong to one wave; the sums over a team's samples for the chains' units exist once per team: team 1 hands its to team 0
    // through LDS (the images are free) ----
    float* EX = smem + SC::P1;                                                        // [TW][MW][D + 1 + O][64] | [O + O + 5]
    constexpr int EXU = D + 1 + O;
    if (q == 1) {
#pragma unroll
        for (int m = 0; m < MW; ++m) {
#pragma unroll
            for (int d = 0; d < D; ++d) EX[((j * MW + m) * EXU + d) * 64 + lane] = dW1a[m][d];
            EX[((j * MW + m) * EXU + D) * 64 + lane] = db1a[m];
#pragma unroll
            for (int o = 0; o < O; ++o) EX[((j * MW + m) * EXU + D + 1 + o) * 64 + lane] = dW3a[m][o];
        }
    }
    float* EXS = EX + TW * MW * EXU * 64;
    {
        float tot[2 * O + 5];
#pragma unroll
        for (int o = 0; o < O; ++o) { tot[o] = half_sum(db3p[o]); tot[O + o] = half_sum(dlsp[o]); }
#pragma unroll
        for (int k = 0; k < 5; ++k) tot[2 * O + k] = half_sum(st[k]);
        if (q == 1 && j == 0 && lane == 0) {
#pragma unroll
            for (int k = 0; k < 2 * O + 5; ++k) EXS[k] = tot[k];
        }
        __syncthreads();
        const int SL = HEAD == HEAD_VALUE ? a.slab_c : a.slab_a;
        const int o_w1 = 0, o_b1 = H * D, o_w2 = o_b1 + H, o_b2 = o_w2 + H * H, o_w3 = o_b2 + H, o_b3 = o_w3 + O * H;
        const int o_ls = o_b3 + O, o_st = SL - 8;
        float* slab = (HEAD == HEAD_VALUE ? a.slabs_critic : a.slabs_actor) + (size_t)g * SL;
#pragma unroll
        for (int mj = 0; mj < MT; ++mj)
#pragma unroll
            for (int r = 0; r < 16; ++r) slab[o_w2 + 32 * mdw + rowfn(r, h) + (32 * mj + c) * H] = dW2[mj][r] * inv_sa;
        { const float b2 = (db2a + __shfl_xor(db2a, 32)) * inv_sg; if (h == 0) slab[o_b2 + 32 * mdw + c] = b2; }
        if (q == 0) {
#pragma unroll
            for (int m = 0; m < MW; ++m) {
                const int mt = mw0 + m;
#pragma unroll
                for (int d = 0; d < D; ++d) { const float s_ = dW1a[m][d] + EX[((j * MW + m) * EXU + d) * 64 + lane]; const float v = (s_ + __shfl_xor(s_, 32)) * inv_sg; if (h == 0) slab[o_w1 + 32 * mt + c + d * H] = v; }
                { const float s_ = db1a[m] + EX[((j * MW + m) * EXU + D) * 64 + lane]; const float b1 = (s_ + __shfl_xor(s_, 32)) * inv_sg; if (h == 0) slab[o_b1 + 32 * mt + c] = b1; }
#pragma unroll
                for (int o = 0; o < O; ++o) { const float s_ = dW3a[m][o] + EX[((j * MW + m) * EXU + D + 1 + o) * 64 + lane]; if ((lane & 16) == 0) slab[o_w3 + o + (32 * mt + rowfn(lane & 15, h)) * O] = s_ * inv_sa; }
            }
            if (j == 0 && lane == 0) {
#pragma unroll
                for (int o = 0; o < O; ++o) {
                    slab[o_b3 + o] = (tot[o] + EXS[o]) * inv_sg;
                    if (HEAD == HEAD_GAUSSIAN) slab[o_ls + o] = (tot[O + o] + EXS[O + o]) * inv_sg;
                }
#pragma unroll
                for (int k = 0; k < 5; ++k) slab[o_st + k] = tot[2 * O + k] + EXS[2 * O + k];
            }
        }
        if (w == 0 && lane < 3) slab[o_st + 5 + lane] = 0.f;
        for (int i = (HEAD == HEAD_GAUSSIAN ? o_ls + O : o_ls) + tid; i < o_st; i += blockDim.x) slab[i] = 0.f;   // padding
    }
}

template <int KIND, int H>
__global__ __launch_bounds__(H * 2, 2) void ppo_grad_wide_teams_kernel(GradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (*a.stop_flag) return;
    constexpr int A = EnvSpec<KIND>::A;
    const bool actor = blockIdx.x < (unsigned)a.G;
    if (actor) grad_body_wide_teams<KIND, H, A, EnvSpec<KIND>::discrete ? HEAD_CATEGORICAL : HEAD_GAUSSIAN>(a, smem);
    else grad_body_wide_teams<KIND, H, 1, HEAD_VALUE>(a, smem);
}

}  // namespace dril
