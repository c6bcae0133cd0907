// dril_split_pieces.h — bf16 piece images of activation tiles in LDS (row reads + ds_read_b64_tr_b16 transposed reads) shared by ppo_grad_wide_split_kernel, ppo_grad_pair_kernel and build_wimg_split_kernel
#pragma once
#include "dril_internal.h"

namespace dril {

// (and only 8 chunks exist): bit 2 (the 64-byte window) from n bit 1, bits 0-1 from n bits 2-3.
template <int H> __device__ __forceinline__ int wimg_g(int n) { return H >= 128 ? (((n & 3) << 2) | ((n >> 2) & 3)) : ((((n >> 1) & 1) << 2) | ((n >> 2) & 3)); }

// pre-split fragment streams of one net: forward A[i][k] = kTanhScale W2[32mo + i][k], reverse A[i][k] = W2[k][32mo + i]; k = 32mi + 16s + 8(lane>>5) + j
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
// tm = tbase ^ (64 m) selects the m-tile (units 32 m ..), tm16 = tm ^ 16 the second half of the fragment (samples + 4 .. + 7: the row's chunk swizzle flips bit 0 with
// (n >> 2) & 1).  Both are passed in so that every ds_read_b64_tr_b16 of a phase is `register + immediate`: the offsets of (s, piece) are multiples of 2 048 bytes, which
// cannot carry into bit 4, so ((tm + off) ^ 16) == (tm ^ 16) + off — an identity the compiler cannot know, and without it every fragment half cost an add and a xor
// (60 VALU per wave and tile of ppo_grad_pair_kernel, round 3)
template <int H>
__device__ __forceinline__ bf16x8 load_frag_wide_T(const char* pimg, int tm, int tm16, int piece, int s) {
    constexpr int RB = 2 * H, PS = 32 * RB;
    const int off = 16 * s * RB + piece * PS;
    static_assert((16 * RB) % 32 == 0 && PS % 32 == 0, "offsets must leave bits 0-4 alone");
    return frag8(lds_read_tr16(pimg, tm + off), lds_read_tr16(pimg, tm16 + off + 4 * RB));
}

}  // namespace dril
