// dril_split_pieces.h — f16 piece images of activation tiles in LDS (row reads + ds_read_b64_tr_b16 transposed reads) shared by ppo_grad_wide_split_kernel, ppo_grad_pair_kernel and build_wimg_split_kernel
#pragma once
#include "dril_internal.h"

namespace dril {

// chunk swizzle g(row) of the piece images: rows of >= 256 bytes (H >= 128) use four bits, 128-byte rows (H = 64; only 8 chunks exist) bit 2 (the 64-byte window) from n bit 1, bits 0-1 from n bits 2-3
template <int H> __device__ __forceinline__ int wimg_g(int n) { return H >= 128 ? (((n & 3) << 2) | ((n >> 2) & 3)) : ((((n >> 1) & 1) << 2) | ((n >> 2) & 3)); }

// ---- XOR form of the swizzled addresses (ppo_grad_pair_kernel, ppo_update_small_kernel; round 3).  When every image starts at a multiple of 512 bytes of LDS, bits 4-6
// of a swizzled address ARE the chunk field chunk ^ g(row), and stepping the chunk by a constant is an XOR of the whole address with that constant: one VALU per
// access, image and piece offsets in the instruction's immediate, instead of xor / shift / add / add.  Addresses are byte offsets into the dynamic LDS segment.
typedef __attribute__((address_space(3))) char lds_char;
template <class T> __device__ __forceinline__ T pl_read(const lds_char* lds, int byte) { return *reinterpret_cast<const __attribute__((address_space(3))) T*>(lds + byte); }
template <class T> __device__ __forceinline__ void pl_write(lds_char* lds, int byte, T v) { *reinterpret_cast<__attribute__((address_space(3))) T*>(lds + byte) = v; }
// the lane's own chunks of m-tile w of a pair image (two f16 pieces, at a + IMG and a + IMG + 4096): t = pair base + row + chunk 4w of the row + 8 (lane >> 5);
// IMG = byte offset of the image within the pair's block (ppo_grad_pair_kernel, ppo_update_small_kernel)
template <int IMG> __device__ __forceinline__ void pair_store_pieces2(lds_char* lds, int t, const f32x16& x) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        unsigned hi[2], lo[2];
        split2_pair(x[4 * g], x[4 * g + 1], hi[0], lo[0]); split2_pair(x[4 * g + 2], x[4 * g + 3], hi[1], lo[1]);
        const int a = t ^ (g << 4);
        pl_write(lds, a + IMG, u32x2{hi[0], hi[1]}); pl_write(lds, a + IMG + 4096, u32x2{lo[0], lo[1]});
    }
}
// (float)hi.half + (float)lo.half of two packed f16 pairs in ONE instruction per element: v_fma_mix_f32 takes f16 sources from either half of a register (the compiler
// emits two v_cvt_f32_f16 and an add — 48 instead of 16 vector instructions per 16-register tile; it has no pattern from `fpext + fpext` to the mixed form)
__device__ __forceinline__ void pieces_sum2(unsigned hk, unsigned lk, float& x0, float& x1) {
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(x0) : "v"(hk), "v"(lk));
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(x1) : "v"(hk), "v"(lk));
}
template <int IMG> __device__ __forceinline__ void pair_load_pieces2(const lds_char* lds, int t, f32x16& x) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int a = t ^ (g << 4);
        const u32x2 hi = pl_read<u32x2>(lds, a + IMG), lo = pl_read<u32x2>(lds, a + IMG + 4096);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const unsigned hk = k ? hi.y : hi.x, lk = k ? lo.y : lo.x;                         // (NOT __builtin_bit_cast(f16x2_t, hi[k]): hipcc 7.2 folds that to element 0 for every k)
            float x0, x1; pieces_sum2(hk, lk, x0, x1);                                         // the value that was split, to 2^-24 relative
            x[4 * g + 2 * k] = x0; x[4 * g + 2 * k + 1] = x1;
        }
    }
}
// both half-waves get v(lower half) + v(upper half), in that order: one v_permlane32_swap instead of an LDS-crossbar permute (ds_bpermute + its address + its wait)
__device__ __forceinline__ float both_halves_sum(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// two-piece f16 forms (ppo_grad_wide_split_kernel since the end of round 3; dril_device.h "the same on f16 pieces"): pieces at a and PS + a
template <int H>
__device__ __forceinline__ void store_tile_pieces2(char* pimg, int w, const f32x16& x, int lane) {
    constexpr int RB = 2 * H, PS = 32 * RB;
    const int c = lane & 31, h = lane >> 5, rowb = c * RB + 8 * h, gsw = wimg_g<H>(c);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        unsigned hi[2], lo[2];
        split2_pair(x[4 * g], x[4 * g + 1], hi[0], lo[0]); split2_pair(x[4 * g + 2], x[4 * g + 3], hi[1], lo[1]);
        const int a = rowb + (((4 * w + g) ^ gsw) << 4);
        *reinterpret_cast<u32x2*>(pimg + a) = u32x2{hi[0], hi[1]}; *reinterpret_cast<u32x2*>(pimg + PS + a) = u32x2{lo[0], lo[1]};
    }
}
template <int H>
__device__ __forceinline__ void load_tile_pieces2(const char* pimg, int w, f32x16& x, int lane) {
    constexpr int RB = 2 * H, PS = 32 * RB;
    const int c = lane & 31, h = lane >> 5, rowb = c * RB + 8 * h, gsw = wimg_g<H>(c);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int a = rowb + (((4 * w + g) ^ gsw) << 4);
        const u32x2 hi = *reinterpret_cast<const u32x2*>(pimg + a), lo = *reinterpret_cast<const u32x2*>(pimg + PS + a);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const unsigned hk = k ? hi.y : hi.x, lk = k ? lo.y : lo.x;                         // (not __builtin_bit_cast(f16x2_t, hi[k]): see pair_load_pieces2)
            float x0, x1; pieces_sum2(hk, lk, x0, x1);
            x[4 * g + 2 * k] = x0; x[4 * g + 2 * k + 1] = x1;
        }
    }
}
// the tile back in the TRANSPOSED accumulator layout (lane = unit 32m + (lane & 31), register r = sample rowfn(r, lane >> 5): what a product with its two operands exchanged
// leaves): register group Q (r = 4Q .. 4Q + 3) = samples 8Q + 4h + {0..3} of the lane's unit = ONE ds_read_b64_tr_b16 per piece, whose 16-lane group addresses rows
// (samples) 8Q + 4h + q, q = e >> 2, and columns (units) 16 gm + 4 (e & 3) .. + 3.  g(8Q + 4h + q) = (q << 2) | ((2Q + h) & 3) = g(4h + q) ^ (2Q & 3): Q enters the address
// as `+ 8 Q RB` (an immediate) and, for odd Q, `^ 32` — two base registers (tm, tm ^ 32) for the four groups.  tm = wide_trm_base ^ (64 m).  H >= 128 (the four-bit swizzle)
template <int H>
__device__ __forceinline__ int wide_trm_base(int lane) {
    static_assert(H >= 128, "wide_trm_base: the four-bit chunk swizzle of rows >= 256 bytes");
    constexpr int RB = 2 * H;
    const int h = lane >> 5, gm = (lane >> 4) & 1, e = lane & 15, q = e >> 2, p = e & 3, n = 4 * h + q;
    return n * RB + ((((2 * gm + (p >> 1)) ^ wimg_g<H>(n)) & 15) << 4) + 8 * (p & 1);
}
template <int H>
__device__ __forceinline__ void load_tile_pieces2_T(const char* pimg, int tm, f32x16& x) {
    constexpr int RB = 2 * H, PS = 32 * RB;
    static_assert((8 * RB) % 64 == 0 && PS % 64 == 0, "offsets must leave bits 0-5 alone");
    const int tm32 = tm ^ 32;
#pragma unroll
    for (int Q = 0; Q < 4; ++Q) {
        const int a = ((Q & 1) ? tm32 : tm) + 8 * Q * RB;
        const u32x2 hi = __builtin_bit_cast(u32x2, lds_read_tr16(pimg, a)), lo = __builtin_bit_cast(u32x2, lds_read_tr16(pimg, a + PS));
        float x0, x1, x2, x3;
        pieces_sum2(hi.x, lo.x, x0, x1); pieces_sum2(hi.y, lo.y, x2, x3);                        // the value that was split, to 2^-24 relative
        x[4 * Q] = x0; x[4 * Q + 1] = x1; x[4 * Q + 2] = x2; x[4 * Q + 3] = x3;
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
