// check of half_reduce16_lane (dril_device.h, round 3): the full reduce-scatter of 16 registers over the 32 lanes of each half-wave — lane l ends with the sum over its
// half of register (l & 15).  Variant 0 closes with __shfl_xor(.., 16), variant 1 with v_permlane16_swap.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CTRL, int BANK = 0xf> __device__ __forceinline__ float dpp_mov(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, 0xf, BANK, false));
}
template <int VAR> __device__ __forceinline__ float half_reduce16_lane(const f32x16& x, int lane) {
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
    float y[8], z[4], u[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float e = x[2 * i], o = x[2 * i + 1];
        asm volatile("" : "+v"(e), "+v"(o));
        const float keep = b0 ? o : e, send = b0 ? e : o;
        y[i] = keep + dpp_mov<0xB1>(0.f, send);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float keep = b1 ? y[2 * i + 1] : y[2 * i], send = b1 ? y[2 * i] : y[2 * i + 1]; z[i] = keep + dpp_mov<0x4E>(0.f, send); }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float keep = b2 ? z[2 * i + 1] : z[2 * i], send = b2 ? z[2 * i] : z[2 * i + 1];
        float t = dpp_mov<0x104, 0x5>(0.f, send);          // row_shl:4 -> the lanes of banks 0 and 2 receive lane + 4
        t = dpp_mov<0x114, 0xa>(t, send);                  // row_shr:4 -> the lanes of banks 1 and 3 receive lane - 4
        u[i] = keep + t;
    }
    const float keep = b3 ? u[1] : u[0], send = b3 ? u[0] : u[1];
    float v = keep + dpp_mov<0x128>(0.f, send);            // row_ror:8 = lane ^ 8
    if (VAR == 0) v += __shfl_xor(v, 16);
    else { const unsigned q = __float_as_uint(v); const auto r = __builtin_amdgcn_permlane16_swap(q, q, false, false); v = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
    return v;
}
template <int VAR> __global__ void k(const float* in, float* o) {
    f32x16 x;
    for (int r = 0; r < 16; ++r) x[r] = in[r * 64 + threadIdx.x];
    o[threadIdx.x] = half_reduce16_lane<VAR>(x, threadIdx.x);
}
int main() {
    float hin[16 * 64], hout[64]; float *din, *dout;
    for (int i = 0; i < 16 * 64; ++i) hin[i] = (float)((i * 7919) % 1000) / 10.0f;
    hipMalloc(&din, sizeof(hin)); hipMalloc(&dout, sizeof(hout));
    hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice);
    int total = 0;
    for (int var = 0; var < 2; ++var) {
        if (var == 0) k<0><<<1, 64>>>(din, dout); else k<1><<<1, 64>>>(din, dout);
        hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int lane = 0; lane < 64; ++lane) {
            const int r = lane & 15, h = lane >> 5;
            float ref = 0; for (int c = 0; c < 32; ++c) ref += hin[r * 64 + 32 * h + c];
            if (fabsf(ref - hout[lane]) > 1e-2f) { if (bad < 5) printf("var %d lane %d got %f want %f\n", var, lane, hout[lane], ref); ++bad; }
        }
        printf("variant %d: bad %d\n", var, bad); total += bad;
    }
    return total != 0;
}
