#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CTRL, int BANK = 0xf> __device__ __forceinline__ float dpp_mov(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, 0xf, BANK, false));
}
__device__ __forceinline__ void half_reduce16(const f32x16& x, int lane, float (&out)[4]) {
    const bool b0 = lane & 1, b1 = lane & 2;
    float y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float keep = b0 ? x[2 * i + 1] : x[2 * i], send = b0 ? x[2 * i] : x[2 * i + 1]; y[i] = keep + dpp_mov<0xB1>(0.f, send); }
    float z[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float keep = b1 ? y[2 * i + 1] : y[2 * i], send = b1 ? y[2 * i] : y[2 * i + 1]; z[i] = keep + dpp_mov<0x4E>(0.f, send); }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float t = dpp_mov<0x104, 0x5>(0.f, z[i]);          // row_shl:4 -> lanes of banks 0, 2 receive lane + 4
        t = dpp_mov<0x114, 0xa>(t, z[i]);                  // row_shr:4 -> lanes of banks 1, 3 receive lane - 4
        z[i] += t;
        z[i] += dpp_mov<0x128>(0.f, z[i]);                 // row_ror:8 = lane ^ 8
        out[i] = z[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] += __shfl_xor(out[i], 16);      // the other row of the half (four independent LDS-crossbar permutes)
}
__global__ void k(const float* in, float* o) {
    f32x16 x;
    for (int r = 0; r < 16; ++r) x[r] = in[r * 64 + threadIdx.x];
    float out[4];
    half_reduce16(x, threadIdx.x, out);
    for (int i = 0; i < 4; ++i) o[i * 64 + threadIdx.x] = out[i];
}
int main() {
    float hin[16 * 64], hout[4 * 64]; float *din, *dout;
    for (int i = 0; i < 16 * 64; ++i) hin[i] = (float)((i * 7919) % 1000) / 10.0f;
    hipMalloc(&din, sizeof(hin)); hipMalloc(&dout, sizeof(hout));
    hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice);
    k<<<1, 64>>>(din, dout);
    hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) for (int i = 0; i < 4; ++i) {
        const int r = 4 * i + (lane & 3), h = lane >> 5;
        float ref = 0; for (int c = 0; c < 32; ++c) ref += hin[r * 64 + 32 * h + c];
        if (fabsf(ref - hout[i * 64 + lane]) > 1e-2f) { if (bad < 5) printf("lane %d i %d got %f want %f\n", lane, i, hout[i * 64 + lane], ref); ++bad; }
    }
    printf("bad %d\n", bad);
    return bad != 0;
}
