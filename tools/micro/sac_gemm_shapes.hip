// microbenchmark (round 4): the contractions of SAC update! / collection one at a time, as launch_gemm / launch_gemm_multi run them, timed with HIP events over
// `reps` back-to-back launches; DRIL_GEMM_DBG ablation bits (16 no operand loads, 32 no MFMA, 64 no reduction / epilogue) say where a launch's time goes.
// build (on the GPU box): hipcc -O3 -std=c++17 --offload-arch=gfx950 -I dril.jl_amd/csrc -o tools/micro/sac_gemm_shapes tools/micro/sac_gemm_shapes.hip
#include "../../dril.jl_amd/csrc/dril_gemm.hip"
#include <cstdio>
#include <vector>
using namespace dril;
static float* dev(size_t n, float v) { float* p; hipMalloc(&p, n * 4); std::vector<float> h(n, v); for (size_t i = 0; i < n; ++i) h[i] = v * (float)((i * 2654435761u) % 1000) / 1000.f - v / 2; hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice); return p; }
template <class F> static float timeit(F&& f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) f();
    hipDeviceSynchronize(); hipEventRecord(a, nullptr);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b, nullptr); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b); return 1e3f * ms / reps;
}
int main(int argc, char** argv) {
    const int H = 512, B = argc > 1 ? atoi(argv[1]) : 256, E = 4096, reps = 200;
    float* W = dev((size_t)4 * (H * H + H), 0.1f); float* X = dev((size_t)4 * E * H, 1.0f); float* Y = dev((size_t)4 * E * H, 0.f); float* G = dev((size_t)4 * (H * (H + 1)), 0.f);
    // forward layer 2, Z nets: C[z] (H x B) = relu(W2[z] (H x H, column-major) . h1[z] (B rows of H) + b2)
    auto fwd = [&](int n, int Z) { GemmArgs g = gemm_args(); g.A = W; g.sAm = 1; g.sAk = H; g.zA = H * H + H; g.B = X; g.sBk = 1; g.sBn = H; g.zB = (long long)E * H;
        g.C = Y; g.sCm = 1; g.sCn = H; g.zC = (long long)E * H; g.bias = W + H * H; g.zBias = H * H + H; g.M = H; g.N = n; g.K = H; g.epi = EPI_RELU; return std::make_pair(g, Z); };
    // [dW2 | db2] = dz2 . [h1' | 1]  (contraction over samples)
    auto dw = [&](int n, int Z) { GemmArgs w = gemm_args(); w.A = Y; w.sAm = 1; w.sAk = H; w.zA = (long long)E * H; w.B = X; w.sBk = H; w.sBn = 1; w.zB = (long long)E * H; w.ones_n = 1;
        w.C = G; w.sCm = 1; w.sCn = H; w.zC = H * (H + 1); w.M = H; w.N = H + 1; w.K = n; return std::make_pair(w, Z); };
    // dz1 = (W2' dz2) .* relu'(h1)
    auto dz = [&](int n, int Z) { GemmArgs g = gemm_args(); g.A = W; g.sAm = H; g.sAk = 1; g.zA = H * H + H; g.B = Y; g.sBk = 1; g.sBn = H; g.zB = (long long)E * H;
        g.C = Y + (size_t)2 * E * H; g.sCm = 1; g.sCn = H; g.zC = (long long)E * H; g.aux = X; g.zAux = (long long)E * H; g.M = H; g.N = n; g.K = H; g.epi = EPI_MASK_RELU; return std::make_pair(g, Z); };
    struct Case { const char* name; std::pair<GemmArgs, int> c; double flops; };
    std::vector<Case> cases = {
        {"#2  actor L2 fwd   512 x 2B x 512, Z 1", fwd(2 * B, 1), 2.0 * H * 2 * B * H},
        {"#4  Q L2 fwd       512 x B x 512,  Z 4", fwd(B, 4), 4 * 2.0 * H * B * H},
        {"#9b Q L2 fwd       512 x B x 512,  Z 2", fwd(B, 2), 2 * 2.0 * H * B * H},
        {"#11 dz1            512 x B x 512,  Z 2", dz(B, 2), 2 * 2.0 * H * B * H},
        {"    dW2|db2        512 x 513 x B,  Z 2", dw(B, 2), 2 * 2.0 * H * (H + 1) * B},
        {"    dz1            512 x B x 512,  Z 1", dz(B, 1), 2.0 * H * B * H},
        {"    dW2|db2        512 x 513 x B,  Z 1", dw(B, 1), 2.0 * H * (H + 1) * B},
        {"col L2 fwd         512 x 4096 x 512", fwd(E, 1), 2.0 * H * E * H},
    };
    for (auto& cs : cases) {
        const float us = timeit([&] { launch_gemm(cs.c.first, cs.c.second, nullptr); }, reps);
        printf("%-44s %7.2f us  %6.1f TFLOP/s\n", cs.name, us, cs.flops / us * 1e-6);
    }
    {   // #6: [dW2 | db2] and dz1 of two critics in one launch (the tiny dW3 left out)
        auto a = dw(B, 2), b = dz(B, 2); GemmArgs gs[2] = {a.first, b.first}; int zs[2] = {2, 2};
        const float us = timeit([&] { launch_gemm_multi(gs, zs, 2, nullptr); }, reps);
        printf("%-44s %7.2f us  %6.1f TFLOP/s\n", "#6  multi: dW2|db2 + dz1, Z 2 each", us, (2 * 2.0 * H * (H + 1) * B + 2 * 2.0 * H * B * H) / us * 1e-6);
        auto a1 = dw(B, 1), b1 = dz(B, 1); GemmArgs g1[2] = {a1.first, b1.first}; int z1[2] = {1, 1};
        const float us1 = timeit([&] { launch_gemm_multi(g1, z1, 2, nullptr); }, reps);
        printf("%-44s %7.2f us  %6.1f TFLOP/s\n", "#13 multi: dW2|db2 + dz1, Z 1 each", us1, (2.0 * H * (H + 1) * B + 2.0 * H * B * H) / us1 * 1e-6);
    }
    return 0;
}
