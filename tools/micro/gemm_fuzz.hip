// fuzz of launch_gemm / launch_gemm_multi (round 4: the LDS-staged split-K body has two access patterns per operand, ragged tiles in m, n and k, the synthetic ones
// column, batch strides): random shapes and layouts against a float64 host reference.  Not part of the product or the suite.
// build + run (GPU box): hipcc -O3 -std=c++17 --offload-arch=gfx950 -I dril.jl_amd/csrc -Wno-unused-value -o /tmp/gemm_fuzz tools/micro/gemm_fuzz.hip && /tmp/gemm_fuzz 400
#include "../../dril.jl_amd/csrc/dril_gemm.hip"
#include <cstdio>
#include <random>
#include <vector>
#include <cmath>
using namespace dril;
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    std::mt19937 rng(argc > 2 ? atoi(argv[2]) : 1);
    auto ri = [&](int lo, int hi) { return (int)(rng() % (unsigned)(hi - lo + 1)) + lo; };
    int bad = 0, lds_cases = 0; double worst = 0;
    for (int it = 0; it < iters; ++it) {
        const int M = ri(1, 150), N = ri(1, 150), K = (ri(0, 3) ? 4 * ri(16, 180) : ri(64, 700)), Z = ri(1, 3);
        const bool a_kcontig = ri(0, 1), b_kcontig = ri(0, 1), ones = !b_kcontig && ri(0, 3) == 0, bias = ri(0, 1), relu = ri(0, 1);
        const int ldA = (a_kcontig ? K : M) + 4 * ri(0, 2), ldB = (b_kcontig ? K : N) + 4 * ri(0, 2);       // padded leading dimensions (multiples of 4 when K / the row count is)
        const int n_real = N - (ones ? 1 : 0);
        if (n_real < 1) continue;
        const size_t szA = (size_t)(a_kcontig ? M : K) * ldA, szB = (size_t)(b_kcontig ? n_real : K) * ldB, szC = (size_t)N * M;
        std::vector<float> hA(Z * szA), hB(Z * szB), hC(Z * szC, -7.f), hb(Z * M);
        for (auto& x : hA) x = (float)((int)(rng() % 2001) - 1000) / 500.f;
        for (auto& x : hB) x = (float)((int)(rng() % 2001) - 1000) / 500.f;
        for (auto& x : hb) x = (float)((int)(rng() % 2001) - 1000) / 500.f;
        float *dA, *dB, *dC, *db;
        hipMalloc(&dA, hA.size() * 4); hipMalloc(&dB, hB.size() * 4); hipMalloc(&dC, hC.size() * 4); hipMalloc(&db, hb.size() * 4);
        hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dC, hC.data(), hC.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
        GemmArgs g = gemm_args();
        g.A = dA; g.sAm = a_kcontig ? ldA : 1; g.sAk = a_kcontig ? 1 : ldA; g.zA = (long long)szA;
        g.B = dB; g.sBn = b_kcontig ? ldB : 1; g.sBk = b_kcontig ? 1 : ldB; g.zB = (long long)szB; g.ones_n = ones ? 1 : 0;
        g.C = dC; g.sCm = 1; g.sCn = M; g.zC = (long long)szC; g.bias = bias ? db : nullptr; g.zBias = M; g.M = M; g.N = N; g.K = K; g.epi = relu ? EPI_RELU : EPI_NONE;
        GemmArgs probe = g; gemm_prepare(probe); lds_cases += probe.use_lds;
        if (launch_gemm(g, Z, nullptr) != hipSuccess) { printf("launch failed M %d N %d K %d\n", M, N, K); return 1; }
        hipDeviceSynchronize();
        hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
        double err = 0, mag = 0;
        for (int z = 0; z < Z; ++z) for (int n = 0; n < N; ++n) for (int m = 0; m < M; ++m) {
            double s = 0;
            for (int k = 0; k < K; ++k) {
                const double a = a_kcontig ? hA[z * szA + (size_t)m * ldA + k] : hA[z * szA + (size_t)k * ldA + m];
                const double b = (ones && n == N - 1) ? 1.0 : (b_kcontig ? hB[z * szB + (size_t)n * ldB + k] : hB[z * szB + (size_t)k * ldB + n]);
                s += a * b;
            }
            if (bias) s += hb[z * M + m];
            if (relu) s = s > 0 ? s : 0;
            err = std::max(err, std::fabs(s - (double)hC[z * szC + (size_t)n * M + m])); mag = std::max(mag, std::fabs(s));
        }
        const double rel = err / std::max(1.0, mag);
        worst = std::max(worst, rel);
        if (!(rel < 2e-6)) { ++bad; printf("MISMATCH M %d N %d K %d Z %d a_k %d b_k %d ones %d ldsA %d ldsB %d use_lds %d: rel %.3e\n", M, N, K, Z, a_kcontig, b_kcontig, ones, probe.ldsA, probe.ldsB, probe.use_lds, rel); }
        hipFree(dA); hipFree(dB); hipFree(dC); hipFree(db);
    }
    printf("%d cases (%d on the LDS-staged body), %d mismatches, worst relative error %.2e\n", iters, lds_cases, bad, worst);
    return bad ? 1 : 0;
}
