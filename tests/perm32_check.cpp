// CPU check driven by tests/test_abi_and_host.py::test_keyed_bijection_32bit_form: the 32-bit forms of the keyed DataLoader bijection (mix_bij32 / mix_bij_inv32,
// dril_device.h) give the identical permutation as the 64-bit forms for every bits <= 32.  The test cuts the functions out of dril_device.h into
// perm32_extract.inc (no HIP needed) and compiles this file with g++.
#include <cstdint>
#include <cstdio>
#include <initializer_list>
#define __host__
#define __device__
#define PERM_ONLY
namespace dril {
#include "perm32_extract.inc"
}
int main() {
    using namespace dril;
    const uint64_t keys[4] = {0x123456789abcdef0ull, 0xffffffffffffffffull, 0ull, 0xdeadbeefcafebabeull};
    long bad = 0;
    for (int bits = 1; bits <= 32; ++bits) for (uint64_t key : keys) {
        const uint64_t n = bits >= 32 ? 0x100000000ull : (1ull << bits);
        const uint64_t stride = n > 200000 ? n / 200000 : 1;
        for (uint64_t x = 0; x < n; x += stride) {
            const uint64_t y = mix_bij(x, key, bits);
            if (y != (uint64_t)mix_bij32((uint32_t)x, key, bits)) ++bad;
            if (mix_bij_inv(y, key, bits) != x || (uint64_t)mix_bij_inv32((uint32_t)y, key, bits) != x) ++bad;
        }
    }
    // ragged n (cycle walking) through perm_index / perm_position
    for (int64_t n : {1LL, 2LL, 3LL, 130LL, 8192LL, 100003LL}) {
        int b = 1; while (((int64_t)1 << b) < n) ++b;
        for (int64_t p = 0; p < n && p < 5000; ++p) { const int64_t i = perm_index(p, n, keys[0], b); if (i < 0 || i >= n || perm_position(i, n, keys[0], b) != p) ++bad; }
    }
    printf("mismatches %ld\n", bad);
    return bad != 0;
}
