// test_ref_polinomial.cpp -- only where the reference tree is present: the reference's OWN polinomial.hpp (header-only, on the include
// path ahead of host/polinomial.hpp) compiled against the Level-0 field classes, and its lookup / grand-product bodies
//     Polinomial::calculateH1H2_opt1 / _opt3 / calculateH1H2_   (polinomial.hpp:303-584; starks.cpp:106-124 calls them)
//     Polinomial::calculateZ                                     (polinomial.hpp:586-607; starks.cpp:179-185)
// and ZhInv::ZhInv (zhInv.cpp:7-31), run beside the CPU oracle's restatements (glo_calculate_h1h2, glo_calculate_z, glo_zhinv) on the same data.  The GPU kernels are tested against
// the oracle bit for bit (tests/test_lookup.py); this is what says the oracle -- and through it the kernels -- follow the reference's
// conventions: which of equal table rows takes the repeats, the order of h1 / h2, z[0] = 1 and the running quotient.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "polinomial.hpp"
#include "zhInv.hpp"
#include "merkleTreeGL.hpp"
#include "../../oracle/gl_oracle.h"

// (zkassert.hpp brings the reference's exit_process.hpp, declarations only: the function comes from the stand-in header, the flag from here)
bool bExitingProcess = false;

static std::mt19937_64 rng(77);
static uint64_t fe() { return rng() % GOLDILOCKS_PRIME; }

static int lookups()
{
    int bad = 0;
    const uint64_t sizes[] = {8, 64, 1024, 1 << 14};
    for (uint64_t n : sizes)
        for (int dim : {1, 3})
            for (int shape = 0; shape < 3; shape++) { // 0: all rows of t distinct; 1: runs of equal rows; 2: one heavy value + padding
                Polinomial t(n, dim), f(n, dim), h1(n, dim), h2(n, dim);
                for (uint64_t i = 0; i < n; i++) {
                    const uint64_t src = shape == 0 ? i : shape == 1 ? i - i % 4 : (i < n / 2 ? i : n / 2);
                    for (int d = 0; d < dim; d++) t[i][d].fe = src == i ? fe() : t[src][d].fe;
                }
                for (uint64_t i = 0; i < n; i++) {
                    const uint64_t r = shape == 2 && (rng() & 1) ? n - 1 : rng() % n;
                    for (int d = 0; d < dim; d++) f[i][d] = t[r][d];
                }
                // the buffers as starks.cpp:101-122 sizes them
                std::vector<uint64_t> buffer(8 * n);
                const uint64_t values = (dim == 1 ? 3 : 5) * n, keys = 8 * n - values;
                if (dim == 1) Polinomial::calculateH1H2_opt1(h1, h2, f, t, 0, buffer.data(), keys, values);
                else Polinomial::calculateH1H2_opt3(h1, h2, f, t, 0, buffer.data(), keys, values);
                std::vector<uint64_t> o1(n * dim), o2(n * dim);
                const int64_t rc = glo_calculate_h1h2(o1.data(), dim, o2.data(), dim, (const uint64_t *)f.address(), dim, (const uint64_t *)t.address(), dim, dim, n);
                uint64_t diff = rc != 0;
                for (uint64_t i = 0; i < n * (uint64_t)dim; i++) diff += o1[i] != Goldilocks::toU64(h1.address()[i]) || o2[i] != Goldilocks::toU64(h2.address()[i]);
                // the plain form (polinomial.hpp:303-347) must agree with the optimised ones as well
                Polinomial g1(n, dim), g2(n, dim);
                Polinomial::calculateH1H2_(g1, g2, f, t, 0);
                for (uint64_t i = 0; i < n * (uint64_t)dim; i++) diff += o1[i] != Goldilocks::toU64(g1.address()[i]) || o2[i] != Goldilocks::toU64(g2.address()[i]);
                std::printf("h1h2 n=%llu dim=%d shape=%d: %llu differences\n", (unsigned long long)n, dim, shape, (unsigned long long)diff);
                bad += diff != 0;
            }
    return bad;
}

static int products()
{
    int bad = 0;
    const uint64_t sizes[] = {1, 2, 9, 1000, 1 << 14};
    for (uint64_t n : sizes)
        for (int closing = 0; closing < 2; closing++) {
            if (n == 1 && !closing) continue; // (calculateZ asserts that the product closes)
            Polinomial num(n, 3), den(n, 3), z(n, 3);
            for (uint64_t i = 0; i < n; i++)
                for (int d = 0; d < 3; d++) { num[i][d].fe = fe(); den[i][d].fe = fe(); }
            if (closing) { // the numerators are a rotation of the denominators
                for (uint64_t i = 0; i < n; i++)
                    for (int d = 0; d < 3; d++) num[i][d] = den[(i + 1) % n][d];
            }
            std::vector<uint64_t> oz(n * 3);
            const int closes = glo_calculate_z(oz.data(), 3, (const uint64_t *)num.address(), 3, (const uint64_t *)den.address(), 3, n);
            uint64_t diff = closes != closing;
            if (closing) { // the reference's body ends in zkassert(product == 1): only closing products can run through it
                Polinomial::calculateZ(z, num, den);
                for (uint64_t i = 0; i < n * 3; i++) diff += oz[i] != Goldilocks::toU64(z.address()[i]);
            } else {       // z itself without the final assertion: the same loop (polinomial.hpp:590-600)
                Polinomial denI(n, 3);
                Polinomial::batchInverse(denI, den);
                Goldilocks3::copy((Goldilocks3::Element *)z[0], &Goldilocks3::one());
                for (uint64_t i = 1; i < n; i++) {
                    Polinomial tmp(1, 3);
                    Polinomial::mulElement(tmp, 0, num, i - 1, denI, i - 1);
                    Polinomial::mulElement(z, i, z, i - 1, tmp, 0);
                }
                for (uint64_t i = 0; i < n * 3; i++) diff += oz[i] != Goldilocks::toU64(z.address()[i]);
            }
            std::printf("z n=%llu closing=%d: %llu differences\n", (unsigned long long)n, closing, (unsigned long long)diff);
            bad += diff != 0;
        }
    return bad;
}

// ZhInv::ZhInv (zhInv.cpp:7-31, the reference's translation unit, linked in) beside glo_zhinv
static int zhinvs()
{
    int bad = 0;
    const unsigned cases[][2] = {{3, 4}, {10, 11}, {17, 20}, {20, 22}, {23, 24}, {5, 13}};
    for (auto &c : cases) {
        ZhInv zi(c[0], c[1]);
        std::vector<uint64_t> o(1ULL << (c[1] - c[0]));
        glo_zhinv(o.data(), c[0], c[1]);
        uint64_t diff = 0;
        for (uint64_t i = 0; i < 3 * o.size(); i++) diff += o[i % o.size()] != Goldilocks::toU64(zi.zhInv(i)); // (periodic: zhInv.hpp:22-25)
        std::printf("zhinv nBits=%u nBitsExt=%u: %llu differences\n", c[0], c[1], (unsigned long long)diff);
        bad += diff != 0;
    }
    return bad;
}

// MerkleTreeGL::getGroupProof (merkleTreeGL.cpp:12-35, the reference's translation unit) over a node array the oracle built, beside
// glo_merkle_group_proof: the proof layout (row, then the sibling of every level, level-major node array)
static int openings()
{
    int bad = 0;
    const uint64_t shapes[][2] = {{2, 5}, {8, 1}, {64, 18}, {1024, 39}, {4096, 3}};
    for (auto &sh : shapes) {
        const uint64_t h = sh[0], w = sh[1];
        MerkleTreeGL tree(h, w, NULL);
        for (uint64_t i = 0; i < h * w; i++) tree.source[i].fe = fe();
        glo_merkletree((uint64_t *)tree.nodes, (const uint64_t *)tree.source, w, h);
        const uint64_t plen = w + tree.MerkleProofSize() * HASH_SIZE;
        uint64_t diff = 0;
        for (uint64_t q = 0; q < 16; q++) {
            const uint64_t idx = q == 0 ? 0 : q == 1 ? h - 1 : rng() % h;
            std::vector<Goldilocks::Element> p(plen);
            std::vector<uint64_t> o(plen);
            tree.getGroupProof(p.data(), idx);
            glo_merkle_group_proof(o.data(), (const uint64_t *)tree.nodes, (const uint64_t *)tree.source, h, w, idx);
            for (uint64_t i = 0; i < plen; i++) diff += o[i] != p[i].fe;
        }
        std::printf("openings height=%llu width=%llu: %llu differences\n", (unsigned long long)h, (unsigned long long)w, (unsigned long long)diff);
        bad += diff != 0;
    }
    return bad;
}

int main()
{
    const int bad = lookups() + products() + zhinvs() + openings();
    std::printf(bad ? "FAIL\n" : "OK\n");
    return bad ? 1 : 0;
}
