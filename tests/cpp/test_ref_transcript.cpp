// test_ref_transcript.cpp -- only where the reference tree is present: the reference's transcript.{hpp,cpp}, compiled unchanged over a
// CPU permutation (tests/cpp/cpu_poseidon/poseidon_goldilocks.hpp: the oracle's), driven through random interleavings of put /
// getField / getFields1 / getPermutations beside the oracle's restatement of the same state machine (glo_transcript_*): every value
// either hands out must agree.  SURVEY 8(c): the golden proofs cannot replay a transcript (the verification key that goes into it is
// absent), so this is what pins Transcript's buffering, cursor and bit-extraction rules.
#include <cstdio>
#include <random>
#include <vector>
#include "transcript.hpp"

int main()
{
    std::mt19937_64 rng(4242);
    uint64_t diff = 0, handed = 0;
    for (int trial = 0; trial < 40; trial++) {
        Transcript ref;
        glo_transcript ora;
        glo_transcript_init(&ora);
        for (int step = 0; step < 60; step++) {
            switch (rng() % 5) {
            case 0: case 1: { // put 0 .. 40 elements (4 = a root, 3 = an evaluation, 8 / 16 = exactly full buffers, 192 = a last polynomial)
                static const uint64_t sizes[] = {0, 1, 3, 4, 7, 8, 9, 16, 24, 40, 192};
                const uint64_t n = sizes[rng() % 11];
                std::vector<Goldilocks::Element> in(n + 1);
                for (auto &e : in) e.fe = rng() % GOLDILOCKS_PRIME;
                ref.put(in.data(), n);
                glo_transcript_put(&ora, (const uint64_t *)in.data(), n);
                break;
            }
            case 2: {
                Goldilocks::Element a[3];
                uint64_t b[3];
                ref.getField(a);
                glo_transcript_get_field(&ora, b);
                for (int d = 0; d < 3; d++) { diff += Goldilocks::toU64(a[d]) != b[d]; handed++; }
                break;
            }
            case 3: {
                diff += Goldilocks::toU64(ref.getFields1()) != glo_transcript_get_fields1(&ora);
                handed++;
                break;
            }
            default: { // query indices: (43, 20) the recursive STARKs', (128, 24) the zkEVM's, and sizes that straddle the 63-bit field boundary;
                       // at most 31 bits an index: the reference forms it with `1 << j` on an int (transcript.cpp:77), the oracle and the product with 1ULL << j
                static const uint64_t qs[][2] = {{43, 20}, {128, 24}, {1, 1}, {3, 21}, {9, 7}, {63, 31}, {5, 30}};
                const uint64_t *q = qs[rng() % 7];
                std::vector<uint64_t> a(q[0]), b(q[0]);
                ref.getPermutations(a.data(), q[0], q[1]);
                glo_transcript_get_permutations(&ora, b.data(), q[0], q[1]);
                for (uint64_t i = 0; i < q[0]; i++) { diff += a[i] != b[i]; handed++; }
                break;
            }
            }
        }
    }
    std::printf("transcript: %llu values handed out, %llu differences\n", (unsigned long long)handed, (unsigned long long)diff);
    std::printf(diff || !handed ? "FAIL\n" : "OK\n");
    return diff || !handed ? 1 : 0;
}
