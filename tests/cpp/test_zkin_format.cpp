// test_zkin_format.cpp -- CPU only: rebuilds a FRIProof from the VALUES of a reference-produced zkin file (handed over as a
// flat u64 blob by tests/test_cpp_shims.py) and writes it back through host/proof2zkinStark.hpp.  The python side compares
// the text with the reference's file byte for byte: key order, nesting, number -> decimal string.
// blob: nq, nSteps, polN, nEvals, nPublics, then per commitment tree t < 5: width, levels; then per FRI step tree i >= 1: width,
// levels; then root1..4 (16), evals (nEvals * 3), per step i >= 1: root (4), per query: vals, siblings (levels * 4);
// per query, per commitment tree: vals, siblings; finalPol (polN * 3); publics.
#include <cstdio>
#include <fstream>
#include <vector>
#include "proof2zkinStark.hpp"

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<uint64_t> b((std::istreambuf_iterator<char>(f)), {});
    f.close();
    std::ifstream g(argv[1], std::ios::binary | std::ios::ate);
    const size_t bytes = (size_t)g.tellg();
    g.seekg(0);
    std::vector<uint64_t> w(bytes / 8);
    g.read((char *)w.data(), bytes);
    size_t k = 0;
    auto next = [&]() { return w[k++]; };
    const uint64_t nq = next(), nSteps = next(), polN = next(), nEvals = next(), nPublics = next();
    uint64_t tw[5], tl[5];
    for (int t = 0; t < 5; t++) { tw[t] = next(); tl[t] = next(); }
    std::vector<uint64_t> sw(nSteps), sl(nSteps);
    for (uint64_t i = 1; i < nSteps; i++) { sw[i] = next(); sl[i] = next(); }
    FRIProof fp(polN, 3, nSteps, nEvals, nPublics);
    Goldilocks::Element *p = (Goldilocks::Element *)&w[k];
    for (int i = 0; i < 4; i++) { fp.proofs.root1[i] = p[i]; fp.proofs.root2[i] = p[4 + i]; fp.proofs.root3[i] = p[8 + i]; fp.proofs.root4[i] = p[12 + i]; }
    k += 16;
    fp.proofs.setEvals((Goldilocks::Element *)&w[k]);
    k += nEvals * 3;
    for (uint64_t i = 1; i < nSteps; i++) {
        fp.proofs.fri.trees[i].setRoot((Goldilocks::Element *)&w[k]);
        k += 4;
        for (uint64_t q = 0; q < nq; q++) {
            fp.proofs.fri.trees[i].polQueries.push_back({MerkleProof(sw[i], sl[i], (Goldilocks::Element *)&w[k])});
            k += sw[i] + sl[i] * 4;
        }
    }
    for (uint64_t q = 0; q < nq; q++) {
        std::vector<MerkleProof> v;
        for (int t = 0; t < 5; t++) {
            v.push_back(MerkleProof(tw[t], tl[t], (Goldilocks::Element *)&w[k]));
            k += tw[t] + tl[t] * 4;
        }
        fp.proofs.fri.trees[0].polQueries.push_back(v);
    }
    fp.proofs.fri.setPol((Goldilocks::Element *)&w[k]);
    k += polN * 3;
    for (uint64_t i = 0; i < nPublics; i++) fp.publics[i] = ((Goldilocks::Element *)&w[k])[i];
    k += nPublics;
    if (k != w.size()) { std::printf("blob size mismatch %zu vs %zu\n", k, w.size()); return 3; }
    std::ofstream(argv[2]) << proof2zkinStark(fp, true);
    return 0;
}
