// friProve.hpp -- FRIProve::prove with the reference's control flow (friProve.cpp:5-190) on device-resident
// data: the FRI polynomial stays in HBM across steps (fold -> transpose -> step tree), only step roots, the
// final polynomial and the query openings come back.  StarkInfo carries just the fields this path reads
// (stark_info.hpp:21-35); the proof container is friProof.hpp.
#ifndef FRI_PROVE
#define FRI_PROVE
#include <utility>
#include <vector>
#include "transcript.hpp"
#include "polinomial.hpp"
#include "merkleTreeGL.hpp"
#include "ntt_goldilocks.hpp"
#include "friProof.hpp"
#include "stark_info.hpp"

class FRIProve
{
public:
    // treesGL: the 5 commitment trees opened at step 0 (null entries are skipped); friPol: 2^polBits cubic-extension
    // elements in HOST memory (as in the reference) -- it is uploaded once and folded in HBM.
    static void prove(FRIProof &fproof, MerkleTreeGL **treesGL, Transcript transcript, Polinomial &friPol, uint64_t polBits, StarkInfo starkInfo,
                      uint64_t nTrees = 5)
    {
        proveImpl(fproof, treesGL, transcript, &friPol, nullptr, polBits, starkInfo, nTrees);
    }
    // the same with the FRI polynomial already in HBM (f_2ns as step52ns leaves it in the image of host/starks.hpp: 2^polBits x 3, read only): no
    // upload, and nothing is written back but the proof
    static void prove(FRIProof &fproof, MerkleTreeGL **treesGL, Transcript transcript, const uint64_t *d_friPol, uint64_t polBits, StarkInfo starkInfo,
                      uint64_t nTrees = 5)
    {
        proveImpl(fproof, treesGL, transcript, nullptr, d_friPol, polBits, starkInfo, nTrees);
    }

private:
    static void proveImpl(FRIProof &fproof, MerkleTreeGL **treesGL, Transcript &transcript, Polinomial *friPol, const uint64_t *d_friPol, uint64_t polBits,
                          StarkInfo &starkInfo, uint64_t nTrees)
    {
        mi_ctx *c = mi::ctx();
        const std::vector<StepStruct> &steps = starkInfo.starkStruct.steps;
        const uint64_t nBitsExt = polBits;
        uint64_t *d_pol = mi::devAlloc(3ULL << polBits, "FRIProve::prove (alloc)"), *d_next = mi::devAlloc(3ULL << polBits, "FRIProve::prove (alloc)");
        uint64_t *d_aux = mi::devAlloc(3ULL << polBits, "FRIProve::prove (alloc)");
        if (friPol) mi::check(mi_copy_h2d(c, d_pol, friPol->address(), (3ULL << polBits) * 8), "FRIProve::prove (h2d)");
        else mi::check(mi_copy_2d_dev(c, d_pol, 3, d_friPol, 3, 1ULL << polBits, 3), "FRIProve::prove (device copy)");
        std::vector<MerkleTreeGL *> treesFRIGL(steps.size(), nullptr);
        std::vector<uint64_t *> owned_sources; // device leaves of the step trees, released after the query phase
        uint64_t pol2N = 0;
        for (uint64_t si = 0; si < steps.size(); si++) {
            const uint64_t curBits = steps[si].nBits;
            pol2N = 1ULL << curBits;
            Goldilocks::Element special_x[FIELD_EXTENSION];
            transcript.getField(special_x); // friProve.cpp:30
            mi::check(mi_fri_fold_dev(c, d_next, d_pol, (unsigned)polBits, (unsigned)curBits, (unsigned)nBitsExt, (const uint64_t *)special_x),
                      "FRIProve::prove (fold)"); // friProve.cpp:44-108 (a copy when si == 0)
            if (si < steps.size() - 1) { // friProve.cpp:110-126
                const uint64_t nGroups = 1ULL << steps[si + 1].nBits, groupSize = pol2N / nGroups;
                mi::check(mi_fri_transpose_dev(c, d_aux, d_next, pol2N, (unsigned)steps[si + 1].nBits), "FRIProve::prove (transpose)");
                MerkleTreeGL *t = new MerkleTreeGL(nGroups, groupSize * FIELD_EXTENSION, NULL);
                uint64_t *d_src = mi::devAlloc(pol2N * 3, "FRIProve::prove (alloc tree source)"); // the tree keeps its own leaves for the query phase
                mi::check(mi_copy_2d_dev(c, d_src, groupSize * 3, d_aux, groupSize * 3, nGroups, groupSize * 3), "FRIProve::prove (copySource)");
                t->setDeviceSource(d_src);
                t->merkelize();
                Goldilocks::Element rootGL[HASH_SIZE];
                t->getRoot(rootGL);
                transcript.put(rootGL, HASH_SIZE);
                fproof.proofs.fri.trees[si + 1].setRoot(rootGL);
                treesFRIGL[si + 1] = t;
                owned_sources.push_back(d_src);
            } else { // friProve.cpp:128-134: the last polynomial goes into the transcript and the proof
                std::vector<Goldilocks::Element> last(pol2N * 3);
                mi::check(mi_copy_d2h(c, last.data(), d_next, pol2N * 3 * 8), "FRIProve::prove (final pol)");
                transcript.put(&last[0], pol2N * FIELD_EXTENSION); // friProve.cpp:131-134 element by element: the same elements in the same order
                fproof.proofs.fri.setPol(last.data());
                // the reference copies every step's folded polynomial over the head of friPol (friProve.cpp:136-140); a
                // caller can only observe the final state, of which the head -- the last polynomial -- is reproduced
                if (friPol) std::memcpy(friPol->address(), last.data(), pol2N * 3 * 8);
            }
            std::swap(d_pol, d_next);
            polBits = curBits;
        }
        // ---- query phase (friProve.cpp:155-187)
        std::vector<uint64_t> ys(starkInfo.starkStruct.nQueries);
        transcript.getPermutations(ys.data(), starkInfo.starkStruct.nQueries, steps[0].nBits);
        for (uint64_t si = 0; si < steps.size(); si++) {
            if (si == 0) {
                std::vector<std::vector<Goldilocks::Element>> buffs(nTrees);
                for (uint64_t t = 0; t < nTrees; t++) {
                    if (!treesGL[t]) continue;
                    const uint64_t stride = treesGL[t]->width + treesGL[t]->MerkleProofSize() * HASH_SIZE;
                    buffs[t].resize(stride * ys.size());
                    treesGL[t]->getGroupProofs(buffs[t].data(), ys.data(), ys.size());
                }
                for (uint64_t i = 0; i < ys.size(); i++) {
                    std::vector<MerkleProof> vMkProof;
                    vMkProof.reserve(nTrees);
                    for (uint64_t t = 0; t < nTrees; t++) {
                        if (!treesGL[t]) continue;
                        const uint64_t stride = treesGL[t]->width + treesGL[t]->MerkleProofSize() * HASH_SIZE;
                        vMkProof.push_back(MerkleProof(treesGL[t]->width, treesGL[t]->MerkleProofSize(), &buffs[t][i * stride]));
                    }
                    fproof.proofs.fri.trees[0].polQueries.push_back(std::move(vMkProof)); // (moved: a MerkleProof holds one small vector per opened value, as the reference's does)
                }
            } else {
                MerkleTreeGL *t = treesFRIGL[si];
                const uint64_t stride = t->width + t->MerkleProofSize() * HASH_SIZE;
                std::vector<Goldilocks::Element> buff(stride * ys.size());
                t->getGroupProofs(buff.data(), ys.data(), ys.size());
                for (uint64_t i = 0; i < ys.size(); i++) {
                    std::vector<MerkleProof> vMkProof;
                    vMkProof.push_back(MerkleProof(t->width, t->MerkleProofSize(), &buff[i * stride]));
                    fproof.proofs.fri.trees[si].polQueries.push_back(std::move(vMkProof)); // (moved: a MerkleProof holds one small vector per opened value, as the reference's does)
                }
            }
            if (si < steps.size() - 1)
                for (uint64_t i = 0; i < ys.size(); i++) ys[i] = ys[i] % (1ULL << steps[si + 1].nBits);
        }
        for (MerkleTreeGL *t : treesFRIGL) delete t;
        for (uint64_t *p : owned_sources) mi::devFree(p);
        mi::devFree(d_pol); mi::devFree(d_next); mi::devFree(d_aux);
    }
};
#endif
