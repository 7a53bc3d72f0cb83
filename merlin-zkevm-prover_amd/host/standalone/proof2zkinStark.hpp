// proof2zkinStark.hpp -- the zkin layout the recursive verifier circuits consume (src/starkpil/fri/proof2zkinStark.cpp:8-82),
// written straight from the FRIProof container: root1..4, evals, s{i}_root / s{i}_vals / s{i}_siblings for the FRI step
// trees, s0_vals{1,2,3,4,C} / s0_siblings{...} for the five commitment trees (2 and 3 only when that stage has
// columns), finalPol, and optionally publics.  The reference's golden files
// (testvectors/aggregatedProof/recursive1.zkin.proof_*.json) have exactly this shape.
#ifndef PROOF2ZKIN__STARK_HPP
#define PROOF2ZKIN__STARK_HPP
#include "friProof.hpp"

inline std::string proof2zkinStark(const FRIProof &fproof, bool withPublics = false)
{
    const Proofs &p = fproof.proofs;
    const Fri &fri = p.fri;
    std::string o = "{\"root1\":" + mi_json::arr(p.root1) + ",\"root2\":" + mi_json::arr(p.root2) + ",\"root3\":" + mi_json::arr(p.root3) +
                    ",\"root4\":" + mi_json::arr(p.root4) + ",\"evals\":" + p.evalsJson();
    const size_t nq = fri.trees[0].polQueries.size();
    for (size_t i = 1; i < fri.trees.size(); i++) { // proof2zkinStark.cpp:18-28
        std::vector<std::string> vals, sibs;
        for (size_t q = 0; q < nq; q++) {
            vals.push_back(fri.trees[i].polQueries[q][0].valuesJson());
            sibs.push_back(fri.trees[i].polQueries[q][0].siblingsJson());
        }
        const std::string s = "s" + std::to_string(i);
        o += ",\"" + s + "_root\":" + mi_json::arr(fri.trees[i].root) + ",\"" + s + "_vals\":" + mi_json::arr(vals) + ",\"" + s +
             "_siblings\":" + mi_json::arr(sibs);
    }
    // step-0 openings: one MerkleProof per commitment tree and query (proof2zkinStark.cpp:30-77).  Key order as the
    // reference's ordered_json produces it: every s0_vals<t> first, then every s0_siblings<t> (trees 2 and 3 only when that
    // stage has columns)
    static const char *names[5] = {"1", "2", "3", "4", "C"};
    const size_t nTrees = nq ? fri.trees[0].polQueries[0].size() : 0;
    for (int pass = 0; pass < 2; pass++)
        for (size_t t = 0; t < nTrees; t++) {
            if ((t == 1 || t == 2) && fri.trees[0].polQueries[0][t].v.empty()) continue; // stage without columns
            std::vector<std::string> items;
            for (size_t q = 0; q < nq; q++)
                items.push_back(pass == 0 ? fri.trees[0].polQueries[q][t].valuesJson() : fri.trees[0].polQueries[q][t].siblingsJson());
            const char *nm = nTrees == 5 ? names[t] : names[t < 4 ? t : 4];
            o += std::string(pass == 0 ? ",\"s0_vals" : ",\"s0_siblings") + nm + "\":" + mi_json::arr(items);
        }
    o += ",\"finalPol\":" + fri.polJson();
    if (withPublics) o += ",\"publics\":" + mi_json::arr(fproof.publics);
    return o + "}";
}
#endif
