// friProof.hpp -- proof container with the reference's shape (src/starkpil/fri/friProof.hpp:9-236): MerkleProof,
// ProofTree, Fri, Proofs{root1..4, evals, fri}, FRIProof{proofs, publics}.  The JSON writers produce the same
// layout as the reference's nlohmann::ordered_json output (field elements as decimal strings), as text.
#ifndef FRI_PROOF
#define FRI_PROOF
#include <cstring>
#include <string>
#include <vector>
#include "goldilocks_base_field.hpp"
#include "goldilocks_cubic_extension.hpp"
#include "merklehash_goldilocks.hpp"

namespace mi_json {
inline std::string str(const Goldilocks::Element &e) { return "\"" + Goldilocks::toString(e) + "\""; }
inline std::string arr(const std::vector<std::string> &items)
{
    std::string s = "[";
    for (size_t i = 0; i < items.size(); i++) s += (i ? "," : "") + items[i];
    return s + "]";
}
inline std::string arr(const std::vector<Goldilocks::Element> &v)
{
    std::vector<std::string> it;
    for (const auto &e : v) it.push_back(str(e));
    return arr(it);
}
} // namespace mi_json

class MerkleProof
{
public:
    std::vector<std::vector<Goldilocks::Element>> v;
    std::vector<std::vector<Goldilocks::Element>> mp;
    MerkleProof(uint64_t nLinears, uint64_t elementsTree, Goldilocks::Element *pointer)
        : v(nLinears, std::vector<Goldilocks::Element>(1, Goldilocks::zero())), mp(elementsTree, std::vector<Goldilocks::Element>(HASH_SIZE, Goldilocks::zero()))
    {
        for (uint64_t i = 0; i < nLinears; i++) v[i][0] = pointer[i];
        for (uint64_t j = 0; j < elementsTree; j++) std::memcpy(&mp[j][0], &pointer[nLinears + j * HASH_SIZE], HASH_SIZE * sizeof(Goldilocks::Element));
    };
    std::string valuesJson() const // friProof.hpp:31-48
    {
        std::vector<std::string> it;
        for (const auto &e : v) it.push_back(e.size() > 1 ? mi_json::arr(e) : mi_json::str(e[0]));
        return mi_json::arr(it);
    }
    std::string siblingsJson() const // friProof.hpp:50-60
    {
        std::vector<std::string> it;
        for (const auto &e : mp) it.push_back(mi_json::arr(e));
        return mi_json::arr(it);
    }
    std::string merkleProof2json() const { return "[" + valuesJson() + "," + siblingsJson() + "]"; }
};

class ProofTree
{
public:
    std::vector<Goldilocks::Element> root;
    std::vector<std::vector<MerkleProof>> polQueries;
    ProofTree() : root(HASH_SIZE){};
    void setRoot(Goldilocks::Element *_root) { std::memcpy(&root[0], _root, HASH_SIZE * sizeof(Goldilocks::Element)); };
    std::string ProofTree2json() const // friProof.hpp:77-110
    {
        std::vector<std::string> q;
        for (const auto &pq : polQueries) {
            if (pq.size() != 1) {
                std::vector<std::string> e;
                for (const auto &m : pq) e.push_back(m.merkleProof2json());
                q.push_back(mi_json::arr(e));
            } else {
                q.push_back(pq[0].merkleProof2json());
            }
        }
        return "{\"root\":" + mi_json::arr(root) + ",\"polQueries\":" + mi_json::arr(q) + "}";
    }
};

class Fri
{
public:
    std::vector<std::vector<Goldilocks::Element>> pol;
    std::vector<ProofTree> trees;
    Fri(uint64_t polN, uint64_t dim, uint64_t numSteps) : pol(polN, std::vector<Goldilocks::Element>(dim, Goldilocks::zero())), trees(numSteps){};
    void setPol(Goldilocks::Element *pPol)
    {
        for (uint64_t i = 0; i < pol.size(); i++) std::memcpy(&pol[i][0], &pPol[i * pol[i].size()], pol[i].size() * sizeof(Goldilocks::Element));
    }
    std::string polJson() const
    {
        std::vector<std::string> it;
        for (const auto &e : pol) it.push_back(mi_json::arr(e));
        return mi_json::arr(it);
    }
    std::string FriP2json() const // friProof.hpp:130-151: the step trees followed by the final polynomial
    {
        std::vector<std::string> it;
        for (const auto &t : trees) it.push_back(t.ProofTree2json());
        it.push_back(polJson());
        return mi_json::arr(it);
    }
};

class Proofs
{
public:
    std::vector<Goldilocks::Element> root1, root2, root3, root4;
    Fri fri;
    std::vector<std::vector<Goldilocks::Element>> evals;
    Proofs(uint64_t polN, uint64_t dim, uint64_t numSteps, uint64_t evalSize)
        : root1(HASH_SIZE, Goldilocks::zero()), root2(HASH_SIZE, Goldilocks::zero()), root3(HASH_SIZE, Goldilocks::zero()),
          root4(HASH_SIZE, Goldilocks::zero()), fri(polN, dim, numSteps), evals(evalSize, std::vector<Goldilocks::Element>(dim, Goldilocks::zero())){};
    void setEvals(Goldilocks::Element *_evals)
    {
        for (uint64_t i = 0; i < evals.size(); i++) std::memcpy(&evals[i][0], &_evals[i * evals[i].size()], evals[i].size() * sizeof(Goldilocks::Element));
    }
    std::string evalsJson() const
    {
        std::vector<std::string> it;
        for (const auto &e : evals) it.push_back(mi_json::arr(e));
        return mi_json::arr(it);
    }
    std::string proof2json() const // friProof.hpp:176-218
    {
        return "{\"root1\":" + mi_json::arr(root1) + ",\"root2\":" + mi_json::arr(root2) + ",\"root3\":" + mi_json::arr(root3) +
               ",\"root4\":" + mi_json::arr(root4) + ",\"evals\":" + evalsJson() + ",\"fri\":" + fri.FriP2json() + "}";
    }
};

class FRIProof
{
public:
    Proofs proofs;
    std::vector<Goldilocks::Element> publics;
    FRIProof(uint64_t polN, uint64_t dim, uint64_t numTrees, uint64_t evalSize, uint64_t nPublics)
        : proofs(polN, dim, numTrees, evalSize), publics(nPublics){};
};
#endif
