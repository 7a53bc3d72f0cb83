// build_const_tree.hpp -- the `bctree` tool's buildConstTree (tools/starkpil/bctree/build_const_tree.cpp:333-449,
// GL hash type) on the GPU: LDE of the constant polynomials (its in-tree interpolate(), :198-331, is the LDE
// specification this repo's mi_lde follows) + Poseidon Merkle tree, written in the reference's file format
//     [ nPols, nExt, pols (nExt x nPols, row-major), nodes ((2 nExt - 1) x 4) ]   all uint64 little-endian
// which MerkleTreeGL(Goldilocks::Element *tree) maps back (merkleTreeGL.hpp:24-32, starks.hpp:141-143), plus the
// verification key {"constRoot": [4 numbers]}.
#ifndef BUILD_CONST_TREE_HPP
#define BUILD_CONST_TREE_HPP
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include "goldilocks_base_field.hpp"
#include "merklehash_goldilocks.hpp"
#include "mi_runtime.hpp"

namespace bctree_detail {
inline std::string slurp(const std::string &path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) { std::fprintf(stderr, "buildConstTree: cannot open %s\n", path.c_str()); std::exit(-1); }
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}
// Value of a TOP-LEVEL key of the starkstruct JSON object ({"nBits": 23, "nBitsExt": 24, "verificationHashType": "GL",
// "steps": [{"nBits": 24}, ...]}): keys nested inside "steps" (or any other array / object) are skipped by tracking the
// bracket depth, strings are skipped as strings.  Numbers and strings only -- all this tool reads.
inline bool json_field(const std::string &js, const std::string &key, std::string &out)
{
    int depth = 0;
    for (size_t p = 0; p < js.size(); p++) {
        const char ch = js[p];
        if (ch == '{' || ch == '[') { depth++; continue; }
        if (ch == '}' || ch == ']') { depth--; continue; }
        if (ch != '"') continue;
        size_t e = p + 1;
        while (e < js.size() && js[e] != '"') e += (js[e] == '\\') ? 2 : 1;
        if (e >= js.size()) return false;
        const std::string tok = js.substr(p + 1, e - p - 1);
        size_t q = e + 1;
        while (q < js.size() && (js[q] == ' ' || js[q] == '\n' || js[q] == '\t' || js[q] == '\r')) q++;
        if (depth == 1 && q < js.size() && js[q] == ':' && tok == key) { // a key of the outermost object
            q++;
            while (q < js.size() && (js[q] == ' ' || js[q] == '\n' || js[q] == '\t' || js[q] == '\r')) q++;
            if (q >= js.size()) return false;
            if (js[q] == '"') {
                size_t v = q + 1;
                while (v < js.size() && js[v] != '"') v += (js[v] == '\\') ? 2 : 1;
                out = js.substr(q + 1, v - q - 1);
                return true;
            }
            size_t v = q;
            while (v < js.size() && js[v] != ',' && js[v] != '}' && js[v] != '\n' && js[v] != ' ' && js[v] != '\r' && js[v] != '\t') v++;
            out = js.substr(q, v - q);
            return !out.empty();
        }
        p = e; // skip the string (a key at another depth, or a string value)
    }
    return false;
}
} // namespace bctree_detail

inline void buildConstTree(const std::string constFile, const std::string starkStructFile, const std::string constTreeFile,
                           const std::string verKeyFile)
{
    using namespace bctree_detail;
    const std::string js = slurp(starkStructFile);
    std::string v, hashType;
    if (!json_field(js, "nBits", v)) { std::fprintf(stderr, "buildConstTree: nBits missing\n"); std::exit(-1); }
    const uint64_t nBits = std::stoull(v);
    if (!json_field(js, "nBitsExt", v)) { std::fprintf(stderr, "buildConstTree: nBitsExt missing\n"); std::exit(-1); }
    const uint64_t nBitsExt = std::stoull(v);
    if (!json_field(js, "verificationHashType", hashType)) { // the reference reads it unconditionally (build_const_tree.cpp:343)
        std::fprintf(stderr, "buildConstTree: verificationHashType missing\n");
        std::exit(-1);
    }
    if (hashType != "GL") { // build_const_tree.cpp:404-443: BN128 trees belong to the recursiveF/SNARK side (out of scope)
        std::fprintf(stderr, "Invalid Hash Type: %s (only GL is built here)\n", hashType.c_str());
        std::exit(-1);
    }
    const uint64_t n = 1ULL << nBits, nExt = 1ULL << nBitsExt;
    const std::string raw = slurp(constFile);
    const uint64_t nPols = raw.size() / (n * sizeof(Goldilocks::Element));
    if (nPols == 0 || raw.size() != nPols * n * sizeof(Goldilocks::Element)) {
        std::fprintf(stderr, "buildConstTree: %s is not a multiple of 2^nBits elements\n", constFile.c_str());
        std::exit(-1);
    }
    const uint64_t numElementsTree = MerklehashGoldilocks::getTreeNumElements(nExt);
    const uint64_t header = MERKLEHASHGOLDILOCKS_HEADER_SIZE, numElementsCopy = header + nPols * nExt;
    const uint64_t numElements = numElementsCopy + numElementsTree;
    std::vector<Goldilocks::Element> constTree(numElements);
    constTree[0] = Goldilocks::fromU64(nPols);
    constTree[1] = Goldilocks::fromU64(nExt);

    mi_ctx *c = mi::ctx();
    uint64_t *d_in = (uint64_t *)mi_dev_alloc(c, raw.size()), *d_ext = (uint64_t *)mi_dev_alloc(c, nPols * nExt * 8);
    uint64_t *d_nodes = (uint64_t *)mi_dev_alloc(c, numElementsTree * 8);
    if (!d_in || !d_ext || !d_nodes) mi::fail("buildConstTree (alloc)");
    mi::check(mi_copy_h2d(c, d_in, raw.data(), raw.size()), "buildConstTree (h2d)");
    mi::check(mi_lde_dev(c, d_ext, nPols, d_in, nPols, nExt, n, nPols), "buildConstTree (interpolate)");
    mi::check(mi_merkle_build_dev(c, d_nodes, d_ext, nPols, nPols, nExt), "buildConstTree (merkletree)");
    mi::check(mi_copy_d2h(c, &constTree[header], d_ext, nPols * nExt * 8), "buildConstTree (d2h pols)");
    mi::check(mi_copy_d2h(c, &constTree[numElementsCopy], d_nodes, numElementsTree * 8), "buildConstTree (d2h nodes)");
    mi_dev_free(c, d_in); mi_dev_free(c, d_ext); mi_dev_free(c, d_nodes);

    if (verKeyFile != "") {
        std::ofstream fk(verKeyFile);
        fk << "{\n    \"constRoot\": [\n";
        for (int i = 0; i < 4; i++)
            fk << "        " << Goldilocks::toU64(constTree[numElements - 4 + i]) << (i < 3 ? ",\n" : "\n");
        fk << "    ]\n}\n";
    }
    std::ofstream fw(constTreeFile.c_str(), std::fstream::out | std::fstream::binary);
    fw.write((const char *)constTree.data(), numElements * sizeof(Goldilocks::Element));
    fw.close();
}
#endif
