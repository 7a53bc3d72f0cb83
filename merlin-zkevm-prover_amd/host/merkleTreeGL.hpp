// merkleTreeGL.hpp -- MerkleTreeGL with the reference's public surface (merkleTreeGL.hpp:9-79,
// merkleTreeGL.cpp:5-44): height, width, source, nodes, copySource, merkelize, getTreeNumElements, getRoot,
// getGroupProof, MerkleProofSize.  Level-1 re-implementation: merkelize() builds the tree in HBM
// and keeps source + nodes device-resident; getRoot / getGroupProof fetch 4 / width+4*levels elements.  The
// host `nodes` array of the reference is still allocated (same ownership rules) and filled on demand by
// syncNodesToHost() for callers that walk it directly.
#ifndef MERKLETREEGL
#define MERKLETREEGL
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "goldilocks_base_field.hpp"
#include "poseidon_goldilocks.hpp"
#include "merklehash_goldilocks.hpp"
#include "mi_runtime.hpp"

#define MERKLEHASHGL_ARITY 2
class MerkleTreeGL
{
    uint64_t *d_source = nullptr, *d_nodes = nullptr; // HBM copies
    bool d_source_borrowed = false, d_nodes_borrowed = false, d_source_tiled = false;
    mi_multi_tree *multiTree = nullptr;               // set: the nodes are subtrees on several devices
    mi_multi *shardMulti = nullptr;                   // with shardSources: the rows of other shards are read on their devices
    std::vector<const uint64_t *> shardSources;
    uint64_t rowsPerShard = 0;
    void releaseDevice()
    {
        if (multiTree) mi_multi_tree_free(multiTree);
        multiTree = nullptr;
        if (d_source && !d_source_borrowed) mi::devFree(d_source);
        if (d_nodes && !d_nodes_borrowed) mi::devFree(d_nodes);
        d_source = d_nodes = nullptr;
        d_source_borrowed = d_nodes_borrowed = d_source_tiled = false;
    }

public:
    uint64_t height = 0;
    uint64_t width = 0;
    Goldilocks::Element *source = NULL;
    Goldilocks::Element *nodes = NULL;
    bool isSourceAllocated = false;
    bool isNodesAllocated = false;
    MerkleTreeGL(){};
    MerkleTreeGL(Goldilocks::Element *tree) // constant tree mapped from a file: [width, height, source, nodes]
    {
        width = Goldilocks::toU64(tree[0]);
        height = Goldilocks::toU64(tree[1]);
        source = &tree[2];
        nodes = &tree[2 + height * width];
    };
    MerkleTreeGL(uint64_t _height, uint64_t _width, Goldilocks::Element *_source) : height(_height), width(_width), source(_source)
    {
        if (source == NULL) {
            source = (Goldilocks::Element *)calloc(height * width, sizeof(Goldilocks::Element));
            isSourceAllocated = true;
        }
        nodes = (Goldilocks::Element *)calloc(getTreeNumElements(), sizeof(Goldilocks::Element));
        isNodesAllocated = true;
    };
    MerkleTreeGL(const MerkleTreeGL &) = delete;
    MerkleTreeGL &operator=(const MerkleTreeGL &) = delete;
    ~MerkleTreeGL()
    {
        releaseDevice();
        if (isSourceAllocated) free(source);
        if (isNodesAllocated) free(nodes);
    };
    void copySource(Goldilocks::Element *_source) { std::memcpy(source, _source, height * width * sizeof(Goldilocks::Element)); }
    // device-resident source (e.g. an LDE output that never left HBM): borrowed, pitch == width
    void setDeviceSource(uint64_t *dev_source)
    {
        releaseDevice();
        d_source = dev_source;
        d_source_borrowed = true;
    }
    // a tree that was built in HBM by someone else (host/starks.hpp): leaves and nodes borrowed, ready for getRoot / getGroupProofs
    // tiled: the leaves' rows lie TILE-MAJOR ([height / 64][width][64], mi_lde_merkle_dev_tiled) -- the openings gather a row's values
    void setDeviceTree(uint64_t *dev_source, uint64_t *dev_nodes, bool tiled = false)
    {
        releaseDevice();
        d_source = dev_source; d_nodes = dev_nodes;
        d_source_borrowed = d_nodes_borrowed = true;
        d_source_tiled = tiled;
    }
    // a tree whose subtrees live on several devices (csrc/multi.hip) while the rows are read from a row-major image on this one: the view
    // owns the sharded tree from here on
    void setMultiTree(uint64_t *dev_source, mi_multi_tree *tree)
    {
        releaseDevice();
        d_source = dev_source;
        d_source_borrowed = true;
        multiTree = tree;
        shardSources.clear();
    }
    // ... and when this device's image holds only ITS rows of the section (row-sharded evaluation, host/starks.hpp): row idx lives on shard
    // idx / rowsPerShard, at sources[shard] + idx * width in that device's memory (sources[0] unused: shard 0 is this device)
    void setShardSources(mi_multi *mm, const std::vector<const uint64_t *> &sources, uint64_t rowsPerShard_)
    {
        shardMulti = mm; shardSources = sources; rowsPerShard = rowsPerShard_;
    }
    uint64_t *deviceNodes() { return d_nodes; }
    uint64_t *deviceSource() { return d_source; }

    void merkelize() // merkleTreeGL.cpp:37-44
    {
        mi_ctx *c = mi::ctx();
        if (!d_source) {
            d_source = mi::devAlloc(height * width, "MerkleTreeGL::merkelize (alloc source)");
        }
        if (!d_source_borrowed) mi::check(mi_copy_h2d(c, d_source, source, height * width * 8), "MerkleTreeGL::merkelize (h2d)");
        if (!d_nodes) {
            d_nodes = mi::devAlloc(getTreeNumElements(), "MerkleTreeGL::merkelize (alloc nodes)");
        }
        mi::check(mi_merkle_build_dev(c, d_nodes, d_source, width, width, height), "MerkleTreeGL::merkelize");
    }
    void syncNodesToHost()
    {
        if (d_nodes) mi::check(mi_copy_d2h(mi::ctx(), nodes, d_nodes, getTreeNumElements() * 8), "MerkleTreeGL::syncNodesToHost");
    }
    uint64_t getTreeNumElements() { return height * HASH_SIZE + (height - 1) * HASH_SIZE; }
    void getRoot(Goldilocks::Element *root)
    {
        if (d_nodes) mi::check(mi_copy_d2h(mi::ctx(), root, d_nodes + getTreeNumElements() - HASH_SIZE, HASH_SIZE * 8), "MerkleTreeGL::getRoot");
        else std::memcpy(root, &nodes[getTreeNumElements() - HASH_SIZE], HASH_SIZE * sizeof(Goldilocks::Element));
    }
    uint64_t MerkleProofSize()
    {
        if (height > 1) return (uint64_t)ceil(log10(height) / log10(MERKLEHASHGL_ARITY));
        return 0;
    }
    // proof = width values of row idx, then the sibling of every level (merkleTreeGL.cpp:12-35)
    void getGroupProof(Goldilocks::Element *proof, uint64_t idx) { getGroupProofs(proof, &idx, 1); }
    void getGroupProofs(Goldilocks::Element *proofs, const uint64_t *idx, uint64_t nq)
    {
        const uint64_t stride = width + MerkleProofSize() * HASH_SIZE;
        if (multiTree) { // siblings from the shards' subtrees, the rows' values from the image on this device
            mi_ctx *c = mi::ctx();
            mi::check(mi_multi_group_proofs(multiTree, (uint64_t *)proofs, idx, nq, 0), "MerkleTreeGL::getGroupProofs (sharded tree)");
            std::vector<uint64_t> sib((uint64_t *)proofs, (uint64_t *)proofs + nq * stride);
            uint64_t *d_out = mi::devAlloc(nq * stride, "MerkleTreeGL::getGroupProofs (alloc)");
            // rows another device holds are NOT read here: with row shards this device's image has memory under its own rows only (a sparse
            // address range, host/starks.hpp mi::Arena): row 0 stands in for them in this launch, their values come from their holder below
            std::vector<uint64_t> local(idx, idx + nq);
            for (uint64_t q = 0; q < nq && !shardSources.empty(); q++) {
                const uint64_t g = idx[q] / rowsPerShard;
                if (g != 0 && g < shardSources.size() && shardSources[g]) local[q] = 0;
            }
            mi::check(mi_merkle_group_proofs_dev(c, d_out, nullptr, d_source, width, height, width, local.data(), nq), "MerkleTreeGL::getGroupProofs (rows of a sharded tree)");
            mi::check(mi_copy_d2h(c, proofs, d_out, nq * stride * 8), "MerkleTreeGL::getGroupProofs (d2h)");
            mi::devFree(d_out);
            for (uint64_t q = 0; q < nq; q++) std::memcpy((uint64_t *)proofs + q * stride + width, &sib[q * stride + width], (stride - width) * 8);
            bool away = false;
            for (uint64_t q = 0; q < nq && !shardSources.empty(); q++) { // rows another device holds: their values from there
                const uint64_t g = idx[q] / rowsPerShard;
                if (g == 0 || g >= shardSources.size() || !shardSources[g]) continue;
                mi::check(mi_multi_set_device(shardMulti, (int)g), "MerkleTreeGL::getGroupProofs (row of another device)");
                mi::check(mi_copy_d2h(mi_multi_ctx(shardMulti, (int)g), (uint64_t *)proofs + q * stride, shardSources[g] + idx[q] * width, width * 8), "MerkleTreeGL::getGroupProofs (row of another device)");
                away = true;
            }
            if (away) mi::check(mi_multi_set_device(shardMulti, 0), "MerkleTreeGL::getGroupProofs (device)");
            return;
        }
        if (d_nodes && d_source) {
            mi_ctx *c = mi::ctx();
            uint64_t *d_out = mi::devAlloc(nq * stride, "MerkleTreeGL::getGroupProofs (alloc)");
            if (d_source_tiled) mi::check(mi_merkle_group_proofs_tiled_dev(c, d_out, d_nodes, d_source, width, height, width, idx, nq), "MerkleTreeGL::getGroupProofs (tile-major rows)");
            else mi::check(mi_merkle_group_proofs_dev(c, d_out, d_nodes, d_source, width, height, width, idx, nq), "MerkleTreeGL::getGroupProofs");
            mi::check(mi_copy_d2h(c, proofs, d_out, nq * stride * 8), "MerkleTreeGL::getGroupProofs (d2h)");
            mi::devFree(d_out);
            return;
        }
        for (uint64_t q = 0; q < nq; q++) { // tree loaded from a file (constant tree): plain copies, no arithmetic
            Goldilocks::Element *p = proofs + q * stride;
            std::memcpy(p, &source[idx[q] * width], width * sizeof(Goldilocks::Element));
            uint64_t id = idx[q], offset = 0, n = height;
            p += width;
            while (n > 1) {
                std::memcpy(p, &nodes[offset + (id ^ 1) * HASH_SIZE], HASH_SIZE * sizeof(Goldilocks::Element));
                p += HASH_SIZE; offset += n * HASH_SIZE; n >>= 1; id >>= 1;
            }
        }
    }
};
#endif
