// merklehash_goldilocks.hpp -- constants + MerklehashGoldilocks::{getTreeNumElements, root}
// (stark_info.hpp:329-335, build_const_tree.cpp:366-403, merkleTreeGL.hpp:24-32).
#ifndef MERKLEHASH_GOLDILOCKS
#define MERKLEHASH_GOLDILOCKS
#include "goldilocks_base_field.hpp"
#ifndef HASH_SIZE
#define HASH_SIZE 4
#endif
#define MERKLEHASHGOLDILOCKS_HEADER_SIZE 2
#define MERKLEHASHGOLDILOCKS_ARITY 2

class MerklehashGoldilocks
{
public:
    // [width, height, source (height*width), nodes ((2*height-1)*4)]
    static inline uint64_t getTreeNumElements(uint64_t degree) { return degree * HASH_SIZE + (degree - 1) * HASH_SIZE; }
    static inline uint64_t getTreeNumElements(uint64_t numCols, uint64_t degree)
    {
        return numCols * degree + getTreeNumElements(degree) + MERKLEHASHGOLDILOCKS_HEADER_SIZE;
    }
    static inline void root(Goldilocks::Element *root, Goldilocks::Element *tree, uint64_t numElementsTree)
    {
        std::memcpy(root, &tree[numElementsTree - HASH_SIZE], HASH_SIZE * sizeof(Goldilocks::Element));
    }
};
#endif
