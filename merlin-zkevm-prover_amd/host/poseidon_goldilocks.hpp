// poseidon_goldilocks.hpp -- PoseidonGoldilocks with the upstream signatures used by the reference
// (transcript.cpp:23,46; merkleTreeGL.cpp:37-44; build_const_tree.cpp:382; smt.cpp:1080), forwarding to
// libmi_stark.  Host pointers in, host pointers out, exactly like the CPU library; MerkleTreeGL below keeps
// big trees device-resident instead.
#ifndef POSEIDON_GOLDILOCKS
#define POSEIDON_GOLDILOCKS
#include "goldilocks_base_field.hpp"
#include "mi_runtime.hpp"
#define SPONGE_WIDTH 12
#define RATE 8
#define CAPACITY 4
#ifndef HASH_SIZE
#define HASH_SIZE 4 // the reference's merkleTreeGL.hpp:63 gets it from this header (it includes no merklehash header)
#endif

class PoseidonGoldilocks
{
public:
    static inline void hash_full_result(Goldilocks::Element *state, const Goldilocks::Element (&input)[SPONGE_WIDTH])
    {
        mi::check(mi_poseidon_hash_full_result(mi::ctx(), (uint64_t *)state, (const uint64_t *)input), "hash_full_result");
    }
    static inline void hash_full_result(Goldilocks::Element *state, const Goldilocks::Element *input)
    {
        mi::check(mi_poseidon_hash_full_result(mi::ctx(), (uint64_t *)state, (const uint64_t *)input), "hash_full_result");
    }
    static inline void hash(Goldilocks::Element (&state)[CAPACITY], const Goldilocks::Element (&input)[SPONGE_WIDTH])
    {
        mi::check(mi_poseidon_hash(mi::ctx(), (uint64_t *)state, (const uint64_t *)input), "hash");
    }
    static inline void linear_hash(Goldilocks::Element *output, Goldilocks::Element *input, uint64_t size)
    {
        mi::check(mi_poseidon_linear_hash(mi::ctx(), (uint64_t *)output, (const uint64_t *)input, size), "linear_hash");
    }
    static inline void merkletree(Goldilocks::Element *tree, Goldilocks::Element *input, uint64_t num_cols, uint64_t num_rows,
                                  int /*nThreads*/ = 0, uint64_t dim = 1)
    {
        mi::check(mi_merkle_build(mi::ctx(), (uint64_t *)tree, (const uint64_t *)input, num_cols * dim, num_rows), "merkletree");
    }
    static inline void merkletree_avx(Goldilocks::Element *tree, Goldilocks::Element *input, uint64_t num_cols, uint64_t num_rows,
                                      int nThreads = 0, uint64_t dim = 1)
    {
        merkletree(tree, input, num_cols, num_rows, nThreads, dim);
    }
    static inline void merkletree_avx512(Goldilocks::Element *tree, Goldilocks::Element *input, uint64_t num_cols, uint64_t num_rows,
                                         int nThreads = 0, uint64_t dim = 1)
    {
        merkletree(tree, input, num_cols, num_rows, nThreads, dim);
    }
};
#endif
