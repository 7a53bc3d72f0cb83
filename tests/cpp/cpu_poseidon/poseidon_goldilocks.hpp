// poseidon_goldilocks.hpp for ONE test that must run on the CPU (tests/test_reference_compiles.py): the reference's transcript.cpp is
// compiled unchanged and its state machine run beside the oracle's; the permutation underneath both is the oracle's
// (glo_hash_full_result, itself pinned by the golden proofs).  Not a product header: the product's PoseidonGoldilocks
// (host/poseidon_goldilocks.hpp) runs every permutation on the GPU and has no CPU path.
#ifndef POSEIDON_GOLDILOCKS
#define POSEIDON_GOLDILOCKS
#include "goldilocks_base_field.hpp"
#include "../../../oracle/gl_oracle.h"
#define HASH_SIZE 4
class PoseidonGoldilocks
{
public:
    static void hash_full_result(Goldilocks::Element *out, const Goldilocks::Element *in) { glo_hash_full_result((uint64_t *)out, (const uint64_t *)in); }
};
#endif
