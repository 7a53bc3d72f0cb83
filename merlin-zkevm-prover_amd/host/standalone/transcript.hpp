// transcript.hpp (standalone stand-in) -- Fiat-Shamir transcript, same state machine as transcript.cpp:4-87.  Sequential by nature and
// kept on the host; every permutation runs on the GPU (a whole put in one launch: mi_transcript_put; a squeeze through
// PoseidonGoldilocks::hash_full_result).  Only for builds without the reference tree (the GPU-box tests, libmi_starks.so): the
// reference's own transcript.{hpp,cpp} compile unchanged against Level 0 and are what a maintainer's build uses.
#ifndef TRANSCRIPT_CLASS
#define TRANSCRIPT_CLASS
#include <cmath>
#include <vector>
#include "goldilocks_base_field.hpp"
#include "goldilocks_cubic_extension.hpp"
#include "poseidon_goldilocks.hpp"
#include "mi_runtime.hpp"

#define TRANSCRIPT_STATE_SIZE 4
#define TRANSCRIPT_PENDING_SIZE 8
#define TRANSCRIPT_OUT_SIZE 12

class Transcript
{
    void _updateState()
    {
        Goldilocks::Element inputs[TRANSCRIPT_OUT_SIZE];
        std::memcpy(inputs, pending, TRANSCRIPT_PENDING_SIZE * sizeof(Goldilocks::Element));
        std::memcpy(&inputs[TRANSCRIPT_PENDING_SIZE], state, TRANSCRIPT_STATE_SIZE * sizeof(Goldilocks::Element));
        PoseidonGoldilocks::hash_full_result(out, inputs);
        out_cursor = TRANSCRIPT_OUT_SIZE;
        std::memset(pending, 0, sizeof(pending));
        pending_cursor = 0;
        std::memcpy(state, out, TRANSCRIPT_STATE_SIZE * sizeof(Goldilocks::Element));
    }

public:
    Goldilocks::Element state[TRANSCRIPT_STATE_SIZE];
    Goldilocks::Element pending[TRANSCRIPT_PENDING_SIZE];
    Goldilocks::Element out[TRANSCRIPT_OUT_SIZE];
    unsigned pending_cursor = 0;
    unsigned out_cursor = 0;
    unsigned state_cursor = 0;

    Transcript()
    {
        std::memset(state, 0, sizeof(state));
        std::memset(pending, 0, sizeof(pending));
        std::memset(out, 0, sizeof(out));
    }
    void put(Goldilocks::Element *input, uint64_t size)
    {
        // the same state machine as _add1 element by element, one launch for the whole put (mi_transcript_put)
        uint32_t pc = pending_cursor, oc = out_cursor;
        mi::check(mi_transcript_put(mi::ctx(), (uint64_t *)state, (uint64_t *)pending, (uint64_t *)out, &pc, &oc, (const uint64_t *)input, size), "Transcript::put");
        pending_cursor = pc;
        out_cursor = oc;
    }
    Goldilocks::Element getFields1()
    {
        if (out_cursor == 0) _updateState();
        Goldilocks::Element res = out[(TRANSCRIPT_OUT_SIZE - out_cursor) % TRANSCRIPT_OUT_SIZE];
        out_cursor--;
        return res;
    }
    void getField(Goldilocks::Element *output)
    {
        for (int i = 0; i < 3; i++) output[i] = getFields1();
    }
    void getPermutations(uint64_t *res, uint64_t n, uint64_t nBits)
    {
        uint64_t totalBits = n * nBits;
        uint64_t NFields = (uint64_t)floor((float)(totalBits - 1) / 63) + 1;
        std::vector<Goldilocks::Element> fields(NFields);
        for (uint64_t i = 0; i < NFields; i++) fields[i] = getFields1();
        uint64_t curField = 0, curBit = 0;
        for (uint64_t i = 0; i < n; i++) {
            uint64_t a = 0;
            for (uint64_t j = 0; j < nBits; j++) {
                uint64_t bit = (Goldilocks::toU64(fields[curField]) >> curBit) & 1;
                if (bit) a = a + (1ULL << j);
                curBit++;
                if (curBit == 63) { curBit = 0; curField++; }
            }
            res[i] = a;
        }
    }
};
#endif
