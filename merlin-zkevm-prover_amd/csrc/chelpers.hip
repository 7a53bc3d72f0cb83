// chelpers.hip -- the generated constraint evaluators ("chelpers") of Starks::genProof on the GPU: the translator for
// all five steps (step2prev / step3prev / step3 share one opcode numbering, see role_stepbase), the SIMT interpreter for the two
// extended-domain steps; chelpers_native.hip compiles any of them to gfx950 kernels.
//
// Reference: src/starkpil/zkevm/chelpers/zkevm.chelpers.step42ns.parser.{hpp,cpp}.  The reference ships, per STARK step,
// a PROGRAM (two generated tables: op42[NOPS_] opcodes, args42[NARGS_] operands) and an INTERPRETER
// (ZkevmSteps::step42ns_parser_first_avx, :10-760; its scalar twin step42ns_parser_first, :762-1441) that runs the
// program once per row of the extended domain: ~12 000 opcodes = 19 198 field operations per row, 2^24 rows, reading
// the committed / constant polynomials at the row (and at row + k mod N for the "prime" polynomials), the challenges,
// public inputs, x_2ns and ZhInv, and writing q_2ns.  Starks::genProof calls it at starks.cpp:237-241.
//
// MI355X design (not a port of the AVX interpreter):
//   * mi_chelpers_compile takes the reference's tables AS DATA and translates them once, on the host, into a uniform
//     three-address code: {add | sub | mul | store-q} x {dst: base temp | ext temp | q} x two operand descriptors
//     (temp, polynomial at row, polynomial at shifted row, constant polynomial, number, challenge, public, x, 1/Z_H).
//     The 89 opcodes of the table are just the operand-kind combinations the generator happened to emit (five of them
//     fusions of others); the role table below records, per opcode, what the reference interpreter does with each
//     argument.
//   * The generated program keeps ~975 base and ~170 extension temporaries live per row (11.8 KB) -- fine for a CPU
//     stack, fatal for a GPU lane.  The translator therefore (1) forwards copies (a temp that only renames a polynomial
//     element, a constant or another temp is replaced by its source), (2) puts the program in SSA form and RESCHEDULES
//     it depth-first from the q store, evaluating the operand that needs more registers first (Sethi-Ullman): the
//     expression is a long Horner chain over ~900 independently computable constraint values, which the generator
//     computes up front and the reschedule consumes as they are produced, (3) drops dead code and (4) re-allocates
//     temporaries by linear scan.  For the zkEVM step42ns program: 1 401 live words -> 89.
//   * One row per lane, 64-lane workgroups, temporaries in LDS as [word][lane] (conflict-free 8-byte accesses, the word
//     index is wave-uniform).  The instruction stream is wave-uniform: it is fetched through the scalar cache and
//     every decode branch is scalar.  All arithmetic is the same gl:: code the other kernels use.
//   * Polynomial operands.  The sections are row-major, a lane owns a row: read in place, every operand is a 64-lane
//     gather at a stride of one row (5 320 B for the 665-column section), one 64-byte sector per lane -- 12 000 operand
//     reads per row = 13 TB of sector traffic at 2^24 rows (first version, measured: 6.3 s, exactly that bound).  So the
//     workgroups are PERSISTENT and each, per group of 64 rows, first TRANSPOSES its rows (+ the few following rows the
//     shifted "prime" reads need) of every section into a private column-major scratch in HBM -- rows read coalesced
//     along the row, turned in an LDS tile, written as 576-byte column runs -- and the interpreter then reads operand
//     (column, row) as one contiguous 512-byte run per wave.  The staged operands of 8 instructions are fetched together
//     (16 loads in flight per wave) before those instructions execute, because with the temporaries' LDS footprint only
//     three waves fit a CU and a dependent load per instruction would leave the kernel latency-bound.
//
// Parity status: UNPINNED by reference data (no input / output pair for any chelpers step exists in the reference tree;
// zkevm.starkinfo.json and the constant polynomials are absent).  The oracle (oracle/chelpers_oracle.c) restates the
// reference interpreter opcode by opcode; tests compare this implementation with it on synthetic programs that use
// every opcode, and -- where /root/reference is present -- on the reference's own program tables.
#include "common.h"
#include <algorithm>
#include <array>
#include <set>
#include <string.h>

using gl::E3;

#include "chelpers_ir.h"

namespace chp {

// what one opcode of the reference interpreter does: dst kind, operation, the kinds of its two sources IN ARGUMENT ORDER
// (argument 0 is always the destination temp; then the arguments of source a, then those of source b)
struct Role { Cls cls; Kind dst, a, b; };

// ---- step42ns (zkevm.chelpers.step42ns.parser.cpp): cases 0-83 of step42ns_parser_first_avx (:24-660) = cases 0-83 of the
// scalar step42ns_parser_first (:781-1383); 84-88 are fusions (:661-748)
static bool role_step42ns(uint64_t op, Role &r)
{
#define R(c, d, x, y) r = {c, d, x, y}; return true
    switch (op) {
    case 0: R(C_ADD, K_T1, K_T1, K_T1);        case 1: R(C_ADD, K_T1, K_T1, K_POL);        case 2: R(C_ADD, K_T1, K_T1, K_NUM);
    case 3: R(C_ADD, K_T1, K_T1, K_CONST);     case 4: R(C_ADD, K_T1, K_POL, K_POL);       case 5: R(C_ADD, K_T1, K_POLS, K_POLS);
    case 6: R(C_ADD, K_T1, K_POL, K_CONST);    case 7: R(C_ADD, K_T1, K_POL, K_NUM);       case 8: R(C_ADD, K_T1, K_CONST, K_CONST);
    case 9: R(C_ADD, K_T1, K_CONSTS, K_CONSTS); case 10: R(C_ADD, K_T1, K_CONST, K_NUM);   case 11: R(C_ADD, K_T1, K_CONSTS, K_NUM);
    case 12: R(C_ADD, K_T3, K_T1, K_T3);       case 13: R(C_ADD, K_T3, K_NUM, K_CHAL);     case 14: R(C_ADD, K_T3, K_T1, K_CHAL);
    case 15: R(C_ADD, K_T3, K_POL, K_T3);      case 16: R(C_ADD, K_T3, K_POL, K_CHAL);     case 17: R(C_ADD, K_T3, K_T3, K_T3);
    case 18: R(C_ADD, K_T3, K_T3, K_CHAL);     case 19: R(C_ADD, K_T3, K_POL3, K_T3);      case 20: R(C_ADD, K_T3, K_POL3, K_CHAL);
    case 21: R(C_SUB, K_T1, K_T1, K_T1);       case 22: R(C_SUB, K_T1, K_T1, K_POL);       case 23: R(C_SUB, K_T1, K_T1, K_POLS);
    case 24: R(C_SUB, K_T1, K_POL, K_T1);      case 25: R(C_SUB, K_T1, K_POLS, K_T1);      case 26: R(C_SUB, K_T1, K_T1, K_NUM);
    case 27: R(C_SUB, K_T1, K_NUM, K_T1);      case 28: R(C_SUB, K_T1, K_POL, K_NUM);      case 29: R(C_SUB, K_T1, K_POLS, K_NUM);
    case 30: R(C_SUB, K_T1, K_NUM, K_POL);     case 31: R(C_SUB, K_T1, K_NUM, K_POLS);     case 32: R(C_SUB, K_T1, K_NUM, K_CONST);
    case 33: R(C_SUB, K_T1, K_NUM, K_CONSTS);  case 34: R(C_SUB, K_T1, K_POL, K_PUB);      case 35: R(C_SUB, K_T1, K_POLS, K_POL);
    case 36: R(C_SUB, K_T1, K_POL, K_POLS);    case 37: R(C_SUB, K_T1, K_POL, K_POL);      case 38: R(C_SUB, K_T1, K_POLS, K_POLS);
    case 39: R(C_SUB, K_T1, K_CONST, K_POL);   case 40: R(C_SUB, K_T1, K_T1, K_CONST);     case 41: R(C_SUB, K_T3, K_POL3, K_NUM);
    case 42: R(C_SUB, K_T3, K_T3, K_T3);       case 43: R(C_SUB, K_T3, K_T3, K_CHAL);      case 44: R(C_SUB, K_T3, K_T3, K_POL3);
    case 45: R(C_MUL, K_T1, K_T1, K_T1);       case 46: R(C_MUL, K_T1, K_NUM, K_T1);       case 47: R(C_MUL, K_T1, K_POL, K_T1);
    case 48: R(C_MUL, K_T1, K_POLS, K_T1);     case 49: R(C_MUL, K_T1, K_T1, K_CONST);     case 50: R(C_MUL, K_T1, K_POL, K_POL);
    case 51: R(C_MUL, K_T1, K_POL, K_POLS);    case 52: R(C_MUL, K_T1, K_POLS, K_POLS);    case 53: R(C_MUL, K_T1, K_NUM, K_POL);
    case 54: R(C_MUL, K_T1, K_POL, K_CONST);   case 55: R(C_MUL, K_T1, K_POLS, K_CONST);   case 56: R(C_MUL, K_T1, K_T1, K_POL);
    case 57: R(C_MUL, K_T1, K_T1, K_POLS);     case 58: R(C_MUL, K_T1, K_CONST, K_T1);     case 59: R(C_MUL, K_T3, K_T1, K_CHAL);
    case 60: R(C_MUL, K_T3, K_CONST, K_T3);    case 61: R(C_MUL, K_T3, K_T1, K_T3);        case 62: R(C_MUL, K_T3, K_POL, K_CHAL);
    case 63: R(C_MUL, K_T3, K_POLS, K_CHAL);   case 64: R(C_MUL, K_T3, K_POL, K_T3);       case 65: R(C_MUL, K_T3, K_POLS, K_T3);
    case 66: R(C_MUL, K_T3, K_NUM, K_CHAL);    case 67: R(C_MUL, K_T3, K_X, K_CHAL);       case 68: R(C_MUL, K_T3, K_X, K_T3);
    case 69: R(C_STOREQ, K_Q, K_T3, K_ZHINV);  // q_2ns[i] = zi.zhInv(i) * tmp3[arg0]: the only argument is the SOURCE
    case 70: R(C_MUL, K_T3, K_CHAL, K_T3);     // mul33c(tmp3[a0], tmp3[a2], challenges[a1]): challenge index comes first
    case 71: R(C_MUL, K_T3, K_T3, K_T3);       case 72: R(C_MUL, K_T3, K_POL3, K_POL3);    case 73: R(C_MUL, K_T3, K_POL3S, K_CHAL);
    case 74: R(C_MUL, K_T3, K_POL3S, K_T3);    case 75: R(C_MUL, K_T3, K_POL3, K_T3);      case 76: R(C_MUL, K_T3, K_POL3, K_CHAL);
    case 77: R(C_MUL, K_T3, K_POL3S, K_POL3);  case 78: R(C_COPY, K_T1, K_T1, K_NONE);     case 79: R(C_COPY, K_T1, K_POL, K_NONE);
    case 80: R(C_COPY, K_T1, K_POLS, K_NONE);  case 81: R(C_COPY, K_T1, K_NUM, K_NONE);    case 82: R(C_COPY, K_T1, K_CONST, K_NONE);
    case 83: R(C_COPY, K_T1, K_CONSTS, K_NONE);
    }
#undef R
    return false;
}
static const std::vector<uint64_t> *fused_step42ns(uint64_t op)
{
    static const std::vector<uint64_t> f84 = {12, 70}, f85 = {0, 50}, f86 = {32, 47, 21, 32, 48}, f87 = {12, 70, 12, 70, 12, 70, 12, 70},
                                       f88 = {21, 50, 21, 53, 0, 0, 50, 50, 0, 50, 21, 50};
    switch (op) {
    case 84: return &f84; case 85: return &f85; case 86: return &f86; case 87: return &f87; case 88: return &f88;
    }
    return nullptr;
}

// ---- step52ns (zkevm.chelpers.step52ns.parser.cpp, step52ns_parser_first :520-690 = step52ns_parser_first_avx :10-190): an
// accumulator machine over three extension registers tmp / tmp1 / tmp2 (here the extension temporaries 0 / 1 / 2), the
// challenges 5 and 6, params.evals, xDivXSubXi / xDivXSubWXi at the row, and the output f_2ns.  Only the polynomial
// operands take arguments; registers and challenge indices are fixed by the opcode.
struct Role52 { Cls cls; uint64_t dst; Kind a; uint64_t a_fixed; Kind b; uint64_t b_fixed; };
static bool role_step52ns(uint64_t op, Role52 &r)
{
#define R(c, d, ka, fa, kb, fb) r = {c, d, ka, fa, kb, fb}; return true
    switch (op) {
    case 0: R(C_MUL, 0, K_POL, 0, K_CHAL, 5);    // tmp = pols[..] * challenges[5]            (mul13c)
    case 1: R(C_MUL, 0, K_T3, 0, K_CHAL, 5);     // tmp = tmp * challenges[5]
    case 2: R(C_MUL, 0, K_T3, 0, K_CHAL, 6);     // tmp = tmp * challenges[6]
    case 3: R(C_MUL, 1, K_T3, 0, K_CHAL, 5);     // tmp1 = tmp * challenges[5]
    case 4: R(C_MUL, 0, K_T3, 2, K_CHAL, 6);     // tmp = tmp2 * challenges[6]
    case 5: R(C_MUL, 0, K_T3, 0, K_XD, 0);       // tmp = tmp * xDivXSubXi[i]
    case 6: R(C_MUL, 0, K_T3, 0, K_XDW, 0);      // tmp = tmp * xDivXSubWXi[i]
    case 7: R(C_ADD, 0, K_T3, 0, K_T3, 2);       // tmp = tmp + tmp2
    case 8: R(C_ADD, 0, K_T3, 1, K_T3, 0);       // tmp = tmp1 + tmp
    case 9: R(C_ADD, 0, K_T3, 0, K_POL3, 0);     // tmp = tmp + pols[..] (extension element)
    case 10: R(C_ADD, 0, K_T3, 0, K_POL, 0);     // tmp = tmp + pols[..] (base element)       (add31)
    case 11: R(C_SUB, 2, K_POL, 0, K_EVAL, 0);   // tmp2 = pols[..] - evals[k]                (sub13c)
    case 12: R(C_SUB, 2, K_POL3, 0, K_EVAL, 0);  // tmp2 = pols[..] - evals[k]                (sub33c)
    case 13: R(C_SUB, 2, K_CONST, 0, K_EVAL, 0); // tmp2 = const[c] - evals[k]
    case 14: R(C_SUB, 0, K_CONST, 5, K_EVAL, 0); // tmp = const[5] - evals[0]: both indices are literals in the reference
    case 15: R(C_STOREF, 0, K_T3, 0, K_NONE, 0); // f_2ns[i] = tmp
    }
#undef R
    return false;
}
static const std::vector<uint64_t> *fused_step52ns(uint64_t op)
{
    static const std::vector<uint64_t> f16 = {1, 10}, f17 = {1, 9}, f18 = {2, 11, 7}, f19 = {2, 13, 7}, f20 = {2, 12, 7};
    switch (op) {
    case 16: return &f16; case 17: return &f17; case 18: return &f18; case 19: return &f19; case 20: return &f20;
    }
    return nullptr;
}

// ---- the base-domain steps step2prev / step3prev / step3 (zkevm.chelpers.step{2prev,3prev,3}.parser.cpp, *_parser_first_avx): ONE
// opcode numbering for the three.  Cases 0-83 are step42ns's, case for case, with params.pConstPols / params.x_n in place of
// pConstPols2ns / x_2ns (69, the q store, is not theirs); 84-85 two more temp operations; 86-100 the result goes to a polynomial
// at the row, &params.pols[a0 + i * a1]; 101-114 to a polynomial at a shifted row, offsets1 = a0 + ((i + a1) % a2) * a3; 91, 97,
// 99 are "code not used" asserts; 115 (step3 only) is the fusion [0, 50].  ddim: 3 where the reference calls a Goldilocks3 form.
struct RoleB { Cls cls; Kind dst; int ddim; Kind a, b; };
static bool role_stepbase(uint64_t op, RoleB &r)
{
    if (op <= 83 && op != 69) {
        Role q;
        if (!role_step42ns(op, q)) return false;
        r = {q.cls, q.dst, q.dst == K_T3 ? 3 : 1, q.a, q.b};
        return true;
    }
#define R(c, d, n, x, y) r = {c, d, n, x, y}; return true
    switch (op) {
    case 84: R(C_ADD, K_T1, 1, K_T1, K_POLS);       case 85: R(C_MUL, K_T1, 1, K_POLS, K_NUM);
    case 86: R(C_ADD, K_DPOL, 1, K_T1, K_T1);       case 87: R(C_ADD, K_DPOL, 1, K_T1, K_POL);      case 88: R(C_ADD, K_DPOL, 3, K_T1, K_T3);
    case 89: R(C_ADD, K_DPOL, 3, K_POL3, K_T3);     case 90: R(C_ADD, K_DPOL, 3, K_T3, K_CHAL);     case 92: R(C_SUB, K_DPOL, 1, K_T1, K_T1);
    case 93: R(C_SUB, K_DPOL, 1, K_NUM, K_T1);      case 94: R(C_MUL, K_DPOL, 1, K_T1, K_T1);       case 95: R(C_MUL, K_DPOL, 1, K_POL, K_T1);
    case 96: R(C_MUL, K_DPOL, 1, K_T1, K_CONST);    case 98: R(C_MUL, K_DPOL, 3, K_T3, K_T3);       case 100: R(C_COPY, K_DPOL, 1, K_T1, K_NONE);
    case 101: R(C_ADD, K_DPOLS, 1, K_T1, K_T1);     case 102: R(C_ADD, K_DPOLS, 1, K_T1, K_POL);    case 103: R(C_ADD, K_DPOLS, 3, K_T1, K_T3);
    case 104: R(C_ADD, K_DPOLS, 3, K_POL3, K_T3);   case 105: R(C_ADD, K_DPOLS, 3, K_T3, K_CHAL);   case 106: R(C_SUB, K_DPOLS, 1, K_T1, K_T1);
    case 107: R(C_SUB, K_DPOLS, 1, K_NUM, K_T1);    case 108: R(C_MUL, K_DPOLS, 1, K_T1, K_T1);     case 109: R(C_MUL, K_DPOLS, 1, K_POL, K_T1);
    case 110: R(C_MUL, K_DPOLS, 1, K_T1, K_CONST);  case 111: R(C_MUL, K_DPOLS, 1, K_CONSTS, K_T1); case 112: R(C_MUL, K_DPOLS, 3, K_T3, K_T3);
    case 113: R(C_COPY, K_DPOLS, 1, K_T1, K_NONE);  case 114: R(C_ADD, K_DPOLS, 1, K_T1, K_POLS);
    }
#undef R
    return false;
}
static const std::vector<uint64_t> *fused_stepbase(uint64_t op)
{
    static const std::vector<uint64_t> f115 = {0, 50};
    return op == 115 ? &f115 : nullptr;
}
static bool is_base_step(int step) { return step == MI_CHELPERS_STEP2PREV || step == MI_CHELPERS_STEP3PREV || step == MI_CHELPERS_STEP3; }

} // namespace chp


namespace chp {


// ---- the kernel
__device__ __forceinline__ u64 lds_word(const char *lds, uint32_t field, uint32_t lane8)
{
    return *(const u64 *)(lds + (field & ~LANE_FLAG) + ((field & LANE_FLAG) ? lane8 : 0));
}

__global__ __launch_bounds__(64) void k_chelpers(const GInstr *__restrict__ prog, const GArgs P)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const uint32_t lane = threadIdx.x, lane8 = lane * 8;
    u64 *tile = (u64 *)(lds + P.tile_off);   // [TILE_ROWS][65], only while staging (aliases hot temporaries + prefetch slots)
    u64 *pre = (u64 *)(lds + P.pre_off);     // [2 * BATCH][lane]
    u64 *cst = (u64 *)(lds + P.cst_off);     // challenges, public inputs, ZhInv: one word for all lanes
    u64 *mycol = P.scratch + (uint64_t)blockIdx.x * P.wg_stride; // this workgroup's staged columns, then its cold temporaries
    u64 *mycold = mycol + P.staged_cols * RS;
    const uint32_t *progw = (const uint32_t *)prog; // 16 dwords per instruction
    *(u64 *)(lds + lane8) = 0;                      // the zero word
    for (uint32_t i = lane; i < P.cst_words; i += 64) cst[i] = P.consts[i];

    for (uint64_t g = blockIdx.x; g < P.n_groups; g += gridDim.x) { // persistent: every workgroup drains its share and exits
        const uint64_t r0 = P.row0 + g * 64;
        const uint64_t row = r0 + lane;
        const bool active = row < P.row_end;
        // ---- stage: rows r0 .. r0 + RS - 1 (wrapping at the section's row count) of every section, transposed
        for (uint32_t si = 0; si < P.n_sections; si++) {
            const GSection S = P.sec[si];
            for (uint32_t c0 = 0; c0 < S.ncols; c0 += 64) {
                const uint32_t c = c0 + lane;
                const uint32_t ncol = S.ncols - c0 < 64 ? S.ncols - c0 : 64;
                for (uint32_t rb = 0; rb < RS; rb += TILE_ROWS) {
                    __syncthreads(); // the tile (and, first time round, the temporaries it aliases) is free
                    if (c < S.ncols) {
#pragma unroll 8
                        for (uint32_t rr = 0; rr < TILE_ROWS; rr++) { // a row's 64 columns: one contiguous 512-byte read per wave
                            uint64_t rw = r0 + rb + rr;
                            while (rw >= S.nrows) rw -= S.nrows; // at most the halo past the end (a division here is 60 scalar instructions)
                            tile[rr * 65 + lane] = gl::canon(S.ptr[rw * S.pitch + c]);
                        }
                    }
                    __syncthreads();
                    if (lane < TILE_ROWS)
                        for (uint32_t cc = 0; cc < ncol; cc++) // a column's 24 rows: a contiguous 192-byte run
                            mycol[(uint64_t)(S.col0 + c0 + cc) * RS + rb + lane] = tile[lane * 65 + cc];
                }
            }
        }
        // the wave reads back what it just stored: the stores must have left the wave (the reads bypass its L1: agent scope)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();

        // ---- interpret.  Software pipeline over batches of BATCH instructions: while batch b executes, the prefetch values of
        // batch b + 1 and the instruction words of batch b + 2 are in flight.  A batch's 8 x 64-byte instructions are two
        // coalesced vector loads (lane l holds dwords l and 64 + l of the batch) and their fields are read back with v_readlane:
        // fetching them through the scalar cache cost a dependent L2 round trip per field (an early version of this loop).
        const uint32_t n_batches = (P.n_instr + BATCH - 1) / BATCH;
        const u64 zh_val = cst[P.zh_off + ((active ? row : r0) & (P.n_zhinv - 1))]; // zi.zhInv(i) of this lane's row (table size 2^k)
        struct Words { uint32_t lo, hi; }; // dwords lane and 64 + lane of a batch: instruction j field f at dword 16 j + f
        auto load_words = [&](uint32_t bi) -> Words { // the program buffer is zero-padded by three batches
            const uint32_t *p = progw + (uint64_t)bi * (BATCH * 16);
            return {p[lane], p[64 + lane]};
        };
        auto issue_prefetch = [&](const Words &w, u64 (&v)[2 * BATCH]) {
#pragma unroll
            for (int j = 0; j < BATCH; j++) {
                const uint32_t op = (uint32_t)__builtin_amdgcn_readlane((int)(j < 4 ? w.lo : w.hi), (j & 3) * 16);
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++) {
                    const uint32_t kind = (op >> (8 + 4 * s2)) & 7;
                    const int fl = (j & 3) * 16 + 8 + 2 * s2;
                    const uint64_t imm = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(j < 4 ? w.lo : w.hi), fl + 1) << 32) |
                                         (uint32_t)__builtin_amdgcn_readlane((int)(j < 4 ? w.lo : w.hi), fl);
                    // P_STAGED and P_COLD are the same load (the spill follows the staged columns in the workgroup's scratch):
                    // agent scope, because both were written by this wave a moment ago and its L1 may hold older lines
                    u64 val = kind == P_IMM ? imm : zh_val;
                    if (kind == P_STAGED) val = __hip_atomic_load(mycol + imm + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    v[2 * j + s2] = val;
                }
            }
        };
        Words iw_cur = load_words(0), iw_nxt = load_words(1);
        u64 v_cur[2 * BATCH], v_nxt[2 * BATCH];
        issue_prefetch(iw_cur, v_cur);
        for (uint32_t bi = 0; bi < n_batches; bi++) {
            issue_prefetch(iw_nxt, v_nxt);                 // batch bi + 1: 16 loads in flight during this batch's arithmetic
            const Words iw_nn = load_words(bi + 2);        // batch bi + 2's instruction words
#pragma unroll
            for (int j = 0; j < 2 * BATCH; j++) pre[j * 64 + lane] = v_cur[j];
#pragma unroll 1
            for (uint32_t j = 0; j < BATCH; j++) {
                // instruction j of the batch lives in dwords 16 j .. 16 j + 15: lanes of `lo` for j < 4, of `hi` otherwise
                const uint32_t wsel = j < 4 ? iw_cur.lo : iw_cur.hi, fbase = (j & 3) * 16;
                auto fld = [&](uint32_t f) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)wsel, (int)(fbase + f)); };
                const uint32_t op = fld(0), dst = fld(1);
                const uint32_t cls = op & 15, dk = (op >> 4) & 3;
                u64 o[3];
                const u64 a0 = lds_word(lds, fld(2), lane8), b0 = lds_word(lds, fld(5), lane8);
                bool three = true;
                switch (cls) {
                case G_ADD1: o[0] = gl::add(a0, b0); three = false; break;
                case G_SUB1: o[0] = gl::sub(a0, b0); three = false; break;
                case G_MUL11: o[0] = gl::mul(a0, b0); three = false; break;
                case G_COPY1: o[0] = a0; three = false; break;
                case G_MUL13: { // base times extension
                    const u64 b1 = lds_word(lds, fld(6), lane8), b2 = lds_word(lds, fld(7), lane8);
                    o[0] = gl::mul(a0, b0); o[1] = gl::mul(a0, b1); o[2] = gl::mul(a0, b2);
                    break;
                }
                case G_MUL31: {
                    const u64 a1 = lds_word(lds, fld(3), lane8), a2 = lds_word(lds, fld(4), lane8);
                    o[0] = gl::mul(a0, b0); o[1] = gl::mul(a1, b0); o[2] = gl::mul(a2, b0);
                    break;
                }
                default: {
                    const u64 a1 = lds_word(lds, fld(3), lane8), a2 = lds_word(lds, fld(4), lane8);
                    const u64 b1 = lds_word(lds, fld(6), lane8), b2 = lds_word(lds, fld(7), lane8);
                    if (cls == G_ADD3) { o[0] = gl::add(a0, b0); o[1] = gl::add(a1, b1); o[2] = gl::add(a2, b2); }
                    else if (cls == G_SUB3) { o[0] = gl::sub(a0, b0); o[1] = gl::sub(a1, b1); o[2] = gl::sub(a2, b2); }
                    else if (cls == G_MUL33) {
                        const E3 p = gl::e3_mul(E3{{a0, a1, a2}}, E3{{b0, b1, b2}});
                        o[0] = p.v[0]; o[1] = p.v[1]; o[2] = p.v[2];
                    } else { o[0] = a0; o[1] = a1; o[2] = a2; } // G_COPY3
                    break;
                }
                }
                if (dk == D_HOT) {
                    *(u64 *)(lds + dst + lane8) = o[0];
                    if (three) { *(u64 *)(lds + dst + 512 + lane8) = o[1]; *(u64 *)(lds + dst + 1024 + lane8) = o[2]; }
                    if (op & 64) // read again much later: a second home in the spill (word in field 12)
                        __hip_atomic_store(mycold + (uint64_t)fld(12) * 64 + lane, o[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else if (dk == D_COLD) { // a base-field value that lives long and is read rarely: spilled beside the staged columns
                    __hip_atomic_store(mycold + (uint64_t)dst * 64 + lane, o[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else if (dk == D_Q && active) {
                    P.q[row * 3] = o[0]; P.q[row * 3 + 1] = o[1]; P.q[row * 3 + 2] = o[2];
                }
            }
            iw_cur = iw_nxt;
            iw_nxt = iw_nn;
#pragma unroll
            for (int j = 0; j < 2 * BATCH; j++) v_cur[j] = v_nxt[j];
        }
    }
}

// ------------------------------------------------------------------ translator
static int decode(int step, const uint64_t *ops, uint64_t nops, const uint64_t *args, uint64_t nargs, std::vector<MicroOp> &out,
                  uint64_t &max_chal, uint64_t &max_pub, uint64_t &max_eval)
{
    MI_REQUIRE(step == MI_CHELPERS_STEP42NS || step == MI_CHELPERS_STEP52NS || is_base_step(step), "unknown step: the opcode numberings of step2prev / step3prev / step3, step42ns and step52ns are known to this build");
    uint64_t ia = 0;
    auto take = [&](Kind k, HOpd &o) -> bool {
        o.k = k;
        const int n = kind_nargs(k);
        if (ia + n > nargs) return false;
        for (int i = 0; i < n; i++) o.v[i] = args[ia + i];
        ia += n;
        if (k == K_NUM) o.v[0] = gl::canon(o.v[0]); // Goldilocks::fromU64
        if (k == K_CHAL) max_chal = std::max(max_chal, o.v[0] + 1);
        if (k == K_PUB) max_pub = std::max(max_pub, o.v[0] + 1);
        if (k == K_EVAL) max_eval = std::max(max_eval, o.v[0] + 1);
        return true;
    };
    auto one52 = [&](uint64_t op) -> int { // registers and challenge indices are fixed by the opcode, only polynomials take arguments
        Role52 r;
        if (!role_step52ns(op, r)) {
            mi_set_error("mi_chelpers_compile: unknown opcode %llu", (unsigned long long)op);
            return MI_ERR_INVALID;
        }
        MicroOp m;
        m.cls = r.cls;
        m.dst = r.cls == C_STOREF ? K_Q : K_T3;
        m.dst_slot = r.dst;
        const Kind ks[2] = {r.a, r.b};
        const uint64_t fixed[2] = {r.a_fixed, r.b_fixed};
        HOpd *os[2] = {&m.a, &m.b};
        for (int s2 = 0; s2 < 2; s2++) {
            HOpd &o = *os[s2];
            o.k = ks[s2];
            switch (ks[s2]) {
            case K_T3: o.v[0] = fixed[s2]; break;
            case K_CHAL: o.v[0] = fixed[s2]; max_chal = std::max(max_chal, fixed[s2] + 1); break;
            case K_POL: case K_POL3:
                if (ia + 2 > nargs) { mi_set_error("mi_chelpers_compile: argument table too short"); return MI_ERR_INVALID; }
                o.v[0] = args[ia]; o.v[1] = args[ia + 1]; ia += 2;
                break;
            case K_CONST: case K_EVAL:
                if (op == 14) { o.v[0] = fixed[s2]; } // both literals in the reference (const column 5, evals[0])
                else {
                    if (ia + 1 > nargs) { mi_set_error("mi_chelpers_compile: argument table too short"); return MI_ERR_INVALID; }
                    o.v[0] = args[ia++];
                }
                if (ks[s2] == K_EVAL) max_eval = std::max(max_eval, o.v[0] + 1);
                break;
            default: break; // K_XD, K_XDW, K_NONE
            }
        }
        out.push_back(m);
        return MI_OK;
    };
    if (step == MI_CHELPERS_STEP52NS) {
        for (uint64_t k = 0; k < nops; k++) {
            if (const std::vector<uint64_t> *f = fused_step52ns(ops[k])) {
                for (uint64_t sub : *f) MI_TRY(one52(sub));
            } else {
                MI_TRY(one52(ops[k]));
            }
        }
        if (ia != nargs) {
            mi_set_error("mi_chelpers_compile: program consumes %llu arguments, table holds %llu", (unsigned long long)ia, (unsigned long long)nargs);
            return MI_ERR_INVALID;
        }
        return MI_OK;
    }
    // base-domain steps: an opcode whose result goes to a polynomial becomes "fresh temp = a op b" + "store the temp" (C_STOREP, the
    // destination in operand b); a later read of exactly that polynomial element -- same offset, stride, shift and dimension: every
    // such read in the zkEVM programs -- is the temp, so no instruction depends on polynomial memory written by the program itself
    uint64_t fresh = 1ull << 40;
    std::map<std::array<uint64_t, 4>, HOpd> stored; // (offset, stride, shift, dim) -> the temp that was stored there
    std::set<std::array<uint64_t, 3>> read_elems;   // (element offset, stride, shift) read from polynomial memory so far
    auto forward = [&](HOpd &o) -> int {
        if (o.k != K_POL && o.k != K_POL3 && o.k != K_POLS && o.k != K_POL3S) return MI_OK;
        const bool sh = o.k == K_POLS || o.k == K_POL3S;
        const uint64_t dim = (o.k == K_POL3 || o.k == K_POL3S) ? 3 : 1, stride = sh ? o.v[3] : o.v[1], shift = sh ? o.v[1] : 0;
        auto it = stored.find({o.v[0], stride, shift, dim});
        if (it != stored.end()) { o = it->second; return MI_OK; }
        for (uint64_t j = 0; j < dim; j++) read_elems.insert({o.v[0] + j, stride, shift});
        for (auto &kv : stored) // anything else that overlaps a stored element would need the stored value from memory
            if (kv.first[1] == stride && o.v[0] < kv.first[0] + kv.first[3] && kv.first[0] < o.v[0] + dim) {
                mi_set_error("mi_chelpers_compile: a polynomial element the program has written is read back with another shift or dimension");
                return MI_ERR_INVALID;
            }
        return MI_OK;
    };
    auto oneb = [&](uint64_t op) -> int {
        RoleB r;
        if (!role_stepbase(op, r)) {
            mi_set_error("mi_chelpers_compile: unknown opcode %llu", (unsigned long long)op);
            return MI_ERR_INVALID;
        }
        MicroOp m;
        m.cls = r.cls;
        HOpd dst;
        bool ok = true;
        if (r.dst == K_T1 || r.dst == K_T3) {
            ok = ia < nargs;
            m.dst = r.dst;
            m.dst_slot = ok ? args[ia++] : 0;
        } else {
            ok = take(r.dst, dst);
            m.dst = r.ddim == 3 ? K_T3 : K_T1;
            m.dst_slot = fresh++;
        }
        ok = ok && take(r.a, m.a);
        if (ok && r.b != K_NONE) ok = take(r.b, m.b);
        if (!ok) {
            mi_set_error("mi_chelpers_compile: argument table too short");
            return MI_ERR_INVALID;
        }
        MI_TRY(forward(m.a));
        MI_TRY(forward(m.b));
        out.push_back(m);
        if (r.dst == K_DPOL || r.dst == K_DPOLS) {
            MicroOp st;
            st.cls = C_STOREP;
            st.dst = r.dst;
            st.dst_slot = 0;
            st.a.k = m.dst;
            st.a.v[0] = m.dst_slot;
            st.b = dst;
            out.push_back(st);
            const bool sh = r.dst == K_DPOLS;
            // (at ANY row shift: on the device another row's store would race with that read; in the reference's row-by-row loop it sees
            // what an earlier or a later iteration left there -- no generated program does it, host/steps_tracer.hpp refuses it as well)
            for (int j = 0; j < r.ddim; j++) {
                const uint64_t elem = dst.v[0] + (uint64_t)j, stride = sh ? dst.v[3] : dst.v[1];
                auto it = read_elems.lower_bound({elem, stride, 0});
                if (it != read_elems.end() && (*it)[0] == elem && (*it)[1] == stride) {
                    mi_set_error("mi_chelpers_compile: the program overwrites a polynomial element it has read before");
                    return MI_ERR_INVALID;
                }
            }
            stored[{dst.v[0], sh ? dst.v[3] : dst.v[1], sh ? dst.v[1] : 0, (uint64_t)r.ddim}] = st.a;
        }
        return MI_OK;
    };
    if (is_base_step(step)) {
        for (uint64_t k = 0; k < nops; k++) {
            if (const std::vector<uint64_t> *f = fused_stepbase(ops[k])) {
                for (uint64_t sub : *f) MI_TRY(oneb(sub));
            } else {
                MI_TRY(oneb(ops[k]));
            }
        }
        if (ia != nargs) {
            mi_set_error("mi_chelpers_compile: program consumes %llu arguments, table holds %llu", (unsigned long long)ia, (unsigned long long)nargs);
            return MI_ERR_INVALID;
        }
        return MI_OK;
    }
    auto one = [&](uint64_t op) -> int {
        Role r;
        if (!role_step42ns(op, r)) {
            mi_set_error("mi_chelpers_compile: unknown opcode %llu", (unsigned long long)op);
            return MI_ERR_INVALID;
        }
        MicroOp m;
        m.cls = r.cls;
        m.dst = r.dst;
        m.dst_slot = 0;
        bool ok = true;
        if (r.cls != C_STOREQ) { // argument 0 = destination temp
            ok = ia < nargs;
            if (ok) m.dst_slot = args[ia++];
        }
        ok = ok && take(r.a, m.a);
        if (ok && r.b != K_NONE && r.b != K_ZHINV) ok = take(r.b, m.b);
        if (r.b == K_ZHINV) m.b.k = K_ZHINV;
        if (!ok) {
            mi_set_error("mi_chelpers_compile: argument table too short");
            return MI_ERR_INVALID;
        }
        out.push_back(m);
        return MI_OK;
    };
    for (uint64_t k = 0; k < nops; k++) {
        if (const std::vector<uint64_t> *f = fused_step42ns(ops[k])) {
            for (uint64_t sub : *f) MI_TRY(one(sub));
        } else {
            MI_TRY(one(ops[k]));
        }
    }
    if (ia != nargs) { // the reference asserts i_args == NARGS_ after every row (parser.cpp:755)
        mi_set_error("mi_chelpers_compile: program consumes %llu arguments, table holds %llu", (unsigned long long)ia, (unsigned long long)nargs);
        return MI_ERR_INVALID;
    }
    return MI_OK;
}

static uint64_t live_words(const std::vector<MicroOp> &p)
{
    // peak of (live base temps + 3 * live ext temps) over the program as written
    std::map<std::pair<int, uint64_t>, std::pair<uint64_t, uint64_t>> cur; // (kind, slot) -> (def position, last use)
    std::vector<std::pair<uint64_t, int>> ev;
    auto close = [&](std::pair<int, uint64_t> key) {
        auto it = cur.find(key);
        if (it == cur.end()) return;
        const int w = key.first == K_T3 ? 3 : 1;
        ev.push_back({it->second.first * 2 + 1, w});
        ev.push_back({it->second.second * 2 + 2, -w});
        cur.erase(it);
    };
    for (uint64_t pos = 0; pos < p.size(); pos++) {
        for (const HOpd *o : {&p[pos].a, &p[pos].b})
            if (o->k == K_T1 || o->k == K_T3) {
                auto it = cur.find({o->k, o->v[0]});
                if (it != cur.end()) it->second.second = pos;
            }
        if (p[pos].dst == K_T1 || p[pos].dst == K_T3) {
            close({p[pos].dst, p[pos].dst_slot});
            cur[{p[pos].dst, p[pos].dst_slot}] = {pos, pos};
        }
    }
    while (!cur.empty()) close(cur.begin()->first);
    std::sort(ev.begin(), ev.end());
    int64_t c = 0, m = 0;
    for (auto &e : ev) { c += e.second; m = std::max(m, c); }
    return (uint64_t)m;
}

static int translate(mi_chelpers_prog *P, std::vector<MicroOp> &prog)
{
    P->stats[1] = prog.size();
    P->stats[4] = live_words(prog);
    // ---- (1) copy forwarding: a base temp defined by a copy of something that is not a temp is an alias of it
    {
        std::map<uint64_t, HOpd> alias; // T1 slot -> source descriptor
        std::vector<MicroOp> out;
        out.reserve(prog.size());
        for (MicroOp m : prog) {
            for (HOpd *o : {&m.a, &m.b})
                if (o->k == K_T1) {
                    auto it = alias.find(o->v[0]);
                    if (it != alias.end()) *o = it->second;
                }
            if (m.cls == C_COPY && m.dst == K_T1 && m.a.k != K_T1 && m.a.k != K_T3) {
                alias[m.dst_slot] = m.a;
                continue;
            }
            if (m.dst == K_T1) alias.erase(m.dst_slot);
            out.push_back(m);
        }
        prog.swap(out);
    }
    P->stats[2] = prog.size();
    // ---- (2) SSA: instruction i defines value i; sources become value ids
    const size_t n = prog.size();
    std::vector<std::array<int64_t, 2>> src(n, {-1, -1});
    {
        std::map<std::pair<int, uint64_t>, int64_t> cur;
        for (size_t i = 0; i < n; i++) {
            HOpd *os[2] = {&prog[i].a, &prog[i].b};
            for (int s = 0; s < 2; s++)
                if (os[s]->k == K_T1 || os[s]->k == K_T3) {
                    auto it = cur.find({os[s]->k, os[s]->v[0]});
                    if (it == cur.end()) {
                        mi_set_error("mi_chelpers_compile: temporary read before it is written (instruction %zu)", i);
                        return MI_ERR_INVALID;
                    }
                    src[i][s] = it->second;
                }
            if (prog[i].dst == K_T1 || prog[i].dst == K_T3) cur[{prog[i].dst, prog[i].dst_slot}] = (int64_t)i;
        }
    }
    auto weight = [&](size_t i) { return prog[i].dst == K_T3 ? 3 : prog[i].dst == K_T1 ? 1 : 0; };
    // ---- (3) Sethi-Ullman numbers (the DAG treated as a tree) and a depth-first order from the stores
    std::vector<uint32_t> need(n, 0);
    for (size_t i = 0; i < n; i++) {
        int64_t s0 = src[i][0], s1 = src[i][1];
        if (s1 >= 0 && (s0 < 0 || need[s1] > need[s0])) std::swap(s0, s1); // s0 = the hungrier operand, evaluated first
        uint32_t m = (uint32_t)weight(i), held = 0;
        if (s0 >= 0) { m = std::max(m, need[s0]); held += weight(s0); }
        if (s1 >= 0 && s1 != s0) { m = std::max(m, need[s1] + held); held += weight(s1); }
        need[i] = std::max(m, held);
    }
    std::vector<uint32_t> order;
    order.reserve(n);
    {
        std::vector<uint8_t> done(n, 0);
        std::vector<std::pair<size_t, int>> stack;
        for (size_t root = 0; root < n; root++) {
            if (prog[root].cls != C_STOREQ && prog[root].cls != C_STOREF && prog[root].cls != C_STOREP) continue; // only stores have effects: the rest is reached from them or dead
            stack.push_back({root, 0});
            while (!stack.empty()) {
                auto [node, state] = stack.back();
                stack.pop_back();
                if (done[node]) continue;
                if (state == 0) {
                    stack.push_back({node, 1});
                    int64_t s0 = src[node][0], s1 = src[node][1];
                    if (s1 >= 0 && (s0 < 0 || need[s1] > need[s0])) std::swap(s0, s1);
                    if (s1 >= 0 && !done[s1]) stack.push_back({(size_t)s1, 0}); // popped second
                    if (s0 >= 0 && !done[s0]) stack.push_back({(size_t)s0, 0}); // popped first: the hungrier operand
                } else {
                    done[node] = 1;
                    order.push_back((uint32_t)node);
                }
            }
        }
    }
    P->stats[3] = order.size();
    // ---- (4) linear-scan slot allocation over the new order; a destination may take over a source that dies here
    std::vector<uint32_t> uses(n, 0);
    for (uint32_t i : order)
        for (int s = 0; s < 2; s++)
            if (src[i][s] >= 0 && (s == 0 || src[i][1] != src[i][0])) uses[src[i][s]]++;
    std::vector<uint64_t> word(n, 0);
    std::vector<uint64_t> free1, free3;
    uint64_t n1 = 0, n3 = 0;
    // base temps live in words [0, n1), ext temps in 3-word groups after them: the split is fixed up after the scan
    std::vector<std::pair<uint32_t, uint64_t>> t3_fix; // (instruction, ext slot)
    std::vector<uint64_t> slot(n, 0);
    uint64_t live = 0, peak = 0;
    for (uint32_t i : order) {
        for (int s = 0; s < 2; s++) {
            const int64_t v = src[i][s];
            if (v < 0 || (s == 1 && v == src[i][0])) continue;
            if (--uses[v] == 0) {
                (prog[v].dst == K_T3 ? free3 : free1).push_back(slot[v]);
                live -= weight(v);
            }
        }
        if (prog[i].dst == K_T1 || prog[i].dst == K_T3) {
            std::vector<uint64_t> &fr = prog[i].dst == K_T3 ? free3 : free1;
            uint64_t &cnt = prog[i].dst == K_T3 ? n3 : n1;
            if (fr.empty()) slot[i] = cnt++;
            else { slot[i] = fr.back(); fr.pop_back(); }
            live += weight(i);
            peak = std::max(peak, live);
        }
    }
    P->stats[5] = peak;
    P->stats[6] = n1;
    P->stats[7] = n3;
    P->n_words = n1 + 3 * n3;
    auto word_of = [&](size_t v) { return prog[v].dst == K_T3 ? n1 + 3 * slot[v] : slot[v]; };
    // ---- (5) emit
    P->host.clear();
    P->host.reserve(order.size());
    for (uint32_t i : order) {
        const MicroOp &m = prog[i];
        DInstr d = {};
        d.op = (uint32_t)m.cls | ((uint32_t)m.dst << 8) | ((uint32_t)m.a.k << 16) | ((uint32_t)m.b.k << 24);
        d.dst = (m.dst == K_T1 || m.dst == K_T3) ? (uint32_t)word_of(i) : 0;
        const HOpd *hs[2] = {&m.a, &m.b};
        Opd *ds[2] = {&d.a, &d.b};
        for (int s = 0; s < 2; s++) {
            const HOpd &h = *hs[s];
            Opd &o = *ds[s];
            switch (h.k) {
            case K_T1: case K_T3: o.off = word_of((size_t)src[i][s]); break;
            case K_POL: case K_POL3: case K_DPOL:
                MI_REQUIRE(h.v[1] < (1ull << 32), "row stride too large");
                o.off = h.v[0]; o.stride = (uint32_t)h.v[1];
                break;
            case K_POLS: case K_POL3S: case K_DPOLS:
                MI_REQUIRE(h.v[3] < (1ull << 32) && h.v[1] < (1ull << 32) && h.v[2] != 0, "bad shifted-row operand");
                o.off = h.v[0]; o.shift = (uint32_t)h.v[1]; o.mod = h.v[2]; o.stride = (uint32_t)h.v[3];
                break;
            case K_CONSTS:
                MI_REQUIRE(h.v[1] < (1ull << 32) && h.v[2] != 0, "bad shifted-row operand");
                o.off = h.v[0]; o.shift = (uint32_t)h.v[1]; o.mod = h.v[2];
                break;
            default: o.off = h.v[0]; break; // NUM value, CONST column, CHAL / PUB index; nothing for X / ZHINV / NONE
            }
        }
        P->host.push_back(d);
    }
    return MI_OK;
}

// ---- the kernel's form of the program.
//   * every polynomial operand becomes a staged column (+ row shift) of one of the sections; a dimension-3 polynomial operand
//     is first copied, word by word, into an extension temporary;
//   * temporaries are re-allocated: extension values and short-lived base values in LDS ("hot"), base values that live long
//     and are read rarely in a per-workgroup spill in HBM ("cold", fetched through the prefetch slots one batch ahead, so a
//     cold value's first read must lie more than three batches after its definition);
//   * every operand is resolved to LDS address fields (see GInstr).
// LDS layout (bytes): [0,512) the zero word | hot base temps | hot extension temps (3 words each) | prefetch slots | constants.
static constexpr uint32_t HOT_T1_DEFAULT = 24; // hot base-field words (raised when the program's short-lived values need more)
static constexpr uint32_t HOT_SPAN = 64;        // a base value read for the last time within this many instructions stays hot

static int build_staged(mi_chelpers_prog *P, uint64_t n_zhinv_max)
{
    struct Val { int dim; int64_t def = -1, first = -1, last = -1; uint32_t nreads = 0; int loc = 0; uint32_t where = 0; }; // loc 1 hot, 2 cold
    struct GOpd { int kind = 0; int64_t val = -1; uint64_t imm = 0; int dim = 1; uint32_t cst = 0; }; // kind: 0 none, 1 value, 2 staged, 3 imm, 4 challenge, 5 public, 6 zhinv
    struct GOp { uint32_t cls; int dk; int64_t dval; GOpd a, b; };
    std::vector<Val> vals;
    std::vector<GOp> ops;
    auto find = [&](int role, uint64_t off, uint64_t stride, uint64_t width, uint32_t &col) -> const HostSection * {
        for (const HostSection &S : P->sections) {
            if (S.role != role) continue;
            if (role == 0 && (stride != S.ncols || off < S.offset || off - S.offset + width > S.ncols)) continue;
            if (role != 0 && off + width > S.ncols) continue;
            col = S.col0 + (uint32_t)(role == 0 ? off - S.offset : off);
            return &S;
        }
        return nullptr;
    };
    std::map<uint64_t, int64_t> cur; // host temp word -> value id (the host program re-uses words)
    auto new_val = [&](int dim) { vals.push_back(Val{dim}); return (int64_t)vals.size() - 1; };
    for (const DInstr &d : P->host) {
        const uint32_t hcls = d.op & 255, hdk = (d.op >> 8) & 255;
        GOpd go[2];
        const Opd *os[2] = {&d.a, &d.b};
        for (int s2 = 0; s2 < 2; s2++) {
            const uint32_t k = (d.op >> (16 + 8 * s2)) & 255;
            const Opd &o = *os[s2];
            GOpd &g = go[s2];
            uint32_t col = 0;
            const HostSection *S = nullptr;
            switch (k) {
            case K_T1: case K_T3: {
                auto it = cur.find(o.off);
                MI_REQUIRE(it != cur.end(), "internal: temporary without a definition");
                g.kind = 1; g.val = it->second; g.dim = k == K_T3 ? 3 : 1;
                break;
            }
            case K_NUM: g.kind = 3; g.imm = o.off; break;
            case K_CHAL: g.kind = 4; g.cst = (uint32_t)o.off; g.dim = 3; break;
            case K_PUB: g.kind = 5; g.cst = (uint32_t)o.off; break;
            case K_ZHINV: g.kind = 6; break;
            case K_EVAL: { // an extension constant of the running proof: three immediates patched in before every launch
                const int64_t v3 = new_val(3);
                for (int w = 0; w < 3; w++) {
                    GOp c = {};
                    c.cls = G_COPY1; c.dk = D_HOT; c.dval = v3;
                    c.a.kind = 7; c.a.cst = (uint32_t)o.off; c.a.imm = (uint64_t)w; // kind 7: evals[cst][imm]
                    c.b.kind = 0; c.b.imm = (uint64_t)w;
                    ops.push_back(c);
                }
                g.kind = 1; g.val = v3; g.dim = 3;
                break;
            }
            case K_POL: case K_POLS: case K_POL3: case K_POL3S: case K_CONST: case K_CONSTS: case K_X: case K_XD: case K_XDW: {
                const bool three = k == K_POL3 || k == K_POL3S || k == K_XD || k == K_XDW, shifted = k == K_POLS || k == K_POL3S || k == K_CONSTS;
                const int role = (k == K_CONST || k == K_CONSTS) ? 1 : k == K_X ? 2 : k == K_XD ? 3 : k == K_XDW ? 4 : 0;
                S = find(role, role >= 2 ? 0 : o.off, o.stride, three ? 3 : 1, col);
                if (!S) {
                    mi_set_error("mi_chelpers_compile: operand (offset %llu, stride %u) lies in none of the declared sections / constant polynomials / x",
                                 (unsigned long long)o.off, o.stride);
                    return MI_ERR_INVALID;
                }
                if (shifted) MI_REQUIRE(o.mod == S->nrows && o.shift <= HALO, "shifted-row operand: modulus must be the section's row count, shift at most 8");
                const uint64_t elem = (uint64_t)col * RS + (shifted ? o.shift : 0);
                if (!three) { g.kind = 2; g.imm = elem; break; }
                const int64_t v3 = new_val(3); // three staged columns -> one extension temporary, a word per copy
                for (int w = 0; w < 3; w++) {
                    GOp c = {};
                    c.cls = G_COPY1; c.dk = D_HOT; c.dval = v3;
                    c.a.kind = 2; c.a.imm = elem + (uint64_t)w * RS;
                    c.b.kind = 0; c.b.imm = (uint64_t)w; // b.imm: which word of the destination
                    ops.push_back(c);
                }
                g.kind = 1; g.val = v3; g.dim = 3;
                break;
            }
            default: break;
            }
        }
        GOp op = {};
        op.a = go[0]; op.b = go[1];
        const bool a3 = go[0].dim == 3, b3 = go[1].dim == 3;
        const int rdim = (hdk == K_T3 || hdk == K_Q) ? 3 : 1;
        switch (hcls) {
        case C_ADD: op.cls = rdim == 3 ? G_ADD3 : G_ADD1; break;
        case C_SUB: op.cls = rdim == 3 ? G_SUB3 : G_SUB1; break;
        case C_MUL: case C_STOREQ: op.cls = a3 && b3 ? G_MUL33 : a3 ? G_MUL31 : b3 ? G_MUL13 : G_MUL11; break;
        default: op.cls = rdim == 3 ? G_COPY3 : G_COPY1; break; // C_COPY, C_STOREF
        }
        if (hdk == K_T1 || hdk == K_T3) {
            op.dk = D_HOT; // decided below
            op.dval = new_val(rdim);
            cur[d.dst] = op.dval;
            // a T3 destination occupies host words dst .. dst + 2: they belong to this value now
        } else if (hdk == K_Q) { op.dk = D_Q; op.dval = -1; }
        ops.push_back(op);
    }
    // ---- lifetimes.  A base value read again long after its definition gets a COLD home (written at the definition, read
    // through the prefetch slots); the reads that follow the definition closely -- closer than the prefetch runs ahead, or
    // simply soon -- are served from a HOT home that is released after the last of them.
    struct Life { int64_t def = -1, last = -1, early_last = -1; bool late = false; };
    std::vector<Life> life(vals.size());
    for (size_t i = 0; i < ops.size(); i++)
        if (ops[i].dval >= 0 && life[ops[i].dval].def < 0) life[ops[i].dval].def = (int64_t)i;
    for (size_t i = 0; i < ops.size(); i++)
        for (const GOpd *o : {&ops[i].a, &ops[i].b})
            if (o->kind == 1) {
                Life &L = life[o->val];
                L.last = (int64_t)i;
                if (vals[o->val].dim == 3 || (int64_t)i - L.def <= (int64_t)HOT_SPAN) L.early_last = (int64_t)i;
                else L.late = true;
            }
    // ---- allocation by linear scan over three pools: hot base words, hot extension groups, cold base words
    std::vector<uint32_t> free1, free3, freec;
    uint32_t n1 = 0, n3 = 0, nc = 0;
    std::vector<uint32_t> hot_where(vals.size(), 0), cold_where(vals.size(), 0);
    std::vector<uint8_t> has_hot(vals.size(), 0), has_cold(vals.size(), 0);
    std::vector<std::vector<int64_t>> hot_dies(ops.size()), cold_dies(ops.size());
    for (size_t v = 0; v < vals.size(); v++) {
        const Life &L = life[v];
        if (L.def < 0) continue;
        has_cold[v] = vals[v].dim == 1 && L.late;
        has_hot[v] = !has_cold[v] || L.early_last >= 0;
        if (has_hot[v]) hot_dies[has_cold[v] ? L.early_last : (L.last < 0 ? L.def : L.last)].push_back((int64_t)v);
        if (has_cold[v]) cold_dies[L.last].push_back((int64_t)v);
    }
    for (size_t i = 0; i < ops.size(); i++) {
        // homes whose last read is this instruction are free for its destination (operands are read before the result is written)
        for (int64_t v : hot_dies[i]) if (life[v].def < (int64_t)i) (vals[v].dim == 3 ? free3 : free1).push_back(hot_where[v]);
        for (int64_t v : cold_dies[i]) freec.push_back(cold_where[v]);
        const int64_t dv = ops[i].dval;
        if (dv < 0 || life[dv].def != (int64_t)i) continue;
        if (has_hot[dv]) {
            std::vector<uint32_t> &fr = vals[dv].dim == 3 ? free3 : free1;
            uint32_t &cnt = vals[dv].dim == 3 ? n3 : n1;
            if (fr.empty()) hot_where[dv] = cnt++; else { hot_where[dv] = fr.back(); fr.pop_back(); }
            if (life[dv].last < 0) fr.push_back(hot_where[dv]); // never read
        }
        if (has_cold[dv]) {
            if (freec.empty()) cold_where[dv] = nc++; else { cold_where[dv] = freec.back(); freec.pop_back(); }
        }
    }
    // ---- layout
    const uint32_t hot1_off = 512, hot3_off = hot1_off + 512 * n1, hot_end = hot3_off + 3 * 512 * n3;
    const uint32_t tile_bytes = TILE_ROWS * 65 * 8;
    const uint32_t pre_off = hot_end, pre_end = pre_off + 2 * BATCH * 512;
    const uint32_t region_end = std::max(pre_end, hot1_off + tile_bytes); // the staging tile aliases [512, ...)
    const uint32_t cst_off = region_end;
    const uint32_t chal_words = (uint32_t)P->max_chal * 3, pub_words = (uint32_t)P->max_pub;
    P->hot_t1 = n1; P->hot_t3 = n3; P->cold_words = nc;
    (void)HOT_T1_DEFAULT;
    P->tile_off = hot1_off; P->pre_off = pre_off; P->cst_off = cst_off;
    P->lds_fixed = cst_off + (chal_words + pub_words + (uint32_t)n_zhinv_max) * 8;
    // ---- emit
    P->gpu.clear();
    P->gpu.reserve(ops.size());
    P->eval_patches.clear();
    P->cold_reads = P->temp_reads = 0;
    const uint32_t Z = 0; // the zero word, same for all lanes
    for (size_t i = 0; i < ops.size(); i++) {
        const GOp &op = ops[i];
        GInstr g = {};
        uint32_t pk[2] = {P_NONE, P_NONE};
        uint32_t *fa[2] = {g.a, g.b};
        uint64_t *im[2] = {&g.a_imm, &g.b_imm};
        const GOpd *os[2] = {&op.a, &op.b};
        const uint32_t slot_in_batch = (uint32_t)(i % BATCH);
        for (int s2 = 0; s2 < 2; s2++) {
            const GOpd &o = *os[s2];
            uint32_t *f = fa[s2];
            const uint32_t pre_field = (pre_off + (2 * slot_in_batch + s2) * 512) | LANE_FLAG;
            f[0] = f[1] = f[2] = Z;
            switch (o.kind) {
            case 1: {
                const Val &v = vals[o.val];
                P->temp_reads++;
                if (has_cold[o.val] && (int64_t)i - life[o.val].def > (int64_t)HOT_SPAN) {
                    // a load like a staged column: the spill follows the staged columns in the workgroup's scratch
                    pk[s2] = P_STAGED; *im[s2] = P->staged_cols * RS + (uint64_t)cold_where[o.val] * 64; f[0] = pre_field; P->cold_reads++;
                    break;
                }
                const uint32_t base = v.dim == 3 ? hot3_off + 3 * 512 * hot_where[o.val] : hot1_off + 512 * hot_where[o.val];
                f[0] = base | LANE_FLAG;
                if (v.dim == 3) { f[1] = (base + 512) | LANE_FLAG; f[2] = (base + 1024) | LANE_FLAG; }
                break;
            }
            case 2: pk[s2] = P_STAGED; *im[s2] = o.imm; f[0] = pre_field; break;
            case 3: pk[s2] = P_IMM; *im[s2] = o.imm; f[0] = pre_field; break;
            case 7: // word o.imm of evals[o.cst]: an immediate whose value mi_chelpers_run_dev patches in (operand a only)
                pk[s2] = P_IMM; *im[s2] = 0; f[0] = pre_field;
                P->eval_patches.push_back({(uint32_t)i, o.cst, (uint32_t)o.imm});
                break;
            case 4: f[0] = cst_off + (o.cst * 3) * 8; f[1] = f[0] + 8; f[2] = f[0] + 16; break;
            case 5: f[0] = cst_off + (chal_words + o.cst) * 8; break;
            case 6: pk[s2] = P_ZHINV; f[0] = pre_field; break;
            default: break;
            }
        }
        uint32_t dk = (uint32_t)op.dk, also_cold = 0;
        if (op.dval >= 0) {
            const Val &v = vals[op.dval];
            if (has_hot[op.dval]) {
                dk = D_HOT;
                g.dst = v.dim == 3 ? hot3_off + 3 * 512 * hot_where[op.dval] : hot1_off + 512 * hot_where[op.dval];
                if (op.cls == G_COPY1 && v.dim == 3) g.dst += 512 * (uint32_t)op.b.imm; // one word of an extension temporary
                if (has_cold[op.dval]) { also_cold = 1; g.pad[0] = cold_where[op.dval]; }
            } else { dk = D_COLD; g.dst = cold_where[op.dval]; }
        }
        g.op = op.cls | (dk << 4) | (also_cold << 6) | (pk[0] << 8) | (pk[1] << 12);
        P->gpu.push_back(g);
    }
    return MI_OK;
}

} // namespace chp

// what mi_chelpers_compile and mi_chelpers_compile_micro share: sections, translation, the kernel's form, the device copy
static int compile_decoded(mi_ctx *c, mi_chelpers_prog **out, mi_chelpers_prog *P, std::vector<chp::MicroOp> &prog, int st, const mi_chelpers_section *sections,
                           uint64_t n_sections, uint64_t n_const, uint64_t nrows_ext)
{
    P->n_const = n_const;
    P->nrows_ext = nrows_ext;
    uint64_t col0 = 0;
    for (uint64_t i = 0; i < n_sections; i++) {
        P->sections.push_back({sections[i].offset, sections[i].ncols, sections[i].nrows, (uint32_t)col0, 0});
        col0 += sections[i].ncols;
    }
    bool uses_const = false, uses_x = false, uses_xd = false, uses_xdw = false;
    for (const chp::MicroOp &m : prog)
        for (const chp::HOpd *o : {&m.a, &m.b}) {
            uses_const |= o->k == chp::K_CONST || o->k == chp::K_CONSTS;
            uses_x |= o->k == chp::K_X;
            uses_xd |= o->k == chp::K_XD;
            uses_xdw |= o->k == chp::K_XDW;
        }
    if (uses_const && n_const) { P->sections.push_back({0, n_const, nrows_ext, (uint32_t)col0, 1}); col0 += n_const; }
    if (uses_x) { P->sections.push_back({0, 1, nrows_ext, (uint32_t)col0, 2}); col0 += 1; }
    if (uses_xd) { P->sections.push_back({0, 3, nrows_ext, (uint32_t)col0, 3}); col0 += 3; }   // xDivXSubXi: nrows_ext x 3
    if (uses_xdw) { P->sections.push_back({0, 3, nrows_ext, (uint32_t)col0, 4}); col0 += 3; } // xDivXSubWXi
    P->staged_cols = col0;
    if (st == MI_OK) st = chp::translate(P, prog);
    // the kernel's form needs the sections; a null context without sections compiles for the host debug executor only
    bool stores_pols = false;
    for (const chp::DInstr &d : P->host) stores_pols |= (d.op & 255) == chp::C_STOREP;
    P->stores_pols = stores_pols;
    // the base-domain steps run through the compiled kernels only (mi_chelpers_build_native): the interpreter has no store form
    if (st == MI_OK && (c || n_sections) && !stores_pols) st = chp::build_staged(P, 256);
    if (st == MI_OK && c && !stores_pols) {
        std::lock_guard<std::recursive_mutex> lock(c->mu);
        hipError_t e = hipSetDevice(c->device);
        // zero padding: the kernel's pipeline reads up to three batches past the end (an all-zero instruction has no destination)
        const size_t padded = (P->gpu.size() / chp::BATCH + 4) * chp::BATCH * sizeof(chp::GInstr);
        if (e == hipSuccess) e = hipMalloc((void **)&P->dev, padded);
        if (e == hipSuccess) e = hipMemsetAsync(P->dev, 0, padded, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(P->dev, P->gpu.data(), P->gpu.size() * sizeof(chp::GInstr), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            mi_set_error("mi_chelpers_compile: %s", hipGetErrorString(e));
            st = MI_ERR_HIP;
        }
    }
    if (st != MI_OK) {
        if (P->dev) (void)hipFree(P->dev);
        delete P;
        return st;
    }
    *out = P;
    return MI_OK;
}

extern "C" int mi_chelpers_compile(mi_ctx *c, mi_chelpers_prog **out, int step, const uint64_t *ops, uint64_t nops, const uint64_t *args,
                                   uint64_t nargs, const mi_chelpers_section *sections, uint64_t n_sections, uint64_t n_const, uint64_t nrows_ext)
{
    if (!out) return MI_ERR_INVALID;
    *out = nullptr;
    MI_REQUIRE(ops && (args || nargs == 0) && nops > 0, "null program tables");
    MI_REQUIRE(n_sections + 4 <= (uint64_t)chp::MAX_SECTIONS && (sections || n_sections == 0), "at most 4 sections");
    std::vector<chp::MicroOp> prog;
    mi_chelpers_prog *P = new mi_chelpers_prog();
    P->stats[0] = nops;
    P->step = step;
    const int st = chp::decode(step, ops, nops, args, nargs, prog, P->max_chal, P->max_pub, P->max_eval);
    return compile_decoded(c, out, P, prog, st, sections, n_sections, n_const, nrows_ext);
}

// the public numbering of mi_stark.h is the internal one
static_assert(MI_CHP_T1 == chp::K_T1 && MI_CHP_T3 == chp::K_T3 && MI_CHP_POL == chp::K_POL && MI_CHP_POLS == chp::K_POLS && MI_CHP_NUM == chp::K_NUM &&
              MI_CHP_CONST == chp::K_CONST && MI_CHP_CONSTS == chp::K_CONSTS && MI_CHP_CHAL == chp::K_CHAL && MI_CHP_PUB == chp::K_PUB &&
              MI_CHP_POL3 == chp::K_POL3 && MI_CHP_POL3S == chp::K_POL3S && MI_CHP_X == chp::K_X && MI_CHP_ZHINV == chp::K_ZHINV && MI_CHP_Q == chp::K_Q &&
              MI_CHP_EVAL == chp::K_EVAL && MI_CHP_XD == chp::K_XD && MI_CHP_XDW == chp::K_XDW && MI_CHP_DPOL == chp::K_DPOL && MI_CHP_DPOLS == chp::K_DPOLS,
              "operand kinds of mi_stark.h");
static_assert(MI_CHP_ADD == chp::C_ADD && MI_CHP_SUB == chp::C_SUB && MI_CHP_MUL == chp::C_MUL && MI_CHP_COPY == chp::C_COPY && MI_CHP_STOREQ == chp::C_STOREQ &&
              MI_CHP_STOREF == chp::C_STOREF && MI_CHP_STOREP == chp::C_STOREP, "operation classes of mi_stark.h");

extern "C" int mi_chelpers_compile_micro(mi_ctx *c, mi_chelpers_prog **out, int step, const mi_chelpers_microop *mops, uint64_t n_mops,
                                         const mi_chelpers_section *sections, uint64_t n_sections, uint64_t n_const, uint64_t nrows_ext)
{
    using namespace chp;
    if (!out) return MI_ERR_INVALID;
    *out = nullptr;
    MI_REQUIRE(mops && n_mops > 0, "null program");
    MI_REQUIRE(step == MI_CHELPERS_STEP42NS || step == MI_CHELPERS_STEP52NS || is_base_step(step), "unknown step");
    MI_REQUIRE(n_sections + 4 <= (uint64_t)MAX_SECTIONS && (sections || n_sections == 0), "at most 4 sections");
    std::vector<MicroOp> prog;
    prog.reserve(n_mops);
    uint64_t max_chal = 0, max_pub = 0, max_eval = 0;
    auto is_src = [](uint32_t k) { return k <= K_XDW && k != K_Q; };
    for (uint64_t i = 0; i < n_mops; i++) {
        const mi_chelpers_microop &u = mops[i];
        MicroOp m;
        MI_REQUIRE(u.cls <= C_STOREP, "operation class out of range");
        m.cls = (Cls)u.cls;
        m.dst = (Kind)u.dst_kind;
        m.dst_slot = u.dst_slot;
        const mi_chelpers_operand *us[2] = {&u.a, &u.b};
        HOpd *os[2] = {&m.a, &m.b};
        for (int s = 0; s < 2; s++) {
            const bool is_dest = m.cls == C_STOREP && s == 1;
            MI_REQUIRE(is_dest ? (us[s]->kind == K_DPOL || us[s]->kind == K_DPOLS) : is_src(us[s]->kind), "operand kind out of range");
            os[s]->k = (Kind)us[s]->kind;
            for (int j = 0; j < 4; j++) os[s]->v[j] = us[s]->v[j];
            if (os[s]->k == K_NUM) os[s]->v[0] = gl::canon(os[s]->v[0]);
            if (os[s]->k == K_CHAL) max_chal = std::max(max_chal, os[s]->v[0] + 1);
            if (os[s]->k == K_PUB) max_pub = std::max(max_pub, os[s]->v[0] + 1);
            if (os[s]->k == K_EVAL) max_eval = std::max(max_eval, os[s]->v[0] + 1);
        }
        switch (m.cls) {
        case C_STOREQ: // q = zhInv * a
            MI_REQUIRE(step == MI_CHELPERS_STEP42NS && m.dst == K_Q && m.a.k == K_T3 && m.b.k == K_ZHINV, "STOREQ: q = an extension temporary times ZhInv, in step42ns");
            break;
        case C_STOREF:
            MI_REQUIRE(step == MI_CHELPERS_STEP52NS && m.dst == K_Q && m.a.k == K_T3, "STOREF: f = an extension temporary, in step52ns");
            break;
        case C_STOREP:
            MI_REQUIRE(is_base_step(step) && (m.a.k == K_T1 || m.a.k == K_T3) && m.dst == m.b.k, "STOREP: a temporary into params.pols, in the base-domain steps");
            break;
        default:
            MI_REQUIRE(m.dst == K_T1 || m.dst == K_T3, "the destination of an arithmetic operation is a temporary");
            MI_REQUIRE(m.a.k != K_NONE && (m.cls == C_COPY ? m.b.k == K_NONE : m.b.k != K_NONE), "operand count");
            MI_REQUIRE(m.dst == K_T3 || (!kind_is3(m.a.k) && !kind_is3(m.b.k)), "an extension operand needs an extension destination");
            break;
        }
        prog.push_back(m);
    }
    mi_chelpers_prog *P = new mi_chelpers_prog();
    P->stats[0] = n_mops;
    P->step = step;
    P->max_chal = max_chal; P->max_pub = max_pub; P->max_eval = max_eval;
    return compile_decoded(c, out, P, prog, MI_OK, sections, n_sections, n_const, nrows_ext);
}

extern "C" void mi_chelpers_free(mi_ctx *c, mi_chelpers_prog *p)
{
    if (!p) return;
    if (c) {
        std::lock_guard<std::recursive_mutex> lock(c->mu);
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        if (p->dev) (void)hipFree(p->dev);
        chp::native_free(c, p);
    } else {
        chp::native_free(nullptr, p);
    }
    delete p;
}

static int check_params(const mi_chelpers_prog *p, const mi_chelpers_params *a, uint64_t row0, uint64_t nrows);

// Compile the translated program to gfx950 code (chelpers_native.hip).  Needs no GPU; cache_dir (or $MI_CHELPERS_CACHE) keeps the
// code objects between processes.  Afterwards mi_chelpers_run_dev runs the compiled kernels instead of the interpreter.
extern "C" int mi_chelpers_build_native(mi_chelpers_prog *p, const char *cache_dir, uint64_t chunk_cost)
{
    if (!p) return MI_ERR_INVALID;
    return chp::native_build(p, cache_dir, chunk_cost, 0, 1);
}

// A section the caller keeps tile-major in HBM (include/mi_stark.h): the generated kernels read it in place.  Any number of a program's
// sections may be.
extern "C" int mi_chelpers_set_tiled_section(mi_chelpers_prog *p, uint64_t section_offset)
{
    if (!p) return MI_ERR_INVALID;
    MI_REQUIRE(!p->native, "mi_chelpers_set_tiled_section comes before mi_chelpers_build_native");
    HostSection *hit = nullptr;
    for (HostSection &S : p->sections) {
        if (S.role == 0 && S.offset == section_offset) hit = &S;
    }
    MI_REQUIRE(hit, "no declared section starts at this offset");
    MI_REQUIRE(hit->nrows % 64 == 0, "a tile-major section has a multiple of 64 rows");
    hit->tiled = true;
    chp::renumber_staged_columns(p);
    return MI_OK;
}

// ... and the constant polynomials the program reads (const_pols of mi_chelpers_run_dev: [nrows / 64][n_const][64] then): a proving key's
// constants never change, so a caller can keep them that way for good (host/starks.hpp does, for the base-domain steps).  A program that
// reads no constant polynomial is left as it is.
extern "C" int mi_chelpers_set_tiled_consts(mi_chelpers_prog *p)
{
    if (!p) return MI_ERR_INVALID;
    MI_REQUIRE(!p->native, "mi_chelpers_set_tiled_consts comes before mi_chelpers_build_native");
    for (HostSection &S : p->sections)
        if (S.role == 1) {
            MI_REQUIRE(S.nrows % 64 == 0, "a tile-major section has a multiple of 64 rows");
            S.tiled = true;
        }
    chp::renumber_staged_columns(p);
    return MI_OK;
}

namespace chp {
void renumber_staged_columns(mi_chelpers_prog *p)
{
    // the per-batch tile-major copy holds the other sections' columns only (a batch of rows is sized by them): renumber the staged
    // columns, the sections read in place behind them (a column keeps a number of its own: the operand statistics go by it)
    uint32_t c0 = 0;
    for (HostSection &S : p->sections)
        if (!S.tiled) { S.col0 = c0; c0 += (uint32_t)S.ncols; }
    p->staged_cols = c0 ? c0 : 1;
    for (HostSection &S : p->sections)
        if (S.tiled) { S.col0 = c0; c0 += (uint32_t)S.ncols; }
}
} // namespace chp

// Parallel builds: process `shard` of `nshards` compiles every nshards-th kernel into the cache and keeps nothing; a final
// mi_chelpers_build_native then finds every kernel in the cache.
extern "C" int mi_chelpers_precompile_shard(mi_chelpers_prog *p, const char *cache_dir, uint64_t chunk_cost, uint32_t shard, uint32_t nshards)
{
    if (!p) return MI_ERR_INVALID;
    return chp::native_build(p, cache_dir, chunk_cost, shard, nshards);
}

extern "C" int mi_chelpers_lower_stats(const mi_chelpers_prog *p, uint64_t chunk_cost, uint64_t out[12])
{
    if (!p || !out) return MI_ERR_INVALID;
    return chp::native_lower_stats(p, chunk_cost, out);
}

// tests only: the lowered program (Horner chains, pieces, spill lists) run on the CPU, see chp::native_host_run
extern "C" int mi_dbg_host_chelpers_run_lowered(const mi_chelpers_prog *p, const mi_chelpers_params *a, const uint64_t *rows, uint64_t nrows, uint64_t chunk_cost)
{
    MI_TRY(check_params(p, a, 0, 0));
    MI_REQUIRE(rows || nrows == 0, "null row list");
    for (const HostSection &S : p->sections) MI_REQUIRE(!S.tiled, "the host executors read row-major sections");
    return chp::native_host_run(p, a, rows, nrows, chunk_cost);
}

extern "C" int mi_chelpers_native_stats(const mi_chelpers_prog *p, uint64_t out[8])
{
    if (!p || !out) return MI_ERR_INVALID;
    chp::native_stats(p, out);
    return MI_OK;
}

// Allocate what a native run over nrows rows needs (operand copy, spill, constants) now instead of inside the first run.
extern "C" int mi_chelpers_reserve(mi_ctx *c, const mi_chelpers_prog *p, uint64_t nrows)
{
    if (!c || !p) return MI_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    MI_HIP_CHECK(hipSetDevice(c->device));
    if (!p->native) return MI_OK; // the interpreter sizes its staging by what is resident: nothing to reserve ahead
    return chp::native_reserve(c, p, nrows, nullptr);
}

extern "C" int mi_set_chelpers_batch_rows(mi_ctx *c, uint64_t rows)
{
    if (!c) return MI_ERR_INVALID;
    MI_REQUIRE(rows % 64 == 0, "batch must be a multiple of 64 rows");
    c->chelpers_batch_rows = rows;
    return MI_OK;
}

extern "C" int mi_set_chelpers_min_words(mi_ctx *c, uint64_t words)
{
    if (!c) return MI_ERR_INVALID;
    MI_REQUIRE(words * 512 <= 160 * 1024, "more words than the LDS holds");
    c->chelpers_min_words = words;
    return MI_OK;
}

extern "C" int mi_chelpers_stats(const mi_chelpers_prog *p, uint64_t out[16])
{
    if (!p || !out) return MI_ERR_INVALID;
    for (int i = 0; i < 8; i++) out[i] = p->stats[i];
    out[8] = p->gpu.size();   // device instructions per row (dimension-3 polynomial operands cost three copies each)
    out[9] = p->hot_t1;       // base temporaries in LDS
    out[10] = p->hot_t3;      // extension temporaries in LDS
    out[11] = p->cold_words;  // base temporaries spilled to HBM
    out[12] = p->temp_reads;  // reads of temporaries per row ...
    out[13] = p->cold_reads;  // ... of which from the spill
    out[14] = p->cst_off;     // LDS bytes per workgroup before the constant tables
    out[15] = p->staged_cols;
    return MI_OK;
}

static int check_params(const mi_chelpers_prog *p, const mi_chelpers_params *a, uint64_t row0, uint64_t nrows)
{
    MI_REQUIRE(p && a, "null program or parameters");
    MI_REQUIRE(a->pols && (p->stores_pols || (p->step == MI_CHELPERS_STEP52NS ? a->f != nullptr : a->q != nullptr)), "null polynomial memory or output");
    MI_REQUIRE(a->n_evals >= p->max_eval && (a->evals || p->max_eval == 0), "program reads more evaluations than were given");
    MI_REQUIRE(a->n_challenges >= p->max_chal && (a->challenges || p->max_chal == 0), "program reads more challenges than were given");
    MI_REQUIRE(a->n_publics >= p->max_pub && (a->publics || p->max_pub == 0), "program reads more public inputs than were given");
    if (p->step == MI_CHELPERS_STEP42NS)
        MI_REQUIRE(a->zhinv && a->n_zhinv > 0 && a->n_zhinv <= 256 && is_pow2(a->n_zhinv), "ZhInv table must hold 2^k <= 256 values (zhInv.cpp:7-31)");
    MI_REQUIRE(row0 + nrows >= row0, "row range overflows");
    return MI_OK;
}

extern "C" int mi_chelpers_run_dev(mi_ctx *c, const mi_chelpers_prog *p, const mi_chelpers_params *a, uint64_t row0, uint64_t nrows)
{
    if (!c) return MI_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    MI_HIP_CHECK(hipSetDevice(c->device));
    MI_TRY(check_params(p, a, row0, nrows));
    if (mi_check_on()) { // every table of the run belongs to the shard this context works for, and so does the program (its code object is loaded on ONE device)
        // (checked where the run touches them: `pols` itself may be a VIRTUAL base -- a row shard's mirror holds the extended sections only,
        // host/starks.hpp rowBase -- and the full-height tables are read at the run's rows)
        for (const HostSection &S : p->sections)
            if (S.role == 0 && a->pols) MI_TRY(mi_own_check(c->logical, (const u64 *)a->pols + S.offset + (p->stores_pols ? 0 : row0 * S.ncols), "mi_chelpers_run_dev (a section of params.pols)"));
        for (const void *q : {(const void *)a->const_pols, (const void *)(a->x ? a->x + row0 * a->x_stride : nullptr), (const void *)(a->xdiv ? a->xdiv + 3 * row0 : nullptr),
                              (const void *)(a->xdivw ? a->xdivw + 3 * row0 : nullptr), (const void *)(a->q ? a->q + 3 * row0 : nullptr), (const void *)(a->f ? a->f + 3 * row0 : nullptr)})
            MI_TRY(mi_own_check(c->logical, q, "mi_chelpers_run_dev"));
        if (p->run_logical >= 0 && p->run_logical != c->logical) {
            mi_set_error("MI_MULTI_CHECK: mi_chelpers_run_dev: the program was first run for logical shard %d and is now run for shard %d (a program is loaded on one device)", p->run_logical, c->logical);
            fprintf(stderr, "mi_stark: %s\n", mi_last_error());
            return MI_ERR_INVALID;
        }
        const_cast<mi_chelpers_prog *>(p)->run_logical = c->logical;
    }
    if (p->native) return chp::native_run(c, p, a, row0, nrows);
    MI_REQUIRE(!p->stores_pols, "the base-domain steps run through the compiled kernels: call mi_chelpers_build_native first");
    MI_REQUIRE(p->dev, "program was compiled without a context");
    for (const HostSection &S : p->sections) MI_REQUIRE(!S.tiled, "a program with a tile-major section runs through the compiled kernels: call mi_chelpers_build_native first");
    if (nrows == 0) return MI_OK;
    MI_REQUIRE(a->n_const == p->n_const, "number of constant polynomials differs from what the program was compiled for");
    MI_REQUIRE(row0 + nrows <= p->nrows_ext, "rows beyond the extended domain the program was compiled for");
    // LDS: the program's layout (build_staged) + the three small tables; the benchmarking knob can only enlarge it
    const uint32_t chal_words = (uint32_t)p->max_chal * 3, pub_words = (uint32_t)p->max_pub;
    const uint64_t n_zh = p->step == MI_CHELPERS_STEP42NS ? a->n_zhinv : 1; // step52ns reads no ZhInv: one dummy word keeps the mask valid
    const uint32_t cst_words = chal_words + pub_words + (uint32_t)n_zh;
    size_t lds = (size_t)p->cst_off + (size_t)cst_words * 8;
    lds = std::max(lds, (size_t)c->chelpers_min_words * 512);
    MI_REQUIRE(lds <= 160 * 1024, "program needs more temporaries per row than the LDS holds");
    // challenges | public inputs | ZhInv, back to back, in the context's scratch
    MI_REQUIRE(cst_words <= 576, "more challenges / public inputs than the scratch holds");
    if (!c->chelpers_scratch) MI_HIP_CHECK(hipMalloc((void **)&c->chelpers_scratch, 576 * 8));
    u64 stage[576] = {0};
    for (uint64_t i = 0; i < chal_words; i++) stage[i] = gl::canon(a->challenges[i]);
    for (uint64_t i = 0; i < pub_words; i++) stage[chal_words + i] = gl::canon(a->publics[i]);
    for (uint64_t i = 0; i < n_zh && a->zhinv; i++) stage[chal_words + pub_words + i] = gl::canon(a->zhinv[i]);
    MI_HIP_CHECK(hipStreamSynchronize(c->stream)); // an earlier run may still be reading the scratch
    MI_HIP_CHECK(hipMemcpyAsync(c->chelpers_scratch, stage, sizeof(stage), hipMemcpyHostToDevice, c->stream));
    MI_HIP_CHECK(hipStreamSynchronize(c->stream)); // `stage` is on this stack frame
    // persistent workgroups: as many as are resident at once (LDS bound), each with a private staging + spill area
    const uint64_t n_groups = (nrows + 63) / 64;
    const uint64_t per_cu = std::max<uint64_t>(1, std::min<uint64_t>(16, (160 * 1024) / lds));
    const uint64_t grid = std::min<uint64_t>(n_groups, per_cu * (uint64_t)c->cu_count);
    const uint64_t wg_stride = p->staged_cols * chp::RS + (uint64_t)p->cold_words * 64 + 64;
    const uint64_t scratch_bytes = grid * wg_stride * 8 + 4096;
    if (c->pool->chelpers_stage_bytes < scratch_bytes) {
        if (c->pool->chelpers_stage) MI_HIP_CHECK(hipFree(c->pool->chelpers_stage));
        c->pool->chelpers_stage = nullptr;
        c->pool->chelpers_stage_bytes = 0;
        hipError_t e = hipMalloc((void **)&c->pool->chelpers_stage, scratch_bytes);
        if (e != hipSuccess) {
            mi_set_error("cannot allocate %llu bytes of operand staging: %s", (unsigned long long)scratch_bytes, hipGetErrorString(e));
            return MI_ERR_NOMEM;
        }
        c->pool->chelpers_stage_bytes = scratch_bytes;
    }
    chp::GArgs A = {};
    A.n_sections = (uint32_t)p->sections.size();
    for (size_t i = 0; i < p->sections.size(); i++) {
        const HostSection &S = p->sections[i];
        chp::GSection &G = A.sec[i];
        G.ncols = (uint32_t)S.ncols;
        G.col0 = S.col0;
        G.nrows = S.nrows;
        if (S.role == 0) { G.ptr = (const u64 *)a->pols + S.offset; G.pitch = S.ncols; }
        else if (S.role == 1) { MI_REQUIRE(a->const_pols, "null constant polynomials"); G.ptr = (const u64 *)a->const_pols; G.pitch = a->n_const; }
        else if (S.role == 2) { MI_REQUIRE(a->x, "null x"); G.ptr = (const u64 *)a->x; G.pitch = a->x_stride; }
        else if (S.role == 3) { MI_REQUIRE(a->xdiv, "null xDivXSubXi"); G.ptr = (const u64 *)a->xdiv; G.pitch = 3; }
        else { MI_REQUIRE(a->xdivw, "null xDivXSubWXi"); G.ptr = (const u64 *)a->xdivw; G.pitch = 3; }
    }
    A.n_instr = (uint32_t)p->gpu.size();
    A.tile_off = p->tile_off;
    A.pre_off = p->pre_off;
    A.cst_off = p->cst_off;
    A.cst_words = cst_words;
    A.zh_off = chal_words + pub_words;
    A.lds_bytes = (uint32_t)lds;
    A.consts = c->chelpers_scratch;
    A.q = (u64 *)(p->step == MI_CHELPERS_STEP52NS ? a->f : a->q);
    A.scratch = c->pool->chelpers_stage;
    if (!p->eval_patches.empty()) { // this proof's evaluations into the immediates that stand for them
        std::vector<chp::GInstr> patched(p->gpu);
        for (const mi_chelpers_prog::Patch &pt : p->eval_patches) patched[pt.instr].a_imm = gl::canon(a->evals[(uint64_t)pt.eval * 3 + pt.word]);
        MI_HIP_CHECK(hipMemcpyAsync(p->dev, patched.data(), patched.size() * sizeof(chp::GInstr), hipMemcpyHostToDevice, c->stream));
        MI_HIP_CHECK(hipStreamSynchronize(c->stream)); // `patched` dies with this scope
    }
    A.n_zhinv = n_zh;
    A.row0 = row0;
    A.row_end = row0 + nrows;
    A.n_groups = n_groups;
    A.staged_cols = p->staged_cols;
    A.wg_stride = wg_stride;
    if (lds > 48 * 1024)
        MI_HIP_CHECK(hipFuncSetAttribute((const void *)chp::k_chelpers, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(chp::k_chelpers, dim3((unsigned)grid), dim3(64), lds, c->stream, p->dev, A);
    MI_HIP_CHECK(hipGetLastError());
    return MI_OK;
}

// ---- host debug executor (tests only): the SAME translated program and instruction semantics, run on the CPU over host
// pointers, so that the translator (role table, copy forwarding, reschedule, slot allocation) can be checked without a GPU
extern "C" int mi_dbg_host_chelpers_run(const mi_chelpers_prog *p, const mi_chelpers_params *a, const uint64_t *rows, uint64_t nrows)
{
    MI_TRY(check_params(p, a, 0, 0));
    MI_REQUIRE(rows || nrows == 0, "null row list");
    for (const HostSection &S : p->sections) MI_REQUIRE(!S.tiled, "the host executors read row-major sections");
    std::vector<u64> chal(p->max_chal * 3 + 1), pub(p->max_pub + 1), zh(a->n_zhinv + 1), ev(p->max_eval * 3 + 1);
    for (uint64_t i = 0; i < p->max_eval * 3; i++) ev[i] = gl::canon(a->evals[i]);
    for (uint64_t i = 0; i < p->max_chal * 3; i++) chal[i] = gl::canon(a->challenges[i]);
    for (uint64_t i = 0; i < p->max_pub; i++) pub[i] = gl::canon(a->publics[i]);
    for (uint64_t i = 0; i < a->n_zhinv; i++) zh[i] = gl::canon(a->zhinv[i]);
    chp::RunArgs A = {};
    A.pols = (const u64 *)a->pols;
    A.cpols = (const u64 *)a->const_pols;
    A.x = (const u64 *)a->x;
    A.chal = chal.data();
    A.pub = pub.data();
    A.zhinv = zh.data();
    A.q = (u64 *)a->q;
    A.f = (u64 *)a->f;
    A.pols_w = (u64 *)a->pols;
    A.evals = ev.data();
    A.xd = (const u64 *)a->xdiv;
    A.xdw = (const u64 *)a->xdivw;
    A.n_const = a->n_const;
    A.x_stride = a->x_stride;
    A.n_zhinv = a->n_zhinv ? a->n_zhinv : 1;
    std::vector<u64> words(p->n_words + 3);
    chp::HostTmp tmp = {words.data()};
    for (uint64_t k = 0; k < nrows; k++)
        for (const chp::DInstr &I : p->host) chp::exec_instr(I, rows[k], true, A, tmp);
    return MI_OK;
}
