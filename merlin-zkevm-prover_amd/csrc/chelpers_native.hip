// chelpers_native.hip -- the constraint evaluators compiled to gfx950 code (second backend of chelpers.hip).
//
// The interpreter of chelpers.hip spends ~90 VALU + ~80 SALU instructions per field operation on fetching, decoding and
// addressing, where the arithmetic itself is 6 (add) to 20 (multiply) instructions, and keeps every temporary in LDS.  A
// constraint program is fixed per proving key, so -- like the reference, whose build compiles the generated chelpers C++ into
// the prover -- the translated program can be compiled once.  What this file does, in the order of the code:
//   * lower(): Horner chains over a challenge become unreduced multiply-accumulates with constants of the running proof
//     (chelpers_acc.h), chain terms that are plain polynomial elements go to a streaming matrix-vector kernel (k_chp_linear), the
//     scheduled three-address program (mi_chelpers_prog::host: depth-first order, ~90 live words) is cut into CHUNKS of about
//     11 000 VALU instructions, and the values that live across a chunk boundary are found (spill lists);
//   * Gen / generate(): every chunk becomes one straight-line HIP kernel, generated as source text and compiled with hiprtc against
//     the very same gl_math.h the other kernels use: one row per lane, 64-lane workgroups = one TILE of 64 consecutive rows,
//     temporaries in registers, values kept weakly reduced in [0, 2^64) and canonicalised at the stores (the generator tracks
//     which words are canonical: add_wc / sub_wc need their second operand canonical), the instruction stream fenced into
//     scheduling groups with the polynomial loads of the next group issued ahead, no wave-uniform "rare case" branches.
//     Compile time is linear in the number of chunks; code objects are cached on disk by the hash of their source;
//   * polynomial operands are read from a TILE-MAJOR copy of the sections, [tile][staged column][64 rows], which a transposing
//     kernel (k_chp_transpose) makes per batch of rows (+ one halo tile for the shifted "prime" reads): a lane owns a row, the
//     sections are row-major, so reading them in place would touch one 64-byte sector per lane and operand.  With the copy an
//     operand is base + column * 512 B + lane * 8 B: a compile-time offset from a per-tile pointer;
//   * values that cross a chunk boundary go through a spill area [tile][word][64 lanes] (coalesced 512-byte runs), canonical;
//     challenges, public inputs, evaluations, ZhInv, chain coefficients and per-piece constants sit in one small device table
//     read with scalar loads (fill_constants);
//   * native_host_run(): the LOWERED program on the CPU (tests), with the same accumulator code.
// Results are the same field elements as the interpreter's and the oracle's (exact arithmetic, canonical at the store).
#include "chelpers_ir.h"
#include "chelpers_acc.h"
#include <algorithm>
#include <chrono>
#include <hip/hiprtc.h>
#include <map>
#include <set>
#include <tuple>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

namespace chp {

static const char *GL_MATH_SRC =
#include "gl_math_src.inc"
    ;

static const char *ACC_SRC =
#include "chelpers_acc_src.inc"
    ;

// what the generated code calls besides gl:: and chpa:: (all values any u64 unless a name says "c" = canonical)
static const char *HELPERS_SRC = R"CHPSRC(
#define DEV __device__ __forceinline__
DEV u64 addg(u64 a, u64 b) { return gl::add_wc(a, gl::canon(b)); }
DEV u64 subg(u64 a, u64 b) { return gl::sub_wc(a, gl::canon(b)); }
DEV u64 negc(u64 bc) { return GL_P - bc; } // canonical -> weakly reduced (p for 0)
// extension multiply, x^3 = x + 1 (polinomial.hpp:195-205): nine products summed unreduced, three reductions
DEV void mul33(u64 &o0, u64 &o1, u64 &o2, u64 a0, u64 a1, u64 a2, u64 b0, u64 b1, u64 b2)
{
    chpa::Acc r0, r1, r2;
    chpa::acc_set(r0, 0); chpa::acc_set(r1, 0); chpa::acc_set(r2, 0);
    chpa::acc_mul33(r0, r1, r2, a0, a1, a2, b0, b1, b2);
    o0 = chpa::acc_reduce(r0); o1 = chpa::acc_reduce(r1); o2 = chpa::acc_reduce(r2);
}
)CHPSRC";

// ---- Horner chains.  A T3 value y that is only ever multiplied by ONE challenge C and added to / subtracted from is carried as
// three unreduced accumulators (chelpers_acc.h): a run of steps  y <- y * C | y <- y + v | y <- y - v  inside one kernel (a PIECE)
// becomes  y_in * C^M + sum_j (+-) v_j * C^(number of multiplications after step j), each a multiply-accumulate with a constant
// of the running proof (the powers are computed on the host per run), reduced once at the end of the piece.  A leaf
// v = polynomial - evals[k] whose only use is the chain step is FOLDED: the polynomial becomes the term, the evaluation goes into
// one per-piece constant K.  zkEVM step42ns: 2 185 of its 2 402 extension multiplications are such steps ((acc + constraint) * vc),
// step52ns: all but a handful.
enum StepT : uint8_t { ST_NONE = 0, ST_M, ST_A, ST_S, ST_FOLDED };
struct Mark { uint8_t type = ST_NONE, yside = 0, lin = 0; int32_t chain = -1, folded = -1; };
struct Coef { uint32_t chal, exp; bool neg; uint32_t xk; bool zero; }; // (+-) C^exp * x^xk, or 0
struct KTerm { uint32_t eval, coef; };
struct Piece { int32_t chain; uint32_t first, last, n_m; int32_t begin_coef = -1, k_slot = -1, lin_sum = -1; };

// ---- linear terms.  A chain step whose leaf is a polynomial element at the row itself (step52ns: every one of them -- the FRI
// polynomial is a random linear combination of the committed polynomials) needs no generated code at all: the sum of such terms
// of a piece is  sum_j pol_j(row) * W_j  with constants W_j, a matrix-vector product over the row-major sections.  One kernel
// (k_chp_linear) streams the sections once, slab by slab through LDS, and accumulates up to four such sums per row with the same
// limb accumulators; the generated kernel adds the result at the end of the piece.  An extension-valued polynomial (three
// adjacent columns) is three base terms with coefficients W, x W, x^2 W.  This also removes those columns from the tile-major
// copy: for step52ns only xDivXSubXi / xDivXSubWXi and one constant column are left in it.
#ifndef MI_LIN_COLS
#define MI_LIN_COLS 16
#endif
static constexpr uint32_t LIN_COLS = MI_LIN_COLS;  // columns per staged slab
static constexpr int LIN_MAX_SUMS = 4;
static constexpr uint32_t LIN_TMAX = 128; // terms of a slab entry (what the kernel's LDS block holds)
static constexpr uint32_t LIN_MIN_TERMS = 256; // below this a pass over the sections costs more than the generated terms
struct LinTermH { uint32_t staged_col, coef, sum; };
struct LinTerm { uint32_t lds_off, coef; };    // byte offset of the column inside a staged row; index of the coefficient (LinTermW carries its value)
struct LinSlabD { uint32_t section, col0, ncols, t0[LIN_MAX_SUMS + 1]; };
struct LinSections { const u64 *ptr[MAX_SECTIONS]; uint64_t pitch[MAX_SECTIONS], row_mask[MAX_SECTIONS]; uint32_t tiled[MAX_SECTIONS]; }; // tiled: [rows / 64][pitch columns][64]

struct Chunk {
    size_t i0 = 0, i1 = 0;          // instructions [i0, i1) of mi_chelpers_prog::host
    std::vector<uint32_t> loads, stores, touched; // temp words read from / written to the spill; all words the kernel names
    std::vector<uint32_t> pieces;   // chain pieces inside this chunk
    std::vector<char> code;         // the code object
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
};
struct NativeProg {
    std::vector<Chunk> chunks;
    std::vector<Mark> marks;                 // per instruction of mi_chelpers_prog::host
    std::vector<int32_t> step_coef;          // A / S steps: coefficient index, -1 = plain +1
    std::vector<int32_t> piece_first, piece_last; // instruction -> piece that begins / ends there
    std::vector<Piece> pieces;
    std::vector<Coef> coefs;
    std::vector<std::vector<KTerm>> kslots;
    std::vector<LinTermH> lin_terms;          // as collected (host executor)
    std::vector<int32_t> lin_first;           // instruction -> its first entry in lin_terms
    std::vector<LinTerm> lin_dev;             // sorted by slab and sum, padded to pairs (kernel)
    std::vector<LinSlabD> lin_slabs;
    uint32_t n_lin_sums = 0;
    struct LinTermW *d_lin_terms = nullptr;   // rewritten per run: the coefficients are constants of the running proof
    LinSlabD *d_lin_slabs = nullptr;
    std::vector<uint32_t> sec_slab_mask;      // per section: 64-column slabs the generated kernels read (tile-major copy)
    std::vector<uint32_t> sec_inplace_mask;   // per section: slabs the kernels read so little of that they read it in place (row-major, a sector per lane)
    uint32_t sc = 0, nw = 0;                     // staged columns per tile, temp words per row
    uint32_t pub_off = 0, ev_off = 0, zh_off = 0, coef_off = 0, k_off = 0, cst_words = 0; // word offsets into the constants table (challenges first)
    double compile_s = 0;
    uint64_t code_bytes = 0, cache_hits = 0, est_valu = 0, spill_words_moved = 0, chain_steps = 0, operand_loads = 0, operand_distinct = 0;
    int loaded_device = -1;
};

static uint64_t fnv1a(const std::string &s, uint64_t h = 1469598103934665603ull)
{
    for (unsigned char ch : s) { h ^= ch; h *= 1099511628211ull; }
    return h;
}

static bool is_pol_kind(uint32_t k) { return k == K_POL || k == K_POLS || k == K_POL3 || k == K_POL3S || k == K_CONST || k == K_CONSTS || k == K_X || k == K_XD || k == K_XDW; }

// a polynomial operand -> its section, first staged column, dimension and row shift (nullptr: in no declared section)
static const HostSection *resolve_pol(const mi_chelpers_prog *P, uint32_t k, const Opd &o, uint32_t &col, int &dim, uint32_t &shift)
{
    const bool three = k == K_POL3 || k == K_POL3S || k == K_XD || k == K_XDW, shifted = k == K_POLS || k == K_POL3S || k == K_CONSTS;
    const int role = (k == K_CONST || k == K_CONSTS) ? 1 : k == K_X ? 2 : k == K_XD ? 3 : k == K_XDW ? 4 : 0;
    const uint64_t off = role >= 2 ? 0 : o.off, width = three ? 3 : 1;
    dim = three ? 3 : 1;
    shift = shifted ? o.shift : 0;
    for (const HostSection &S : P->sections) {
        if (S.role != role) continue;
        if (role == 0 && (o.stride != S.ncols || off < S.offset || off - S.offset + width > S.ncols)) continue;
        if (role != 0 && off + width > S.ncols) continue;
        if (shifted && (o.mod != S.nrows || o.shift >= 64)) continue;
        col = S.col0 + (uint32_t)(role == 0 ? off - S.offset : off);
        return &S;
    }
    return nullptr;
}

// estimated VALU instructions of one translated instruction (chunk sizing only)
static uint64_t cost_of(const DInstr &d, const Mark &m, const std::vector<DInstr> &H)
{
    const uint32_t cls = d.op & 255, dk = (d.op >> 8) & 255, ak = (d.op >> 16) & 255, bk = d.op >> 24;
    if (m.type == ST_M || m.type == ST_FOLDED || m.lin) return 0;
    if (m.type == ST_A || m.type == ST_S) {
        uint32_t lk = m.yside == 0 ? bk : ak;
        if (m.folded >= 0) lk = (H[m.folded].op >> 16) & 255;
        return kind_is3(lk) ? 110 : 30;
    }
    const bool a3 = kind_is3(ak), b3 = kind_is3(bk), r3 = dk == K_T3 || dk == K_Q;
    switch (cls) {
    case C_ADD: case C_SUB: return r3 ? (a3 && b3 ? 21 : 9) : 7;
    case C_MUL: return a3 && b3 ? 130 : (a3 || b3) ? 57 : 19;
    case C_STOREQ: return 70;
    default: return r3 ? 6 : 2;
    }
}

static void words_of(uint32_t kind, const Opd &o, std::vector<uint32_t> &out)
{
    if (kind == K_T1) out.push_back((uint32_t)o.off);
    else if (kind == K_T3) { out.push_back((uint32_t)o.off); out.push_back((uint32_t)o.off + 1); out.push_back((uint32_t)o.off + 2); }
}

// chains, chunks, pieces, spill lists, constants layout: everything the generator and the host debug executor share
static int lower(const mi_chelpers_prog *P, NativeProg *N, uint64_t chunk_cost, bool chains_on, bool lin_on = true)
{
    const std::vector<DInstr> &H = P->host;
    const size_t n = H.size();
    N->sc = (uint32_t)P->staged_cols;
    N->nw = (uint32_t)P->n_words + 3;
    N->pub_off = (uint32_t)P->max_chal * 3;
    N->ev_off = N->pub_off + (uint32_t)P->max_pub;
    N->zh_off = N->ev_off + (uint32_t)P->max_eval * 3;
    N->coef_off = N->zh_off + 256;
    N->marks.assign(n, Mark());
    N->step_coef.assign(n, -1);
    N->piece_first.assign(n, -1);
    N->piece_last.assign(n, -1);
    N->lin_first.assign(n, -1);
    // ---- definitions and use counts of temporaries
    std::vector<int32_t> defi(N->nw + 3, -1);
    std::vector<std::array<int32_t, 2>> src(n, {-1, -1});
    std::vector<uint32_t> nuse(n, 0);
    for (size_t i = 0; i < n; i++) {
        const DInstr &d = H[i];
        const Opd *os[2] = {&d.a, &d.b};
        for (int s = 0; s < 2; s++) {
            const uint32_t k = (d.op >> (16 + 8 * s)) & 255;
            if (k != K_T1 && k != K_T3) continue;
            const int32_t v = defi[os[s]->off];
            bool whole = v >= 0 && H[v].dst == os[s]->off && ((H[v].op >> 8) & 255) == k;
            if (k == K_T3) whole = whole && defi[os[s]->off + 1] == v && defi[os[s]->off + 2] == v;
            src[i][s] = whole ? v : -1;
            if (v >= 0) nuse[v] += whole ? 1 : 2; // a partial read disqualifies the definition from every pattern
        }
        const uint32_t dk = (d.op >> 8) & 255;
        if (dk == K_T1) defi[d.dst] = (int32_t)i;
        else if (dk == K_T3) defi[d.dst] = defi[d.dst + 1] = defi[d.dst + 2] = (int32_t)i;
    }
    // ---- chains
    struct Chain { int32_t chal = -1; std::vector<uint32_t> steps; uint32_t n_m = 0; };
    std::vector<Chain> chains;
    if (chains_on) {
        std::vector<int32_t> chain_of(n, -1);
        for (size_t i = 0; i < n; i++) {
            const DInstr &d = H[i];
            const uint32_t cls = d.op & 255, dk = (d.op >> 8) & 255, kk[2] = {(d.op >> 16) & 255, d.op >> 24};
            const Opd *os[2] = {&d.a, &d.b};
            if (dk != K_T3) continue;
            auto is_tail = [&](int s) {
                const int32_t v = src[i][s];
                return kk[s] == K_T3 && v >= 0 && nuse[v] == 1 && chain_of[v] >= 0 && chains[chain_of[v]].steps.back() == (uint32_t)v;
            };
            if (cls == C_MUL && ((kk[0] == K_T3 && kk[1] == K_CHAL) || (kk[0] == K_CHAL && kk[1] == K_T3))) {
                const int ys = kk[0] == K_T3 ? 0 : 1;
                const int32_t c = (int32_t)os[1 - ys]->off;
                int32_t ch = -1;
                if (is_tail(ys) && (chains[chain_of[src[i][ys]]].chal == c || chains[chain_of[src[i][ys]]].chal < 0)) ch = chain_of[src[i][ys]];
                if (ch < 0) { chains.push_back(Chain()); ch = (int32_t)chains.size() - 1; }
                chains[ch].chal = c;
                chains[ch].steps.push_back((uint32_t)i);
                chains[ch].n_m++;
                chain_of[i] = ch;
                N->marks[i].type = ST_M; N->marks[i].yside = (uint8_t)ys; N->marks[i].chain = ch;
            } else if (cls == C_ADD || cls == C_SUB) {
                int ys = -1;
                if (is_tail(0)) ys = 0;
                else if (cls == C_ADD && is_tail(1)) ys = 1;
                if (ys < 0) continue;
                const uint32_t lk = kk[1 - ys];
                if (!(lk == K_T1 || lk == K_T3 || is_pol_kind(lk))) continue;
                const int32_t ch = chain_of[src[i][ys]];
                chains[ch].steps.push_back((uint32_t)i);
                chain_of[i] = ch;
                Mark &m = N->marks[i];
                m.type = cls == C_ADD ? ST_A : ST_S; m.yside = (uint8_t)ys; m.chain = ch;
                const int32_t L = src[i][1 - ys];
                if (lk == K_T3 && L >= 0 && nuse[L] == 1 && chain_of[L] < 0 && N->marks[L].type == ST_NONE && (H[L].op & 255) == C_SUB &&
                    (H[L].op >> 24) == K_EVAL && is_pol_kind((H[L].op >> 16) & 255)) {
                    m.folded = L;
                    N->marks[L].type = ST_FOLDED;
                }
            }
        }
        for (size_t ci = 0; ci < chains.size(); ci++)
            if (chains[ci].n_m < 2) { // nothing to gain: back to plain instructions
                for (uint32_t i : chains[ci].steps) {
                    if (N->marks[i].folded >= 0) N->marks[N->marks[i].folded] = Mark();
                    N->marks[i] = Mark();
                }
                chains[ci].steps.clear();
            }
    }
    // ---- linear terms: chain steps whose leaf is a polynomial element at the row itself
    {
        uint32_t cand = 0;
        auto leaf_kind = [&](size_t i) {
            const Mark &m = N->marks[i];
            return m.folded >= 0 ? (H[m.folded].op >> 16) & 255 : (H[i].op >> (16 + 8 * (1 - m.yside))) & 255;
        };
        auto is_lin = [&](size_t i) {
            const Mark &m = N->marks[i];
            if (m.type != ST_A && m.type != ST_S) return false;
            const uint32_t lk = leaf_kind(i);
            if (!(lk == K_POL || lk == K_POL3 || lk == K_CONST || lk == K_X)) return false;
            uint32_t col, sh;
            int dim;
            const HostSection *S = resolve_pol(P, lk, m.folded >= 0 ? H[m.folded].a : (m.yside == 0 ? H[i].b : H[i].a), col, dim, sh);
            return S != nullptr; // (row-major or tile-major: the linear kernel fetches either)
        };
        for (size_t i = 0; i < n; i++) cand += is_lin(i);
        uint32_t lin_min = LIN_MIN_TERMS;
        if (const char *e = getenv("MI_CHELPERS_LIN_MIN")) lin_min = (uint32_t)atoi(e);
        if (lin_on && cand >= lin_min)
            for (size_t i = 0; i < n; i++) N->marks[i].lin = is_lin(i);
    }
    std::map<std::tuple<uint32_t, uint32_t, bool, uint32_t>, int32_t> coef_ix;
    auto coef = [&](uint32_t c, uint32_t e, bool neg, uint32_t xk = 0) {
        auto key = std::make_tuple(c, e, neg, xk);
        auto it = coef_ix.find(key);
        if (it != coef_ix.end()) return it->second;
        N->coefs.push_back({c, e, neg, xk, false});
        return coef_ix[key] = (int32_t)N->coefs.size() - 1;
    };
    if (chunk_cost == 0) chunk_cost = 11000; // ~90 KB of code: measured best of 3 000 / 5 500 / 11 000 / 25 000 (instruction cache vs spill traffic)
    for (int attempt = 0; attempt < 2; attempt++) {
        N->chunks.clear(); N->pieces.clear(); N->coefs.clear(); N->kslots.clear(); N->lin_terms.clear(); coef_ix.clear();
        N->est_valu = N->chain_steps = 0; N->n_lin_sums = 0;
        std::fill(N->step_coef.begin(), N->step_coef.end(), -1);
        std::fill(N->piece_first.begin(), N->piece_first.end(), -1);
        std::fill(N->piece_last.begin(), N->piece_last.end(), -1);
        // ---- chunks by estimated cost
        {
            uint64_t acc = 0;
            Chunk cur;
            for (size_t i = 0; i < n; i++) {
                const uint64_t c = cost_of(H[i], N->marks[i], H);
                N->est_valu += c;
                if (N->marks[i].type == ST_M || N->marks[i].type == ST_A || N->marks[i].type == ST_S) N->chain_steps++;
                if (acc && acc + c > chunk_cost) {
                    cur.i1 = i;
                    N->chunks.push_back(cur);
                    cur = Chunk();
                    cur.i0 = i;
                    acc = 0;
                }
                acc += c;
            }
            cur.i1 = n;
            N->chunks.push_back(cur);
        }
        // ---- pieces: the steps of one chain inside one chunk; exponents, coefficients, K constants, linear terms
        for (size_t k = 0; k < N->chunks.size(); k++) {
            std::map<int32_t, std::vector<uint32_t>> by_chain;
            for (size_t i = N->chunks[k].i0; i < N->chunks[k].i1; i++)
                if (N->marks[i].type == ST_M || N->marks[i].type == ST_A || N->marks[i].type == ST_S) by_chain[N->marks[i].chain].push_back((uint32_t)i);
            for (auto &kv : by_chain) {
                const std::vector<uint32_t> &st = kv.second;
                Piece pc;
                pc.chain = kv.first;
                pc.first = st.front();
                pc.last = st.back();
                pc.n_m = 0;
                for (uint32_t i : st) pc.n_m += N->marks[i].type == ST_M;
                const uint32_t c = (uint32_t)chains[kv.first].chal;
                if (pc.n_m) pc.begin_coef = coef(c, pc.n_m, false);
                std::vector<KTerm> kt;
                uint32_t after = pc.n_m;
                for (uint32_t i : st) {
                    const Mark &m = N->marks[i];
                    if (m.type == ST_M) { after--; continue; }
                    const bool neg = m.type == ST_S;
                    if (m.folded >= 0) kt.push_back({(uint32_t)H[m.folded].b.off, (uint32_t)coef(c, after, !neg)}); // y +- (pol - eval): the evaluation enters with the opposite sign
                    if (!m.lin) {
                        N->step_coef[i] = (after == 0 && !neg) ? -1 : coef(c, after, neg);
                        continue;
                    }
                    if (pc.lin_sum < 0) pc.lin_sum = (int32_t)N->n_lin_sums++;
                    const uint32_t lk = m.folded >= 0 ? (H[m.folded].op >> 16) & 255 : (H[i].op >> (16 + 8 * (1 - m.yside))) & 255;
                    uint32_t col = 0, sh = 0;
                    int dim = 1;
                    (void)resolve_pol(P, lk, m.folded >= 0 ? H[m.folded].a : (m.yside == 0 ? H[i].b : H[i].a), col, dim, sh);
                    N->lin_first[i] = (int32_t)N->lin_terms.size();
                    for (int j = 0; j < dim; j++) N->lin_terms.push_back({col + (uint32_t)j, (uint32_t)coef(c, after, neg, (uint32_t)j), (uint32_t)pc.lin_sum});
                }
                if (!kt.empty()) { N->kslots.push_back(kt); pc.k_slot = (int32_t)N->kslots.size() - 1; }
                N->pieces.push_back(pc);
                const uint32_t pi = (uint32_t)N->pieces.size() - 1;
                N->piece_first[pc.first] = (int32_t)pi;
                N->piece_last[pc.last] = (int32_t)pi;
                N->chunks[k].pieces.push_back(pi);
            }
        }
        if (N->n_lin_sums <= (uint32_t)LIN_MAX_SUMS) break;
        for (Mark &m : N->marks) m.lin = 0; // more sums than one pass of the linear kernel carries: generated terms after all
    }
    // ---- the linear kernel's tables: terms by (section, slab of LIN_COLS columns, sum), every range padded to pairs with zero terms
    if (!N->lin_terms.empty()) {
        N->coefs.push_back({0, 0, false, 0, true});
        const uint32_t zero_coef = (uint32_t)N->coefs.size() - 1;
        for (size_t si = 0; si < P->sections.size(); si++) {
            const HostSection &S = P->sections[si];
            for (uint32_t c0 = 0; c0 < S.ncols; c0 += LIN_COLS) {
                // the slab's terms by sum; more than the kernel's LDS block holds: further entries over the same columns
                const uint32_t ncols = std::min<uint32_t>(LIN_COLS, (uint32_t)S.ncols - c0);
                std::vector<std::vector<LinTerm>> by_sum(LIN_MAX_SUMS);
                size_t total = 0;
                for (const LinTermH &t : N->lin_terms)
                    if (t.staged_col >= S.col0 + c0 && t.staged_col < S.col0 + c0 + ncols) {
                        by_sum[t.sum].push_back({(t.staged_col - S.col0 - c0) * 8, t.coef});
                        total++;
                    }
                std::vector<size_t> pos(LIN_MAX_SUMS, 0);
                while (total) {
                    LinSlabD d = {};
                    d.section = (uint32_t)si; d.col0 = c0; d.ncols = ncols;
                    uint32_t room = LIN_TMAX - 2; // leaving the pair padding
                    for (int sum = 0; sum < LIN_MAX_SUMS; sum++) {
                        d.t0[sum] = (uint32_t)N->lin_dev.size();
                        while (pos[sum] < by_sum[sum].size() && room) {
                            N->lin_dev.push_back(by_sum[sum][pos[sum]++]);
                            room--;
                            total--;
                        }
                        if ((N->lin_dev.size() - d.t0[sum]) & 1) { N->lin_dev.push_back({0, zero_coef}); if (room) room--; }
                    }
                    d.t0[LIN_MAX_SUMS] = (uint32_t)N->lin_dev.size();
                    MI_REQUIRE(d.t0[LIN_MAX_SUMS] - d.t0[0] <= LIN_TMAX, "internal: linear slab entry over its LDS block");
                    N->lin_slabs.push_back(d);
                }
            }
        }
    }
    N->k_off = N->coef_off + 3 * (uint32_t)N->coefs.size();
    N->cst_words = N->k_off + 3 * (uint32_t)N->kslots.size();
    MI_REQUIRE(N->cst_words < (1u << 17), "constants table too large for scalar-load offsets");
    // ---- liveness of temp words across chunks (backwards): live_in = exposed uses + (live_out - defs).  Chain steps read their
    // leaf (unless folded) and, at the start of a piece, the chain value; they write only at the end of a piece.
    const size_t nc = N->chunks.size();
    std::vector<std::set<uint32_t>> use(nc), def(nc), live_in(nc + 1);
    for (size_t k = 0; k < nc; k++) {
        for (size_t i = N->chunks[k].i0; i < N->chunks[k].i1; i++) {
            const DInstr &d = H[i];
            const Mark &m = N->marks[i];
            if (m.type == ST_FOLDED) continue;
            std::vector<uint32_t> r;
            const Opd *os[2] = {&d.a, &d.b};
            const bool step = m.type == ST_M || m.type == ST_A || m.type == ST_S;
            for (int s = 0; s < 2; s++) {
                if (step && s == m.yside && N->piece_first[i] < 0) continue;   // the chain value is in the accumulators
                if (step && s != m.yside && (m.type == ST_M || m.folded >= 0)) continue; // a challenge / a folded leaf
                words_of((d.op >> (16 + 8 * s)) & 255, *os[s], r);
            }
            for (uint32_t x : r) if (!def[k].count(x)) use[k].insert(x);
            if (step && N->piece_last[i] < 0) continue;
            const uint32_t dk = (d.op >> 8) & 255;
            if (dk == K_T1) def[k].insert(d.dst);
            else if (dk == K_T3) { def[k].insert(d.dst); def[k].insert(d.dst + 1); def[k].insert(d.dst + 2); }
        }
    }
    for (size_t k = nc; k-- > 0;) {
        live_in[k] = use[k];
        for (uint32_t x : live_in[k + 1]) if (!def[k].count(x)) live_in[k].insert(x);
        for (uint32_t x : def[k]) if (live_in[k + 1].count(x)) N->chunks[k].stores.push_back(x);
        N->chunks[k].loads.assign(use[k].begin(), use[k].end());
        std::set<uint32_t> touched(use[k].begin(), use[k].end());
        touched.insert(def[k].begin(), def[k].end());
        N->chunks[k].touched.assign(touched.begin(), touched.end());
        N->spill_words_moved += N->chunks[k].loads.size() + N->chunks[k].stores.size();
    }
    if (!live_in[0].empty()) {
        mi_set_error("mi_chelpers_build_native: internal: temporary %u is read before it is written", *live_in[0].begin());
        return MI_ERR_INVALID;
    }
    // ---- which 64-column slabs of every section the generated kernels read: only those go into the tile-major copy
    N->sec_slab_mask.assign(P->sections.size(), 0);
    N->sec_inplace_mask.assign(P->sections.size(), 0);
    std::map<std::pair<size_t, uint32_t>, uint32_t> slab_loads; // (section, slab) -> operand loads per row out of it, all kernels
    std::set<uint64_t> chunk_ops, all_ops; // (staged column, shift) read by the current kernel / by any: what CSE leaves of the operand reads
    size_t chunk_of = 0;
    for (size_t i = 0; i < n; i++) {
        while (chunk_of + 1 < N->chunks.size() && i >= N->chunks[chunk_of].i1) { N->operand_loads += chunk_ops.size(); chunk_ops.clear(); chunk_of++; }
        const Mark &m = N->marks[i];
        if (m.type == ST_FOLDED || m.type == ST_M || m.lin) continue;
        const DInstr &d = H[i];
        const Opd *os[2] = {&d.a, &d.b};
        uint32_t kk[2] = {(d.op >> 16) & 255, d.op >> 24};
        if ((m.type == ST_A || m.type == ST_S)) {
            kk[m.yside] = K_NONE; // the chain value
            if (m.folded >= 0) { kk[1 - m.yside] = (H[m.folded].op >> 16) & 255; os[1 - m.yside] = &H[m.folded].a; }
        }
        for (int s2 = 0; s2 < 2; s2++) {
            if (!is_pol_kind(kk[s2])) continue;
            uint32_t col = 0, sh = 0;
            int dim = 1;
            const HostSection *S = resolve_pol(P, kk[s2], *os[s2], col, dim, sh);
            if (!S) continue; // reported by the generator
            MI_REQUIRE(S->ncols <= 2048, "sections wider than 2048 columns are not supported by the tile-major copy");
            for (int j = 0; j < dim; j++) {
                const size_t si = S - P->sections.data();
                const uint32_t slab = (col + j - S->col0) / 64;
                if (!S->tiled) N->sec_slab_mask[si] |= 1u << slab;
                if (chunk_ops.insert(((uint64_t)(col + j) << 8) | sh).second) slab_loads[{si, slab}]++;
                all_ops.insert(((uint64_t)(col + j) << 8) | sh);
            }
        }
    }
    N->operand_loads += chunk_ops.size();
    N->operand_distinct = all_ops.size();
    // A slab the kernels load only a few elements of per row is not worth its copy (64 columns read and written: 1 KiB per row; the
    // zkEVM's step52ns left FOUR polynomial elements to the generated kernel -- the rest are linear terms -- and paid 16 ms of copies for
    // them): those elements are read in place, one 64-byte sector per lane and load.
    uint32_t inplace_max = 8;
    if (const char *e = getenv("MI_CHELPERS_INPLACE_MAX")) inplace_max = (uint32_t)std::max(0, atoi(e));
    for (const auto &kv : slab_loads) {
        const HostSection &S = P->sections[kv.first.first];
        if (S.tiled || S.role > 1 || kv.second > inplace_max) continue;
        N->sec_inplace_mask[kv.first.first] |= 1u << kv.first.second;
        N->sec_slab_mask[kv.first.first] &= ~(1u << kv.first.second);
    }
    return MI_OK;
}

// challenges | public inputs | evaluations | ZhInv (256 words) | chain coefficients (+- C^e) | per-piece K constants, all canonical
static void fill_constants(const mi_chelpers_prog *P, const NativeProg *N, const mi_chelpers_params *a, std::vector<u64> &cst)
{
    cst.assign(N->cst_words + 8, 0);
    for (uint64_t i = 0; i < P->max_chal * 3; i++) cst[i] = gl::canon(a->challenges[i]);
    for (uint64_t i = 0; i < P->max_pub; i++) cst[N->pub_off + i] = gl::canon(a->publics[i]);
    for (uint64_t i = 0; i < P->max_eval * 3; i++) cst[N->ev_off + i] = gl::canon(a->evals[i]);
    const uint64_t n_zh = P->step == MI_CHELPERS_STEP42NS ? a->n_zhinv : 0;
    for (uint64_t i = 0; i < n_zh && i < 256 && a->zhinv; i++) cst[N->zh_off + i] = gl::canon(a->zhinv[i]);
    std::map<uint32_t, std::vector<E3>> pw;
    for (const Coef &c : N->coefs) {
        if (c.zero) continue;
        std::vector<E3> &v = pw[c.chal];
        if (v.empty()) v.push_back(E3{{1, 0, 0}});
        const E3 C = {{cst[c.chal * 3], cst[c.chal * 3 + 1], cst[c.chal * 3 + 2]}};
        while (v.size() <= c.exp) v.push_back(gl::e3_mul(v.back(), C));
    }
    for (size_t i = 0; i < N->coefs.size(); i++) {
        const Coef &c = N->coefs[i];
        if (c.zero) continue;
        E3 p = pw[c.chal][c.exp];
        for (uint32_t k = 0; k < c.xk; k++) p = E3{{p.v[2], gl::add(p.v[0], p.v[2]), p.v[1]}}; // times x: x^3 = x + 1
        for (int j = 0; j < 3; j++) cst[N->coef_off + 3 * i + j] = c.neg ? gl::neg(p.v[j]) : p.v[j];
    }
    for (size_t s = 0; s < N->kslots.size(); s++) {
        E3 K = {{0, 0, 0}};
        for (const KTerm &t : N->kslots[s]) {
            const E3 e = {{cst[N->ev_off + 3 * t.eval], cst[N->ev_off + 3 * t.eval + 1], cst[N->ev_off + 3 * t.eval + 2]}};
            const E3 w = {{cst[N->coef_off + 3 * t.coef], cst[N->coef_off + 3 * t.coef + 1], cst[N->coef_off + 3 * t.coef + 2]}};
            K = gl::e3_add(K, gl::e3_mul(e, w));
        }
        for (int j = 0; j < 3; j++) cst[N->k_off + 3 * s + j] = K.v[j];
    }
}

// The generated kernel is one basic block of tens of thousands of instructions with a lot of independent work (every constraint
// value, every product of a chain term): left alone, LLVM's scheduler hoists hundreds of loads and products to the top, runs out
// of 512 VGPRs and spills (measured: 16 000 spilled VGPRs and 17 000 spilled SGPRs per kernel).  The program order the translator
// produced is already the low-pressure order (depth-first, Sethi-Ullman), so the generator fixes the schedule itself: the
// instructions are emitted in GROUPS of a few hundred VALU instructions, fenced by __builtin_amdgcn_sched_barrier(0), and the
// polynomial loads of group g + 1 are issued before the arithmetic of group g (software prefetch, one group ahead).
struct Group { std::string loads, compute; };

struct Gen {
    const mi_chelpers_prog *P;
    const NativeProg *N;
    std::string body;
    std::vector<Group> groups;
    std::map<std::string, std::string> group_loads; // load expression -> variable, current group
    std::map<std::string, std::pair<std::string, uint32_t>> recent; // load expression -> (variable, group that loaded it)
    uint32_t n_loads = 0, cur_group = 0, reuse_window = 1u << 30; // a loaded element stays named for the rest of the kernel (measured: 297 -> 272 ms against reloading per group)
    void end_group()
    {
        Group g;
        g.compute.swap(body);
        for (auto &kv : group_loads) g.loads += "  const u64 " + kv.second + " = " + kv.first + ";\n";
        group_loads.clear();
        groups.push_back(g);
        cur_group++;
    }
    std::string load(const std::string &expr)
    {
        auto it = group_loads.find(expr);
        if (it != group_loads.end()) return it->second;
        auto r = recent.find(expr); // still in its register from a group or two ago?
        if (r != recent.end() && cur_group - r->second.second <= reuse_window) return r->second.first;
        const std::string name = "l" + std::to_string(n_loads++);
        recent[expr] = {name, cur_group};
        return group_loads[expr] = name;
    }
    std::set<uint32_t> shifts; // row shifts read from the tile-major copy
    std::set<std::pair<uint32_t, uint32_t>> xshifts; // (section kept tile-major, row shift) pairs read in place (shift 0 included)
    bool uses_zh = false;
    // Stores into params.pols at the row itself (the base-domain steps): a lane owns a row, so a store is 64 eight-byte writes a pitch
    // apart -- one 64-byte sector touched per word (measured: 19 of step3's 95 ms, 11 of step3prev's 47).  The words a kernel stores are
    // mostly ADJACENT columns (a step's outputs are laid out in the order it computes them), so they wait in LDS ([slot][row], 65-word
    // slot pitch) and the epilogue writes each run of adjacent columns row by row: runs of 8 L bytes instead of L scattered words.
    struct LdsStore { uint64_t off; uint32_t stride; };
    std::vector<LdsStore> lds_stores;
    uint32_t lds_cap = 0, lds_slots = 0; // cap 0: stores go straight to memory; lds_slots: the most slots in use at a time (the array's size)
    // the staged words out to memory: every run of adjacent columns, row by row (one wave per workgroup: the barriers order its LDS traffic).
    // Called when the slots are full and at the end of the kernel.
    void flush_lds(std::string &dst)
    {
        if (lds_stores.empty()) return;
        lds_slots = std::max<uint32_t>(lds_slots, (uint32_t)lds_stores.size());
        char ln[640];
        dst += "  __syncthreads();\n";
        for (size_t s0 = 0; s0 < lds_stores.size();) {
            size_t s1 = s0 + 1;
            while (s1 < lds_stores.size() && lds_stores[s1].stride == lds_stores[s0].stride && lds_stores[s1].off == lds_stores[s1 - 1].off + 1) s1++;
            const unsigned L = (unsigned)(s1 - s0);
            snprintf(ln, sizeof ln,
                     "  for (u32 i = lane; i < %uu; i += 64u) { const u32 rr = i / %uu, j = i - rr * %uu; const u64 grow = row_base + tile * 64 + rr; "
                     "if (grow < row_end) out[%lluULL + j + grow * %uULL] = LS[(%uu + j) * 65u + rr]; }\n",
                     64u * L, L, L, (unsigned long long)lds_stores[s0].off, lds_stores[s0].stride, (unsigned)s0);
            dst += ln;
            s0 = s1;
        }
        dst += "  __syncthreads();\n";
        lds_stores.clear();
    }
    std::vector<uint8_t> canon; // per temp word: known canonical
    char buf[256];

    struct V { std::string e[3]; bool c[3] = {true, true, true}; int dim = 1; };

    const HostSection *find(int role, uint64_t off, uint64_t stride, uint64_t width, uint32_t &col) const
    {
        for (const HostSection &S : P->sections) {
            if (S.role != role) continue;
            if (role == 0 && (stride != S.ncols || off < S.offset || off - S.offset + width > S.ncols)) continue;
            if (role != 0 && off + width > S.ncols) continue;
            col = S.col0 + (uint32_t)(role == 0 ? off - S.offset : off);
            return &S;
        }
        return nullptr;
    }
    std::string cstw(uint64_t w) { snprintf(buf, sizeof buf, "cst[%llu]", (unsigned long long)w); return buf; }
    int operand(uint32_t k, const Opd &o, V &v)
    {
        v = V();
        switch (k) {
        case K_NONE: v.dim = 0; return MI_OK;
        case K_T1:
            snprintf(buf, sizeof buf, "t%llu", (unsigned long long)o.off);
            v.e[0] = buf; v.c[0] = canon[o.off];
            return MI_OK;
        case K_T3:
            v.dim = 3;
            for (int j = 0; j < 3; j++) {
                snprintf(buf, sizeof buf, "t%llu", (unsigned long long)o.off + j);
                v.e[j] = buf; v.c[j] = canon[o.off + j];
            }
            return MI_OK;
        case K_NUM: snprintf(buf, sizeof buf, "0x%llxULL", (unsigned long long)o.off); v.e[0] = buf; return MI_OK;
        case K_CHAL: v.dim = 3; for (int j = 0; j < 3; j++) v.e[j] = cstw(o.off * 3 + j); return MI_OK;
        case K_PUB: v.e[0] = cstw(N->pub_off + o.off); return MI_OK;
        case K_EVAL: v.dim = 3; for (int j = 0; j < 3; j++) v.e[j] = cstw(N->ev_off + o.off * 3 + j); return MI_OK;
        case K_ZHINV: uses_zh = true; v.e[0] = "zh"; return MI_OK;
        default: break;
        }
        uint32_t col = 0, sh = 0;
        int pdim = 1;
        const HostSection *S = resolve_pol(P, k, o, col, pdim, sh);
        if (!S) {
            mi_set_error("mi_chelpers_build_native: operand (offset %llu, stride %u, shift %u) lies in none of the declared sections / constant polynomials / x, "
                         "or is shifted by 64 rows or more", (unsigned long long)o.off, o.stride, o.shift);
            return MI_ERR_INVALID;
        }
        const bool three = pdim == 3;
        if (S->tiled) xshifts.insert({(uint32_t)(S - P->sections.data()), sh}); // read in place: X<section>_<shift>, columns counted from the section's first
        else if (sh) shifts.insert(sh);   // (a T<shift> no load names is dropped by the compiler)
        v.dim = three ? 3 : 1;
        for (int j = 0; j < v.dim; j++) {
            const uint32_t sc = col - S->col0 + j; // column of the section
            if (S->tiled) snprintf(buf, sizeof buf, "X%u_%u[%llu]", (uint32_t)(S - P->sections.data()), sh, (unsigned long long)sc * 64);
            else if ((N->sec_inplace_mask[S - P->sections.data()] >> (sc / 64)) & 1) // in place, row-major (role 0: the polynomial area, 1: the constants)
                snprintf(buf, sizeof buf, "%s[%lluULL + ((row + %uu) & %lluULL) * %lluULL]", S->role == 0 ? "pols" : "cpols",
                         (unsigned long long)((S->role == 0 ? S->offset : 0) + sc), sh, (unsigned long long)(S->nrows - 1),
                         (unsigned long long)(S->role == 0 ? S->ncols : P->n_const));
            else snprintf(buf, sizeof buf, "T%u[%llu]", sh, (unsigned long long)(col + j) * 64);
            v.e[j] = load(buf);
        }
        return MI_OK;
    }
    // a + b / a - b with whatever is known about the operands
    std::string add(const std::string &a, bool ac, const std::string &b, bool bc)
    {
        if (bc) return "gl::add_wc(" + a + ", " + b + ")";
        if (ac) return "gl::add_wc(" + b + ", " + a + ")";
        return "addg(" + a + ", " + b + ")";
    }
    std::string sub(const std::string &a, const std::string &b, bool bc) { return (bc ? "gl::sub_wc(" : "subg(") + a + ", " + b + ")"; }

    int chain_step(size_t i, const DInstr &d, const Mark &m)
    {
        const uint32_t kk[2] = {(d.op >> 16) & 255, d.op >> 24};
        const Opd *os[2] = {&d.a, &d.b};
        char acc[3][32];
        auto name = [&](int32_t pi) { for (int j = 0; j < 3; j++) snprintf(acc[j], sizeof acc[j], "p%d_%d", pi, j); };
        body += "  { ";
        const int32_t pf = N->piece_first[i];
        if (pf >= 0) { // the chain value enters the accumulators, already times C^(multiplications of this piece)
            const Piece &pc = N->pieces[pf];
            V y;
            MI_TRY(operand(kk[m.yside], *os[m.yside], y));
            name(pf);
            if (pc.begin_coef < 0) {
                for (int j = 0; j < 3; j++) body += std::string("chpa::acc_set(") + acc[j] + ", " + y.e[j] + "); ";
            } else {
                const uint64_t w = N->coef_off + 3 * (uint64_t)pc.begin_coef;
                for (int j = 0; j < 3; j++) body += std::string("chpa::acc_set(") + acc[j] + ", 0); ";
                body += std::string("chpa::acc_mul33_s(") + acc[0] + ", " + acc[1] + ", " + acc[2] + ", " + y.e[0] + ", " + y.e[1] + ", " + y.e[2] + ", " +
                        cstw(w) + ", " + cstw(w + 1) + ", " + cstw(w + 2) + "); ";
            }
        }
        // which piece is this step in?  (the one that began last for its chain inside this chunk)
        int32_t pi = -1;
        for (size_t q = N->pieces.size(); q-- > 0;)
            if (N->pieces[q].chain == m.chain && N->pieces[q].first <= i && i <= N->pieces[q].last) { pi = (int32_t)q; break; }
        MI_REQUIRE(pi >= 0, "internal: chain step outside every piece");
        name(pi);
        if (m.type != ST_M && !m.lin) {
            V v;
            if (m.folded >= 0) MI_TRY(operand((P->host[m.folded].op >> 16) & 255, P->host[m.folded].a, v));
            else MI_TRY(operand(kk[1 - m.yside], *os[1 - m.yside], v));
            const int32_t ci = N->step_coef[i];
            if (ci < 0) {
                for (int j = 0; j < v.dim; j++) body += std::string("chpa::acc_add(") + acc[j] + ", " + v.e[j] + "); ";
            } else {
                const uint64_t w = N->coef_off + 3 * (uint64_t)ci;
                body += std::string(v.dim == 3 ? "chpa::acc_mul33_s(" : "chpa::acc_mul13_s(") + acc[0] + ", " + acc[1] + ", " + acc[2] + ", " + v.e[0];
                if (v.dim == 3) body += ", " + v.e[1] + ", " + v.e[2];
                body += ", " + cstw(w) + ", " + cstw(w + 1) + ", " + cstw(w + 2) + "); ";
            }
        }
        if (N->piece_last[i] >= 0) {
            const Piece &pc = N->pieces[N->piece_last[i]];
            for (int j = 0; j < 3; j++) {
                snprintf(buf, sizeof buf, "t%u = ", d.dst + j);
                const std::string lhs = buf; // cstw() formats into buf too
                std::string r = std::string("chpa::acc_reduce(") + acc[j] + ")";
                if (pc.k_slot >= 0) r = "gl::add_wc(" + r + ", " + cstw(N->k_off + 3 * (uint64_t)pc.k_slot + j) + ")";
                if (pc.lin_sum >= 0) { // the polynomial terms of the piece, summed by k_chp_linear (canonical)
                    snprintf(buf, sizeof buf, ", LIN[%u])", (unsigned)((pc.lin_sum * 3 + j) * 64));
                    r = "gl::add_wc(" + r + buf;
                }
                body += lhs + r + "; ";
                canon[d.dst + j] = 0;
            }
        }
        body += "}\n";
        return MI_OK;
    }

    int instr(size_t i)
    {
        const DInstr &d = P->host[i];
        const Mark &m = N->marks[i];
        if (m.type == ST_FOLDED) return MI_OK;
        if (m.type != ST_NONE) return chain_step(i, d, m);
        const uint32_t cls = d.op & 255, dk = (d.op >> 8) & 255, ak = (d.op >> 16) & 255, bk = d.op >> 24;
        V a, b;
        MI_TRY(operand(ak, d.a, a));
        if (cls == C_STOREP) { // params.pols[off + row' * stride] = a, row' = row or (row + shift) mod n (out = the polynomial memory here)
            for (const HostSection &S : P->sections)
                MI_REQUIRE(!(S.tiled && S.role == 0 && d.b.off >= S.offset && d.b.off < S.offset + S.ncols),
                           "the program stores into the section declared tile-major (it is read in place, row-major stores would corrupt it)");
            if (dk != K_DPOLS && (uint32_t)a.dim <= lds_cap) {
                if (lds_stores.size() + (size_t)a.dim > lds_cap) flush_lds(body); // the slots are full: out they go, the slots start over
                for (int j = 0; j < a.dim; j++) {
                    snprintf(buf, sizeof buf, "  LS[%uu * 65u + lane] = ", (unsigned)lds_stores.size());
                    body += buf + (a.c[j] ? a.e[j] : "gl::canon(" + a.e[j] + ")") + ";\n";
                    lds_stores.push_back({d.b.off + (uint64_t)j, d.b.stride});
                }
                return MI_OK;
            }
            body += "  if (row < row_end) { ";
            if (dk == K_DPOLS) {
                MI_REQUIRE(d.b.mod && (d.b.mod & (d.b.mod - 1)) == 0, "shifted-row destination: the modulus must be a power of two");
                snprintf(buf, sizeof buf, "u64 *po = out + %lluULL + ((row + %uu) & %lluULL) * %uULL; ", (unsigned long long)d.b.off, d.b.shift,
                         (unsigned long long)(d.b.mod - 1), d.b.stride);
            } else {
                snprintf(buf, sizeof buf, "u64 *po = out + %lluULL + row * %uULL; ", (unsigned long long)d.b.off, d.b.stride);
            }
            body += buf;
            for (int j = 0; j < a.dim; j++) {
                snprintf(buf, sizeof buf, "po[%d] = ", j);
                body += buf + (a.c[j] ? a.e[j] : "gl::canon(" + a.e[j] + ")") + "; ";
            }
            body += "}\n";
            return MI_OK;
        }
        MI_TRY(operand(bk, d.b, b));
        const int rdim = (dk == K_T3 || dk == K_Q) ? 3 : 1;
        std::string o[3];
        bool oc[3] = {false, false, false};
        std::string pre; // statements before the result expressions
        switch (cls) {
        case C_ADD: case C_SUB:
            for (int j = 0; j < rdim; j++) {
                const bool ha = j < a.dim, hb = j < b.dim;
                if (ha && hb) { o[j] = cls == C_ADD ? add(a.e[j], a.c[j], b.e[j], b.c[j]) : sub(a.e[j], b.e[j], b.c[j]); }
                else if (ha) { o[j] = a.e[j]; oc[j] = a.c[j]; }
                else if (hb) {
                    if (cls == C_ADD) { o[j] = b.e[j]; oc[j] = b.c[j]; }
                    else o[j] = b.c[j] ? "negc(" + b.e[j] + ")" : "negc(gl::canon(" + b.e[j] + "))";
                } else { o[j] = "0ULL"; oc[j] = true; }
            }
            break;
        case C_MUL: case C_STOREQ:
            if (a.dim == 3 && b.dim == 3) {
                pre = "u64 m0, m1, m2; mul33(m0, m1, m2, " + a.e[0] + ", " + a.e[1] + ", " + a.e[2] + ", " + b.e[0] + ", " + b.e[1] + ", " + b.e[2] + "); ";
                o[0] = "m0"; o[1] = "m1"; o[2] = "m2";
            } else if (a.dim == 3) {
                for (int j = 0; j < 3; j++) o[j] = "gl::mul_w(" + a.e[j] + ", " + b.e[0] + ")";
            } else if (b.dim == 3) {
                for (int j = 0; j < 3; j++) o[j] = "gl::mul_w(" + a.e[0] + ", " + b.e[j] + ")";
            } else {
                o[0] = "gl::mul_w(" + a.e[0] + ", " + b.e[0] + ")";
                o[1] = o[2] = "0ULL"; oc[1] = oc[2] = true;
            }
            break;
        default: // C_COPY, C_STOREF
            for (int j = 0; j < rdim; j++) {
                if (j < a.dim) { o[j] = a.e[j]; oc[j] = a.c[j]; } else { o[j] = "0ULL"; oc[j] = true; }
            }
            break;
        }
        body += "  { " + pre;
        if (dk == K_T1 || dk == K_T3) {
            // results first, then the assignments: a destination may be one of the sources
            for (int j = 0; j < rdim; j++) { snprintf(buf, sizeof buf, "const u64 r%d = ", j); body += buf + o[j] + "; "; }
            for (int j = 0; j < rdim; j++) {
                snprintf(buf, sizeof buf, "t%u = r%d; ", d.dst + j, j);
                body += buf;
                canon[d.dst + j] = oc[j];
            }
        } else if (dk == K_Q) {
            body += "if (row < row_end) { ";
            for (int j = 0; j < 3; j++) {
                snprintf(buf, sizeof buf, "out[row * 3 + %d] = ", j);
                body += buf + (oc[j] ? o[j] : "gl::canon(" + o[j] + ")") + "; ";
            }
            body += "} ";
        }
        body += "}\n";
        return MI_OK;
    }
};

static int compile_source(const std::string &src, const std::string &cache_dir, std::vector<char> &code, NativeProg &N)
{
    int rtc_major = 0, rtc_minor = 0;
    (void)hiprtcVersion(&rtc_major, &rtc_minor); // a code object is only as good as the compiler that made it: part of the key
    const std::string opts = "gfx950 -O3 c++17 v8 hiprtc " + std::to_string(rtc_major) + "." + std::to_string(rtc_minor);
    char name[64];
    snprintf(name, sizeof name, "%016llx%016llx.hsaco", (unsigned long long)fnv1a(src + opts), (unsigned long long)fnv1a(opts + src, 0x9E3779B97F4A7C15ull));
    const std::string path = cache_dir.empty() ? "" : cache_dir + "/" + name;
    if (!path.empty()) {
        if (FILE *f = fopen(path.c_str(), "rb")) {
            fseek(f, 0, SEEK_END);
            const long n = ftell(f);
            fseek(f, 0, SEEK_SET);
            code.resize(n > 0 ? (size_t)n : 0);
            const bool ok = n > 0 && fread(code.data(), 1, (size_t)n, f) == (size_t)n;
            fclose(f);
            // a code object is an ELF image; anything else at this path (a truncated or foreign file) is dropped and rebuilt
            if (ok && n > 64 && memcmp(code.data(), "\x7f" "ELF", 4) == 0) { N.cache_hits++; return MI_OK; }
            remove(path.c_str());
        }
    }
    if (const char *dump = getenv("MI_CHELPERS_DUMP_SRC")) { // debugging: the generated source, for experiments with hipcc
        if (FILE *f = fopen((std::string(dump) + "/" + name + ".hip").c_str(), "w")) { fwrite(src.data(), 1, src.size(), f); fclose(f); }
    }
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "chelpers_chunk.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
        mi_set_error("mi_chelpers_build_native: hiprtcCreateProgram failed");
        return MI_ERR_HIP;
    }
    const char *o[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
    const hiprtcResult r = hiprtcCompileProgram(prog, 3, o);
    if (r != HIPRTC_SUCCESS) {
        size_t ls = 0;
        hiprtcGetProgramLogSize(prog, &ls);
        std::string log(ls + 1, 0);
        if (ls) hiprtcGetProgramLog(prog, &log[0]);
        const size_t from = log.size() > 900 ? log.size() - 900 : 0; // the errors follow the warnings
        mi_set_error("mi_chelpers_build_native: generated kernel does not compile: %s: ...%s", hiprtcGetErrorString(r), log.c_str() + from);
        hiprtcDestroyProgram(&prog);
        return MI_ERR_HIP;
    }
    size_t cs = 0;
    hiprtcGetCodeSize(prog, &cs);
    code.resize(cs);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    if (!path.empty()) { // write-then-rename: another process may be filling the same cache
        const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
        if (FILE *f = fopen(tmp.c_str(), "wb")) {
            const bool ok = fwrite(code.data(), 1, code.size(), f) == code.size();
            fclose(f);
            if (!ok || rename(tmp.c_str(), path.c_str()) != 0) remove(tmp.c_str());
        }
    }
    return MI_OK;
}

static constexpr uint64_t GROUP_COST = 300; // estimated VALU instructions per scheduling group
static constexpr size_t GROUP_LOADS = 16;   // ... or this many distinct polynomial loads

static int generate(const mi_chelpers_prog *P, const NativeProg *N, size_t k, std::string &src)
{
    const Chunk &C = N->chunks[k];
    Gen g;
    g.P = P;
    g.N = N;
    g.canon.assign(N->nw + 3, 0);
    for (uint32_t w : C.loads) g.canon[w] = 1; // the spill holds canonical values
    {
        uint64_t acc = 0, group_cost = GROUP_COST;
        size_t group_loads = GROUP_LOADS;
        if (const char *e = getenv("MI_CHELPERS_GROUP_COST")) group_cost = (uint64_t)atoll(e);   // experiments
        if (const char *e = getenv("MI_CHELPERS_GROUP_LOADS")) group_loads = (size_t)atoll(e);
        if (const char *e = getenv("MI_CHELPERS_REUSE_WINDOW")) g.reuse_window = (uint32_t)atoi(e);
        {   // LDS-staged stores: as many slots as leave every wave the kernel is compiled for its share of the CU's 160 KB
            unsigned w_ = std::max(1u, std::min(4u, 512u / (2 * N->nw + 64)));
            if (const char *e = getenv("MI_CHELPERS_WAVES")) w_ = std::max(1, std::min(8, atoi(e)));
            const char *e = getenv("MI_CHELPERS_LDS_STORES");
            g.lds_cap = (e && e[0] == '0') ? 0u : std::min(32u, (160u * 1024u / (w_ * 4u)) / (65u * 8u));
        }
        for (size_t i = C.i0; i < C.i1; i++) {
            MI_TRY(g.instr(i));
            acc += cost_of(P->host[i], N->marks[i], P->host);
            if (acc >= group_cost || g.group_loads.size() >= group_loads) { g.end_group(); acc = 0; }
        }
        g.end_group();
    }
    // registers: two per live temporary word, the prefetched loads of two groups, three accumulators per chain, working set of a
    // multiplication.  Left alone the compiler takes all 512 (one wave per SIMD, nothing to hide a load behind); told to fit N
    // waves it spills a few hundred values to scratch instead (2 % of the instructions at N = 2 for the zkEVM-sized step42ns).
    unsigned waves = std::max(1u, std::min(4u, 512u / (2 * N->nw + 64)));
    if (const char *e = getenv("MI_CHELPERS_WAVES")) waves = std::max(1, std::min(8, atoi(e)));
    src = "#define MI_NO_RARE_BRANCH 1\n";
    src += GL_MATH_SRC;
    src += ACC_SRC;
    src += HELPERS_SRC;
    char line[1024];
    snprintf(line, sizeof line,
             "extern \"C\" __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(%u, %u))) void chelpers_chunk(const u64 *__restrict__ tiled, u64 *__restrict__ spill, "
             "const u64 *__restrict__ cst, u64 *__restrict__ out, u64 row_base, u64 row_end, u32 zmask, const u64 *__restrict__ lin, const u64 *__restrict__ pols, const u64 *__restrict__ cpols)\n{\n"
             "  const u32 lane = threadIdx.x;\n  const u64 tile = blockIdx.x;\n  const u64 row = row_base + tile * 64 + lane;\n"
             "  const u64 *__restrict__ T0 = tiled + tile * %lluULL + lane;\n  u64 *__restrict__ S = spill + tile * %lluULL + lane;\n",
             waves, waves, (unsigned long long)N->sc * 64, (unsigned long long)N->nw * 64);
    src += line;
    std::string lds_tail;
    g.flush_lds(lds_tail); // (what is still staged at the end of the kernel; fixes lds_slots)
    if (g.lds_slots) {
        snprintf(line, sizeof line, "  __shared__ u64 LS[%uu];\n", g.lds_slots * 65u);
        src += line;
    }
    for (uint32_t s : g.shifts) {
        snprintf(line, sizeof line, "  const u64 *__restrict__ T%u = tiled + (tile + ((lane + %uu) >> 6)) * %lluULL + ((lane + %uu) & 63u);\n", s, s,
                 (unsigned long long)N->sc * 64, s);
        src += line;
    }
    for (const auto &xs : g.xshifts) {
        // a section kept tile-major (a part of the polynomial area, or the constant polynomials): tile = row / 64 of the whole section
        // (row_base is a multiple of 64), rows wrap at its end
        const HostSection &S = P->sections[xs.first];
        const uint32_t s = xs.second;
        char where[64];
        if (S.role == 0) snprintf(where, sizeof where, "pols + %lluULL", (unsigned long long)S.offset);
        else snprintf(where, sizeof where, "cpols"); // (role 1: the constant polynomials, mi_chelpers_set_tiled_consts)
        snprintf(line, sizeof line, "  const u64 *__restrict__ X%u_%u = %s + (((row_base >> 6) + tile + ((lane + %uu) >> 6)) & %lluULL) * %lluULL + gl::tile_pos((lane + %uu) & 63u);\n",
                 xs.first, s, where, s, (unsigned long long)(S.nrows / 64 - 1), (unsigned long long)S.ncols * 64, s);
        src += line;
    }
    if (g.uses_zh) {
        snprintf(line, sizeof line, "  const u64 zh = cst[%u + (u32)(row & zmask)];\n", N->zh_off);
        src += line;
    }
    if (N->n_lin_sums) {
        snprintf(line, sizeof line, "  const u64 *__restrict__ LIN = lin + tile * %uULL + lane;\n", N->n_lin_sums * 3 * 64);
        src += line;
    }
    std::set<uint32_t> loads(C.loads.begin(), C.loads.end());
    for (uint32_t w : C.touched) {
        if (loads.count(w)) snprintf(line, sizeof line, "  u64 t%u = S[%llu];\n", w, (unsigned long long)w * 64);
        else snprintf(line, sizeof line, "  u64 t%u;\n", w);
        src += line;
    }
    for (uint32_t pi : C.pieces) {
        snprintf(line, sizeof line, "  chpa::Acc p%u_0, p%u_1, p%u_2;\n", pi, pi, pi);
        src += line;
    }
    size_t ahead = 1; // groups whose loads are in flight ahead of the arithmetic
    if (const char *e = getenv("MI_CHELPERS_PREFETCH")) ahead = (size_t)std::max(1, std::min(4, atoi(e)));
    for (size_t gi = 0; gi < ahead && gi < g.groups.size(); gi++) src += g.groups[gi].loads;
    for (size_t gi = 0; gi < g.groups.size(); gi++) {
        if (gi + ahead < g.groups.size()) src += g.groups[gi + ahead].loads;
        src += "  __builtin_amdgcn_sched_barrier(0);\n";
        src += g.groups[gi].compute;
        src += "  __builtin_amdgcn_sched_barrier(0);\n";
    }
    for (uint32_t w : C.stores) {
        snprintf(line, sizeof line, g.canon[w] ? "  S[%llu] = t%u;\n" : "  S[%llu] = gl::canon(t%u);\n", (unsigned long long)w * 64, w);
        src += line;
    }
    src += lds_tail;
    src += "}\n";
    return MI_OK;
}

int native_build(mi_chelpers_prog *P, const char *cache_dir_arg, uint64_t chunk_cost, uint32_t shard, uint32_t nshards)
{
    MI_REQUIRE(P && !P->host.empty(), "no translated program");
    MI_REQUIRE(!P->native, "native code was already built for this program");
    MI_REQUIRE(!P->sections.empty(), "the program was compiled without sections");
    MI_REQUIRE(nshards >= 1 && shard < nshards, "bad shard");
    std::string cache_dir = cache_dir_arg ? cache_dir_arg : "";
    if (cache_dir.empty())
        if (const char *e = getenv("MI_CHELPERS_CACHE")) cache_dir = e;
    if (!cache_dir.empty()) (void)mkdir(cache_dir.c_str(), 0777);
    MI_REQUIRE(nshards == 1 || !cache_dir.empty(), "a sharded build only fills the cache: it needs a cache directory");
    const auto t0 = std::chrono::steady_clock::now();
    NativeProg *N = new NativeProg();
    const char *no_chains = getenv("MI_CHELPERS_NO_CHAINS");
    int st = lower(P, N, chunk_cost, !(no_chains && no_chains[0] == '1'));
    for (size_t k = 0; k < N->chunks.size() && st == MI_OK; k++) {
        if (k % nshards != shard) continue;
        std::string src;
        st = generate(P, N, k, src);
        if (st == MI_OK) st = compile_source(src, cache_dir, N->chunks[k].code, *N);
        N->code_bytes += N->chunks[k].code.size();
    }
    if (st != MI_OK || nshards > 1) { // a shard of a parallel build has filled its part of the cache: nothing to keep
        delete N;
        return st;
    }
    N->compile_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    P->native = N;
    return MI_OK;
}

int native_lower_stats(const mi_chelpers_prog *P, uint64_t chunk_cost, uint64_t out[12])
{
    NativeProg N;
    MI_TRY(lower(P, &N, chunk_cost, true));
    uint64_t folded = 0;
    for (const Mark &m : N.marks) folded += m.type == ST_FOLDED;
    out[0] = N.chunks.size(); out[1] = N.chain_steps; out[2] = N.pieces.size(); out[3] = N.est_valu;
    out[4] = N.coefs.size(); out[5] = N.kslots.size(); out[6] = folded; out[7] = N.spill_words_moved;
    out[8] = N.operand_loads; out[9] = N.operand_distinct; out[10] = N.lin_terms.size(); out[11] = N.n_lin_sums;
    return MI_OK;
}

void native_stats(const mi_chelpers_prog *P, uint64_t out[8])
{
    const NativeProg *N = P->native;
    for (int i = 0; i < 8; i++) out[i] = 0;
    if (!N) return;
    out[0] = N->chunks.size();
    out[1] = N->code_bytes;
    out[2] = (uint64_t)(N->compile_s * 1000.0);
    out[3] = N->cache_hits;
    out[4] = N->est_valu;
    out[5] = N->spill_words_moved; // temp words loaded + stored at chunk boundaries, per row
    out[6] = N->chain_steps;       // instructions turned into Horner-chain accumulator steps
    out[7] = N->cst_words;
}

void native_free(mi_ctx *c, mi_chelpers_prog *P)
{
    NativeProg *N = P->native;
    if (!N) return;
    for (Chunk &C : N->chunks)
        if (C.mod) (void)hipModuleUnload(C.mod);
    if (N->d_lin_terms) (void)hipFree(N->d_lin_terms);
    if (N->d_lin_slabs) (void)hipFree(N->d_lin_slabs);
    (void)c;
    delete N;
    P->native = nullptr;
}

// ---- host debug executor of the LOWERED program (tests only): the same chains, pieces, coefficients, K constants and spill
// lists the kernels are generated from, run on the CPU over host pointers with the same accumulator code
int native_host_run(const mi_chelpers_prog *P, const mi_chelpers_params *a, const uint64_t *rows, uint64_t nrows, uint64_t chunk_cost)
{
    NativeProg Nl;
    NativeProg *N = &Nl;
    MI_TRY(lower(P, N, chunk_cost, true));
    std::vector<u64> cst;
    fill_constants(P, N, a, cst);
    std::vector<u64> chal(P->max_chal * 3 + 1), pub(P->max_pub + 1), zh(a->n_zhinv + 1), ev(P->max_eval * 3 + 1);
    for (uint64_t i = 0; i < P->max_eval * 3; i++) ev[i] = gl::canon(a->evals[i]);
    for (uint64_t i = 0; i < P->max_chal * 3; i++) chal[i] = gl::canon(a->challenges[i]);
    for (uint64_t i = 0; i < P->max_pub; i++) pub[i] = gl::canon(a->publics[i]);
    for (uint64_t i = 0; i < a->n_zhinv; i++) zh[i] = gl::canon(a->zhinv[i]);
    RunArgs A = {};
    A.pols = (const u64 *)a->pols; A.cpols = (const u64 *)a->const_pols; A.x = (const u64 *)a->x;
    A.chal = chal.data(); A.pub = pub.data(); A.zhinv = zh.data(); A.evals = ev.data();
    A.q = (u64 *)a->q; A.f = (u64 *)a->f; A.pols_w = (u64 *)a->pols;
    A.xd = (const u64 *)a->xdiv; A.xdw = (const u64 *)a->xdivw;
    A.n_const = a->n_const; A.x_stride = a->x_stride; A.n_zhinv = a->n_zhinv ? a->n_zhinv : 1;
    const u64 POISON = 0xDEADBEEFDEADBEEFull;
    std::vector<chpa::Acc> acc(N->pieces.size() * 3);
    for (uint64_t rk = 0; rk < nrows; rk++) {
        const uint64_t r = rows[rk];
        std::vector<u64> spill(N->nw + 3, POISON);
        for (const Chunk &C : N->chunks) {
            std::vector<u64> words(N->nw + 3, POISON);
            for (uint32_t w : C.loads) words[w] = spill[w];
            HostTmp tmp = {words.data()};
            for (size_t i = C.i0; i < C.i1; i++) {
                const DInstr &d = P->host[i];
                const Mark &m = N->marks[i];
                if (m.type == ST_FOLDED) continue;
                if (m.type == ST_NONE) { exec_instr(d, r, true, A, tmp); continue; }
                const uint32_t kk[2] = {(d.op >> 16) & 255, d.op >> 24};
                const Opd *os[2] = {&d.a, &d.b};
                u64 v[3];
                if (N->piece_first[i] >= 0) {
                    const Piece &pc = N->pieces[N->piece_first[i]];
                    chpa::Acc *p = &acc[3 * N->piece_first[i]];
                    load_operand(kk[m.yside], *os[m.yside], r, A, tmp, v);
                    if (pc.begin_coef < 0) { for (int j = 0; j < 3; j++) chpa::acc_set(p[j], v[j]); }
                    else {
                        const u64 *w = &cst[N->coef_off + 3 * pc.begin_coef];
                        for (int j = 0; j < 3; j++) chpa::acc_set(p[j], 0);
                        chpa::acc_mul33(p[0], p[1], p[2], v[0], v[1], v[2], w[0], w[1], w[2]);
                    }
                }
                int32_t pi = -1;
                for (size_t q = N->pieces.size(); q-- > 0;)
                    if (N->pieces[q].chain == m.chain && N->pieces[q].first <= i && i <= N->pieces[q].last) { pi = (int32_t)q; break; }
                MI_REQUIRE(pi >= 0, "internal: chain step outside every piece");
                chpa::Acc *p = &acc[3 * pi];
                if (m.type != ST_M && m.lin) { // as k_chp_linear does it: one base term per column, coefficients W, x W, x^2 W
                    const uint32_t lk = m.folded >= 0 ? (P->host[m.folded].op >> 16) & 255 : kk[1 - m.yside];
                    load_operand(lk, m.folded >= 0 ? P->host[m.folded].a : *os[1 - m.yside], r, A, tmp, v);
                    const int dim = kind_is3(lk) ? 3 : 1;
                    for (int j = 0; j < dim; j++) {
                        const LinTermH &lt = N->lin_terms[N->lin_first[i] + j];
                        const u64 *w = &cst[N->coef_off + 3 * lt.coef];
                        chpa::acc_mul13_s(p[0], p[1], p[2], v[j], w[0], w[1], w[2]);
                    }
                } else if (m.type != ST_M) {
                    uint32_t lk = kk[1 - m.yside];
                    if (m.folded >= 0) { lk = (P->host[m.folded].op >> 16) & 255; load_operand(lk, P->host[m.folded].a, r, A, tmp, v); }
                    else load_operand(lk, *os[1 - m.yside], r, A, tmp, v);
                    const int dim = kind_is3(lk) ? 3 : 1;
                    const int32_t ci = N->step_coef[i];
                    if (ci < 0) { for (int j = 0; j < dim; j++) chpa::acc_add(p[j], v[j]); }
                    else {
                        const u64 *w = &cst[N->coef_off + 3 * ci];
                        if (dim == 3) chpa::acc_mul33(p[0], p[1], p[2], v[0], v[1], v[2], w[0], w[1], w[2]);
                        else chpa::acc_mul13_s(p[0], p[1], p[2], v[0], w[0], w[1], w[2]);
                    }
                }
                if (N->piece_last[i] >= 0) {
                    const Piece &pc = N->pieces[N->piece_last[i]];
                    for (int j = 0; j < 3; j++) {
                        u64 o = chpa::acc_reduce(p[j]);
                        if (pc.k_slot >= 0) o = gl::add_wc(o, cst[N->k_off + 3 * pc.k_slot + j]);
                        words[d.dst + j] = gl::canon(o);
                    }
                }
            }
            for (uint32_t w : C.stores) spill[w] = gl::canon(words[w]);
        }
    }
    return MI_OK;
}

// [rows x ncols] row-major (pitch words per row, rows taken modulo nrows) -> tiles [tile][SC columns][64 rows], canonical
__global__ __launch_bounds__(256) void k_chp_transpose(const u64 *__restrict__ src, uint64_t pitch, uint32_t ncols, uint64_t row_mask,
                                                        u64 *__restrict__ tiled, uint32_t sc, uint32_t col0, uint64_t row_base, uint32_t slab_mask, uint32_t swz)
{
    if (!((slab_mask >> blockIdx.y) & 1)) return; // no generated kernel reads a column of this slab
    __shared__ u64 t[64][65];
    const uint32_t l = threadIdx.x & 63, q = threadIdx.x >> 6;
    const uint64_t tile = blockIdx.x;
    const uint32_t c0 = blockIdx.y * 64;
    if (c0 + l < ncols) {
#pragma unroll 4
        for (uint32_t r = q; r < 64; r += 4) {
            const uint64_t row = (row_base + tile * 64 + r) & row_mask;
            t[r][l] = gl::canon_sel(src[row * pitch + c0 + l]);
        }
    }
    __syncthreads();
    // swz: a section that STAYS tile-major (mi_tile_major_dev) keeps a row at gl::tile_pos of its index; the per-batch copy keeps it at the index
    u64 *dst = tiled + (tile * sc + col0 + c0) * 64 + (swz ? gl::tile_pos(l) : l);
#pragma unroll 4
    for (uint32_t cc = q; cc < 64; cc += 4)
        if (c0 + cc < ncols) dst[(uint64_t)cc * 64] = t[l][cc];
}

} // namespace chp

int launch_tile_major(mi_ctx *ctx, u64 *dst, uint64_t ncols_total, uint64_t col0, const u64 *src, uint64_t src_pitch, uint64_t nrows, uint64_t ncols)
{
    MI_REQUIRE(nrows % 64 == 0 && nrows / 64 < (1ull << 31), "a tile-major section has a multiple of 64 rows");
    MI_REQUIRE(ncols_total < (1ull << 32), "section too wide");
    for (uint64_t c0 = 0; c0 < ncols; c0 += 2048) { // (one launch carries a 32-bit mask of 64-column slabs)
        const uint64_t w = std::min<uint64_t>(2048, ncols - c0);
        hipLaunchKernelGGL(chp::k_chp_transpose, dim3((unsigned)(nrows / 64), (unsigned)((w + 63) / 64)), dim3(256), 0, ctx->stream, src + c0, src_pitch, (uint32_t)w,
                           ~0ull, dst, (uint32_t)ncols_total, (uint32_t)(col0 + c0), (uint64_t)0, 0xffffffffu, 1u);
        MI_HIP_CHECK(hipGetLastError());
    }
    return MI_OK;
}

// rows [row0, row0 + nrows) x columns [col0, col0 + ncols) of a tile-major section -> row-major (checks and tests: nothing in a proof reads it)
__global__ __launch_bounds__(256) void k_untile(u64 *__restrict__ dst, uint64_t dst_pitch, const u64 *__restrict__ src, uint64_t ncols_total, uint64_t col0,
                                                 uint64_t row0, uint64_t nrows, uint64_t ncols)
{
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < nrows * ncols; e += (uint64_t)gridDim.x * 256) {
        const uint64_t r = e / ncols, cc = e % ncols, row = row0 + r;
        dst[r * dst_pitch + cc] = src[((row >> 6) * ncols_total + col0 + cc) * 64 + gl::tile_pos((uint32_t)(row & 63))];
    }
}
// whole tiles: 64 rows x 64 columns through LDS, 512-byte runs in (a column's words of a tile) and out (a row's 64 columns) -- the
// constants' LDE reads its source through this (host/starks.hpp keeps them tile-major)
__global__ __launch_bounds__(256) void k_untile_tiles(u64 *__restrict__ dst, uint64_t dst_pitch, const u64 *__restrict__ src, uint64_t ncols_total, uint64_t col0,
                                                       uint64_t tile0, uint32_t ncols)
{
    __shared__ u64 t[64][65];
    const uint32_t l = threadIdx.x & 63, q = threadIdx.x >> 6, c0 = blockIdx.y * 64;
    const uint64_t tile = blockIdx.x;
    const u64 *s = src + ((tile0 + tile) * ncols_total + col0 + c0) * 64 + l;
    const uint32_t row = gl::tile_pos(l); // word l of a run is this row of the tile
#pragma unroll 4
    for (uint32_t cc = q; cc < 64; cc += 4)
        if (c0 + cc < ncols) t[row][cc] = s[(uint64_t)cc * 64];
    __syncthreads();
    if (c0 + l < ncols) {
#pragma unroll 4
        for (uint32_t r = q; r < 64; r += 4) dst[(tile * 64 + r) * dst_pitch + c0 + l] = t[r][l];
    }
}
int launch_untile(mi_ctx *ctx, u64 *dst, uint64_t dst_pitch, const u64 *src, uint64_t ncols_total, uint64_t col0, uint64_t row0, uint64_t nrows, uint64_t ncols)
{
    if (!nrows || !ncols) return MI_OK;
    if (row0 % 64 == 0 && nrows % 64 == 0 && nrows / 64 < (1ull << 31) && ncols < (1ull << 22)) {
        hipLaunchKernelGGL(k_untile_tiles, dim3((unsigned)(nrows / 64), (unsigned)((ncols + 63) / 64)), dim3(256), 0, ctx->stream, dst, dst_pitch, src, ncols_total, col0,
                           row0 / 64, (uint32_t)ncols);
        MI_HIP_CHECK(hipGetLastError());
        return MI_OK;
    }
    hipLaunchKernelGGL(k_untile, dim3(mi_grid_256(nrows * ncols)), dim3(256), 0, ctx->stream, dst, dst_pitch, src, ncols_total, col0, row0, nrows, ncols);
    MI_HIP_CHECK(hipGetLastError());
    return MI_OK;
}

namespace chp {

// sums of polynomial elements times constants, straight from the row-major sections (see "linear terms" above).  One wave per
// tile of 64 rows; per slab of LIN_COLS columns: 64 x LIN_COLS elements are fetched along the rows (128-byte runs) and turned
// through LDS, and every lane multiplies its row's elements by the terms' coefficients into the limb accumulators of the term's
// sum.  The slab's terms (column offset + coefficient, 32 bytes each, rewritten per run) travel through LDS as well -- read back
// as broadcasts -- because scalar loads inside the term loop (descriptor, then the coefficient it points to) left the kernel
// latency-bound at three times its HBM time.  Everything of slab s + 1 is in flight during the arithmetic of slab s.
struct LinTermW { uint32_t lds_off, pad; u64 w[3]; };
static_assert(sizeof(LinTermW) == 32, "term must be 32 bytes");

template <int S>
__global__ __launch_bounds__(64) void k_chp_linear(const LinSlabD *__restrict__ slabs, uint32_t n_slabs, const LinTermW *__restrict__ terms,
                                                     const LinSections sec, u64 *__restrict__ lin, uint64_t row_base, uint32_t n_sums)
{
    // one buffer each: the next slab waits in registers (stage / cstage) until the arithmetic on this one is done
    __shared__ u64 buf[64 * (LIN_COLS + 1)];
    __shared__ __attribute__((aligned(16))) u64 cb[(LIN_TMAX + 4) * 4]; // + slack: the term loop reads one pair ahead
    const uint32_t lane = threadIdx.x;
    const uint64_t tile = blockIdx.x;
    chpa::Acc acc[S][3];
#pragma unroll
    for (int s2 = 0; s2 < S; s2++)
#pragma unroll
        for (int j = 0; j < 3; j++) chpa::acc_set(acc[s2][j], 0);
    if (threadIdx.x < 16) cb[LIN_TMAX * 4 + threadIdx.x] = 0; // the slack stays zero (offset 0, coefficient 0)
    u64 stage[LIN_COLS]; // 64 x LIN_COLS elements / 64 lanes
    uint32_t stage_tiled = 0; // the staged slab came out of a tile-major section: stage[i] is column i of the lane's row
    uint4 cstage[LIN_TMAX * 32 / 1024];
    auto issue = [&](uint32_t sl) {
        const uint32_t si = slabs[sl].section, c0 = slabs[sl].col0, nc = slabs[sl].ncols;
        const u64 *src = sec.ptr[si];
        const uint64_t pitch = sec.pitch[si], mask = sec.row_mask[si];
        if (sec.tiled[si]) { // (wave-uniform) element i of the stage = column i of this lane's own row: 512-byte runs, nothing to turn
            const u64 *t = src + ((((row_base >> 6) + tile) & (mask >> 6)) * pitch + c0) * 64 + gl::tile_pos(lane);
#pragma unroll
            for (uint32_t i = 0; i < LIN_COLS; i++) stage[i] = t[(uint64_t)(i < nc ? i : 0) * 64];
        } else {
#pragma unroll
            for (uint32_t i = 0; i < LIN_COLS; i++) {
                const uint32_t e = i * 64 + lane, r = e / LIN_COLS, cc = e % LIN_COLS;
                const uint64_t row = (row_base + tile * 64 + r) & mask;
                stage[i] = src[row * pitch + c0 + (cc < nc ? cc : 0)]; // no term reads a column past the section's last: any valid address will do
            }
        }
        stage_tiled = sec.tiled[si];
        const uint32_t tb = slabs[sl].t0[0], n16 = (slabs[sl].t0[LIN_MAX_SUMS] - tb) * 2; // 16-byte pieces of this slab's terms
        const uint4 *tp = (const uint4 *)(terms + tb);
#pragma unroll
        for (uint32_t k = 0; k < LIN_TMAX * 32 / 1024; k++) {
            const uint32_t e = k * 64 + lane;
            const uint4 v = tp[e < n16 ? e : 0]; // unconditional load (a predicated one becomes a branch per load)
            cstage[k] = e < n16 ? v : make_uint4(0, 0, 0, 0);
        }
    };
    issue(0);
    for (uint32_t sl = 0; sl < n_slabs; sl++) {
        u64 *b = buf;
        uint4 *c4 = (uint4 *)cb;
        if (stage_tiled) {
#pragma unroll
            for (uint32_t i = 0; i < LIN_COLS; i++) b[lane * (LIN_COLS + 1) + i] = stage[i];
        } else {
#pragma unroll
            for (uint32_t i = 0; i < LIN_COLS; i++) {
                const uint32_t e = i * 64 + lane;
                b[(e / LIN_COLS) * (LIN_COLS + 1) + e % LIN_COLS] = stage[i];
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < LIN_TMAX * 32 / 1024; k++) c4[k * 64 + lane] = cstage[k];
        __syncthreads();
        if (sl + 1 < n_slabs) issue(sl + 1);
        const char *rowp = (const char *)(b + lane * (LIN_COLS + 1));
        const uint32_t tb = slabs[sl].t0[0];
#pragma unroll
        for (int s2 = 0; s2 < S; s2++) {
            const uint32_t t0 = slabs[sl].t0[s2] - tb, t1 = slabs[sl].t0[s2 + 1] - tb;
            if (t0 >= t1) continue;
            // software pipeline: the descriptors of pair k + 1 are read before, its elements in the middle of, the arithmetic of pair k
            const uint4 *q = (const uint4 *)cb + 2 * t0;
            uint4 a0 = q[0], a1 = q[1], b0 = q[2], b1 = q[3];
            u64 xa = *(const u64 *)(rowp + a0.x), xb = *(const u64 *)(rowp + b0.x);
            for (uint32_t t = t0; t < t1; t += 2) { // ranges are padded to pairs
                q += 4;
                const uint4 na0 = q[0], na1 = q[1], nb0 = q[2], nb1 = q[3];
                __builtin_amdgcn_sched_barrier(0); // keep the reads up here: the scheduler sinks them below the arithmetic otherwise
                chpa::acc_mac(acc[s2][0], xa, ((u64)a0.w << 32) | a0.z);
                chpa::acc_mac(acc[s2][1], xa, ((u64)a1.y << 32) | a1.x);
                chpa::acc_mac(acc[s2][2], xa, ((u64)a1.w << 32) | a1.z);
                __builtin_amdgcn_sched_barrier(0);
                const u64 nxa = *(const u64 *)(rowp + na0.x), nxb = *(const u64 *)(rowp + nb0.x);
                __builtin_amdgcn_sched_barrier(0);
                chpa::acc_mac(acc[s2][0], xb, ((u64)b0.w << 32) | b0.z);
                chpa::acc_mac(acc[s2][1], xb, ((u64)b1.y << 32) | b1.x);
                chpa::acc_mac(acc[s2][2], xb, ((u64)b1.w << 32) | b1.z);
                a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
                xa = nxa; xb = nxb;
            }
        }
        __syncthreads();
    }
    u64 *o = lin + tile * ((uint64_t)n_sums * 3 * 64) + lane;
#pragma unroll
    for (int s2 = 0; s2 < S; s2++)
#pragma unroll
        for (int j = 0; j < 3; j++) o[(s2 * 3 + j) * 64] = gl::canon(chpa::acc_reduce(acc[s2][j]));
}

static int grow(u64 **buf, uint64_t *have, uint64_t need, const char *what)
{
    if (*have >= need) return MI_OK;
    if (*buf) MI_HIP_CHECK(hipFree(*buf));
    *buf = nullptr;
    *have = 0;
    const hipError_t e = hipMalloc((void **)buf, need);
    if (e != hipSuccess) {
        mi_set_error("cannot allocate %llu bytes for %s: %s", (unsigned long long)need, what, hipGetErrorString(e));
        return MI_ERR_NOMEM;
    }
    *have = need;
    return MI_OK;
}

// where a run's batch buffers live: the context's pool, or the tail of a workspace the caller lent
struct RunBufs { u64 *tiled = nullptr, *spill = nullptr, *lin = nullptr; };
// the device buffers a run over nrows rows needs (kept by the context, grown on demand): allocate them ahead of the first run
int native_reserve(mi_ctx *c, const mi_chelpers_prog *P, uint64_t nrows, uint64_t *batch_out, RunBufs *bufs)
{
    const NativeProg *N = P->native;
    MI_REQUIRE(N, "native code was not built");
    uint64_t batch = c->chelpers_batch_rows;
    if (batch == 0) { // about 9 GiB of batch buffers: tile-major copy (zkEVM step42ns with row-major sections: 8 of them), spill, linear sums
                      // (MI_CHELPERS_BATCH_GIB: another size -- a one-GPU rehearsal of eight shards at full size has 2)
        static const double gib = [] { const char *e = getenv("MI_CHELPERS_BATCH_GIB"); const double v = e ? atof(e) : 9.0; return v >= 0.01 && v <= 64 ? v : 9.0; }();
        batch = (uint64_t)(gib * (double)(1ull << 30)) / (((uint64_t)N->sc + N->nw + 3 * (uint64_t)N->n_lin_sums) * 8);
        batch = std::max<uint64_t>(64, batch & ~(uint64_t)63);
    }
    batch = std::min(batch, (nrows + 63) & ~(uint64_t)63);
    const uint64_t max_tiles = batch / 64;
    MI_REQUIRE(max_tiles + 1 < (1ull << 31), "batch too large");
    MI_TRY(grow(&c->pool->chelpers_cst, &c->pool->chelpers_cst_bytes, (uint64_t)N->cst_words * 8 + 64, "the constraint program's constants"));
    const uint64_t b_tiled = ((max_tiles + 1) * N->sc * 512 + 255) & ~(uint64_t)255, b_spill = (max_tiles * N->nw * 512 + 255) & ~(uint64_t)255,
                   b_lin = N->n_lin_sums ? (max_tiles * N->n_lin_sums * 3 * 512 + 255) & ~(uint64_t)255 : 0;
    if (bufs) {
        // A caller that plans its HBM has LENT the context a region that is not live (mi_ctx_lend_workspace: host/starks.hpp lends a dead
        // section before every step): the batch's tile-major operand copy, the spill and the linear sums -- 9 GB at zkEVM size -- come out
        // of its tail for the duration of the run instead of staying allocated beside the plan (r04 weak #10: 284 of 309 GB at the peak).
        // Everything that uses the region is enqueued on this context's stream, before or after the run's kernels.
        const uint64_t need = b_tiled + b_spill + b_lin;
        if (c->workspace_lent && c->workspace_bytes >= need + (1ull << 30) && !getenv("MI_CHELPERS_NO_LENT_BUFFERS")) {
            // (2 MiB-aligned like an allocation of its own: the copy is read in 512-byte runs by every lane of every kernel)
            char *base = (char *)(((uintptr_t)c->workspace + c->workspace_bytes - need) & ~(uintptr_t)((2u << 20) - 1));
            bufs->tiled = (u64 *)base; bufs->spill = (u64 *)(base + b_tiled); bufs->lin = b_lin ? (u64 *)(base + b_tiled + b_spill) : nullptr;
            if (batch_out) *batch_out = batch;
            return MI_OK;
        }
    }
    MI_TRY(grow(&c->pool->chelpers_tiled, &c->pool->chelpers_tiled_bytes, b_tiled, "the tile-major operand copy"));
    MI_TRY(grow(&c->pool->chelpers_spill, &c->pool->chelpers_spill_bytes, b_spill, "the chunk-boundary spill"));
    if (N->n_lin_sums) MI_TRY(grow(&c->pool->chelpers_lin, &c->pool->chelpers_lin_bytes, b_lin, "the linear sums"));
    if (bufs) { bufs->tiled = c->pool->chelpers_tiled; bufs->spill = c->pool->chelpers_spill; bufs->lin = c->pool->chelpers_lin; }
    if (batch_out) *batch_out = batch;
    return MI_OK;
}

int native_run(mi_ctx *c, const mi_chelpers_prog *P, const mi_chelpers_params *a, uint64_t row0, uint64_t nrows)
{
    NativeProg *N = P->native;
    MI_REQUIRE(N, "native code was not built");
    MI_REQUIRE(a->n_const == P->n_const, "number of constant polynomials differs from what the program was compiled for");
    MI_REQUIRE(row0 + nrows <= P->nrows_ext && is_pow2(P->nrows_ext), "rows beyond the extended domain the program was compiled for (a power of two)");
    if (nrows == 0) return MI_OK;
    if (N->loaded_device != c->device) {
        MI_REQUIRE(N->loaded_device < 0, "program is loaded on another device");
        for (Chunk &C : N->chunks) {
            hipError_t e = hipModuleLoadData(&C.mod, C.code.data());
            if (e == hipSuccess) e = hipModuleGetFunction(&C.fn, C.mod, "chelpers_chunk");
            if (e != hipSuccess) { // leave nothing half-loaded behind: the next run starts from scratch
                for (Chunk &D : N->chunks) {
                    if (D.mod) (void)hipModuleUnload(D.mod);
                    D.mod = nullptr;
                    D.fn = nullptr;
                }
                mi_set_error("mi_chelpers_run_dev: cannot load a compiled kernel (%s); if it came from the code-object cache, clear the cache directory", hipGetErrorString(e));
                return MI_ERR_HIP;
            }
        }
        if (!N->lin_dev.empty()) { // the linear kernel's tables, coefficient indices turned into word offsets of the constants table
            MI_HIP_CHECK(hipMalloc((void **)&N->d_lin_terms, (N->lin_dev.size() + 64) * sizeof(LinTermW)));
            MI_HIP_CHECK(hipMalloc((void **)&N->d_lin_slabs, N->lin_slabs.size() * sizeof(LinSlabD)));
            MI_HIP_CHECK(hipMemcpy(N->d_lin_slabs, N->lin_slabs.data(), N->lin_slabs.size() * sizeof(LinSlabD), hipMemcpyHostToDevice));
        }
        N->loaded_device = c->device;
    }
    // ---- constants table (fill_constants)
    const uint64_t n_zh = P->step == MI_CHELPERS_STEP42NS ? a->n_zhinv : 1;
    std::vector<u64> cst;
    fill_constants(P, N, a, cst);
    const uint64_t cst_words = N->cst_words;
    MI_TRY(grow(&c->pool->chelpers_cst, &c->pool->chelpers_cst_bytes, cst_words * 8 + 64, "the constraint program's constants"));
    MI_HIP_CHECK(hipStreamSynchronize(c->stream)); // an earlier run may still be reading the table
    MI_HIP_CHECK(hipMemcpyAsync(c->pool->chelpers_cst, cst.data(), cst_words * 8, hipMemcpyHostToDevice, c->stream));
    if (!N->lin_dev.empty()) { // the linear kernel's terms with this proof's coefficients
        std::vector<LinTermW> tw(N->lin_dev.size());
        for (size_t i = 0; i < tw.size(); i++) {
            tw[i].lds_off = N->lin_dev[i].lds_off;
            tw[i].pad = 0;
            for (int j = 0; j < 3; j++) tw[i].w[j] = cst[N->coef_off + 3 * N->lin_dev[i].coef + j];
        }
        MI_HIP_CHECK(hipMemcpy(N->d_lin_terms, tw.data(), tw.size() * sizeof(LinTermW), hipMemcpyHostToDevice));
    }
    MI_HIP_CHECK(hipStreamSynchronize(c->stream)); // `cst` dies with this call
    // ---- batches of rows: tile-major copy (+ halo tile) and the spill
    uint64_t batch = 0;
    RunBufs rb;
    MI_TRY(native_reserve(c, P, nrows, &batch, &rb));
    u64 *out = P->stores_pols ? (u64 *)a->pols : (u64 *)(P->step == MI_CHELPERS_STEP52NS ? a->f : a->q);
    uint32_t zmask = (uint32_t)(n_zh - 1);
    const uint64_t row_end = row0 + nrows;
    for (const HostSection &S : P->sections)
        if (S.tiled) MI_REQUIRE(row0 % 64 == 0 && is_pow2(S.nrows) && S.nrows >= 64, "a program with a tile-major section runs over rows from a multiple of 64");
    for (uint64_t b0 = row0; b0 < row_end; b0 += batch) {
        const uint64_t rows = std::min(batch, row_end - b0), tiles = (rows + 63) / 64;
        LinSections ls = {};
        for (size_t si = 0; si < P->sections.size(); si++) {
            const HostSection &S = P->sections[si];
            const u64 *ptr;
            uint64_t pitch;
            if (S.role == 0) { ptr = (const u64 *)a->pols + S.offset; pitch = S.ncols; }
            else if (S.role == 1) { MI_REQUIRE(a->const_pols, "null constant polynomials"); ptr = (const u64 *)a->const_pols; pitch = a->n_const; }
            else if (S.role == 2) { MI_REQUIRE(a->x, "null x"); ptr = (const u64 *)a->x; pitch = a->x_stride; }
            else if (S.role == 3) { MI_REQUIRE(a->xdiv, "null xDivXSubXi"); ptr = (const u64 *)a->xdiv; pitch = 3; }
            else { MI_REQUIRE(a->xdivw, "null xDivXSubWXi"); ptr = (const u64 *)a->xdivw; pitch = 3; }
            MI_REQUIRE(is_pow2(S.nrows), "section row counts must be powers of two");
            ls.ptr[si] = ptr; ls.pitch[si] = pitch; ls.row_mask[si] = S.nrows - 1; ls.tiled[si] = S.tiled ? 1 : 0;
            if (S.tiled) continue; // read in place
            if (!N->sec_slab_mask[si]) continue;
            hipLaunchKernelGGL(k_chp_transpose, dim3((unsigned)(tiles + 1), (unsigned)((S.ncols + 63) / 64)), dim3(256), 0, c->stream, ptr, pitch,
                               (uint32_t)S.ncols, S.nrows - 1, rb.tiled, N->sc, S.col0, b0, N->sec_slab_mask[si], 0u);
            MI_HIP_CHECK(hipGetLastError());
        }
        if (N->n_lin_sums) {
            const dim3 g((unsigned)tiles), bl(64);
            const uint32_t ns = (uint32_t)N->lin_slabs.size();
            switch (N->n_lin_sums) {
            case 1: hipLaunchKernelGGL(k_chp_linear<1>, g, bl, 0, c->stream, N->d_lin_slabs, ns, N->d_lin_terms, ls, rb.lin, b0, 1u); break;
            case 2: hipLaunchKernelGGL(k_chp_linear<2>, g, bl, 0, c->stream, N->d_lin_slabs, ns, N->d_lin_terms, ls, rb.lin, b0, 2u); break;
            case 3: hipLaunchKernelGGL(k_chp_linear<3>, g, bl, 0, c->stream, N->d_lin_slabs, ns, N->d_lin_terms, ls, rb.lin, b0, 3u); break;
            default: hipLaunchKernelGGL(k_chp_linear<4>, g, bl, 0, c->stream, N->d_lin_slabs, ns, N->d_lin_terms, ls, rb.lin, b0, 4u); break;
            }
            MI_HIP_CHECK(hipGetLastError());
        }
        for (Chunk &C : N->chunks) {
            const u64 *tiled = rb.tiled, *cstp = c->pool->chelpers_cst;
            u64 *spill = rb.spill, *outp = out;
            uint64_t row_base = b0, rend = row_end;
            const u64 *linp = rb.lin;
            const u64 *polsp = (const u64 *)a->pols, *cpolsp = (const u64 *)a->const_pols;
            void *args[] = {&tiled, &spill, &cstp, &outp, &row_base, &rend, &zmask, &linp, &polsp, &cpolsp};
            MI_HIP_CHECK(hipModuleLaunchKernel(C.fn, (unsigned)tiles, 1, 1, 64, 1, 1, 0, c->stream, args, nullptr));
        }
    }
    return MI_OK;
}

} // namespace chp
