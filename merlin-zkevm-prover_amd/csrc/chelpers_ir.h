// chelpers_ir.h -- the intermediate forms of a constraint program shared by the translator / interpreter (chelpers.hip) and the
// native-code backend (chelpers_native.hip).  See chelpers.hip for what each form is.
#pragma once
#include "common.h"
#include <array>
#include <map>
#include <string>
#include <vector>

using gl::E3;

namespace chp {

enum Kind : uint32_t { K_NONE = 0, K_T1, K_T3, K_POL, K_POLS, K_NUM, K_CONST, K_CONSTS, K_CHAL, K_PUB, K_POL3, K_POL3S, K_X, K_ZHINV, K_Q,
                       K_EVAL, K_XD, K_XDW, // step52ns: params.evals[k], params.xDivXSubXi[i], params.xDivXSubWXi[i] (all dimension 3)
                       K_DPOL, K_DPOLS };   // base-domain steps: a DESTINATION pols[off + i * stride] / pols[off + ((i + shift) % n) * stride]
enum Cls : uint32_t { C_ADD = 0, C_SUB, C_MUL, C_COPY, C_STOREQ, C_STOREF, C_STOREP }; // STOREQ: q = zhInv * a; STOREF: f = a

static inline int kind_nargs(Kind k)
{
    switch (k) {
    case K_T1: case K_T3: case K_NUM: case K_CONST: case K_CHAL: case K_PUB: case K_EVAL: return 1;
    case K_POL: case K_POL3: case K_DPOL: return 2;   // offset, row stride
    case K_CONSTS: return 3;             // column, row shift, modulus
    case K_POLS: case K_POL3S: case K_DPOLS: return 4; // offset, row shift, modulus, row stride
    default: return 0;
    }
}
MI_HD bool kind_is3(uint32_t k) { return k == K_T3 || k == K_CHAL || k == K_POL3 || k == K_POL3S || k == K_Q || k == K_EVAL || k == K_XD || k == K_XDW; }

// ---- device instruction (64 bytes, wave-uniform, fetched through the scalar cache)
struct Opd { uint64_t off; uint32_t stride, shift; uint64_t mod; }; // meaning by kind, see load_operand
struct DInstr {
    uint32_t op;  // bits 0-7 class, 8-15 dst kind, 16-23 kind of a, 24-31 kind of b
    uint32_t dst; // word of the destination temp (ext temps take 3 consecutive words)
    Opd a, b;
    uint64_t pad;
};
static_assert(sizeof(DInstr) == 64, "instruction must be 64 bytes");

struct RunArgs { // host debug executor: operands read in place
    const u64 *pols, *cpols, *x, *zhinv, *chal, *pub, *evals, *xd, *xdw;
    u64 *q, *f, *pols_w; // pols_w: the same polynomial memory, written by the base-domain steps
    uint64_t n_const, x_stride, n_zhinv, row0, row_end;
};

// ---- device form.  Every operand of an instruction is, by the time it executes, up to three 64-bit words in LDS:
//   a hot temporary ([word][lane]), a prefetch slot ([slot][lane]: a staged polynomial value, an immediate, a cold
//   temporary or 1/Z_H, put there one batch ahead), a challenge / public input (one word for all lanes), or the all-zero
//   word that pads a base-field operand to an extension element.  So the executing loop has no operand-kind branches: it
//   reads addresses.  An LDS address field is a byte offset with bit 31 = "add 8 x lane".
enum GCls : uint32_t { G_ADD1 = 0, G_ADD3, G_SUB1, G_SUB3, G_MUL11, G_MUL13, G_MUL31, G_MUL33, G_COPY1, G_COPY3 };
enum GDst : uint32_t { D_NONE = 0, D_HOT, D_COLD, D_Q };
enum GPre : uint32_t { P_NONE = 0, P_STAGED, P_IMM, P_ZHINV };
struct GInstr {
    uint32_t op;      // bits 0-3 GCls, 4-5 GDst, 6 "also write pad[0] of the spill", 8-10 GPre of a, 12-14 GPre of b
    uint32_t dst;     // D_HOT: LDS byte offset (lane-indexed, 1 or 3 consecutive words); D_COLD: spill word
    uint32_t a[3], b[3]; // LDS address fields of the operands' words
    uint64_t a_imm, b_imm; // P_STAGED: element index into the workgroup's scratch (staged column or spilled temporary); P_IMM: the value
    uint32_t pad[4];
};
static_assert(sizeof(GInstr) == 64, "device instruction must be 64 bytes");
static constexpr uint32_t LANE_FLAG = 0x80000000u;

static constexpr int MAX_SECTIONS = 8;
static constexpr uint32_t HALO = 8;             // rows staged beyond the group's 64 (largest row shift a program may use)
static constexpr uint32_t RS = 64 + HALO;       // rows per staged column
static constexpr uint32_t TILE_ROWS = 24;       // rows transposed per pass of the staging tile (RS = 3 passes)
static constexpr int BATCH = 8;                 // instructions whose prefetch slots are filled together
struct GSection { const u64 *ptr; uint64_t pitch, nrows; uint32_t ncols, col0; };
struct GArgs {
    GSection sec[MAX_SECTIONS];
    uint32_t n_sections, n_instr;
    uint32_t tile_off, pre_off, cst_off, cst_words, zh_off, lds_bytes; // LDS layout (bytes / words), see build_staged
    const u64 *consts;          // device copy of [challenges | public inputs | ZhInv] = the cst region
    u64 *q, *scratch;           // scratch: per resident workgroup, [staged column][RS rows] then [cold word][64 lanes]
    uint64_t n_zhinv, row0, row_end, n_groups, staged_cols, wg_stride;
};

// ---- host-side intermediate form
struct HOpd { Kind k = K_NONE; uint64_t v[4] = {0, 0, 0, 0}; };
struct MicroOp { Cls cls; Kind dst; uint64_t dst_slot; HOpd a, b; };

struct NativeProg;

// one operand at row r.  Memory is read through M so that the same code runs in the kernel (LDS temporaries) and in the
// host debug executor (a plain array).
template <typename Tmp>
MI_HD void load_operand(uint32_t kind, const Opd &o, uint64_t r, const RunArgs &P, const Tmp &tmp, u64 (&v)[3])
{
    v[1] = v[2] = 0;
    switch (kind) {
    case K_T1: v[0] = tmp.get(o.off); break;
    case K_T3: v[0] = tmp.get(o.off); v[1] = tmp.get(o.off + 1); v[2] = tmp.get(o.off + 2); break;
    case K_NUM: v[0] = o.off; break; // canonicalised by the translator (Goldilocks::fromU64)
    case K_POL: v[0] = gl::canon(P.pols[o.off + r * o.stride]); break;
    case K_POL3: {
        const u64 *p = P.pols + o.off + r * o.stride;
        v[0] = gl::canon(p[0]); v[1] = gl::canon(p[1]); v[2] = gl::canon(p[2]);
        break;
    }
    case K_POLS: case K_POL3S: {
        // ((i + shift) % modulus): the modulus is the extended domain size in every generated program, a power of two
        const uint64_t rs = (o.mod & (o.mod - 1)) == 0 ? ((r + o.shift) & (o.mod - 1)) : ((r + o.shift) % o.mod);
        const u64 *p = P.pols + o.off + rs * o.stride;
        v[0] = gl::canon(p[0]);
        if (kind == K_POL3S) { v[1] = gl::canon(p[1]); v[2] = gl::canon(p[2]); }
        break;
    }
    case K_CONST: v[0] = gl::canon(P.cpols[o.off + r * P.n_const]); break;
    case K_CONSTS: {
        const uint64_t rs = (o.mod & (o.mod - 1)) == 0 ? ((r + o.shift) & (o.mod - 1)) : ((r + o.shift) % o.mod);
        v[0] = gl::canon(P.cpols[o.off + rs * P.n_const]);
        break;
    }
    case K_CHAL: v[0] = P.chal[o.off * 3]; v[1] = P.chal[o.off * 3 + 1]; v[2] = P.chal[o.off * 3 + 2]; break;
    case K_PUB: v[0] = P.pub[o.off]; break;
    case K_X: v[0] = gl::canon(P.x[r * P.x_stride]); break;
    case K_EVAL: v[0] = P.evals[o.off * 3]; v[1] = P.evals[o.off * 3 + 1]; v[2] = P.evals[o.off * 3 + 2]; break;
    case K_XD: case K_XDW: {
        const u64 *p = (kind == K_XD ? P.xd : P.xdw) + r * 3;
        v[0] = gl::canon(p[0]); v[1] = gl::canon(p[1]); v[2] = gl::canon(p[2]);
        break;
    }
    case K_ZHINV: v[0] = P.zhinv[r % P.n_zhinv]; break;
    default: v[0] = 0; break;
    }
}

template <typename Tmp>
MI_HD void exec_instr(const DInstr &I, uint64_t r, bool active, const RunArgs &P, Tmp &tmp)
{
    const uint32_t cls = I.op & 255, dk = (I.op >> 8) & 255, ak = (I.op >> 16) & 255, bk = I.op >> 24;
    u64 a[3], b[3], o[3];
    load_operand(ak, I.a, r, P, tmp, a);
    load_operand(bk, I.b, r, P, tmp, b);
    const bool a3 = kind_is3(ak), b3 = kind_is3(bk);
    switch (cls) {
    case C_ADD: // a dimension-1 operand is (v, 0, 0): Goldilocks3::add13 / add31 / add1c3c
        o[0] = gl::add(a[0], b[0]); o[1] = gl::add(a[1], b[1]); o[2] = gl::add(a[2], b[2]);
        break;
    case C_SUB: // Goldilocks3::sub31c / sub13c: component-wise on (v, 0, 0)
        o[0] = gl::sub(a[0], b[0]); o[1] = gl::sub(a[1], b[1]); o[2] = gl::sub(a[2], b[2]);
        break;
    case C_MUL: case C_STOREQ:
        if (a3 && b3) {
            const E3 p = gl::e3_mul(E3{{a[0], a[1], a[2]}}, E3{{b[0], b[1], b[2]}});
            o[0] = p.v[0]; o[1] = p.v[1]; o[2] = p.v[2];
        } else if (a3) { // Goldilocks3::mul31 / mul13: scalar times extension element
            o[0] = gl::mul(a[0], b[0]); o[1] = gl::mul(a[1], b[0]); o[2] = gl::mul(a[2], b[0]);
        } else if (b3) {
            o[0] = gl::mul(a[0], b[0]); o[1] = gl::mul(a[0], b[1]); o[2] = gl::mul(a[0], b[2]);
        } else {
            o[0] = gl::mul(a[0], b[0]); o[1] = o[2] = 0;
        }
        break;
    default: // C_COPY, C_STOREF
        o[0] = a[0]; o[1] = a[1]; o[2] = a[2];
        break;
    }
    if (dk == K_T1) {
        tmp.set(I.dst, o[0]);
    } else if (dk == K_T3) {
        tmp.set(I.dst, o[0]); tmp.set(I.dst + 1, o[1]); tmp.set(I.dst + 2, o[2]);
    } else if (active && dk == K_Q) { // (Goldilocks3::Element &)params.q_2ns[i * 3] (step42ns) / params.f_2ns[i * 3] (step52ns)
        u64 *out = cls == C_STOREF ? P.f : P.q;
        out[r * 3] = o[0]; out[r * 3 + 1] = o[1]; out[r * 3 + 2] = o[2];
    } else if (active && (dk == K_DPOL || dk == K_DPOLS)) { // params.pols[off + row * stride]: a is the value, b describes the destination
        const uint64_t row = dk == K_DPOLS ? ((I.b.mod & (I.b.mod - 1)) == 0 ? ((r + I.b.shift) & (I.b.mod - 1)) : ((r + I.b.shift) % I.b.mod)) : r;
        u64 *out = P.pols_w + I.b.off + row * I.b.stride;
        out[0] = a[0];
        if (a3) { out[1] = a[1]; out[2] = a[2]; }
    }
}

struct HostTmp {
    u64 *base;
    u64 get(uint64_t w) const { return base[w]; }
    void set(uint64_t w, u64 v) { base[w] = v; }
};

} // namespace chp

// role 0: section of pols, 1: constant polynomials, 2: x, 3 / 4: xDivXSubXi / xDivXSubWXi.  tiled: the section already lies in HBM as
// [tile of 64 rows][column][row in tile] (mi_chelpers_set_tiled_section): the generated kernels read it in place, nothing is copied
struct HostSection { uint64_t offset, ncols, nrows; uint32_t col0; int role; bool tiled = false; };
struct mi_chelpers_prog {
    int run_logical = -1;          // MI_MULTI_CHECK: the logical shard the program first ran for (common.h)
    std::vector<chp::DInstr> host; // the translated program, operands in place (host debug executor)
    std::vector<chp::GInstr> gpu;  // the same program over staged columns (kernel)
    chp::GInstr *dev = nullptr;
    std::vector<HostSection> sections; // what the kernel stages per group of rows, in staged-column order
    uint64_t staged_cols = 0;
    uint64_t n_const = 0, nrows_ext = 0;
    // LDS layout of the kernel (bytes) and the cold spill, see build_staged
    uint32_t hot_t1 = 0, hot_t3 = 0, cold_words = 0, tile_off = 0, pre_off = 0, cst_off = 0, lds_fixed = 0;
    uint64_t cold_reads = 0, temp_reads = 0;
    uint64_t n_words = 0;          // LDS words per row
    uint64_t stats[8] = {0};       // ops in, micro-ops, after copy forwarding, scheduled, live words before, after, t1 slots, t3 slots
    uint64_t max_chal = 0, max_pub = 0, max_eval = 0;
    int step = 0;
    bool stores_pols = false;      // a base-domain step: results go into polynomial memory (C_STOREP)
    struct Patch { uint32_t instr, eval, word; };
    std::vector<Patch> eval_patches; // device instructions whose a_imm is evals[eval][word] of the running proof
    chp::NativeProg *native = nullptr; // the program compiled to gfx950 code (chelpers_native.hip), when mi_chelpers_build_native was called
};

namespace chp {
void renumber_staged_columns(mi_chelpers_prog *p); // chelpers.hip: after a section was declared tile-major
// chelpers_native.hip
int native_build(mi_chelpers_prog *P, const char *cache_dir, uint64_t chunk_cost, uint32_t shard, uint32_t nshards);
int native_host_run(const mi_chelpers_prog *P, const mi_chelpers_params *a, const uint64_t *rows, uint64_t nrows, uint64_t chunk_cost);
int native_run(mi_ctx *c, const mi_chelpers_prog *P, const mi_chelpers_params *a, uint64_t row0, uint64_t nrows);
void native_free(mi_ctx *c, mi_chelpers_prog *P);
struct RunBufs;
int native_reserve(mi_ctx *c, const mi_chelpers_prog *P, uint64_t nrows, uint64_t *batch_out, RunBufs *bufs = nullptr);
int native_lower_stats(const mi_chelpers_prog *P, uint64_t chunk_cost, uint64_t out[12]);
void native_stats(const mi_chelpers_prog *P, uint64_t out[8]);
} // namespace chp
