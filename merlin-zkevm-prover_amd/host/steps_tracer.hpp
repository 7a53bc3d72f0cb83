// steps_tracer.hpp -- a Steps class's generated PER-ROW code, run on the device.
//
// The recursive STARKs of the reference (c12a, recursive1, recursive2: prover.cpp:577,611 prove them with the same class Starks)
// ship their constraint evaluators not as opcode tables but as generated straight-line C++:
//     void Recursive1Steps::step3_first(StepsParams &params, uint64_t i) {
//          Goldilocks::sub(tmp_123, params.pols[0 + i*18], params.pols[4 + i*18]);
//          Goldilocks3::mul(tmp_1430, tmp_3364, (Goldilocks3::Element &)*params.challenges[3]); ...
// (recursive1.chelpers.step3.cpp:7-...; ~7 000 field operations a row) and Starks::genProof calls them row by row
// (starks.cpp:84-88,166-170,204-208,252-256,382-386: nrowsStepBatch == 1).  Over a polynomial area that lives in HBM that loop
// would bring every section down and back.  Instead the function is run ONCE, on the host, at one row, with a recorder installed
// in the field classes (MiFieldRecorder, goldilocks_base_field.hpp): every operation reports the addresses of its operands, the
// recorder recognises each address -- a column of a section of params.pols at the row or a shifted row, a constant polynomial, a
// challenge, an evaluation, a public input, x, x/(x - xi), a named temporary of the function (an address an earlier operation
// wrote), or else a literal -- and writes the row's computation down as field operations (mi_chelpers_microop).  That program goes
// through the same translator and native-code backend as the zkEVM's tables (mi_chelpers_compile_micro) and runs over the device
// image.  Nothing of the generated code is re-typed or parsed: it is compiled as it is, against these headers.
//
// The function is recorded at row 0 AND at the last row (where every shifted read wraps); both recordings must give the same
// program, which also tells a literal from the one value that is passed by value and varies with the row, zi.zhInv(i).
#ifndef MI_STEPS_TRACER_HPP
#define MI_STEPS_TRACER_HPP
#include <cstdint>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>
#include "goldilocks_base_field.hpp"
#include "../../include/mi_stark.h"

namespace mi {

struct TraceLayout
{
    int step = 0;                      // MI_CHELPERS_STEP*
    uint64_t rows = 0;                 // rows of the step's domain (N or NExtended)
    const uint64_t *pols = nullptr;    // params.pols and its sections (StarkInfo::mapOffsets / mapSectionsN), rows each
    struct Sec { uint64_t offset, cols, rows; };
    std::vector<Sec> secs;             // every section of the area
    uint64_t qOffset = ~0ULL, fOffset = ~0ULL; // the sections q_2ns / f_2ns (3 columns each)
    const uint64_t *constPols = nullptr; uint64_t nConst = 0; // the constant polynomials of this domain, rows x nConst
    const uint64_t *chal = nullptr; uint64_t nChal = 0;
    const uint64_t *evals = nullptr; uint64_t nEvals = 0;
    const uint64_t *pub = nullptr; uint64_t nPub = 0;
    const uint64_t *x = nullptr; uint64_t xStride = 1;       // x_n / x_2ns
    const uint64_t *xd = nullptr, *xdw = nullptr;            // xDivXSubXi / xDivXSubWXi, rows x 3
};

class StepRecorder : public MiFieldRecorder
{
public:
    std::vector<mi_chelpers_microop> ops;
    std::string error;

    StepRecorder(const TraceLayout &layout, uint64_t row, uint64_t zhinvAtRow) : L(layout), row(row), zhinv(zhinvAtRow) {}

    void op(int cls, void *r, int rdim, const void *a, int adim, const void *b, int bdim) override
    {
        if (!error.empty()) return;
        settle();
        mi_chelpers_microop u;
        std::memset(&u, 0, sizeof u);
        u.cls = (uint32_t)cls;
        if (!source(a, adim, u.a)) return;
        if (cls != MI_CHP_COPY && !source(b, bdim, u.b)) return;
        if (rdim == 1 && (adim == 3 || bdim == 3)) { fail("an extension operand with a base-field destination"); return; }
        const uint64_t *rp = (const uint64_t *)r;
        const uint32_t tk = rdim == 3 ? MI_CHP_T3 : MI_CHP_T1;
        uint64_t idx = 0;
        if (!inPols(rp, idx)) {
            if (known(rp)) { fail("a step writes into its constants, challenges or tables"); return; }
            u.dst_kind = tk;
            u.dst_slot = nextSlot++;
            ops.push_back(u);
            define(rp, rdim, u.dst_slot);
            return;
        }
        // a destination in params.pols
        const TraceLayout::Sec *S = section(idx);
        if (!S) { fail("a destination in params.pols outside every section"); return; }
        const uint64_t rel = idx - S->offset, prow = rel / S->cols, col = rel % S->cols;
        if (col + (uint64_t)rdim > S->cols || S->rows != L.rows) { fail("a destination that straddles a row, or a section of the other domain"); return; }
        const uint64_t shift = (prow + L.rows - row) % L.rows;
        if (S->offset == L.qOffset || S->offset == L.fOffset) {
            if (shift || col || rdim != 3) { fail("q_2ns / f_2ns are written one extension element at the row"); return; }
            if (S->offset == L.qOffset) { // q = zhInv * t  (the reference's only form: recursive1.chelpers.step42ns.cpp:6972)
                const bool za = u.a.kind == MI_CHP_ZHINV && u.b.kind == MI_CHP_T3, zb = u.b.kind == MI_CHP_ZHINV && u.a.kind == MI_CHP_T3;
                if (cls != MI_CHP_MUL || !(za || zb)) { fail("q_2ns is written by something other than zhInv(i) times an extension temporary"); return; }
                mi_chelpers_microop st;
                std::memset(&st, 0, sizeof st);
                st.cls = MI_CHP_STOREQ; st.dst_kind = MI_CHP_Q;
                st.a = za ? u.b : u.a;
                st.b.kind = MI_CHP_ZHINV;
                ops.push_back(st);
                return;
            }
            mi_chelpers_microop st;
            std::memset(&st, 0, sizeof st);
            st.cls = MI_CHP_STOREF; st.dst_kind = MI_CHP_Q;
            if (cls == MI_CHP_COPY && u.a.kind == MI_CHP_T3) st.a = u.a;
            else {
                u.dst_kind = MI_CHP_T3; u.dst_slot = nextSlot++;
                ops.push_back(u);
                st.a.kind = MI_CHP_T3; st.a.v[0] = u.dst_slot;
            }
            ops.push_back(st);
            return;
        }
        // an ordinary polynomial: the value goes to a fresh temporary, is stored, and later reads of that element are the temporary
        u.dst_kind = tk;
        u.dst_slot = nextSlot++;
        ops.push_back(u);
        mi_chelpers_microop st;
        std::memset(&st, 0, sizeof st);
        st.cls = MI_CHP_STOREP;
        st.a.kind = tk; st.a.v[0] = u.dst_slot;
        if (shift == 0) { st.b.kind = MI_CHP_DPOL; st.b.v[0] = S->offset + col; st.b.v[1] = S->cols; }
        else { st.b.kind = MI_CHP_DPOLS; st.b.v[0] = S->offset + col; st.b.v[1] = shift; st.b.v[2] = L.rows; st.b.v[3] = S->cols; }
        st.dst_kind = st.b.kind;
        ops.push_back(st);
        define(rp, rdim, u.dst_slot);
        for (int j = 0; j < rdim; j++) written.push_back({S->offset + col + (uint64_t)j, S->cols});
    }
    // after the traced call has returned
    void finish()
    {
        if (!error.empty()) return;
        settle();
        // a column the step writes may not be read from memory at all -- not even BEFORE the write in program order: in the reference's
        // row-by-row loop such a read sees what an earlier or a later iteration left there, on the device it would race with another row's store
        for (const Col &r : memoryReads)
            for (const Col &w : written)
                if (w.stride == r.stride && w.elem == r.elem) { fail("a polynomial the step writes is also read from memory (another row's value)"); return; }
    }

private:
    struct Val { uint64_t slot; int dim; uint64_t v[3]; bool settled; };
    const TraceLayout &L;
    const uint64_t row, zhinv;
    uint64_t nextSlot = 0;
    std::unordered_map<const uint64_t *, Val> vals; // address of a value the function has written -> which temporary it is
    const uint64_t *pending = nullptr;
    struct Col { uint64_t elem, stride; };
    std::vector<Col> written, memoryReads; // columns (first-row element index, row stride) stored / read from polynomial memory

    void fail(const char *what)
    {
        if (error.empty()) error = std::string("steps tracer: ") + what + " (operation " + std::to_string(ops.size()) + ")";
    }
    // the previous operation has computed by now: remember what its destination holds
    void settle()
    {
        if (!pending) return;
        Val &v = vals[pending];
        for (int j = 0; j < v.dim; j++) v.v[j] = pending[j];
        v.settled = true;
        pending = nullptr;
    }
    void define(const uint64_t *p, int dim, uint64_t slot)
    {
        // a value written over part of an older one makes the older one unreadable
        for (int j = -2; j <= 2; j++) {
            if (j == 0) continue;
            auto it = vals.find(p + j);
            if (it != vals.end() && ((j < 0 && it->second.dim > -j) || (j > 0 && dim > j))) vals.erase(it);
        }
        Val v = {slot, dim, {0, 0, 0}, false};
        vals[p] = v;
        pending = p;
    }
    bool within(const uint64_t *p, const uint64_t *base, uint64_t words, uint64_t &idx) const
    {
        if (!base || p < base || p >= base + words) return false;
        idx = (uint64_t)(p - base);
        return true;
    }
    uint64_t polsWords() const
    {
        uint64_t e = 0;
        for (const TraceLayout::Sec &S : L.secs) e = std::max(e, S.offset + S.cols * S.rows);
        return e;
    }
    bool inPols(const uint64_t *p, uint64_t &idx) const { return within(p, L.pols, polsWords(), idx); }
    bool known(const uint64_t *p) const
    {
        uint64_t i;
        return within(p, L.constPols, L.nConst * L.rows, i) || within(p, L.chal, L.nChal * 3, i) || within(p, L.evals, L.nEvals * 3, i) ||
               within(p, L.pub, L.nPub, i) || within(p, L.x, L.rows * L.xStride, i) || within(p, L.xd, L.rows * 3, i) || within(p, L.xdw, L.rows * 3, i);
    }
    const TraceLayout::Sec *section(uint64_t idx) const
    {
        for (const TraceLayout::Sec &S : L.secs)
            if (S.cols && idx >= S.offset && idx < S.offset + S.cols * S.rows) return &S;
        return nullptr;
    }
    bool source(const void *ptr, int dim, mi_chelpers_operand &o)
    {
        const uint64_t *p = (const uint64_t *)ptr;
        uint64_t idx = 0;
        auto it = vals.find(p);
        if (it != vals.end()) {
            const Val &v = it->second;
            if (v.dim != dim) { fail("a temporary read with another dimension than it was written with"); return false; }
            if (v.settled)
                for (int j = 0; j < dim; j++)
                    if (p[j] != v.v[j]) { fail("memory the function wrote holds another value now: its temporaries cannot be told apart by address"); return false; }
            o.kind = dim == 3 ? MI_CHP_T3 : MI_CHP_T1;
            o.v[0] = v.slot;
            return true;
        }
        for (int j = 1; j <= 2; j++) { // the middle of an extension value
            auto mid = vals.find(p - j);
            if (mid != vals.end() && mid->second.dim > j) { fail("a word of an extension temporary read on its own"); return false; }
        }
        if (inPols(p, idx)) {
            const TraceLayout::Sec *S = section(idx);
            if (!S) { fail("an operand in params.pols outside every section"); return false; }
            const uint64_t rel = idx - S->offset, prow = rel / S->cols, col = rel % S->cols;
            if (col + (uint64_t)dim > S->cols) { fail("an operand that straddles a row"); return false; }
            if (S->rows != L.rows) { fail("an operand in a section of the other domain"); return false; }
            for (const Col &w : written)
                if (w.stride == S->cols && w.elem >= S->offset + col && w.elem < S->offset + col + (uint64_t)dim) {
                    fail("a polynomial the step writes is read back at another row or with another dimension");
                    return false;
                }
            const uint64_t shift = (prow + L.rows - row) % L.rows;
            for (int j = 0; j < dim; j++) memoryReads.push_back({S->offset + col + (uint64_t)j, S->cols});
            if (shift == 0) { o.kind = dim == 3 ? MI_CHP_POL3 : MI_CHP_POL; o.v[0] = S->offset + col; o.v[1] = S->cols; }
            else { o.kind = dim == 3 ? MI_CHP_POL3S : MI_CHP_POLS; o.v[0] = S->offset + col; o.v[1] = shift; o.v[2] = L.rows; o.v[3] = S->cols; }
            return true;
        }
        if (within(p, L.constPols, L.nConst * L.rows, idx)) {
            if (dim != 1) { fail("a constant polynomial read as an extension element"); return false; }
            const uint64_t prow = idx / L.nConst, col = idx % L.nConst, shift = (prow + L.rows - row) % L.rows;
            if (shift == 0) { o.kind = MI_CHP_CONST; o.v[0] = col; }
            else { o.kind = MI_CHP_CONSTS; o.v[0] = col; o.v[1] = shift; o.v[2] = L.rows; }
            return true;
        }
        if (within(p, L.chal, L.nChal * 3, idx)) {
            if (dim != 3 || idx % 3) { fail("a challenge read as a base-field element"); return false; }
            o.kind = MI_CHP_CHAL; o.v[0] = idx / 3;
            return true;
        }
        if (within(p, L.evals, L.nEvals * 3, idx)) {
            if (dim != 3 || idx % 3) { fail("an evaluation read as a base-field element"); return false; }
            o.kind = MI_CHP_EVAL; o.v[0] = idx / 3;
            return true;
        }
        if (within(p, L.pub, L.nPub, idx)) {
            if (dim != 1) { fail("a public input read as an extension element"); return false; }
            o.kind = MI_CHP_PUB; o.v[0] = idx;
            return true;
        }
        if (within(p, L.x, L.rows * L.xStride, idx)) {
            if (dim != 1 || idx != row * L.xStride) { fail("x read at another row"); return false; }
            o.kind = MI_CHP_X;
            return true;
        }
        if (within(p, L.xd, L.rows * 3, idx) || within(p, L.xdw, L.rows * 3, idx)) {
            if (dim != 3 || idx != row * 3) { fail("xDivXSubXi read at another row"); return false; }
            o.kind = within(p, L.xd, L.rows * 3, idx) ? MI_CHP_XD : MI_CHP_XDW;
            return true;
        }
        // anything else is a value the caller materialised for the call: Goldilocks::fromU64(c), or zi.zhInv(i) (returned by value,
        // zhInv.hpp:22-25).  The two recordings at different rows tell them apart for good: a literal does not change with the row.
        if (dim != 1) { fail("an extension operand that is neither a temporary nor in any table"); return false; }
        if (L.step == MI_CHELPERS_STEP42NS && p[0] == zhinv) { o.kind = MI_CHP_ZHINV; return true; }
        o.kind = MI_CHP_NUM;
        o.v[0] = p[0];
        return true;
    }
};

// Run `call(row)` (the Steps member for this step, e.g. [&](uint64_t i) { steps->step3_first(params, i); }) under a recorder at
// row 0 and at the last row; on success `out` is the row program.  zhinvAt(row) = zi.zhInv(row).
template <typename Call, typename ZhInvAt>
inline bool traceStep(const TraceLayout &L, Call call, ZhInvAt zhinvAt, std::vector<mi_chelpers_microop> &out, std::string &error)
{
    std::vector<mi_chelpers_microop> recs[2];
    const uint64_t rows[2] = {0, L.rows - 1};
    for (int t = 0; t < 2; t++) {
        StepRecorder rec(L, rows[t], zhinvAt(rows[t]));
        // the rows of every section a per-row function at this row can reach (itself and the shifted rows around it), as they are before the call:
        // a function that records nothing must not have changed them either (it would be arithmetic that bypasses the hooked forms -- a plain
        // Element assignment, a memcpy, a translation unit built against another field header -- and the step would be dropped silently)
        std::vector<uint64_t> snapshot;
        auto window = [&](bool compare) -> bool {
            size_t k = 0;
            for (const TraceLayout::Sec &S : L.secs) {
                if (!S.cols || !S.rows || !L.pols) continue;
                for (int64_t d = -8; d <= 8; d++) {
                    const uint64_t r = (rows[t] + S.rows + (uint64_t)(d + 64) - 64) % S.rows;
                    const uint64_t *p = L.pols + S.offset + r * S.cols;
                    if (!compare) snapshot.insert(snapshot.end(), p, p + S.cols);
                    else if (std::memcmp(snapshot.data() + k, p, S.cols * 8)) return false;
                    k += S.cols;
                }
            }
            return true;
        };
        window(false);
        MiFieldRecorder *before = mi_field_recorder;
        mi_field_recorder = &rec;
        mi_field_recording.fetch_add(1, std::memory_order_relaxed); // (a count: another thread may be tracing too)
        call(rows[t]);
        mi_field_recorder = before;
        mi_field_recording.fetch_sub(1, std::memory_order_relaxed);
        rec.finish();
        if (rec.error.empty() && rec.ops.empty() && !window(true))
            rec.error = "steps tracer: the function recorded nothing but changed params.pols: its arithmetic does not go through Goldilocks:: / Goldilocks3:: "
                        "(set MI_STEPS_ON_HOST=1 to run such a Steps class on the host)";
        if (rec.error.empty() && rec.untracked)
            rec.error = "steps tracer: the function computes with Goldilocks operators or value-returning forms, which leave no addresses to follow";
        if (rec.error.empty() && rec.ops.empty() && (L.step == MI_CHELPERS_STEP42NS || L.step == MI_CHELPERS_STEP52NS))
            rec.error = "steps tracer: the function recorded nothing, yet every STARK writes q_2ns / f_2ns here: its arithmetic does not go through "
                        "Goldilocks:: / Goldilocks3:: (set MI_STEPS_ON_HOST=1 to run such a Steps class on the host)";
        if (!rec.error.empty()) { error = rec.error + " at row " + std::to_string(rows[t]); return false; }
        recs[t].swap(rec.ops);
    }
    if (recs[0].size() != recs[1].size() || (recs[0].size() && std::memcmp(recs[0].data(), recs[1].data(), recs[0].size() * sizeof(mi_chelpers_microop)))) {
        size_t k = 0;
        while (k < recs[0].size() && k < recs[1].size() && !std::memcmp(&recs[0][k], &recs[1][k], sizeof(mi_chelpers_microop))) k++;
        error = "steps tracer: the function does not compute the same program at row 0 and at the last row (first difference at operation " +
                std::to_string(k) + "): it branches on the row or reads a table this tracer does not know";
        return false;
    }
    out.swap(recs[0]);
    return true;
}

} // namespace mi
#endif
