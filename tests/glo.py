"""ctypes binding of the CPU oracle (oracle/libgl_oracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product."""
import ctypes, os, subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 0xFFFFFFFF00000001
u64 = ctypes.c_uint64
_p64 = ctypes.POINTER(u64)


def build(flavour=""):
    """flavour "baseline": oracle/libgl_baseline_avx2.so (cpu_baseline_avx2.c: the hand-vectorised CPU-baseline leg, plus the oracle's
    symbols it is checked against)."""
    so = os.path.join(ROOT, "oracle", "libgl_baseline_avx2.so" if flavour == "baseline" else "libgl_oracle%s.so" % ("_" + flavour if flavour else ""))
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("gl_oracle.c", "chelpers_oracle.c", "gl_oracle.h", "poseidon_constants.h", "Makefile", "cpu_baseline_avx2.c")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return so


_L = {}


def lib(flavour=""):
    """flavour "": the checker build; "avx2": the same source built -O3 -mavx2 (bench.py's cpu_baseline leg only)."""
    if flavour not in _L:
        L = ctypes.CDLL(build(flavour))
        for f in ("glo_canon", "glo_add", "glo_sub", "glo_mul", "glo_pow", "glo_inv", "glo_w", "glo_shift",
                  "glo_transcript_get_fields1"):
            getattr(L, f).restype = u64
        L.glo_add.argtypes = L.glo_sub.argtypes = L.glo_mul.argtypes = L.glo_pow.argtypes = [u64, u64]
        L.glo_inv.argtypes = L.glo_canon.argtypes = [u64]
        L.glo_w.argtypes = [ctypes.c_uint]
        _L[flavour] = L
    return _L[flavour]


def ptr(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_p64)


def A(x):
    return np.ascontiguousarray(x, dtype=np.uint64)


def rand_fe(rng, shape, canonical=True):
    a = rng.integers(0, 1 << 64, size=shape, dtype=np.uint64)
    if canonical:
        a = np.where(a >= np.uint64(P), a - np.uint64(P), a)
    return A(a)


def splitmix64(seed, n):
    """Deterministic synthetic workload generator: splitmix64 stream reduced mod p."""
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return A(np.where(z >= np.uint64(P), z - np.uint64(P), z))


def perm(state):
    s = A(state).copy()
    lib().glo_poseidon_perm(ptr(s))
    return s


def linear_hash(vals):
    v = A(vals)
    out = np.zeros(4, dtype=np.uint64)
    lib().glo_linear_hash(ptr(out), ptr(v) if v.size else None, u64(v.size))
    return out


def merkletree(src, ncols, nrows, flavour=""):
    src = A(src)
    nodes = np.zeros((2 * nrows - 1) * 4, dtype=np.uint64)
    lib(flavour).glo_merkletree(ptr(nodes), ptr(src), u64(ncols), u64(nrows))
    return nodes


def merkle_group_proof(nodes, src, height, width, idx):
    nlev = max(height - 1, 0).bit_length()
    proof = np.zeros(width + 4 * nlev, dtype=np.uint64)
    lib().glo_merkle_group_proof(ptr(proof), ptr(A(nodes)), ptr(A(src)), u64(height), u64(width), u64(idx))
    return proof


def merkle_verify(root, vals, sibs, idx):
    sibs = A(sibs).reshape(-1, 4)
    vals = A(vals)
    return lib().glo_merkle_verify(ptr(A(root)), ptr(vals), u64(vals.size), ptr(sibs), u64(sibs.shape[0]), u64(idx)) == 1


def ntt(src, n, ncols, inverse=False):
    src = A(src)
    dst = np.zeros(n * ncols, dtype=np.uint64)
    lib().glo_ntt(ptr(dst), ptr(src), u64(n), u64(ncols), ctypes.c_int(int(inverse)))
    return dst.reshape(n, ncols)


def extend_pol(src, n_ext, n, ncols, flavour=""):
    src = A(src)
    out = np.zeros(n_ext * ncols, dtype=np.uint64)
    lib(flavour).glo_extend_pol(ptr(out), ptr(src), u64(n_ext), u64(n), u64(ncols))
    return out.reshape(n_ext, ncols)


def dft_naive(col, inverse=False):
    col = A(col)
    out = np.zeros_like(col)
    lib().glo_dft_naive(ptr(out), ptr(col), u64(col.size), ctypes.c_int(int(inverse)))
    return out


def fri_fold(pol, prev_bits, cur_bits, nbits_ext, x):
    pol = A(pol)
    out = np.zeros((1 << cur_bits) * 3, dtype=np.uint64)
    lib().glo_fri_fold(ptr(out), ptr(pol), ctypes.c_uint(prev_bits), ctypes.c_uint(cur_bits), ctypes.c_uint(nbits_ext), ptr(A(x)))
    return out.reshape(-1, 3)


def fri_fold_group(vals, nx_bits, prev_bits, nbits_ext, g, x):
    out = np.zeros(3, dtype=np.uint64)
    lib().glo_fri_fold_group(ptr(out), ptr(A(vals)), ctypes.c_uint(nx_bits), ctypes.c_uint(prev_bits),
                             ctypes.c_uint(nbits_ext), u64(g), ptr(A(x)))
    return out


def fri_transpose(pol, degree, tbits):
    pol = A(pol)
    aux = np.zeros(degree * 3, dtype=np.uint64)
    lib().glo_fri_transpose(ptr(aux), ptr(pol), u64(degree), ctypes.c_uint(tbits))
    return aux


def e3_mul(a, b):
    out = np.zeros(3, dtype=np.uint64)
    lib().glo3_mul(ptr(out), ptr(A(a)), ptr(A(b)))
    return out


def e3_inv(a):
    out = np.zeros(3, dtype=np.uint64)
    lib().glo3_inv(ptr(out), ptr(A(a)))
    return out


class GloTranscript(ctypes.Structure):
    _fields_ = [("state", u64 * 4), ("pending", u64 * 8), ("out", u64 * 12),
                ("pending_cursor", ctypes.c_uint), ("out_cursor", ctypes.c_uint)]


class Transcript:
    def __init__(self):
        self.t = GloTranscript()
        lib().glo_transcript_init(ctypes.byref(self.t))

    def put(self, vals):
        v = A(vals).ravel()
        lib().glo_transcript_put(ctypes.byref(self.t), ptr(v), u64(v.size))

    def get_fields1(self):
        return int(lib().glo_transcript_get_fields1(ctypes.byref(self.t)))

    def get_field(self):
        out = np.zeros(3, dtype=np.uint64)
        lib().glo_transcript_get_field(ctypes.byref(self.t), ptr(out))
        return out

    def get_permutations(self, n, nbits):
        res = np.zeros(n, dtype=np.uint64)
        lib().glo_transcript_get_permutations(ctypes.byref(self.t), ptr(res), u64(n), u64(nbits))
        return res


def q_split(qq1, n, qdeg):
    qq1 = A(qq1)
    out = np.zeros(2 * n * qdeg * 3, dtype=np.uint64)  # n_ext = 2n rows x (qdeg*3); rows >= n stay zero
    lib().glo_q_split(ptr(out), ptr(qq1), u64(n), ctypes.c_uint(qdeg))
    return out


def batch_inverse3(src):
    src = A(src)
    out = np.zeros_like(src)
    lib().glo_batch_inverse3(ptr(out), ptr(src), u64(src.size // 3))
    return out


def geom_seq(n, start, ratio):
    out = np.zeros(n, dtype=np.uint64)
    lib().glo_geom_seq(ptr(out), u64(n), u64(start), u64(ratio))
    return out


def geom_seq3(n, ratio):
    out = np.zeros(3 * n, dtype=np.uint64)
    lib().glo_geom_seq3(ptr(out), u64(n), ptr(A(ratio)))
    return out


def zhinv(nbits, nbits_ext):
    out = np.zeros(1 << (nbits_ext - nbits), dtype=np.uint64)
    lib().glo_zhinv(ptr(out), ctypes.c_uint(nbits), ctypes.c_uint(nbits_ext))
    return out


def evmap(pols, prime, lev, lpev, n, ext_bits):
    """pols: list of (array, offset, dim, stride) views; returns evals [len(pols),3]"""
    k = len(pols)
    ptrs = (ctypes.c_void_p * k)()
    dims = np.zeros(k, dtype=np.uint32)
    strides = np.zeros(k, dtype=np.uint64)
    keep = []
    for i, (arr, off, dim, stride) in enumerate(pols):
        arr = A(arr)
        keep.append(arr)
        ptrs[i] = arr.ctypes.data + 8 * off
        dims[i] = dim
        strides[i] = stride
    pr = np.ascontiguousarray(prime, dtype=np.uint8)
    evals = np.zeros(3 * k, dtype=np.uint64)
    lib().glo_evmap(ptr(evals), u64(k), u64(n), ctypes.c_uint(ext_bits), ptrs,
                    dims.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), ptr(strides),
                    pr.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), ptr(A(lev)), ptr(A(lpev)))
    return evals.reshape(k, 3)


def chelpers_step42ns(ops, args, pols, const_pols, n_const, challenges, publics, x, x_stride, zhinv, q, row0, nrows):
    """The reference's step42ns interpreter restated (oracle/chelpers_oracle.c); q (host array) receives the rows."""
    ops, args = A(ops), A(args)
    ch, pb, zh = A(challenges).reshape(-1), A(publics).reshape(-1), A(zhinv).reshape(-1)
    f = lib().glo_chelpers_step42ns
    f.restype = ctypes.c_int
    st = f(ptr(ops), u64(ops.size), ptr(args) if args.size else None, u64(args.size), ctypes.c_void_p(pols.ctypes.data),
           ctypes.c_void_p(const_pols.ctypes.data), u64(n_const), ptr(ch) if ch.size else None, ptr(pb) if pb.size else None,
           ctypes.c_void_p(x.ctypes.data), u64(x_stride), ptr(zh), u64(zh.size), ctypes.c_void_p(q.ctypes.data), u64(row0), u64(nrows))
    if st != 0:
        raise RuntimeError("glo_chelpers_step42ns: " + {-1: "unknown opcode", -2: "argument count mismatch"}.get(st, str(st)))


def chelpers_step52ns(ops, args, pols, const_pols, n_const, challenges, evals, xdiv, xdivw, f, row0, nrows):
    """The reference's step52ns interpreter restated (oracle/chelpers_oracle.c); f (host array) receives the rows."""
    ops, args = A(ops), A(args)
    ch, ev = A(challenges).reshape(-1), A(evals).reshape(-1)
    fn = lib().glo_chelpers_step52ns
    fn.restype = ctypes.c_int
    st = fn(ptr(ops), u64(ops.size), ptr(args) if args.size else None, u64(args.size), ctypes.c_void_p(pols.ctypes.data),
            ctypes.c_void_p(const_pols.ctypes.data), u64(n_const), ptr(ch), ptr(ev), ctypes.c_void_p(xdiv.ctypes.data),
            ctypes.c_void_p(xdivw.ctypes.data), ctypes.c_void_p(f.ctypes.data), u64(row0), u64(nrows))
    if st != 0:
        raise RuntimeError("glo_chelpers_step52ns: " + {-1: "unknown opcode", -2: "argument count mismatch"}.get(st, str(st)))


def chelpers_stepbase(ops, args, pols, const_pols, n_const, challenges, publics, x, x_stride, rows):
    """The reference's step2prev / step3prev / step3 interpreter restated (oracle/chelpers_oracle.c); pols (host array) is
    read and written; rows are evaluated in the given order."""
    ops, args, rows = A(ops), A(args), A(rows)
    ch, pb = A(challenges).reshape(-1), A(publics).reshape(-1)
    fn = lib().glo_chelpers_stepbase
    fn.restype = ctypes.c_int
    st = fn(ptr(ops), u64(ops.size), ptr(args) if args.size else None, u64(args.size), ctypes.c_void_p(pols.ctypes.data),
            ctypes.c_void_p(const_pols.ctypes.data), u64(n_const), ptr(ch), ptr(pb), ctypes.c_void_p(x.ctypes.data), u64(x_stride),
            ptr(rows), u64(rows.size))
    if st != 0:
        raise RuntimeError("glo_chelpers_stepbase: " + {-1: "unknown opcode", -2: "argument count mismatch"}.get(st, str(st)))


def calculate_h1h2(area, h1_off, h1_stride, h2_off, h2_stride, f_off, f_stride, t_off, t_stride, dim, n):
    """polinomial.hpp:303-347 over strided views of one host array (offsets in elements); h1 / h2 are written into `area`.  Returns 0
    or 1 + the first row of f that is not in t."""
    assert area.dtype == np.uint64 and area.flags.c_contiguous
    fn = lib().glo_calculate_h1h2
    fn.restype = ctypes.c_int64
    base = area.ctypes.data
    P = lambda off: ctypes.c_void_p(base + 8 * off)
    return int(fn(P(h1_off), u64(h1_stride), P(h2_off), u64(h2_stride), P(f_off), u64(f_stride), P(t_off), u64(t_stride), ctypes.c_uint(dim),
                  u64(n)))


def calculate_z(area, z_off, z_stride, num_off, num_stride, den_off, den_stride, n):
    """polinomial.hpp:586-607 over strided views of one host array; returns whether the product closes."""
    assert area.dtype == np.uint64 and area.flags.c_contiguous
    fn = lib().glo_calculate_z
    fn.restype = ctypes.c_int
    base = area.ctypes.data
    P = lambda off: ctypes.c_void_p(base + 8 * off)
    return int(fn(P(z_off), u64(z_stride), P(num_off), u64(num_stride), P(den_off), u64(den_stride), u64(n)))
