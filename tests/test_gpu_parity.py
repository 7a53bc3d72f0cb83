"""GPU parity tests proper: every call goes through the C ABI (libmi_stark.so) and is compared
bit-for-bit with the CPU oracle on the same seeded inputs, with the golden fixtures of the reference,
and -- at BASELINE sizes -- through size-independent properties (round trips, linearity, Merkle paths)."""
import glob, os
import numpy as np
import pytest
import glo

pytestmark = pytest.mark.gpu
P = glo.P
BITS = [20, 16, 12, 9, 6]


@pytest.fixture(scope="module")
def ctx():
    import mi_stark
    c = mi_stark.Context(0)
    yield c
    c.close()


def dev_roundtrip(ctx, a):
    return ctx.to_host(ctx.to_device(a))


# ------------------------------------------------------------------ field arithmetic (device build)
def test_device_field_arithmetic_on_adversarial_operands(ctx):
    """mul / add / sub / shift-twiddles exactly as the kernels compute them (inline-asm multiply-add with carry-out,
    borrow-chain reduction, wave-uniform rare branches), against Python integers: every pair of a set of boundary
    values (0, 1, 2^32 +- 1, p - 1, p, p + 1, 2^64 - 1, powers of two and their complements), plus random pairs."""
    special = [0, 1, 2, 0xFFFFFFFE, 0xFFFFFFFF, 1 << 32, (1 << 32) + 1, P - 2, P - 1, P, P + 1, P + 2, (1 << 64) - 2, (1 << 64) - 1,
               0xFFFFFFFF00000000, 0x00000000FFFFFFFF, 0x8000000000000000, 0x7FFFFFFFFFFFFFFF, 0xFFFFFFFEFFFFFFFF]
    special += [1 << k for k in range(0, 64, 3)] + [P - (1 << k) for k in range(0, 64, 5)] + [((1 << 64) - (1 << k)) for k in range(1, 64, 7)]
    special = sorted(set(v % (1 << 64) for v in special))
    rng = np.random.default_rng(99)
    a = np.array([x for x in special for _ in special], dtype=np.uint64)
    b = np.array([y for _ in special for y in special], dtype=np.uint64)
    ra = rng.integers(0, 1 << 64, size=200000, dtype=np.uint64)
    rb = rng.integers(0, 1 << 64, size=200000, dtype=np.uint64)
    rb[:50000] = ra[:50000]                                    # squares
    ra[50000:60000] &= np.uint64(0xFFFFFFFF)                   # small operands: products with empty high words
    a, b = np.concatenate([a, ra]), np.concatenate([b, rb])
    n = a.size
    out = ctx.empty(5 * n)
    ctx.dbg_field_ops(out, ctx.to_device(a), ctx.to_device(b), n)
    got = ctx.to_host(out).reshape(5, n)
    ai, bi = [int(v) for v in a], [int(v) for v in b]
    want = np.array([[x * y % P for x, y in zip(ai, bi)], [(x + y) % P for x, y in zip(ai, bi)], [(x - y) % P for x, y in zip(ai, bi)],
                     [(-x) % P for x in ai], [(x << 40) % P for x in ai]], dtype=np.uint64)
    for k, name in enumerate(("mul", "add", "sub", "neg via 2^96", "shift 40")):
        bad = np.nonzero(got[k] != want[k])[0]
        assert bad.size == 0, (name, hex(ai[bad[0]]), hex(bi[bad[0]]), hex(int(got[k][bad[0]])), hex(int(want[k][bad[0]])))


# ------------------------------------------------------------------ Poseidon
@pytest.mark.parametrize("variant", [0, 1, 2])
def test_permute_matches_oracle(ctx, variant):
    ctx.set_poseidon_variant(variant)
    rng = np.random.default_rng(10 + variant)
    st = glo.rand_fe(rng, (1000, 12), canonical=False)
    st[0] = 0
    st[1] = np.arange(12)
    st[2] = P - 1
    st[3] = 2**64 - 1
    # powers of two: their squares / 4th powers have all-zero low words, which drives the rarely-taken
    # "lo < hh" and "x >= p" correction branches of the device arithmetic
    for k in range(64):
        st[4 + k] = np.uint64(1) << np.uint64(k)
        st[68 + k] = [(1 << ((k + 5 * j) % 64)) for j in range(12)]
        st[132 + k] = [P - (1 << ((k + 3 * j) % 63)) for j in range(12)]
    d_in = ctx.to_device(st)
    d_out = ctx.empty(st.size)
    ctx.permute(d_out, d_in, 1000)
    got = ctx.to_host(d_out).reshape(-1, 12)
    for i in range(1000):
        assert np.array_equal(got[i], glo.perm(st[i])), i
    assert [hex(x) for x in got[0][:4]] == ["0x3c18a9786cb0b359", "0xc4055e3364a246c3", "0x7953db0ab48808f4", "0xc71603f33a1144ca"]
    ctx.set_poseidon_variant(0)


def test_host_pointer_hashes(ctx):
    rng = np.random.default_rng(11)
    x = glo.rand_fe(rng, 12)
    assert np.array_equal(ctx.hash_full_result(x), glo.perm(x))
    assert np.array_equal(ctx.hash(x), glo.perm(x)[:4])
    for size in (0, 1, 3, 4, 5, 8, 9, 16, 17, 18, 39, 52, 665):
        v = glo.rand_fe(rng, size)
        assert np.array_equal(ctx.linear_hash(v), glo.linear_hash(v)), size


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_linear_hash_rows_all_widths(ctx, variant):
    ctx.set_poseidon_variant(variant)
    rng = np.random.default_rng(12)
    for w in list(range(0, 26)) + [39, 48, 52, 96, 128, 371, 665]:
        h = 70 if w < 100 else 9
        pitch = w + 3
        src = glo.rand_fe(rng, (h, pitch), canonical=(w % 2 == 0))
        d = ctx.to_device(src)
        out = ctx.empty(h * 4)
        ctx.linear_hash_rows(out, d, w, h, pitch=pitch)
        got = ctx.to_host(out).reshape(h, 4)
        for r in range(h):
            assert np.array_equal(got[r], glo.linear_hash(src[r, :w])), (w, r)
    ctx.set_poseidon_variant(0)


def test_linear_hash_rows_line_ring_corner_cases(ctx):
    """The leaf kernel streams whole aligned 128-byte lines through a 24-slot LDS ring: exercise every in-line start
    offset (matrix base not line-aligned, odd pitches), rows on both sides of a 1024-row wave group, row counts that
    leave idle lanes, and rows short enough that the stream ends inside the first or second line."""
    rng = np.random.default_rng(1234)
    for (h, w, pitch, off) in [(1, 5, 5, 0), (1, 5, 5, 1), (3, 7, 9, 5), (17, 8, 8, 3), (33, 9, 9, 15), (1030, 13, 14, 7),
                               (1100, 16, 17, 2), (70, 23, 23, 9), (70, 24, 29, 11), (2050, 31, 31, 13), (40, 33, 35, 6),
                               (130, 47, 47, 1), (20, 100, 101, 4), (18, 665, 665, 3), (5, 666, 667, 14)]:
        buf = glo.rand_fe(rng, off + h * pitch)
        d = ctx.to_device(buf)
        out = ctx.empty(h * 4)
        ctx.linear_hash_rows(out, d, w, h, pitch=pitch, src_off=off)
        got = ctx.to_host(out).reshape(h, 4)
        rows = buf[off:].reshape(h, pitch)
        for r in list(range(min(h, 40))) + list(range(max(0, h - 40), h)):
            assert np.array_equal(got[r], glo.linear_hash(np.ascontiguousarray(rows[r, :w]))), (h, w, pitch, off, r)


def test_linear_hash_absorb_column_windows(ctx):
    """Streaming leaf sponge: the columns of a row arrive as separate column windows (own pitch, own base), over
    several calls; the result must equal linear_hash of the concatenated row."""
    import mi_stark
    rng = np.random.default_rng(4321)
    for (h, widths_per_call) in [(70, [[32, 32, 32], [32, 25]]), (1100, [[8], [16, 8], [3]]), (33, [[64, 32, 19]]),
                                 (5, [[40], [40], [40], [7]]), (2050, [[32, 32], [32, 20]]), (17, [[8, 0, 8], [0], [16, 1]])]:
        cols = [glo.rand_fe(rng, (h, w + 2)) for call in widths_per_call for w in call]     # each window: pitch = w + 2
        flat = [w for call in widths_per_call for w in call]
        row_cat = np.concatenate([c[:, :w] for c, w in zip(cols, flat)], axis=1)
        dev = [ctx.to_device(np.concatenate([np.zeros(3, dtype=np.uint64), c.reshape(-1)])) for c in cols]   # base offset 3
        dig = ctx.zeros(h * 4)
        i = 0
        for ci, call in enumerate(widths_per_call):
            wins = [(dev[i + j], 3, w, w + 2) for j, w in enumerate(call)]
            ctx.linear_hash_absorb(dig, wins, h, first=(ci == 0), final=(ci == len(widths_per_call) - 1))
            i += len(call)
        got = ctx.to_host(dig).reshape(h, 4)
        for r in list(range(min(h, 30))) + list(range(max(0, h - 30), h)):
            assert np.array_equal(got[r], glo.linear_hash(np.ascontiguousarray(row_cat[r]))), (h, widths_per_call, r)
    with pytest.raises(mi_stark.MiStarkError):     # a window that is not the row's last must be a multiple of 8 wide
        d = ctx.zeros(8 * 12)
        ctx.linear_hash_absorb(ctx.zeros(8 * 4), [(d, 0, 12, 12)], 8, first=True, final=False)
    with pytest.raises(mi_stark.MiStarkError):
        d = ctx.zeros(8 * 20)
        ctx.linear_hash_absorb(ctx.zeros(8 * 4), [(d, 0, 12, 12), (d, 0, 8, 8)], 8, first=True, final=True)


@pytest.mark.parametrize("h,w", [(1, 5), (2, 3), (4, 1), (8, 9), (64, 18), (512, 21), (1024, 6), (2048, 39), (4096, 4), (1 << 13, 12)])
def test_merkle_tree_matches_oracle(ctx, h, w):
    rng = np.random.default_rng(h + w)
    src = glo.rand_fe(rng, (h, w))
    nodes = ctx.empty((2 * h - 1) * 4)
    ctx.merkle_build(nodes, ctx.to_device(src), w, h)
    got = ctx.to_host(nodes)
    want = glo.merkletree(src, w, h)
    assert np.array_equal(got, want)
    assert np.array_equal(ctx.merkle_build_host(src, w, h), want)


def test_merkle_rejects_bad_sizes(ctx):
    import mi_stark
    nodes = ctx.empty(64)
    with pytest.raises(mi_stark.MiStarkError):
        ctx.merkle_build(nodes, ctx.zeros(30), 5, 6)          # height must be a power of two
    ctx.merkle_build(nodes, ctx.zeros(8), 5, 0)                # zero rows: returns immediately like merkletree()


def test_group_proofs_match_oracle_and_verify(ctx):
    rng = np.random.default_rng(14)
    h, w = 1 << 10, 18
    src = glo.rand_fe(rng, (h, w))
    d_src = ctx.to_device(src)
    nodes = ctx.empty((2 * h - 1) * 4)
    ctx.merkle_build(nodes, d_src, w, h)
    idx = np.array([0, 1, 2, 511, 512, 1023, 700], dtype=np.uint64)
    stride = w + 4 * 10
    proofs = ctx.empty(idx.size * stride)
    ctx.merkle_group_proofs(proofs, nodes, d_src, h, w, idx)
    got = ctx.to_host(proofs).reshape(idx.size, stride)
    h_nodes = ctx.to_host(nodes)
    for q, i in enumerate(idx):
        assert np.array_equal(got[q], glo.merkle_group_proof(h_nodes, src, h, w, int(i)))
        assert glo.merkle_verify(h_nodes[-4:], got[q][:w], got[q][w:], int(i))


FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_golden_merkle_paths_on_gpu(ctx, path):
    """Replays the reference's golden proofs with GPU hashing only: linear_hash of the opened row, then the
    2->1 climb with mi_poseidon_hash, must land on the roots stored in the proof."""
    d = np.load(path)
    for q in range(min(3, len(d["q_index"]))):
        idx0 = int(d["q_index"][q])
        trees = [(d["root1"], d["s0_vals1"][q], d["s0_siblings1"][q], idx0),
                 (d["root3"], d["s0_vals3"][q], d["s0_siblings3"][q], idx0),
                 (d["root4"], d["s0_vals4"][q], d["s0_siblings4"][q], idx0)]
        for s in range(1, 5):
            trees.append((d[f"s{s}_root"], d[f"s{s}_vals"][q], d[f"s{s}_siblings"][q], idx0 % (1 << BITS[s])))
        for root, vals, sibs, idx in trees:
            cur = ctx.linear_hash(vals)
            for lvl in range(sibs.shape[0]):
                inp = np.zeros(12, dtype=np.uint64)
                if idx & 1:
                    inp[:4], inp[4:8] = sibs[lvl], cur
                else:
                    inp[:4], inp[4:8] = cur, sibs[lvl]
                cur = ctx.hash(inp)
                idx >>= 1
            assert np.array_equal(cur, root)


def test_transcript_put_in_one_launch_matches_the_oracle_state_machine(ctx):
    """mi_transcript_put (a whole Transcript::put per launch, wave-cooperative permutation) against the oracle's transcript: the same
    members after every put, for puts that complete no block, exactly one, many, and that start mid-block."""
    rng = np.random.default_rng(77)
    tr = {"state": np.zeros(4, dtype=np.uint64), "pending": np.zeros(8, dtype=np.uint64), "out": np.zeros(12, dtype=np.uint64), "pending_cursor": 0, "out_cursor": 0}
    ref = glo.Transcript()
    for n in (3, 4, 1, 8, 16, 5, 354, 7, 1, 192, 0, 9):
        v = glo.rand_fe(rng, n)
        ctx.transcript_put(tr, v)
        ref.put(v)
        t = ref.t
        assert [int(x) for x in t.state] == [int(x) for x in tr["state"]] and [int(x) for x in t.pending] == [int(x) for x in tr["pending"]], n
        assert int(t.pending_cursor) == tr["pending_cursor"] and int(t.out_cursor) == tr["out_cursor"], n
        if tr["out_cursor"]:
            assert [int(x) for x in t.out] == [int(x) for x in tr["out"]], n
    # and both permutation forms give the same single hash
    st = glo.rand_fe(rng, 12)
    a = ctx.hash_full_result(st)
    ctx.set_poseidon_coop_max(0)
    b = ctx.hash_full_result(st)
    ctx.set_poseidon_coop_max(16384)
    assert np.array_equal(a, b) and np.array_equal(a, glo.perm(st))


@pytest.mark.parametrize("coop_max", [0, 16384])
def test_merkle_levels_both_permutation_forms(ctx, coop_max):
    """Tree levels through the one-state-per-lane kernels only (coop_max 0) and with the wave-cooperative kernels for the small levels
    and the top: the same node array as the oracle's, heights around every switch-over (64, 512, 1024, 16384 nodes)."""
    ctx.set_poseidon_coop_max(coop_max)
    try:
        for h in (1, 2, 32, 64, 128, 1024, 2048, 1 << 15, 1 << 16):
            src = glo.splitmix64(h, h * 9).reshape(h, 9)
            nodes = ctx.empty((2 * h - 1) * 4)
            ctx.merkle_build(nodes, ctx.to_device(src), 9, h)
            assert np.array_equal(ctx.to_host(nodes), glo.merkletree(src, 9, h)), h
    finally:
        ctx.set_poseidon_coop_max(16384)


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_golden_final_polynomial_is_low_degree_under_the_gpu_intt(ctx, path):
    """The reference's own data through the device INTT (tests/test_oracle_golden.py has the argument): finalPol's 64 evaluations must
    transform to eight coefficients and 56 zeros."""
    fp = np.ascontiguousarray(np.load(path)["finalPol"].reshape(64, 3))
    d = ctx.to_device(fp)
    out = ctx.empty(64 * 3)
    ctx.ntt(out, d, 64, 3, inverse=True)
    c = ctx.to_host(out).reshape(64, 3)
    assert c[:8].any(axis=1).all() and not c[8:].any()
    assert np.array_equal(c, glo.ntt(fp, 64, 3, inverse=True).reshape(64, 3))


def test_golden_root2_is_the_tree_without_columns_on_gpu(ctx):
    """Every golden proof's root2 is the root of MerkleTreeGL(2^20 rows, 0 columns) -- the recursive STARKs commit nothing in stage 2
    (tests/test_oracle_golden.py): the tree builder on a width of 0 at the golden height must produce exactly it."""
    d = np.load(FILES[0])
    n = 1 << 20
    nodes = ctx.empty((2 * n - 1) * 4)
    ctx.merkle_build(nodes, ctx.empty(16), 0, n)
    assert np.array_equal(ctx.to_host(nodes[-4:]), d["root2"])
    assert all(np.array_equal(np.load(f)["root2"], d["root2"]) for f in FILES)
    # (such a tree is one value per level: the builder computes log2(n) hashes and fills) -- every node, against the oracle's tree
    for m in (1, 2, 4, 8, 1024, 1 << 16):
        got = ctx.empty((2 * m - 1) * 4 + 4)
        got.fill_(-1)
        ctx.merkle_build(got, ctx.empty(16), 0, m)
        h = ctx.to_host(got)
        assert np.array_equal(h[:-4], glo.merkletree(np.zeros(1, dtype=np.uint64), 0, m)), m
        assert (h[-4:] == np.uint64(0xFFFFFFFFFFFFFFFF)).all(), m       # nothing past the last node


# ------------------------------------------------------------------ NTT / LDE
NTT_CASES = [(1, 1), (1, 5), (2, 1), (2, 3), (4, 6), (8, 1), (16, 3), (32, 7), (64, 33), (128, 2), (256, 37), (512, 3),
             (1024, 1), (1024, 65), (2048, 6), (4096, 32), (1 << 13, 5), (1 << 16, 3), (1 << 17, 1)]


@pytest.mark.parametrize("n,ncols", NTT_CASES)
def test_ntt_matches_oracle(ctx, n, ncols):
    rng = np.random.default_rng(n * 131 + ncols)
    x = glo.rand_fe(rng, (n, ncols), canonical=(ncols % 2 == 1))
    d = ctx.to_device(x)
    out = ctx.empty(n * ncols)
    for inverse in (False, True):
        ctx.ntt(out, d, n, ncols, inverse=inverse)
        assert np.array_equal(ctx.to_host(out).reshape(n, ncols), glo.ntt(x, n, ncols, inverse=inverse)), inverse


@pytest.mark.parametrize("n,ncols", [(2, 1), (8, 1), (16, 3), (128, 2), (256, 1), (512, 1), (512, 3), (1024, 2), (1 << 13, 1)])
def test_small_transforms_stay_inside_their_buffers(ctx, n, ncols):
    """Transforms with fewer beta rows than a tile has lanes for (TJ * TCP < tile width): the surplus lanes must neither
    load nor store.  dst is followed by a canary region 40x its size (the overrun reached 32x), src sits at the very end of
    its allocation, and the transform runs out of place and in place, forward and inverse, with both tile widths."""
    rng = np.random.default_rng(n * 7 + ncols)
    x = glo.rand_fe(rng, (n, ncols))
    canary = np.uint64(0xC0FFEE0DDBA11AD5)
    try:
        for log_b in (5, 4):
            ctx.set_ntt_tile(log_b)
            for inverse in (False, True):
                want = glo.ntt(x, n, ncols, inverse=inverse).reshape(-1)
                buf = np.full(n * ncols * 41, canary, dtype=np.uint64)
                d = ctx.to_device(buf)
                ctx.ntt(d, ctx.to_device(x), n, ncols, inverse=inverse)
                got = ctx.to_host(d)
                assert np.array_equal(got[:n * ncols], want), (log_b, inverse)
                assert np.all(got[n * ncols:] == canary), (log_b, inverse, int(np.nonzero(got[n * ncols:] != canary)[0][0]))
                buf[:n * ncols] = x.reshape(-1)
                d = ctx.to_device(buf)
                ctx.ntt(d, d, n, ncols, inverse=inverse)                      # in place
                got = ctx.to_host(d)
                assert np.array_equal(got[:n * ncols], want), (log_b, inverse, "in place")
                assert np.all(got[n * ncols:] == canary), (log_b, inverse, "in place")
    finally:
        ctx.set_ntt_tile(5)


@pytest.mark.parametrize("n,n_ext,ncols", [(16, 32, 2), (32, 64, 2), (32, 128, 3), (64, 128, 1), (128, 256, 5), (256, 512, 2), (2, 4, 1)])
def test_small_ldes_stay_inside_their_buffers(ctx, n, n_ext, ncols):
    """Same for extendPol, fused middle pass on and off: nothing is written past row n_ext of the output."""
    rng = np.random.default_rng(n + 3 * n_ext + ncols)
    x = glo.rand_fe(rng, (n, ncols))
    want = glo.extend_pol(x, n_ext, n, ncols).reshape(-1)
    canary = np.uint64(0xC0FFEE0DDBA11AD5)
    try:
        for fuse in (1, 0):
            ctx.set_lde_fuse(fuse)
            d = ctx.to_device(np.full(n_ext * ncols * 41, canary, dtype=np.uint64))
            ctx.lde(d, ctx.to_device(x), n_ext, n, ncols)
            got = ctx.to_host(d)
            assert np.array_equal(got[:n_ext * ncols], want), fuse
            assert np.all(got[n_ext * ncols:] == canary), (fuse, int(np.nonzero(got[n_ext * ncols:] != canary)[0][0]))
    finally:
        ctx.set_lde_fuse(1)


def test_absorb_first_call_without_windows_leaves_zero_capacity(ctx):
    """A first absorb whose windows are all empty must still initialise the running capacity, and a kernel with fewer
    than 16 rows and a carried-in capacity must not read digests of rows that do not exist."""
    rng = np.random.default_rng(4242)
    h, w = 5, 24
    src = glo.rand_fe(rng, (h, w))
    d = ctx.to_device(src)
    dig = ctx.to_device(np.full(h * 4, 0xDEADBEEF, dtype=np.uint64))
    ctx.linear_hash_absorb(dig, [(d, 0, 0, w)], h, True, False)          # nothing to absorb yet
    ctx.linear_hash_absorb(dig, [(d, 0, 16, w)], h, False, False)
    ctx.linear_hash_absorb(dig, [(d, 16, 8, w)], h, False, True)
    got = ctx.to_host(dig).reshape(h, 4)
    for r in range(h):
        assert np.array_equal(got[r], glo.linear_hash(src[r])), r


def test_ntt_and_lde_with_16_wide_tiles(ctx):
    """The NTT tile width is a tuning knob (128-byte or 256-byte row segments); results must not depend on it."""
    ctx.set_ntt_tile(4)
    try:
        rng = np.random.default_rng(77)
        for n, ncols in ((64, 33), (1024, 1), (4096, 17), (1 << 13, 5), (1 << 16, 3)):
            x = glo.rand_fe(rng, (n, ncols))
            out = ctx.empty(n * ncols)
            for inverse in (False, True):
                ctx.ntt(out, ctx.to_device(x), n, ncols, inverse=inverse)
                assert np.array_equal(ctx.to_host(out).reshape(n, ncols), glo.ntt(x, n, ncols, inverse=inverse)), (n, ncols, inverse)
        x = glo.rand_fe(rng, (1 << 12, 37))
        out = ctx.empty((1 << 13) * 37)
        ctx.lde(out, ctx.to_device(x), 1 << 13, 1 << 12, 37)
        assert np.array_equal(ctx.to_host(out).reshape(1 << 13, 37), glo.extend_pol(x, 1 << 13, 1 << 12, 37))
    finally:
        ctx.set_ntt_tile(5)


def test_ntt_in_place_pitched_and_host_variant(ctx):
    rng = np.random.default_rng(21)
    n, ncols, pitch = 1 << 12, 3, 7
    buf = glo.rand_fe(rng, (n, pitch))
    d = ctx.to_device(buf)
    ctx.ntt(d, d, n, ncols, inverse=True, dst_pitch=pitch, src_pitch=pitch)       # LEv-style in-place INTT (starks.cpp:325)
    got = ctx.to_host(d).reshape(n, pitch)
    assert np.array_equal(got[:, :ncols], glo.ntt(np.ascontiguousarray(buf[:, :ncols]), n, ncols, inverse=True))
    assert np.array_equal(got[:, ncols:], buf[:, ncols:])                           # untouched columns
    x = glo.rand_fe(rng, (1 << 9, 4))
    assert np.array_equal(ctx.ntt_host(x, 1 << 9, 4), glo.ntt(x, 1 << 9, 4))


def test_ntt_identities(ctx):
    n = 1 << 14
    imp = np.zeros(n, dtype=np.uint64); imp[0] = 1
    out = ctx.empty(n)
    ctx.ntt(out, ctx.to_device(imp), n, 1)
    assert np.all(ctx.to_host(out) == 1)
    w = glo.lib().glo_w(14)
    geo = glo.geom_seq(n, 1, glo.lib().glo_inv(w))
    ctx.ntt(out, ctx.to_device(geo), n, 1)
    want = np.zeros(n, dtype=np.uint64); want[1] = n
    assert np.array_equal(ctx.to_host(out), want)
    for fill in (0, P - 1):
        c = np.full(n, fill, dtype=np.uint64)
        ctx.ntt(out, ctx.to_device(c), n, 1)
        assert np.array_equal(ctx.to_host(out), glo.ntt(c, n, 1)[:, 0])
    # sparse power-of-two inputs (rare correction branches), forward and inverse
    c = np.array([(1 << (i % 64)) if i % 7 == 0 else 0 for i in range(n)], dtype=np.uint64)
    for inverse in (False, True):
        ctx.ntt(out, ctx.to_device(c), n, 1, inverse=inverse)
        assert np.array_equal(ctx.to_host(out), glo.ntt(c, n, 1, inverse=inverse)[:, 0])


def test_config1_forward_ntt_2pow20(ctx):
    """BASELINE config 1 on the GPU: forward NTT over 2^20 elements, 1 column, bit-exact vs the CPU oracle."""
    n = 1 << 20
    x = glo.splitmix64(0x5EED0001, n)
    out = ctx.empty(n)
    ctx.ntt(out, ctx.to_device(x), n, 1)
    assert np.array_equal(ctx.to_host(out), glo.ntt(x, n, 1)[:, 0])


def test_config2_roundtrip_2pow23(ctx):
    """BASELINE config 2: 2^23-row x 1-column NTT + iNTT round trip returns the input bit-exactly."""
    n = 1 << 23
    d = ctx.empty(n)
    ctx.fill_synthetic(d, n, 0x5EED0002)
    x = ctx.to_host(d).copy()
    assert np.array_equal(x[:1000], glo.splitmix64(0x5EED0002, 1000))
    y = ctx.empty(n)
    ctx.ntt(y, d, n, 1)
    hy = ctx.to_host(y)
    # spot-check a few outputs against the definition X[k] = sum x[i] w^(ik) via Horner on the host
    w = glo.lib().glo_w(23)
    xi = [int(v) for v in x]
    assert sum(xi) % P == int(hy[0])
    wk, acc = pow(w, 12345, P), 0
    for v in reversed(xi):
        acc = (acc * wk + v) % P
    assert acc == int(hy[12345])
    ctx.ntt(y, y, n, 1, inverse=True)
    assert np.array_equal(ctx.to_host(y), x)


# sizes where the fused middle pass (last INTT pass + first NTT pass in one kernel) applies -- radix r1 -> r2:
# 16->32 .. 128->256 single pass, (5,4)->(5,5), (6,6)->(7,6), (7,7)->(8,7), (8,7)->(8,8), blowup 4: 64->256,
# (5,4)->(6,5), (7,6)->(8,7), three-pass splits (6,6,5)->(6,6,6), (6,6,6)->(7,6,6), (6,6,5)->(7,6,6) -- and where it does not (n < 16, one column, 256->(5,4), (8,8)->(6,6,5), n_ext = n).
LDE_CASES = [(1, 2, 3), (2, 4, 1), (4, 8, 5), (16, 32, 37), (32, 64, 2), (64, 128, 9), (128, 256, 33), (64, 256, 5),
             (256, 512, 3), (256, 1024, 6), (512, 1024, 70), (512, 2048, 35), (1024, 2048, 1), (1024, 2048, 4),
             (1 << 12, 1 << 13, 33), (1 << 13, 1 << 15, 3), (1 << 14, 1 << 15, 7), (1 << 15, 1 << 16, 6),
             (1 << 16, 1 << 17, 2), (1 << 17, 1 << 18, 5), (1 << 18, 1 << 19, 3), (1 << 17, 1 << 19, 2), (8, 8, 4)]


@pytest.mark.parametrize("n,n_ext,ncols", LDE_CASES)
def test_lde_matches_oracle(ctx, n, n_ext, ncols):
    rng = np.random.default_rng(n + n_ext + ncols)
    x = glo.rand_fe(rng, (n, ncols))
    want = glo.extend_pol(x, n_ext, n, ncols)
    try:
        for fuse in (1, 0):
            ctx.set_lde_fuse(fuse)
            out = ctx.empty(n_ext * ncols)
            ctx.lde(out, ctx.to_device(x), n_ext, n, ncols)
            assert np.array_equal(ctx.to_host(out).reshape(n_ext, ncols), want), fuse
    finally:
        ctx.set_lde_fuse(1)


def test_lde_column_chunks_and_pitches(ctx):
    """Wide traces are processed in column chunks sized to the workspace; chunking must not change results."""
    import mi_stark
    rng = np.random.default_rng(33)
    n, n_ext, ncols, in_pitch, out_pitch = 1 << 10, 1 << 11, 100, 103, 101
    src = glo.rand_fe(rng, (n, in_pitch))
    small = mi_stark.Context(0, workspace_limit=1 << 20)      # 1 MiB -> chunks of 32 columns
    out = small.zeros(n_ext * out_pitch)
    small.lde(out, small.to_device(src), n_ext, n, ncols, out_pitch=out_pitch, in_pitch=in_pitch)
    got = small.to_host(out).reshape(n_ext, out_pitch)
    assert np.array_equal(got[:, :ncols], glo.extend_pol(np.ascontiguousarray(src[:, :ncols]), n_ext, n, ncols))
    assert not got[:, ncols:].any()
    x = glo.rand_fe(rng, (1 << 11, 70))
    o2 = small.empty(x.size)
    small.ntt(o2, small.to_device(x), 1 << 11, 70)
    assert np.array_equal(small.to_host(o2).reshape(1 << 11, 70), glo.ntt(x, 1 << 11, 70))
    assert np.array_equal(small.lde_host(x[:256, :5], 512, 256, 5), glo.extend_pol(np.ascontiguousarray(x[:256, :5]), 512, 256, 5))
    small.close()
    # 3 MiB: chunks of 64 columns AND room for a second compact ping-pong buffer, so that the strided output is only
    # written by the last pass (the configuration the full-size run takes with the default 32 GiB)
    mid = mi_stark.Context(0, workspace_limit=3 << 20)
    for (n, n_ext, ncols, in_pitch, out_pitch) in ((1 << 10, 1 << 11, 200, 203, 201), (1 << 9, 1 << 11, 150, 150, 150)):
        src = glo.rand_fe(rng, (n, in_pitch))
        out = mid.zeros(n_ext * out_pitch)
        mid.lde(out, mid.to_device(src), n_ext, n, ncols, out_pitch=out_pitch, in_pitch=in_pitch)
        got = mid.to_host(out).reshape(n_ext, out_pitch)
        assert np.array_equal(got[:, :ncols], glo.extend_pol(np.ascontiguousarray(src[:, :ncols]), n_ext, n, ncols))
        assert not got[:, ncols:].any()
    mid.close()


@pytest.mark.parametrize("n,n_ext,ncols,chunk,pinned", [(1 << 10, 1 << 11, 70, 32, True), (1 << 10, 1 << 11, 70, 8, False), (1 << 12, 1 << 13, 100, 0, True),
                                                     (1 << 9, 1 << 11, 33, 16, True), (1 << 8, 1 << 9, 5, 8, False), (1 << 11, 1 << 12, 64, 32, True),
                                                     (1 << 16, 1 << 17, 70, 32, False)])   # (16 MiB chunks: the packing really runs on several threads)
@pytest.mark.parametrize("pack", [0, 3])
def test_stage_driver_streams_a_host_trace(ctx, n, n_ext, ncols, chunk, pinned, pack):
    """mi_lde_merkle_host: host trace in column chunks on a copy stream, LDE + streaming leaf absorption behind it; the
    resident extension and the whole node array equal the oracle's, with pinned and pageable host memory, ragged last chunk,
    one chunk only, a pitched output, and when called twice in a row (buffers and events are reused).  pack: the chunks as strided
    2-D copies (0) or packed by that many host threads into page-locked staging and sent as contiguous copies."""
    import torch
    ctx.set_host_pack_threads(pack)
    rng = np.random.default_rng(n + ncols + chunk)
    trace = glo.rand_fe(rng, (n, ncols))
    want_ext = glo.extend_pol(trace, n_ext, n, ncols)
    want_nodes = glo.merkletree(want_ext, ncols, n_ext)
    host = torch.from_numpy(trace.view(np.int64).reshape(-1).copy())
    if pinned:
        host = host.pin_memory()
    pitch = ncols + 3
    for _ in range(2):
        ext = ctx.to_device(np.full(n_ext * pitch, 0x5151, dtype=np.uint64))
        nodes = ctx.empty((2 * n_ext - 1) * 4)
        ctx.lde_merkle_host(nodes, ext, host.data_ptr(), n, n_ext, ncols, ext_pitch=pitch, chunk_cols=chunk)
        ctx.sync()
        got = ctx.to_host(ext).reshape(n_ext, pitch)
        assert np.array_equal(got[:, :ncols], want_ext)
        assert np.all(got[:, ncols:] == 0x5151)
        assert np.array_equal(ctx.to_host(nodes), want_nodes)
    ctx.set_host_pack_threads(0)


def test_config3_shape_at_2pow18_rows_bit_exact():
    """BASELINE configs[2] with fewer rows: 2^18 x 665 trace -> LDE to 2^19 -> Poseidon Merkle tree, the whole result
    compared with the oracle (Merkle root = a checksum of all 2^19 x 665 extended values, plus sampled rows).  A small
    workspace forces the production column chunking (21 chunks of 32 columns + a ragged one)."""
    import mi_stark
    c = mi_stark.Context(0, workspace_limit=256 << 20)
    n, n_ext, ncols = 1 << 18, 1 << 19, 665
    trace = glo.splitmix64(0x5EED0003, n * ncols).reshape(n, ncols)
    d_ext = c.empty(n_ext * ncols)
    c.lde(d_ext, c.to_device(trace), n_ext, n, ncols)
    nodes = c.empty((2 * n_ext - 1) * 4)
    c.merkle_build(nodes, d_ext, ncols, n_ext)
    want_ext = glo.extend_pol(trace, n_ext, n, ncols)
    got_ext = c.to_host(d_ext).reshape(n_ext, ncols)
    for r in (0, 1, 2, n - 1, n, n_ext - 1, 12345, 400001):
        assert np.array_equal(got_ext[r], want_ext[r]), r
    assert np.array_equal(got_ext, want_ext)
    want_nodes = glo.merkletree(want_ext, ncols, n_ext)
    assert np.array_equal(c.to_host(nodes)[-4:], want_nodes[-4:])
    c.close()


def test_lde_merkle_large_properties(ctx):
    """2^20 x 32 LDE + tree: linearity of the LDE, even rows reproduce the trace's coset-free part, and a
    Merkle path opened from the device tree verifies on the CPU against the device root."""
    n, n_ext, ncols = 1 << 20, 1 << 21, 32
    a, b = ctx.empty(n * ncols), ctx.empty(n * ncols)
    ctx.fill_synthetic(a, n * ncols, 1)
    ctx.fill_synthetic(b, n * ncols, 2)
    ea, eb, es = ctx.empty(n_ext * ncols), ctx.empty(n_ext * ncols), ctx.empty(n_ext * ncols)
    ctx.lde(ea, a, n_ext, n, ncols)
    ctx.lde(eb, b, n_ext, n, ncols)
    ha, hb = ctx.to_host(a), ctx.to_host(b)
    hs = np.array([(int(u) + int(v)) % P for u, v in zip(ha[:4096], hb[:4096])], dtype=np.uint64)
    # full-size modular sum with uint64 wrap-around arithmetic (python ints would take minutes)
    sw = ha + hb
    carry = sw < ha
    sw = np.where(carry, sw + np.uint64(0xFFFFFFFF), sw)
    sw = np.where(sw >= np.uint64(P), sw - np.uint64(P), sw)
    assert np.array_equal(sw[:4096], hs)
    ctx.lde(es, ctx.to_device(sw), n_ext, n, ncols)
    hea, heb, hes = ctx.to_host(ea), ctx.to_host(eb), ctx.to_host(es)
    t = hea + heb
    t = np.where(t < hea, t + np.uint64(0xFFFFFFFF), t)
    t = np.where(t >= np.uint64(P), t - np.uint64(P), t)
    assert np.array_equal(t, hes)                                    # LDE(a+b) = LDE(a) + LDE(b)
    # tree + path
    nodes = ctx.empty((2 * n_ext - 1) * 4)
    ctx.merkle_build(nodes, ea, ncols, n_ext)
    idx = np.array([0, 123456, n_ext - 1], dtype=np.uint64)
    stride = ncols + 4 * 21
    proofs = ctx.empty(idx.size * stride)
    ctx.merkle_group_proofs(proofs, nodes, ea, n_ext, ncols, idx)
    pr = ctx.to_host(proofs).reshape(idx.size, stride)
    root = ctx.to_host(nodes[-4:])
    rows = hea.reshape(n_ext, ncols)
    for q, i in enumerate(idx):
        assert np.array_equal(pr[q][:ncols], rows[int(i)])
        assert glo.merkle_verify(root, pr[q][:ncols], pr[q][ncols:], int(i))


# ------------------------------------------------------------------ FRI
@pytest.mark.parametrize("prev,cur", [(6, 5), (8, 6), (9, 6), (10, 6), (12, 7), (14, 8), (10, 10)])
def test_fri_fold_matches_oracle(ctx, prev, cur):
    rng = np.random.default_rng(prev * 17 + cur)
    pol = glo.rand_fe(rng, (1 << prev, 3))
    x = glo.rand_fe(rng, 3)
    out = ctx.empty((1 << cur) * 3)
    ctx.fri_fold(out, ctx.to_device(pol), prev, cur, 16, x)
    assert np.array_equal(ctx.to_host(out).reshape(-1, 3), glo.fri_fold(pol, prev, cur, 16, x))


def test_fri_transpose_and_step_tree(ctx):
    rng = np.random.default_rng(40)
    prev, nxt = 12, 8
    pol = glo.rand_fe(rng, (1 << prev) * 3)
    aux = ctx.empty(pol.size)
    ctx.fri_transpose(aux, ctx.to_device(pol), 1 << prev, nxt)
    want = glo.fri_transpose(pol, 1 << prev, nxt)
    assert np.array_equal(ctx.to_host(aux), want)
    # friProve.cpp:110-126: MerkleTreeGL(nGroups, groupSize*3) over the transposed polynomial
    groups, gsize = 1 << nxt, (1 << (prev - nxt)) * 3
    nodes = ctx.empty((2 * groups - 1) * 4)
    ctx.merkle_build(nodes, aux, gsize, groups)
    assert np.array_equal(ctx.to_host(nodes), glo.merkletree(want, gsize, groups))


def test_golden_fri_fold_on_gpu(ctx, golden_dir):
    """The reference's own proofs, folded on the GPU: plant the 43 opened groups of a step into an otherwise
    random polynomial, fold, and the planted outputs must equal the next step's opened values / finalPol."""
    d = np.load(os.path.join(golden_dir, "recursive1_proof_0.npz"))
    rng = np.random.default_rng(41)
    for s in range(1, 5):
        prev, cur = BITS[s - 1], BITS[s]
        nx = 1 << (prev - cur)
        pol = glo.rand_fe(rng, (1 << prev, 3))
        gs = []
        for q in range(len(d["q_index"])):
            g = int(d["q_index"][q]) % (1 << cur)
            vals = d[f"s{s}_vals"][q].reshape(nx, 3)
            for i in range(nx):
                pol[i * (1 << cur) + g] = vals[i]
            gs.append(g)
        out = ctx.empty((1 << cur) * 3)
        ctx.fri_fold(out, ctx.to_device(pol), prev, cur, 20, d["special_x"][s - 1])
        got = ctx.to_host(out).reshape(-1, 3)
        for q, g in enumerate(gs):
            if s < 4:
                j = g >> BITS[s + 1]
                want = d[f"s{s + 1}_vals"][q][3 * j:3 * j + 3]
            else:
                want = d["finalPol"][g]
            assert np.array_equal(got[g], want), (s, q)


# ------------------------------------------------------------------ the rest of genProof's loops
def test_q_split_batch_inverse_tables(ctx):
    rng = np.random.default_rng(50)
    n, qdeg = 1 << 10, 2
    qq1 = glo.rand_fe(rng, (2 * n) * 3)
    qq2 = ctx.empty(2 * n * qdeg * 3)
    ctx.q_split(qq2, ctx.to_device(qq1), n, 2 * n, qdeg)
    assert np.array_equal(ctx.to_host(qq2), glo.q_split(qq1, n, qdeg))
    src = glo.rand_fe(rng, 999 * 3)
    src[:3] = 0
    d = ctx.to_device(src)
    ctx.batch_inverse3(d, d, 999)
    assert np.array_equal(ctx.to_host(d), glo.batch_inverse3(src))
    for n_ in (1, 2, 3, 4, 5, 1025, 70000):                          # (four elements of a thread share one inversion) ragged sizes, zeros anywhere
        src = glo.rand_fe(rng, n_ * 3, canonical=False)
        src.reshape(-1, 3)[rng.integers(0, n_, 3)] = 0
        d = ctx.to_device(np.concatenate([src, np.full(3, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)]))
        ctx.batch_inverse3(d, d, n_)
        got = ctx.to_host(d)
        assert np.array_equal(got[:-3], glo.batch_inverse3(src)) and (got[-3:] == np.uint64(0xFFFFFFFFFFFFFFFF)).all(), n_
    w = glo.lib().glo_w(13)
    out = ctx.empty(1 << 13)
    ctx.geom_seq(out, 1 << 13, 49, w)                                 # x_2ns (starks.hpp:154-159)
    assert np.array_equal(ctx.to_host(out), glo.geom_seq(1 << 13, 49, w))
    r = glo.rand_fe(rng, 3)
    o3 = ctx.empty(3 * 5000)
    ctx.geom_seq3(o3, 5000, r)                                        # LEv (starks.cpp:311-323)
    assert np.array_equal(ctx.to_host(o3), glo.geom_seq3(5000, r))
    assert np.array_equal(ctx.zhinv(10, 11), glo.zhinv(10, 11))
    assert np.array_equal(ctx.zhinv(8, 11), glo.zhinv(8, 11))
    # xDivXSubXi (starks.cpp:350-365)
    xs = glo.geom_seq(4096, 49, glo.lib().glo_w(12))
    xi = glo.rand_fe(rng, 3)
    o = ctx.empty(4096 * 3)
    ctx.x_div_x_sub(o, ctx.to_device(xs), 4096, xi)
    got = ctx.to_host(o).reshape(-1, 3)
    for k in (0, 1, 4095):
        den = np.array([(int(xs[k]) - int(xi[0])) % P, (-int(xi[1])) % P, (-int(xi[2])) % P], dtype=np.uint64)
        want = glo.e3_mul(glo.e3_inv(den), np.array([xs[k], 0, 0], dtype=np.uint64))
        assert np.array_equal(got[k], want)
    # (four elements of a thread share one inversion) every row, sizes that leave a thread's batch ragged, and a denominator that is zero:
    # xi in the base field and equal to one of the x (its inverse is defined as 0, like Goldilocks3::inv's)
    for n_ in (1, 2, 3, 5, 257, 1023, 4096):
        for xi_ in (xi, np.array([xs[min(2, n_ - 1)], 0, 0], dtype=np.uint64)):
            o = ctx.empty(n_ * 3 + 3)
            o.fill_(-1)
            ctx.x_div_x_sub(o, ctx.to_device(xs[:n_]), n_, xi_)
            got = ctx.to_host(o)
            assert (got[-3:] == np.uint64(0xFFFFFFFFFFFFFFFF)).all()
            den = np.zeros((n_, 3), dtype=np.uint64)
            den[:, 0] = [(int(v) - int(xi_[0])) % P for v in xs[:n_]]
            den[:, 1], den[:, 2] = (-int(xi_[1])) % P, (-int(xi_[2])) % P
            inv = glo.batch_inverse3(den.reshape(-1)).reshape(-1, 3)
            want = np.array([glo.e3_mul(inv[k], np.array([xs[k], 0, 0], dtype=np.uint64)) for k in range(n_)], dtype=np.uint64)
            assert np.array_equal(got[:-3].reshape(-1, 3), want), (n_, xi_)


def test_evmap_matches_oracle(ctx):
    rng = np.random.default_rng(51)
    n, ext_bits, width = 1 << 12, 1, 23
    cm = glo.rand_fe(rng, ((n << ext_bits), width))
    q = glo.rand_fe(rng, ((n << ext_bits), 3))
    lev, lpev = glo.rand_fe(rng, (n, 3)), glo.rand_fe(rng, (n, 3))
    d_cm, d_q = ctx.to_device(cm), ctx.to_device(q)
    pols_h = [(cm, c, 1, width) for c in (0, 5, 5, 22, 7)] + [(cm, 10, 3, width), (q, 0, 3, 3)]
    pols_d = [(d_cm, c, 1, width) for c in (0, 5, 5, 22, 7)] + [(d_cm, 10, 3, width), (d_q, 0, 3, 3)]
    prime = [0, 1, 0, 1, 1, 0, 1]
    ev = ctx.empty(len(pols_d) * 3)
    ctx.evmap(ev, pols_d, prime, ctx.to_device(lev), ctx.to_device(lpev), n, ext_bits)
    want = glo.evmap(pols_h, prime, lev, lpev, n, ext_bits)
    assert np.array_equal(ctx.to_host(ev).reshape(-1, 3), want)
    # row shards' shares (mi_evmap_range_dev): the partial sums of a ragged partition of the rows add up, in F_p^3, to the evaluations
    P = (1 << 64) - (1 << 32) + 1
    total = np.zeros((len(pols_d), 3), dtype=object)
    for row0, nrows in ((0, 1), (1, 1023), (1024, 2048), (3072, 1000), (4072, 24)):
        part = ctx.zeros(len(pols_d) * 3)
        ctx.evmap(part, pols_d, prime, ctx.to_device(lev), ctx.to_device(lpev), n, ext_bits, row0=row0, nrows=nrows)
        total = (total + ctx.to_host(part).reshape(-1, 3).astype(object)) % P
    assert np.array_equal(total.astype(np.uint64), want)


def _device_ops(ctx):
    from shard import device_ops
    return device_ops(ctx)


@pytest.mark.parametrize("world,ncols,tile", [(2, 37, 8), (4, 150, 32), (8, 665, 32), (4, 70, 8)])
def test_sharded_path_emulated_on_one_gpu(ctx, world, ncols, tile):
    """The multi-GPU orchestration (shard.py) with the REAL device ops, all ranks emulated one after the other on
    this GPU and the point-to-point exchange / all-gather done by tensor copies following the plan's message list:
    the root must equal the single-GPU root, and every rank's leaf digests its slice of level 0."""
    import torch
    from shard import ShardPlan, exchange_messages, phase_lde, phase_absorb, phase_subtree, phase_top
    n = 1 << 10
    n_ext = 2 * n
    Ops = _device_ops(ctx)
    # single-GPU reference
    full = ctx.empty(n * ncols)
    ctx.fill_synthetic_2d(full, n, ncols, ncols, 0, 0x5EED0003)
    ext1, nodes1 = ctx.empty(n_ext * ncols), ctx.empty((2 * n_ext - 1) * 4)
    ctx.lde(ext1, full, n_ext, n, ncols)
    ctx.merkle_build(nodes1, ext1, ncols, n_ext)
    want_root = ctx.to_host(nodes1[-4:])
    want_leaves = ctx.to_host(nodes1[:n_ext * 4])
    # emulated ranks
    plans = [ShardPlan(n=n, n_ext=n_ext, ncols=ncols, world=world, rank=r, tile=tile) for r in range(world)]
    bufs, traces = [], []
    for p in plans:
        trace = ctx.empty(n * max(p.my_cols, 1))
        for k, (c0, w) in enumerate(p.my_tile_cols()):                                # my tiles of the same trace
            ctx.fill_synthetic_2d(trace, n, w, ncols, c0, 0x5EED0003, out_pitch=p.my_cols, out_off=p.local_col(k))
        traces.append(trace)
        bufs.append({"ext": ctx.empty(p.ext_elems()), "nodes": ctx.empty((2 * p.rows_per_rank - 1) * 4),
                     "recv": ctx.zeros(p.recv_elems()), "roots": ctx.empty((2 * world - 1) * 4)})
    for k in range(plans[0].n_rounds):
        for p in plans:
            phase_lde(p, Ops, traces[p.rank], bufs[p.rank], k)
        torch.cuda.synchronize()
        for p in plans:                                                                # the round's messages, by copies
            for msgs in exchange_messages(p, k):
                for (peer, s_off, s_cnt, r_off, r_cnt) in msgs:
                    # what p sends to peer lands where peer expects p's window
                    pr_off, pr_cnt = None, None
                    for pm in exchange_messages(plans[peer], k)[exchange_messages(p, k).index(msgs)]:
                        if pm[0] == p.rank:
                            pr_off, pr_cnt = pm[3], pm[4]
                    assert pr_cnt == s_cnt
                    bufs[peer]["recv"][pr_off:pr_off + pr_cnt] = bufs[p.rank]["ext"][s_off:s_off + s_cnt]
        for p in plans:
            phase_absorb(p, Ops, bufs[p.rank], k)
    roots = [phase_subtree(p, Ops, bufs[p.rank]).clone() for p in plans]
    for p in plans:                                                                    # all-gather by copies
        assert np.array_equal(ctx.to_host(bufs[p.rank]["nodes"][:p.rows_per_rank * 4]),
                              want_leaves[p.row0 * 4:(p.row0 + p.rows_per_rank) * 4]), p.rank
        for r in range(world):
            bufs[p.rank]["roots"][4 * r:4 * r + 4] = roots[r]
        got = phase_top(p, Ops, bufs[p.rank])
        assert np.array_equal(ctx.to_host(got), want_root), p.rank
    # query openings over the row-sharded tree: each emulated rank contributes the rows it owns, the all-reduce is their sum
    from shard import group_proofs_sharded

    class NoComm:
        @staticmethod
        def all_reduce(t):
            pass
    idx = sorted({0, 1, n_ext - 1, n_ext // 2, plans[0].rows_per_rank - 1, plans[0].rows_per_rank % n_ext, 777 % n_ext})
    total = sum(group_proofs_sharded(p, Ops, NoComm, bufs[p.rank], idx) for p in plans)
    levels = 11
    want = ctx.empty(len(idx) * (ncols + 4 * levels))
    ctx.merkle_group_proofs(want, nodes1, ext1, n_ext, ncols, np.array(idx, dtype=np.uint64))
    assert torch.equal(total.reshape(-1), want)


def test_pipelined_path_with_real_rccl_calls_on_one_rank(ctx):
    """lde_merkle_sharded itself on a one-rank RCCL communicator: no peers, so every window is read in place, but
    the all_gather and the stream ordering between the library's kernels and the collective are the real ones;
    run twice because bench.py reuses the buffers across steps."""
    import socket
    import torch
    import torch.distributed as dist
    from shard import ShardPlan, lde_merkle_sharded
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        n, ncols = 1 << 12, 150
        n_ext = 2 * n
        plan = ShardPlan(n=n, n_ext=n_ext, ncols=ncols, world=1, rank=0)
        assert plan.n_rounds == 5                       # 4 x 32 + 22 columns
        trace = ctx.empty(n * ncols)
        ctx.fill_synthetic_2d(trace, n, ncols, ncols, 0, 0x5EED0003)
        bufs = {"ext": ctx.empty(plan.ext_elems()), "nodes": ctx.empty((2 * n_ext - 1) * 4), "recv": ctx.empty(plan.recv_elems()),
                "roots": ctx.empty(4)}
        for _ in range(2):
            root = lde_merkle_sharded(plan, _device_ops(ctx), dist, trace, bufs, always_exchange=True)
            torch.cuda.synchronize()
        want_ext = glo.extend_pol(ctx.to_host(trace).reshape(n, ncols), n_ext, n, ncols)
        want = glo.merkletree(want_ext, ncols, n_ext)
        assert np.array_equal(ctx.to_host(root), want[-4:])
        assert np.array_equal(ctx.to_host(bufs["nodes"]), want)
        # openings and the FRI commit through the same orchestration and the real collectives (one rank)
        from shard import group_proofs_sharded, fri_commit_sharded
        Ops = _device_ops(ctx)
        idx = [0, 5, n_ext - 1, 4097]
        pr = ctx.to_host(group_proofs_sharded(plan, Ops, dist, bufs, idx)).reshape(len(idx), -1)
        for j, i in enumerate(idx):
            assert np.array_equal(pr[j], glo.merkle_group_proof(want, want_ext, n_ext, ncols, i)), i
        fb, steps = 13, [13, 9, 6, 3]
        hpol = glo.splitmix64(0xF00D, 3 << fb)
        t_dev, t_ref = glo.Transcript(), glo.Transcript()
        final, trees, _ = fri_commit_sharded(1, 0, Ops, dist, t_dev, ctx.to_device(hpol), steps, fb)
        cur, bits = hpol, fb
        for si, cb in enumerate(steps):                                   # the oracle's fold / transpose / tree, same transcript
            cur = glo.fri_fold(cur, bits, cb, fb, t_ref.get_field()).reshape(-1)
            if si < len(steps) - 1:
                nb = steps[si + 1]
                wn = glo.merkletree(glo.fri_transpose(cur, 1 << cb, nb), (1 << (cb - nb)) * 3, 1 << nb)
                assert np.array_equal(ctx.to_host(trees[si][0]), wn), si
                t_ref.put(wn[-4:])
            else:
                t_ref.put(cur)
            bits = cb
        assert np.array_equal(ctx.to_host(final)[:cur.size], cur) and t_dev.get_fields1() == t_ref.get_fields1()
        t = torch.arange(8, dtype=torch.int64, device="cuda")        # and one real collective on the same stream
        o = torch.zeros(8, dtype=torch.int64, device="cuda")
        dist.all_gather_into_tensor(o, t)
        assert torch.equal(o, t)
    finally:
        dist.destroy_process_group()


def test_element_wise_kernels_beyond_2pow32_elements(ctx):
    """A HIP launch of 2^32 or more threads is truncated silently; the element-wise kernels (synthetic fills, 2-D
    copy) must cover buffers larger than that -- the BASELINE trace has 5.6e9 elements.  Checks the tail, which a
    truncated launch leaves untouched."""
    import torch
    nrows, ncols = 1 << 22, 1030
    count = nrows * ncols                                      # 4.32e9 > 2^32
    buf = ctx.zeros(count)
    ctx.fill_synthetic(buf, count, 0x5EED0003)
    want = glo.splitmix64(0x5EED0003, 64)                      # elements 0..63
    assert np.array_equal(ctx.to_host(buf[:64]), want)
    idx = np.arange(count - 64, count, dtype=np.uint64) + np.uint64(1)
    with np.errstate(over="ignore"):
        z = np.uint64(0x5EED0003) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    tail = np.where(z >= np.uint64(glo.P), z - np.uint64(glo.P), z)
    assert np.array_equal(ctx.to_host(buf[count - 64:]), tail)
    # 2-D fill of the same stream in two column windows, then a strided copy: both > 2^32 elements
    a = ctx.zeros(nrows * ncols)
    ctx.fill_synthetic_2d(a, nrows, ncols, ncols, 0, 0x5EED0003)
    assert torch.equal(a, buf[:nrows * ncols])                # the 2-D fill of all columns is the flat stream
    del buf
    b = ctx.zeros(nrows * ncols)
    ctx.copy_2d(b, a, nrows, ncols, dst_pitch=ncols, src_pitch=ncols)
    assert torch.equal(a, b)


@pytest.mark.gpu
def test_persistent_ntt_pass_matches_oracle():
    """MI_NTT_PERSISTENT=1 selects the persistent double-buffered radix-256 pass (csrc/ntt.hip k_ntt_pass_pers; measured slower than the
    default and therefore off: profiles/r04_pmc_ntt_persistent.txt).  It stays bit-exact against the oracle (LDE, INTT / NTT at sizes
    where it is taken); the variable is read once per process, hence a child."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "ntt_persistent_check.py")], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, MI_NTT_PERSISTENT="1"))
    assert r.returncode == 0 and r.stdout.count("OK") == 6, r.stdout + r.stderr[-2000:]
