"""What genProof computes between the base-domain constraint steps: the plookup columns h1 / h2 (Polinomial::calculateH1H2*,
polinomial.hpp:303-584, called at starks.cpp:92-128) and the grand products z (Polinomial::calculateZ, polinomial.hpp:586-607,
starks.cpp:174-187).

CPU part: the oracle's restatements against the definitions written out in Python.  GPU part: mi_calculate_h1h2_dev /
mi_calculate_z_dev against the oracle, bit for bit, on strided views of one polynomial area, over the shapes the domain has (runs
of equal values, duplicate table rows, one value taking every lookup, values absent from the table, 1 .. 2^20 rows), and at 2^23
rows through properties that need no oracle run (multiset of h1 u h2 == multiset of f u t, order follows t, the recurrence of z)."""
import numpy as np
import pytest
import glo

P = glo.P


def make_lookup(rng, n, dim, kind):
    """(f, t) as (n, dim) arrays.  kinds: 'runs' small alphabet (long runs, duplicate rows in t), 'distinct' all rows of t differ,
    'one' every row of f is the same value, 'padded' t = a short table repeated to the end with its last row."""
    if kind == "distinct":
        t = glo.rand_fe(rng, (n, dim))
        t[:, 0] = (np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) % np.uint64(P)  # distinct first words
        f = t[rng.integers(0, n, size=n)]
    elif kind == "runs":
        alphabet = glo.rand_fe(rng, (max(1, min(n, 37)), dim))
        t = alphabet[rng.integers(0, alphabet.shape[0], size=n)]
        f = t[np.sort(rng.integers(0, n, size=n))]
    elif kind == "one":
        t = glo.rand_fe(rng, (n, dim))
        f = np.repeat(t[rng.integers(0, n)][None, :], n, axis=0)
    elif kind == "padded":
        m = max(1, n // 5)
        t = glo.rand_fe(rng, (n, dim))
        t[m:] = t[m - 1]
        f = t[rng.integers(0, n, size=n)]
        f[rng.integers(0, n, size=n // 2)] = t[0]
    else:
        raise ValueError(kind)
    return np.ascontiguousarray(f, dtype=np.uint64), np.ascontiguousarray(t, dtype=np.uint64)


def lay_out(rng, n, cols, views):
    """One (n x cols) row-major area with random content; views = {name: (column, array (n, dim))} written into it.  Returns the flat
    area."""
    area = glo.rand_fe(rng, (n, cols))
    for col, arr in views.values():
        area[:, col:col + arr.shape[1]] = arr
    return np.ascontiguousarray(area.reshape(-1))


def h1h2_definition(f, t):
    """polinomial.hpp:303-347 with Python containers."""
    n = t.shape[0]
    last = {}
    for i in range(n):
        last[tuple(int(v) for v in t[i])] = i
    counter = [1] * n
    for i in range(n):
        k = tuple(int(v) for v in f[i])
        if k not in last:
            return i + 1, None, None
        counter[last[k]] += 1
    s = [t[i] for i in range(n) for _ in range(counter[i])]
    return 0, np.array(s[0::2], dtype=np.uint64), np.array(s[1::2], dtype=np.uint64)


def e3(a):
    return [int(v) for v in a]


def e3_mul(a, b):
    return e3(glo.e3_mul(np.array(a, dtype=np.uint64), np.array(b, dtype=np.uint64)))


# ------------------------------------------------------------------ the oracle against the definitions (CPU)
@pytest.mark.parametrize("dim", [1, 3])
@pytest.mark.parametrize("kind", ["runs", "distinct", "one", "padded"])
def test_oracle_h1h2_is_the_definition(dim, kind):
    rng = np.random.default_rng(100 + dim)
    for n in (1, 2, 7, 64, 301):
        f, t = make_lookup(rng, n, dim, kind)
        cols = 4 * dim + 3
        area = lay_out(rng, n, cols, {"t": (0, t), "f": (dim + 1, f)})
        c1, c2 = 2 * dim + 2, 3 * dim + 2
        bad = glo.calculate_h1h2(area, c1, cols, c2, cols, dim + 1, cols, 0, cols, dim, n)
        wbad, h1, h2 = h1h2_definition(f, t)
        assert bad == wbad == 0
        a = area.reshape(n, cols)
        assert np.array_equal(a[:, c1:c1 + dim], h1) and np.array_equal(a[:, c2:c2 + dim], h2)


def test_oracle_h1h2_reports_the_first_missing_row():
    rng = np.random.default_rng(5)
    n = 50
    f, t = make_lookup(rng, n, 1, "runs")
    f[17, 0] = np.uint64(P - 1)
    f[30, 0] = np.uint64(P - 2)
    area = lay_out(rng, n, 4, {"t": (0, t), "f": (1, f)})
    assert glo.calculate_h1h2(area, 2, 4, 3, 4, 1, 4, 0, 4, 1, n) == 18 == h1h2_definition(f, t)[0]


def test_oracle_z_is_the_recurrence():
    rng = np.random.default_rng(6)
    n = 40
    num, den = glo.rand_fe(rng, (n, 3)), glo.rand_fe(rng, (n, 3))
    area = lay_out(rng, n, 11, {"num": (0, num), "den": (4, den)})
    closes = glo.calculate_z(area, 8, 11, 0, 11, 4, 11, n)
    z = area.reshape(n, 11)[:, 8:11]
    assert e3(z[0]) == [1, 0, 0] and not closes
    for i in range(n - 1):
        assert e3_mul(z[i + 1], den[i]) == e3_mul(z[i], num[i])  # polinomial.hpp:597-601
    # a product that closes: den = a permutation of num
    den2 = num[rng.permutation(n)]
    area = lay_out(rng, n, 11, {"num": (0, num), "den": (4, den2)})
    assert glo.calculate_z(area, 8, 11, 0, 11, 4, 11, n) == 1


# ------------------------------------------------------------------ the device path against the oracle
@pytest.fixture(scope="module")
def ctx():
    import mi_stark
    c = mi_stark.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dim", [1, 3])
@pytest.mark.parametrize("kind", ["runs", "distinct", "one", "padded"])
def test_h1h2_matches_oracle(ctx, dim, kind):
    rng = np.random.default_rng(200 + dim)
    for n in (1, 2, 5, 63, 64, 65, 255, 256, 257, 4095, 4097, (1 << 16) + 3, 1 << 18):
        f, t = make_lookup(rng, n, dim, kind)
        cols = 4 * dim + 3
        ct, cf, c1, c2 = 0, dim + 1, 2 * dim + 2, 3 * dim + 2
        area = lay_out(rng, n, cols, {"t": (ct, t), "f": (cf, f)})
        d = ctx.to_device(area)
        ctx.calculate_h1h2(d[c1:], cols, d[c2:], cols, d[cf:], cols, d[ct:], cols, dim, n)
        assert glo.calculate_h1h2(area, c1, cols, c2, cols, cf, cols, ct, cols, dim, n) == 0
        got = ctx.to_host(d)
        assert np.array_equal(got, area), (n, kind, dim)  # h1, h2 equal and nothing else in the area touched


@pytest.mark.gpu
def test_h1h2_views_with_their_own_strides_and_constant_table(ctx):
    """f in one section, t in another array (a constant polynomial), h1 / h2 in a third: four different strides."""
    rng = np.random.default_rng(11)
    n = 3000
    f, t = make_lookup(rng, n, 1, "runs")
    sec_f, sec_t, sec_h = lay_out(rng, n, 5, {"f": (3, f)}), lay_out(rng, n, 2, {"t": (1, t)}), lay_out(rng, n, 7, {})
    df, dt, dh = ctx.to_device(sec_f), ctx.to_device(sec_t), ctx.to_device(sec_h)
    ctx.calculate_h1h2(dh[2:], 7, dh[6:], 7, df[3:], 5, dt[1:], 2, 1, n)
    _, h1, h2 = h1h2_definition(f, t)
    want = sec_h.reshape(n, 7).copy()
    want[:, 2:3], want[:, 6:7] = h1, h2
    assert np.array_equal(ctx.to_host(dh).reshape(n, 7), want)


@pytest.mark.gpu
@pytest.mark.parametrize("dim", [1, 3])
def test_h1h2_missing_value_fails_like_the_reference(ctx, dim):
    import mi_stark
    rng = np.random.default_rng(12)
    n = 70000
    f, t = make_lookup(rng, n, dim, "runs")
    f[40000, dim - 1] ^= np.uint64(1)  # (a value one bit away from a table row: same first words for dim 3)
    f[66000, 0] = np.uint64(P - 5)
    cols = 4 * dim
    area = lay_out(rng, n, cols, {"t": (0, t), "f": (dim, f)})
    d = ctx.to_device(area)
    with pytest.raises(mi_stark.MiStarkError, match="number not included: w=40000"):
        ctx.calculate_h1h2(d[2 * dim:], cols, d[3 * dim:], cols, d[dim:], cols, d, cols, dim, n)
    assert glo.calculate_h1h2(area.copy(), 2 * dim, cols, 3 * dim, cols, dim, cols, 0, cols, dim, n) == 40001
    assert np.array_equal(ctx.to_host(d), area)  # h1 / h2 untouched


@pytest.mark.gpu
def test_z_matches_oracle(ctx):
    rng = np.random.default_rng(13)
    for n in (1, 2, 7, 8, 9, 2047, 2048, 2049, 70001, 1 << 20):
        num, den = glo.rand_fe(rng, (n, 3)), glo.rand_fe(rng, (n, 3))
        if n > 8:
            den[5] = 0  # inverse of zero is zero on both sides
            den[6] = 0  # (two in one thread's run of rows: they share one batched inversion)
        if n > 2000:
            den[n - 1] = 0
            den[1024] = 0
        cols = 11
        area = lay_out(rng, n, cols, {"num": (0, num), "den": (4, den)})
        d = ctx.to_device(area)
        closes = ctx.calculate_z(d[8:], cols, d, cols, d[4:], cols, n)
        want_closes = glo.calculate_z(area, 8, cols, 0, cols, 4, cols, n)
        assert np.array_equal(ctx.to_host(d), area), n
        assert closes == bool(want_closes)


@pytest.mark.gpu
def test_z_batch_matches_oracle(ctx):
    """mi_calculate_z_batch_dev: the products of a stage in one pass over a shared area (numerators / denominators in one wide section,
    z columns in another, like tmpExp_n and cm3_n) = the oracle's product by product; more products than one launch takes (32)."""
    rng = np.random.default_rng(15)
    for n, k in ((1, 3), (9, 2), (2049, 5), (40000, 35), (1 << 16, 13)):
        wt, wz = 6 * k + 1, 3 * k + 2
        area = glo.rand_fe(rng, (n * (wt + wz),))
        t_off, z_off = 0, n * wt
        src = area[:n * wt].reshape(n, wt)
        closing = k - 1                       # the last product's numerators are a permutation of its denominators: it closes
        if n > 8:
            src[5, 3:6] = 0                   # zero denominators in product 0 (inverse of zero is zero on both sides)
            src[n - 1, 3:6] = 0
            src[:, 6 * closing:6 * closing + 3] = src[rng.permutation(n), 6 * closing + 3:6 * closing + 6]
        d = ctx.to_device(area)
        prods = [(d[z_off + 3 * i:], wz, d[t_off + 6 * i:], wt, d[t_off + 6 * i + 3:], wt) for i in range(k)]
        closes = ctx.calculate_z_batch(prods, n)
        want = [bool(glo.calculate_z(area, z_off + 3 * i, wz, t_off + 6 * i, wt, t_off + 6 * i + 3, wt, n)) for i in range(k)]
        assert np.array_equal(ctx.to_host(d), area), (n, k)
        assert closes == want, (n, k)
        if n > 8:
            assert closes[closing] is True and not any(closes[:closing])
    assert ctx.calculate_z_batch([], 10) == []


@pytest.mark.gpu
def test_z_closes_when_the_numerators_are_a_permutation_of_the_denominators(ctx):
    rng = np.random.default_rng(14)
    n = 100000
    num = glo.rand_fe(rng, (n, 3))
    den = num[rng.permutation(n)]
    dn, dd, dz = ctx.to_device(num.reshape(-1)), ctx.to_device(den.reshape(-1)), ctx.empty(n * 3)
    assert ctx.calculate_z(dz, 3, dn, 3, dd, 3, n) is True
    den[123, 1] ^= np.uint64(1)
    assert ctx.calculate_z(dz, 3, dn, 3, ctx.to_device(den.reshape(-1)), 3, n) is False


@pytest.mark.gpu
def test_full_size_properties(ctx):
    """2^23 rows (the zkEVM's N).  h1 / h2: the multiset of h1 u h2 is the multiset of f u t, consecutive values follow the order of
    t, and 4096 sampled rows match the definition through the counts; z: the recurrence on sampled rows, z[0] = 1."""
    torch = ctx.torch
    n = 1 << 23
    g = torch.Generator(device=ctx.device)
    g.manual_seed(77)
    # t: 2^20 distinct values each repeated 8 times in a row; f: random rows of t, half of them one heavy value
    tvals = torch.randint(0, 1 << 62, (n >> 3,), generator=g, device=ctx.device, dtype=torch.int64)
    tvals = torch.unique(tvals)
    assert tvals.numel() == n >> 3  # (a collision among 2^20 62-bit draws would only weaken the test; it does not happen with this seed)
    tvals = tvals[torch.randperm(n >> 3, generator=g, device=ctx.device)]
    t = tvals.repeat_interleave(8)
    f = t[torch.randint(0, n, (n,), generator=g, device=ctx.device)]
    f[: n // 2] = t[12345]
    area = ctx.zeros(n * 4)
    area[0::4], area[1::4] = t, f
    ctx.calculate_h1h2(area[2:], 4, area[3:], 4, area[1:], 4, area, 4, 1, n)
    h1, h2 = area[2::4], area[3::4]
    s = torch.stack([h1, h2], dim=1).reshape(-1)
    assert torch.equal(torch.sort(s).values, torch.sort(torch.cat([f, t])).values)
    # order follows t: the position in t (in units of its runs of 8) of each value of s never decreases
    order = torch.argsort(tvals)
    pos = order[torch.searchsorted(tvals[order], s)]
    assert bool((pos[1:] >= pos[:-1]).all())
    del s, pos
    # z: random numerators / denominators, the recurrence on sampled rows
    cols = 9
    zarea = torch.randint(0, 1 << 62, (n * cols,), generator=g, device=ctx.device, dtype=torch.int64)
    ctx.calculate_z(zarea[6:], cols, zarea, cols, zarea[3:], cols, n)
    rows = np.unique(np.concatenate([[0, 1, 2046, 2047, 2048, n - 2], np.random.default_rng(3).integers(0, n - 1, size=500)]))
    idx = torch.from_numpy(rows).to(ctx.device)
    a = ctx.to_host(zarea.reshape(n, cols)[idx]), ctx.to_host(zarea.reshape(n, cols)[idx + 1])
    assert e3(ctx.to_host(zarea[6:9])) == [1, 0, 0]
    for k in range(rows.size):
        num, den, z0, z1 = a[0][k][0:3], a[0][k][3:6], a[0][k][6:9], a[1][k][6:9]
        assert e3_mul(z1, den) == e3_mul(z0, num), rows[k]


@pytest.mark.gpu
def test_zero_rows_are_a_no_op(ctx):
    d = ctx.to_device(np.arange(12, dtype=np.uint64))
    ctx.calculate_h1h2(d[2:], 4, d[3:], 4, d[1:], 4, d, 4, 1, 0)
    assert ctx.calculate_z(d[6:], 9, d, 9, d[3:], 9, 0) is True
    assert np.array_equal(ctx.to_host(d), np.arange(12, dtype=np.uint64))
