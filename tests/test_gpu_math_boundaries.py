"""GPU: the kernels' check-node / metric arithmetic on the operands where a table form could differ from the
reference's compare chain (CHK SCL_1024.c:343-374, T :352-359, PHI :481-502), through the test-only probe kernel
(include/polar_hip_testing.h: the same device functions the decoders inline), bit for bit against the oracle.

polar_lut.h picks a table cell from the operand's exponent and three mantissa bits and finishes with ONE compare;
AWGN data practically never lands on a threshold, on |a| == |b|, on a signed zero or outside the table, so the
decoders' parity tests exercise that arithmetic only statistically.  Here every such operand is constructed."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

THR = (0.196, 0.433, 0.71, 1.05, 1.508, 2.252, 4.5)


def _bits_equal(x, y):
    u = np.uint32 if x.dtype == np.float32 else np.uint64
    return np.array_equal(x.view(u), y.view(u))


def _operands(dtype):
    """(a, b) pairs: |a + b| or |a - b| exactly on / one ulp beside every threshold, equal magnitudes, signed zeros,
    denormals, beyond the table, huge; plus a broad random set."""
    R = dtype
    rng = np.random.default_rng(1234)
    fi = np.finfo(R)
    A, B = [], []
    targets = []
    for t in THR:
        t = R(t)
        targets += [np.nextafter(t, R(0)), t, np.nextafter(t, R(np.inf))]
    # cell boundaries of the table (8 per binade over [0.125, 8)) and their neighbours
    for e in range(-4, 4):
        for m in range(8):
            v = R(2.0 ** e * (1 + m / 8))
            targets += [np.nextafter(v, R(0)), v, np.nextafter(v, R(np.inf))]
    targets = np.array(targets, dtype=R)
    for v in targets:
        # a + b == v exactly, and a - b == v exactly, with b on a coarse grid so that the sum is exact
        for b in (R(0), R(0.0625), R(0.5), R(1.75), R(-0.375), R(-3.0), R(7.5), v, R(v / 2)):
            for sgn in (R(1), R(-1)):
                a = R(v - b)
                if R(a + b) == v:
                    A += [sgn * a, sgn * a]; B += [sgn * b, sgn * b]
                a2 = R(v + b)
                if R(a2 - b) == v:
                    A += [sgn * a2]; B += [sgn * b]
                    A += [sgn * b]; B += [sgn * a2]
    special = [R(0), R(-0.0), fi.smallest_subnormal, -fi.smallest_subnormal, fi.tiny, R(fi.tiny / 2), R(1e-30),
               R(0.098), R(0.2165), R(1.0), R(-1.0), R(2.25), R(4.0), R(8.0), np.nextafter(R(8), R(0)), R(9.0), R(100.0),
               R(999.0), R(-999.0), R(1e30), R(fi.max / 4), R(-fi.max / 4)] + [R(t) for t in THR] + [R(-t) for t in THR]
    for x in special:
        for y in special:
            A.append(x); B.append(y)
    # equal magnitudes over many scales
    mags = (2.0 ** rng.uniform(-20, 6, 4000)).astype(R)
    for s1, s2 in ((1, 1), (1, -1), (-1, 1), (-1, -1)):
        A += list(s1 * mags); B += list(s2 * mags)
    # broad random: normal LLR-like, log-uniform magnitudes, and a fine sweep across every threshold
    n = 400000
    A += list((rng.standard_normal(n) * 4).astype(R)); B += list((rng.standard_normal(n) * 4).astype(R))
    A += list((rng.choice([-1, 1], n) * 2.0 ** rng.uniform(-30, 8, n)).astype(R))
    B += list((rng.choice([-1, 1], n) * 2.0 ** rng.uniform(-30, 8, n)).astype(R))
    for t in THR:
        k = np.arange(-2000, 2001)
        sweep = np.full(k.size, R(t), dtype=R)
        # 2000 ulps either side
        u = np.uint32 if R == np.float32 else np.uint64
        sweep = (sweep.view(u).astype(np.int64) + k).astype(u).view(R)
        half = (sweep / R(2)).astype(R)
        A += list(half); B += list((sweep - half).astype(R))           # a + b ~ t
        A += list(sweep + R(1)); B += list(np.full(k.size, R(1)))      # a - b ~ t
    return np.array(A, dtype=R), np.array(B, dtype=R)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_check_node_forms_bit_identical_to_oracle(dtype, oracle):
    from polardecoding_amd import testing as T
    a, b = _operands(dtype)
    assert a.size > 800000
    ref = oracle.math(0, a, b, dtype)
    assert np.isfinite(ref).all()
    for op in (T.OP_CHK, T.OP_CHK_LUT, T.OP_CHK_LUT1, T.OP_CHK_CNT, T.OP_CHK_IDX, T.OP_CHK_TAB):
        got = T.math(op, a, b, dtype)
        # -0.0 + (+0.0): the sign of a zero result is the only place the forms may legitimately be compared by value
        # -- they are not allowed to differ there either
        u = np.uint32 if dtype == np.float32 else np.uint64
        bad = np.flatnonzero(got.view(u) != ref.view(u))
        assert bad.size == 0, f"op {op}: {bad.size} differ, e.g. a={a[bad[:3]]!r} b={b[bad[:3]]!r} got={got[bad[:3]]!r} want={ref[bad[:3]]!r}"


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_staircase_and_metric_increment_bit_identical_to_oracle(dtype, oracle):
    from polardecoding_amd import testing as T
    a, b = _operands(dtype)
    x = np.concatenate([a, b, a + b, a - b]).astype(dtype)
    x = x[np.isfinite(x)]
    assert _bits_equal(T.math(T.OP_TABV, x, 0, dtype), oracle.math(1, x, 0, dtype))
    for u in (0, 1):
        ref = oracle.math(2, x, u, dtype)
        assert _bits_equal(T.math(T.OP_PHI, x, u, dtype), ref)
        assert _bits_equal(T.math(T.OP_PHI_LUT, x, u, dtype), ref)


def test_oracle_staircase_is_the_published_table(oracle):
    """the oracle's T itself against the literal thresholds and levels of SCL_1024.c:352-359"""
    lv = (0.65, 0.55, 0.45, 0.35, 0.25, 0.15, 0.05, 0.0)
    for i, t in enumerate(THR):
        below, at = np.nextafter(t, 0.0), t
        assert oracle.math(1, [below], 0)[0] == lv[i] and oracle.math(1, [at], 0)[0] == lv[i + 1]
    assert oracle.math(1, [0.0], 0)[0] == 0.65 and oracle.math(1, [1e9], 0)[0] == 0.0
