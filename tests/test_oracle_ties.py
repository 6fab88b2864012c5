"""Median ties in the list decoders (SCL_1024.c:619-633, "Oops!"): what the reference does, pinned.

tests/golden/ties_<program>.npz were produced by the COMPILED reference (tests/golden/make_golden.py: make_ties) on
frames built to collide path metrics.  What these tests establish, without a GPU:

  * oracle/polar_oracle_literal.c -- the reference's node records as a persistent object -- reproduces the compiled
    reference on every such frame: decisions, chosen metric and the number of "Oops!" / "Wrong propagation order!" /
    "Error!" lines, both from cleared records and frame after frame without clearing; and it predicts the frames
    from which the reference never returns (Partition() loops for ever on three equal candidates, :518-544).
  * the build's tie rule (DESIGN.md "Median ties"; oracle po_scl_decode_*, and the kernels through it): the tie flag
    is raised on exactly the frames where the reference prints "Oops!" (or does not return), and decisions and
    metric EQUAL the reference's on every tied frame of the fixtures from which the reference returns.
  * on those frames the reference's result does not depend on what earlier frames left in its records (probed with
    arbitrary leftovers), so comparing a stateless decoder with it is meaningful there.
"""
import os

import numpy as np
import pytest

from oracle import oracle_py as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PROGRAMS = ("SCL_128", "CASCL_128", "SCL_1024", "CASCL_1024_L8")


def load(name):
    z = np.load(os.path.join(GOLD, f"ties_{name}.npz"))
    N, K, taps, algo, L = O.REF_PROGRAMS[name]
    code = O.Code(N, K, taps)
    sigma = float(z["sigma"])
    llr = np.stack([O.llr_from_y(y, sigma) for y in z["y"]])
    return z, code, algo, L, sigma, llr


@pytest.mark.parametrize("name", PROGRAMS)
def test_fixture_has_every_kind(name):
    z = np.load(os.path.join(GOLD, f"ties_{name}.npz"))
    kinds = z["kind"]
    assert all((kinds == k).sum() >= 2 for k in range(4))
    assert ((kinds == 1) == (z["returns"] == 0)).all()
    assert (z["diag"][kinds == 0, 0] > 0).all(), "tie frames: the reference printed Oops!"
    assert (z["diag"][kinds == 0, 1] > 0).all(), "tie frames: and then Wrong propagation order!"
    assert (z["diag"][kinds >= 2] == 0).all() and (z["diag"][:, 2] == 0).all()


@pytest.mark.parametrize("name", PROGRAMS)
def test_literal_model_reproduces_the_compiled_reference(name):
    z, code, algo, L, sigma, llr = load(name)
    lit = O.Literal(code, L, crc=(algo == "CASCL"))
    for f in range(len(llr)):      # (a) records cleared before every frame
        lit.reset()
        uh, pm, d = lit.decode(llr[f])
        if not z["returns"][f]:
            assert lit.last_rc == -5, f"frame {f}: the reference does not return, the model must say so"
            continue
        assert lit.last_rc == 0
        assert np.array_equal(uh, z["u_hat"][f]) and pm == z["pm"][f] and d == z["diag"][f].tolist(), f"frame {f}"
    lit.reset()
    for f in range(len(llr)):      # (b) frame after frame like main(), nothing cleared in between
        if not z["returns"][f]:
            continue
        uh, pm, d = lit.decode(llr[f])
        assert np.array_equal(uh, z["u_hat_seq"][f]) and pm == z["pm_seq"][f] and d == z["diag_seq"][f].tolist(), f"frame {f}"


@pytest.mark.parametrize("name", PROGRAMS)
def test_reference_result_on_these_frames_does_not_depend_on_history(name):
    z, code, algo, L, sigma, llr = load(name)
    ret = z["returns"] == 1
    assert np.array_equal(z["u_hat"][ret], z["u_hat_seq"][ret]) and np.array_equal(z["pm"][ret], z["pm_seq"][ret])
    lit = O.Literal(code, L, crc=(algo == "CASCL"))
    for f in np.flatnonzero(ret)[: (8 if code.N == 128 else 3)]:
        for seed in (1, 2):
            lit.reset()
            lit.poison(seed)
            uh, pm, _ = lit.decode(llr[f])
            assert lit.last_rc == 0 and np.array_equal(uh, z["u_hat"][f]) and pm == z["pm"][f], f"frame {f}"


@pytest.mark.parametrize("name", PROGRAMS)
def test_tie_rule_of_the_build_against_the_reference(name):
    z, code, algo, L, sigma, llr = load(name)
    st = np.zeros((len(llr), 2), dtype=np.int32)
    uh, pm, ties = O.decode(code, llr, algo, L=L, stats=st)
    kinds, ret = z["kind"], z["returns"] == 1
    # the flag: raised exactly where the reference prints "Oops!" or hangs in its sort
    assert np.array_equal(ties > 0, (z["diag"][:, 0] > 0) | ~ret)
    # decisions and metric: identical wherever the reference returns -- tied or not
    assert np.array_equal(uh[ret], z["u_hat"][ret]) and np.array_equal(pm[ret], z["pm"][ret])
    # kind 2: no tie, but the 32-bit keys the kernels rank with are not enough; kind 3: plain frames
    assert (st[kinds == 2, 0] > 0).all() and (ties[kinds == 2] == 0).all()
    assert (st[kinds == 3, 0] == 0).all()


@pytest.mark.parametrize("name", PROGRAMS)
def test_compiled_reference_still_gives_the_fixture(name):
    if not O.ref_available(name):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    z, code, algo, L, sigma, llr = load(name)
    ref = O.Ref(name)
    for f in np.flatnonzero(z["returns"] == 1)[:: (2 if code.N == 128 else 3)]:
        ref.reset_state()
        ref.diag()
        uh, pm = ref.decode(z["y"][f], sigma)
        assert np.array_equal(uh, z["u_hat"][f]) and pm == z["pm"][f] and ref.diag() == z["diag"][f].tolist()


def test_literal_model_equals_the_restatement_without_ties():
    """On ordinary AWGN frames (no tie) the two statements of the reference are the same function."""
    code = O.Code(128, 64, O.CRC6_TAPS)
    sim = O.Sim(5)
    sig = O.sigma_from_db(1.5)
    lit = O.Literal(code, 8, crc=True)
    for _ in range(40):
        _, y = sim.frame(code, sig)
        llr = O.llr_from_y(y, sig)
        a, pa, d = lit.decode(llr)
        b, pb, t = O.decode(code, llr, "CASCL", L=8)
        assert d == [0, 0, 0] and t == 0 and np.array_equal(a, b) and pa == pb
