#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference decoders.

Run in the build container (needs /root/reference and `make -C oracle ref`):

    python tests/golden/make_golden.py

For every reference simulator (SC_128 ... CASCL_1024_L8) it draws seeded frames with the
reference's own transmit chain (restated in oracle/polar_oracle.c and pinned by the published run
counts), calls the reference's decode function compiled from /root/reference (oracle/_ref/), and
stores inputs + outputs:

    <name>.npz : sigma[F], y[F][N] f64, u[F][N] u8 (sent), u_hat[F][N] u8 (reference decision),
                 pm[F] f64 (metric of the chosen path; 0 for SC/BP)

BPr_128_main_seed7.txt is the first 43 lines (three Eb/N0 points) of `oracle/_ref/BPr_128_main 7`, the compiled
BPr_128.c main() with its time() seed pinned to 7 (4 minutes of CPU for all seven points).
BP_128_main_seed7.txt is everything `oracle/_ref/BP_128_main 7` prints (BP_128.c's main(), 100 iterations, 200 block
errors per point, 1.0 .. 4.0 dB: 144 640 frames, 4.5 minutes of CPU) -- what tests/test_gpu_dropin.py expects from the
same main() with its decode call bound to the library.

ties_<name>.npz (SCL_128, CASCL_128, SCL_1024, CASCL_1024_L8): frames on which the list decoders meet a MEDIAN TIE
(SCL_1024.c:619-633), which AWGN inputs in double precision practically never produce: observations on a grid of
1/64 with std = 0.5, so that the channel LLR 2*y/std/std = 8*y is exact and path metrics collide.  Per frame the
compiled reference's u_hat, chosen-path metric and the number of "Oops!" / "Wrong propagation order!" / "Error!"
lines it printed, decoded (a) from freshly calloc()ed node records and (b) one after the other without clearing
anything, like main() does; frames from which the reference NEVER RETURNS (three equal candidates send its
Partition(), SCL_1024.c:518-544, into an endless loop; found with a time limit in a child process) are kept with
returns = 0; plus frames without a tie on which the high 32 bits of the candidates do not decide the ranking.
make_ties() documents the fields.

It also parses the reference's published fixed-seed logs (myResult_*.zip: data, not code) into
published_runs.json: the known-answer run counts used by tests/test_oracle_kat.py.
Only data is written here; no reference source text.
"""
import json
import os
import re
import sys
import zipfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from oracle import oracle_py as O  # noqa: E402

REF = "/root/reference"
SNRS = (1.0, 2.0, 3.0)


def make_vectors(only=None):
    for name, (N, K, taps, algo, L) in O.REF_PROGRAMS.items():
        if only and name not in only:
            continue
        code = O.Code(N, K, taps, systematic=name in O.SYSTEMATIC_PROGRAMS)
        if code.systematic:
            check_systematic_rows(name, code)
        ref = O.Ref(name)
        per = 8 if N == 1024 else 32
        if algo == "BP" and N == 1024:
            per = 4
        sim = O.Sim(20261004 + N + len(name))
        sig, ys, us, uhs, pms = [], [], [], [], []
        for db in SNRS:
            s = O.sigma_from_db(db)
            for _ in range(per):
                u, y = sim.frame(code, s)
                uh, pm = ref.decode(y, s)
                sig.append(s); ys.append(y); us.append(u); uhs.append(uh); pms.append(pm)
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"),
                            sigma=np.array(sig), y=np.array(ys), u=np.array(us, dtype=np.uint8),
                            u_hat=np.array(uhs, dtype=np.uint8), pm=np.array(pms))
        nerr = sum(int((a != b).any()) for a, b in zip(us, uhs))
        print(f"{name}: {len(sig)} frames, {nerr} in error")


def check_systematic_rows(name, code):
    """The systematic program carries the redundant part of its generator matrix as a K x r literal
    (CASCL_1024_sys.c:48-561).  The oracle derives the rows as D^(r+i) mod g; compare all of them with the
    literal here, where the reference can be read (nothing of it is stored)."""
    text = open(os.path.join(REF, name + ".c")).read()
    body = text[text.index("Gc[K][r]"):]
    body = body[body.index("{") + 1:body.index("};")]
    rows = [[int(t) for t in re.findall(r"[01]", row)] for row in re.findall(r"\{([^{}]*)\}", body)]
    assert len(rows) == code.K and all(len(r) == code.r for r in rows), (len(rows), code.K)
    import ctypes as C
    for i, row in enumerate(rows):
        v = np.zeros(code.K, dtype=np.int32)
        v[i] = 1
        w = np.zeros(code.A, dtype=np.int32)
        O.lib().po_crc_encode(code._h, v.ctypes.data_as(C.POINTER(C.c_int)), w.ctypes.data_as(C.POINTER(C.c_int)))
        assert w[:code.r].tolist() == row, f"generator row {i} differs from the reference literal"
        assert w[code.r + i] == 1 and w[code.r:].sum() == 1
    print(f"{name}: {len(rows)} generator rows equal D^(r+i) mod g")


def make_bpr():
    """BPr_128.c: BP (iterMax 90) with per-stage read-outs after 3, 6, 10, 20, 40, 80 iterations.
    BPr_128.npz: sigma[F], y[F][N], u[F][N], u_hat[F][N], iters, checkpoints[6], E[F][6][n+1] (per frame)."""
    name = "BPr_128"
    ref = O.Ref(name)
    iters, cp = ref.readout_iters()
    code = O.Code(128, 64)
    sim = O.Sim(20261004 + 128 + len(name))
    sig, ys, us, uhs, Es = [], [], [], [], []
    ref.readout(reset=True)
    for db in SNRS:
        s = O.sigma_from_db(db)
        for _ in range(16):
            u, y = sim.frame(code, s)
            uh = ref.decode_u(y, s, u)
            sig.append(s); ys.append(y); us.append(u); uhs.append(uh); Es.append(ref.readout(reset=True))
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), sigma=np.array(sig), y=np.array(ys),
                        u=np.array(us, dtype=np.uint8), u_hat=np.array(uhs, dtype=np.uint8), iters=iters,
                        checkpoints=np.array(cp), E=np.array(Es))
    print(f"{name}: {len(sig)} frames, iterMax {iters}, checkpoints {cp}, E total\n{np.array(Es).sum(axis=0)}")


TIE_PROGRAMS = {"SCL_128": (16, 3, 8, 4), "CASCL_128": (16, 3, 8, 4), "SCL_1024": (6, 2, 4, 2),
                "CASCL_1024_L8": (6, 2, 4, 2)}   # frames wanted: tie, never-returns, near-tie, clean
TIE_GRID = 64.0
TIE_SIGMA = 0.5


def make_main_outputs():
    """stdout of unmodified reference programs with the seed pinned (text the programs printed: data, not source)."""
    import subprocess
    for exe, seed, fname in (("BP_128_main", 7, "BP_128_main_seed7.txt"),):
        out = subprocess.run([os.path.join(REPO, "oracle", "_ref", exe), str(seed)], capture_output=True, text=True,
                             timeout=3600, check=True).stdout
        with open(os.path.join(HERE, fname), "w") as f:
            f.write(out)
        print(fname, len(out.splitlines()), "lines")


def _probe_returns(name, y, limit=20.0):
    """Does the compiled reference return from this frame (fresh records)?  Run in a child with a time limit."""
    import subprocess
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".npy") as f:
        np.save(f.name, y)
        code = (f"import sys; sys.path.insert(0, {REPO!r}); import numpy as np; from oracle import oracle_py as O; "
                f"r = O.Ref({name!r}); r.reset_state(); r.decode(np.load({f.name!r}), {TIE_SIGMA})")
        try:
            subprocess.run([sys.executable, "-c", code], timeout=limit, check=True)
            return True
        except subprocess.TimeoutExpired:
            return False


def make_ties(only=None):
    """ties_<name>.npz:
      sigma (scalar), y[F][N], kind[F] (0 tie, 1 reference never returns, 2 no tie but the candidates' high words do not
      decide the ranking, 3 neither), returns[F],
      u_hat[F][N], pm[F], diag[F][3]                : compiled reference, node records cleared before the frame
      u_hat_seq[F][N], pm_seq[F], diag_seq[F][3]    : compiled reference, the frames with returns = 1 decoded in file
                                                      order without clearing anything (records cleared once, before the first)
    (rows of frames with returns = 0 are zero)."""
    for name, want in TIE_PROGRAMS.items():
        if only and name not in only and "ties" not in only:
            continue
        N, K, taps, algo, L = O.REF_PROGRAMS[name]
        code = O.Code(N, K, taps)
        ref = O.Ref(name)
        lit = O.Literal(code, L, crc=(algo == "CASCL"))
        sim = O.Sim(977 + N + len(name))
        got = {0: [], 1: [], 2: [], 3: []}
        tries = 0
        while any(len(got[k]) < want[k] for k in range(4)) and tries < 20000:
            tries += 1
            _, y0 = sim.frame(code, O.sigma_from_db(1.0 if tries % 2 else 2.0))
            y = np.round(y0 * TIE_GRID) / TIE_GRID
            llr = O.llr_from_y(y, TIE_SIGMA)
            assert np.array_equal(llr, 8 * y)
            lit.reset()
            _, _, d = lit.decode(llr)
            st = np.zeros((1, 2), dtype=np.int32)
            _, _, t = O.decode(code, llr, algo, L=L, stats=st)
            if lit.last_rc == -5:
                kind = 1
            elif lit.last_rc != 0:
                continue
            elif d[0] > 0:
                kind = 0
            elif st[0, 0] > 0:
                kind = 2
            else:
                kind = 3
            if len(got[kind]) < want[kind]:
                got[kind].append(y)
        ys = np.array([y for k in range(4) for y in got[k]])
        kinds = np.array([k for k in range(4) for _ in got[k]], dtype=np.int8)
        F = len(ys)
        ret = np.zeros(F, dtype=np.int8)
        uh = np.zeros((F, N), dtype=np.uint8); pm = np.zeros(F); dg = np.zeros((F, 3), dtype=np.int64)
        for f in range(F):
            if kinds[f] == 1:
                assert not _probe_returns(name, ys[f]), "the model says the reference loops here, the reference returned"
                continue
            ret[f] = 1
            ref.reset_state(); ref.diag()
            u, p = ref.decode(ys[f], TIE_SIGMA)
            uh[f], pm[f], dg[f] = u, p, ref.diag()
        uhs = np.zeros_like(uh); pms = np.zeros(F); dgs = np.zeros_like(dg)
        ref.reset_state(); ref.diag()
        for f in range(F):
            if ret[f]:
                u, p = ref.decode(ys[f], TIE_SIGMA)
                uhs[f], pms[f], dgs[f] = u, p, ref.diag()
        np.savez_compressed(os.path.join(HERE, f"ties_{name}.npz"), sigma=TIE_SIGMA, y=ys, kind=kinds, returns=ret,
                            u_hat=uh, pm=pm, diag=dg, u_hat_seq=uhs, pm_seq=pms, diag_seq=dgs)
        print(f"ties_{name}: {[len(got[k]) for k in range(4)]} frames (tie, never returns, near, clean) from {tries} draws; "
              f"Oops per tie frame {dg[kinds == 0, 0].tolist()}, wrong-order {dg[kinds == 0, 1].tolist()}; "
              f"fresh == in-sequence on {int(sum(np.array_equal(uh[f], uhs[f]) and pm[f] == pms[f] for f in range(F) if ret[f]))}/{int(ret.sum())}")


def parse_log(text):
    """-> list of blocks {seed, L, rows:[(snr, errblock, run)]}"""
    blocks, cur = [], None
    seed = None
    for line in text.replace("\r", "").split("\n"):
        m = re.match(r"\s*SEED = (\d+)", line)
        if m:
            seed = int(m.group(1))
            cur = None
            continue
        m = re.search(r"bSNR = ([\d.]+)\s+error block = (\d+)\s+run = (\d+)", line)
        if not m:
            continue
        lm = re.match(r"\s*L = (\d+)", line)
        Lv = int(lm.group(1)) if lm else 1
        if cur is None or cur["L"] != Lv or cur["seed"] != seed:
            cur = {"seed": seed, "L": Lv, "rows": []}
            blocks.append(cur)
        cur["rows"].append((float(m.group(1)), int(m.group(2)), int(m.group(3))))
    return blocks


def make_published():
    out = {}
    for zn, members in (("myResult_1024", ["SC1024out.dat", "SCL1024out.dat", "CASCL_L8.dat"]),
                        ("myResult_128", ["SC128out.txt", "SCL128out_errblock50.dat", "CASCL_128_L8.txt"])):
        z = zipfile.ZipFile(os.path.join(REF, zn + ".zip"))
        for mname in members:
            text = z.read(f"{zn}/{mname}").decode("utf-8", "replace")
            out[f"{zn}/{mname}"] = parse_log(text)
    # Logs made with a stop rule that is not in the repository's sources: "error block = 487 run = 2000" can only come
    # from `errBlock < BLE || run < 2000`.  With that rule (oracle po_run_sweep_min) the restated harness reproduces them,
    # which pins CA-SCL at L = 32 (CRC-24C at N = 1024, CRC-6 at N = 128) on reference-held data.
    for zn, mname, ble in (("myResult_1024", "CASCL_L32.dat", 50), ("myResult_128", "CASCL_128_L32.txt", 200)):
        z = zipfile.ZipFile(os.path.join(REF, zn + ".zip"))
        blocks = parse_log(z.read(f"{zn}/{mname}").decode("utf-8", "replace"))
        for b in blocks:
            b["min_run"] = 2000
            b["ble"] = ble
        out[f"{zn}/{mname}"] = blocks
    # seeds the logs do not print but the sources fix (SC_128.c:35, SCL_128.c:36: SEED = 1024)
    for key in ("myResult_128/SC128out.txt", "myResult_128/SCL128out_errblock50.dat"):
        for b in out[key]:
            if b["seed"] is None:
                b["seed"] = 1024
    with open(os.path.join(HERE, "published_runs.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("published_runs.json:", {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    only = sys.argv[1:] or None          # optional: names of the programs to regenerate ("ties": only the tie files)
    if only != ["ties"]:
        make_vectors(only)
    make_ties(only)
    if not only or "BPr_128" in only:
        make_bpr()
    if not only or "mains" in only:
        make_main_outputs()
    if not only:
        make_published()
