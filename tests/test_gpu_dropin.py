"""GPU: the drop-in claim, literally.  oracle/_ref/<program>_dropin_main is the reference's OWN main() -- its PN source,
CRC, encoder, Ranq1 / normal(), stop rule and printf lines, compiled from the source where it lies -- with exactly one line
changed: the decode call `X(y, u_hat);` goes to libpolar_hip.so through the binding of INTEGRATION.md 2 (oracle/Makefile
target `dropin`, oracle/ref_wrap.c -DREF_DROPIN).  Run beside the unmodified program (oracle/_ref/<program>_main) with the
same seed it must print the same text, character for character; for the seeds the reference published, that text holds the
published run counts.  (Executables built in the container from /root/reference travel to the GPU box with oracle/_ref.)"""
import json
import os
import re
import subprocess

import pytest

from conftest import GOLDEN, REPO

pytestmark = pytest.mark.gpu
REF = os.path.join(REPO, "oracle", "_ref")


def _run(exe, seed, timeout=900):
    path = os.path.join(REF, exe)
    if not os.path.exists(path):
        pytest.skip(f"{path} not built (make -C oracle ref dropin, needs /root/reference)")
    out = subprocess.run([path, str(seed)], capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout


def test_cascl_128_main_with_the_library_prints_the_published_log():
    """CASCL_128.c (CRC-6, L = 8, BLE 200, 1.0 .. 3.0 dB) with its seed pinned to 8392: 134 328 frames, one polar_decode()
    per frame, against the unmodified program and against myResult_128.zip:CASCL_128_L8.txt"""
    mine = _run("CASCL_128_dropin_main", 8392)
    theirs = _run("CASCL_128_main", 8392)
    assert mine == theirs
    with open(os.path.join(GOLDEN, "published_runs.json")) as f:
        pub = [b for b in json.load(f)["myResult_128/CASCL_128_L8.txt"] if b["seed"] == 8392][0]
    assert [int(x) for x in re.findall(r"run = (\d+)", mine)] == [r[2] for r in pub["rows"][:5]]


def test_cascl_1024_main_with_the_library():
    """CASCL_1024_L8.c as it stands (CRC-24C, L = 8, BLE 200, 1.0 and 1.5 dB), seed 1242: same output as the unmodified program"""
    mine = _run("CASCL_1024_L8_dropin_main", 1242)
    theirs = _run("CASCL_1024_L8_main", 1242)
    assert mine == theirs and mine.count("BLER") == 2


def test_sc_128_main_with_the_library_prints_the_published_log():
    """SC_128.c (fixed SEED 1024, 100 block errors, 1.0 .. 4.0 dB): myResult_128.zip:SC128out.txt"""
    mine = _run("SC_128_dropin_main", 0)
    theirs = _run("SC_128_main", 0)
    assert mine == theirs
    with open(os.path.join(GOLDEN, "published_runs.json")) as f:
        pub = json.load(f)["myResult_128/SC128out.txt"][0]
    assert [int(x) for x in re.findall(r"run = (\d+)", mine)] == [r[2] for r in pub["rows"]]


def _first_lines(exe, seed, nlines, timeout=600):
    """stdout of exe up to its first nlines lines: the wrapper around the reference's main() (oracle/ref_wrap.c) leaves the
    program after that many printed lines when REF_STOP_AFTER_LINES is set -- the rest of the sweep is not needed"""
    path = os.path.join(REF, exe)
    if not os.path.exists(path):
        pytest.skip(f"{path} not built (make -C oracle ref dropin, needs /root/reference)")
    out = subprocess.run([path, str(seed)], capture_output=True, text=True, timeout=timeout,
                         env=dict(os.environ, REF_STOP_AFTER_LINES=str(nlines)))
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout.splitlines(keepends=True)


def test_scl_1024_main_with_the_library_prints_the_published_log():
    """SCL_1024.c as it stands (L = 8, fixed SEED 1024, 50 block errors per point): the first four points, 28 695 frames with
    one polar_decode() each, are the lines of myResult_1024.zip:SCL1024out.dat (L = 8 block) in the program's own format
    (SCL_1024.c:277-278); the fifth point (178 842 frames) is left to tests/test_gpu_kat.py, which decodes it in batches."""
    with open(os.path.join(GOLDEN, "published_runs.json")) as f:
        pub = [b for b in json.load(f)["myResult_1024/SCL1024out.dat"] if b["L"] == 8][0]
    want = ["L = 8\tbSNR = %.2f\terror block = %d\trun = %d\tBLER = %fe-2\n" % (snr, eb, run, eb * 100.0 / run)
            for snr, eb, run in pub["rows"][:4]]
    assert _first_lines("SCL_1024_dropin_main", 0, 4) == want


def test_scl_128_main_with_the_library():
    """SCL_128.c (L = 8, 1.0 .. 2.5 dB): same stdout as the unmodified program, which is the published log's L = 8 block"""
    mine = _run("SCL_128_dropin_main", 0)
    theirs = _run("SCL_128_main", 0)
    assert mine == theirs
    with open(os.path.join(GOLDEN, "published_runs.json")) as f:
        pub = [b for b in json.load(f)["myResult_128/SCL128out_errblock50.dat"] if b["L"] == 8][0]
    assert [int(x) for x in re.findall(r"run = (\d+)", mine)] == [r[2] for r in pub["rows"][:4]]


def test_bp_128_main_with_the_library():
    """BP_128.c (flooding BP, 100 iterations, 200 block errors per point, 1.0 .. 4.0 dB) with its time() seed pinned to 7:
    144 640 frames, one polar_decode() each, print what the unmodified program printed (BLER and BER lines; fixture
    tests/golden/BP_128_main_seed7.txt, made by make_golden.py from oracle/_ref/BP_128_main: 4.5 minutes of CPU)"""
    mine = _run("BP_128_dropin_main", 7)
    with open(os.path.join(GOLDEN, "BP_128_main_seed7.txt")) as f:
        assert mine == f.read()
