"""Developer tool: run bench.py (or another script) against an alternative build of the library, e.g. the
-DPOLAR_STAMPS build:   python tools/run_with_lib.py build/libpolar_hip_stamps.so bench.py --steps 3 --no-cpu-baseline"""
import os
import runpy
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import polardecoding_amd.api as A  # noqa: E402

lib = os.path.abspath(sys.argv[1])
A.lib_path = lambda testing=False: lib
sys.argv = sys.argv[2:]
runpy.run_path(os.path.join(REPO, sys.argv[0]) if not os.path.isabs(sys.argv[0]) else sys.argv[0], run_name="__main__")
