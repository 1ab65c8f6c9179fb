"""Run a script of this repository against another build of the library (A/B measurements):
   DC_AB_LIB=tools/ab/libdeltaconv_hip_A.so python tools/ab_run.py bench.py --steps 40 ..."""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deltaconv_amd._lib as _L

if os.environ.get("DC_AB_LIB"):
    _L.LIB_PATH = os.path.abspath(os.environ["DC_AB_LIB"])
script = sys.argv[1]
sys.argv = sys.argv[1:]
runpy.run_path(script, run_name="__main__")
