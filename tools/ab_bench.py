import os, sys, runpy
sys.path.insert(0, os.getcwd())
from slim_switch_moe_vit_amd import _lib
if os.environ.get("SMOE_LIB"):
    _lib.LIB_PATH = os.environ["SMOE_LIB"]
sys.argv = ["bench.py", "--steps", "30", "--warmup", "5", "--no-cpu-baseline"]
runpy.run_path("bench.py", run_name="__main__")
