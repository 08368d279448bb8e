"""A/B of two builds of libmmtpsm.so on one box: `MMT_LIB=path python tools/bench_with_lib.py <bench.py arguments>` runs bench.py
with the binding pointed at that library (tools only; the product loads mmt-psm_amd/libmmtpsm.so and nothing else)."""
import os, runpy, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(ROOT, "mmt-psm_amd"))
from maskrcnn_benchmark import _hip
if os.environ.get("MMT_LIB"):
    _hip.LIB_PATH = os.path.abspath(os.environ["MMT_LIB"])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
