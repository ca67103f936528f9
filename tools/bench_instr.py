"""bench.py on the INSTRUMENTATION library (libsvb_hip_instr.so: env A/B switches of the kernels' launch policies are live there,
e.g. SVB_WG_BLOCKS / SVB_WG_SMALL_BLOCKS = split-K targets of the weight gradients).  tools only: the package itself never loads
that library.    SVB_WG_BLOCKS=256 python tools/bench_instr.py --steps 30 --warmup 8 --no-cpu-baseline ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neuralsvb_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "neuralsvb_amd", "libsvb_hip_instr.so")
import bench  # noqa: E402

if __name__ == "__main__":
    bench.main()
