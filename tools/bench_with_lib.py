"""A/B helper: run bench.py against another build of the library (same ABI), e.g. an overhead A/B of one
compile-time switch:  python tools/bench_with_lib.py mcm_amd/libmcm_hip_nosat.so --no-drift --cpu-seconds 0"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mcm_amd.engine as eng  # noqa: E402

eng.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
import bench  # noqa: E402

bench.main()
