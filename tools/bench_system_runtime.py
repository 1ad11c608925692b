#!/usr/bin/env python3
"""bench.py with the IMAGE's HIP runtime (ROCm 7.2, /opt/rocm/lib) loaded before torch's bundled one (7.0.2, same
soname): the A/B of DESIGN.md section 7 -- which engine carries the result delivery is the runtime's choice.  Same
arguments as bench.py.  Measured (1 GPU): mono 161.5 vs 160.2 k, stereo 64.3 vs 63.3 k frames/s."""
import ctypes, os, runpy, sys
ctypes.CDLL('/opt/rocm/lib/libamdhip64.so.7', mode=ctypes.RTLD_GLOBAL)
bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py')
sys.argv = [bench] + sys.argv[1:]
runpy.run_path(bench, run_name='__main__')
