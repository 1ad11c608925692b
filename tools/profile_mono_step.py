#!/usr/bin/env python3
"""cProfile of the host side of bench.py's mono step (GPU box)."""
import cProfile, pstats, sys, io, os
sys.argv = ["bench.py", "--steps", "60", "--warmup", "5", "--no-cpu-baseline"]
sys.path.insert(0, "/root/repo")
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
