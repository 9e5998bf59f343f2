import cProfile, pstats, sys, os, io, contextlib
sys.path.insert(0, os.getcwd())
sys.argv = ["bench.py", "--config", "5", "--rows", "1024", "--cols", "256", "--no-cpu-baseline", "--no-kernel-timing", "--steps", "2"]
import bench
pr = cProfile.Profile()
pr.enable()
try:
    bench.main()
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
