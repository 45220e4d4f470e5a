"""One-off (GPU): cProfile of the single-solver Plaza1 run (scripts/run_plaza1.py), by self time."""
import cProfile, pstats, os, sys, io
path = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "scripts", "run_plaza1.py")
sys.argv = ["run_plaza1.py"]
os.environ["EVERY"] = "1000"
__file__ = path
pr = cProfile.Profile()
pr.enable()
try:
    exec(compile(open(path).read(), path, "exec"))
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats(os.environ.get("SORT", "tottime")).print_stats(45)
print(s.getvalue()[:9000])
