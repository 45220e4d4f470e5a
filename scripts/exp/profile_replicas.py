"""One-off (GPU): cProfile of eight Plaza1 replicas in lock-step (first 40 updates)."""
import cProfile, pstats, os, sys, io
__file__ = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "scripts", "run_plaza1.py")
sys.argv = ["run_plaza1.py", os.environ.get("UPDATES", "40")]
os.environ["REPLICAS"] = "8"; os.environ["EVERY"] = "1000"
pr = cProfile.Profile()
pr.enable()
try:
    exec(compile(open(__file__).read(), __file__, "exec"))
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats(os.environ.get("SORT", "cumulative")).print_stats(70)
print(s.getvalue()[:16000])
if os.environ.get("CALLERS"):
    s = io.StringIO()
    st = pstats.Stats(pr, stream=s).sort_stats("tottime")
    for pat in os.environ["CALLERS"].split(","):
        st.print_callers(pat)
    print(s.getvalue()[:30000])
