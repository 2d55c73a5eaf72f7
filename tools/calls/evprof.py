import cProfile, pstats, sys, os
sys.argv=["evaluate_time.py"] + (sys.argv[1:] or ["128","32","10"])
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/tools")
import evaluate_time
cProfile.run("evaluate_time.main()", "/tmp/ev.prof")
p=pstats.Stats("/tmp/ev.prof"); p.sort_stats("tottime").print_stats(28)
