import cProfile, pstats, sys, os, io
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.argv = ["e2e", "--dtype", "bf16", "--steps", "30", "--warmup", "10", "--scans", "16"]
os.environ["IRX_E2E_WORKER"] = "0"
sys.path.insert(0, root); sys.path.insert(0, root + "/tools")
src = open(root + "/tools/e2e_train_bench.py").read()
pr = cProfile.Profile(); pr.enable()
exec(compile(src, root + "/tools/e2e_train_bench.py", "exec"), {"__name__": "__main__", "__file__": root + "/tools/e2e_train_bench.py"})
pr.disable()
buf = io.StringIO(); pstats.Stats(pr, stream=buf).sort_stats("cumulative").print_stats(90)
open(root + "/gpurun_out/r05_e2e_cprofile.txt", "w").write(buf.getvalue())
