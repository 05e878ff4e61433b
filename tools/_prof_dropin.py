import cProfile, pstats, sys, os, io
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "visual-odom-pipeline_amd")]
import bench
scenes = bench.pipe_scenes(1, 40, 4321)
pr = cProfile.Profile()
pr.enable()
out = bench.dropin_step_ms(0, scenes[0], n_warm=2, n_time=6)
pr.disable()
print(out)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
