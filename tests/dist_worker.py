"""Worker for tests/test_dist_cpu.py: exercises bench.py's multi-rank plumbing (gloo, CPU) without a GPU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

d = bench.Dist()
assert d.world == int(os.environ["WORLD_SIZE"]) and d.rank == int(os.environ["RANK"])
d.barrier()
# max over ranks of the per-rank elapsed time, sum of the per-rank frame counts (weak scaling aggregate)
t = d.max(1.0 + d.rank)
n = d.sum(100.0 * 4)
assert t == float(d.world), t
assert n == 400.0 * d.world, n
# independent sequences per rank: different seeds, no shared state
seeds = [100 * d.rank + i for i in range(4)]
allseeds = d.sum(float(sum(seeds)))
assert allseeds == float(sum(100 * r + i for r in range(d.world) for i in range(4)))
assert bench.pingpong(0, 8) == 0 and bench.pingpong(7, 8) == 7 and bench.pingpong(8, 8) == 6 and bench.pingpong(14, 8) == 0
d.barrier()
d.close()
print("rank %d ok" % d.rank)
