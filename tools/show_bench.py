import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["ba_lm_iterations_histogram"], d["config"]["ba_solves_stopped_by_lm_max_iters"])
r=d["roofline"]; print(r["kernel"], r["frac"], r["traffic"], r["traffic_source"][:80]); print(r["kernels"]["source"], r["kernels"]["stale"])
for k in r["kernels"]["kernels"][:6]: print(k)
print(d.get("config5_n1")); print(d["cpu_baseline"]["value"])
ps=d["pipeline_step"]
for k,v in ps.items():
    if isinstance(v, dict): print(k, v.get("frames_per_s"), v.get("ms_per_step"), v.get("ba_solves_cut_by_the_budget"), v.get("capacity_policy_frames"), v.get("pose_error_vs_ground_truth"))
    else: print(k, v)
print(d["single_sequence"]["frames_per_s"], d["dropin_step"]["lazy_views_frames_per_s"], d["layout_3_contexts_of_32"]["frames_per_s"])
