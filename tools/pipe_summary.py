import json,sys
for f in sys.argv[1:]:
    try:
        d=json.load(open(f))["pipeline"]
    except Exception as e:
        print(f, "ERR", e); continue
    print(f, {k:d[k] for k in ("frames_per_s","sequences_alive_at_end","ba_window","max_tracked_keypoints","mean_tracked_keypoints","mean_landmark_entries","mean_candidates","mean_pnp_inliers","mean_new_landmarks","mean_resurrected","mean_detected","ba_iterations_histogram","capacity_policy_frames","pose_error_vs_ground_truth","pnp_bound_not_reached")})
