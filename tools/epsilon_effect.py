"""What libnabo's configured approximation does to a registration: the CPU oracle run twice on the same inputs, once with
the exact lowest-index search the GPU path is compared with (epsilon = 0 in the product's sense) and once with libnabo's
KDTREE_LINEAR_HEAP search restated from its published algorithm at the epsilon of icp.yaml:11-15 (0.01) — and at
epsilon 0 with libnabo's own tie-break, which isolates the approximation from the tie rule.

Reported per case: correspondences whose id differs at the FIRST iteration's pose (same pose on both sides) and the worst
ratio of their distances, the first-iteration trim limit and kept count, and over the whole chain the iteration count
and the final pose difference.  CPU only (oracle/ is test infrastructure); writes JSON to stdout.

    python tools/epsilon_effect.py > profiles/r03/e_epsilon_effect.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402
from open3d_slam_advanced_rss_2024_public_amd import synthetic as syn  # noqa: E402


def first_iteration_matches(o, scan_xyz, T_init):
    """ids / d2 of the matcher at the pose the first iteration sees (reading moved by T_refMean^-1 * T_init)."""
    mean = o.reference_mean()
    p = (np.asarray(T_init, np.float64)[:3, :3] @ scan_xyz.T.astype(np.float64)).T + np.asarray(T_init, np.float64)[:3, 3]
    q = (p - mean.astype(np.float64)).astype(np.float32)
    return o.find_closests(q)


def run_case(name, cfg, map_xyz, map_n, scan_xyz, scan_n, T_init, threads):
    out = {"case": name, "reading_points": int(len(scan_xyz)), "reference_points": int(len(map_xyz))}
    res = {}
    for label, eps in (("exact", -1.0), ("nabo_eps0", 0.0), ("nabo_eps0.01", 0.01)):
        o = orc.OracleIcp(cfg, threads=threads)
        o.set_nabo_epsilon(eps)
        assert o.init_reference(map_xyz, map_n) == orc.OK
        ids, d2 = first_iteration_matches(o, scan_xyz, T_init)
        T = o.compute(scan_xyz, scan_n, T_init)
        res[label] = dict(ids=ids, d2=d2, T=T, iters=o.stats.iterations, limits=o.trace_limit.copy(), kept=o.trace_kept.copy())
    ex = res["exact"]
    for label in ("nabo_eps0", "nabo_eps0.01"):
        r = res[label]
        diff = r["ids"] != ex["ids"]
        both = diff & (ex["ids"] >= 0) & (r["ids"] >= 0)
        ratio = np.sqrt(r["d2"][both].astype(np.float64) / np.maximum(ex["d2"][both].astype(np.float64), 1e-30)) if both.any() else np.zeros(0)
        dt, ang = orc.pose_error(ex["T"], r["T"])
        n = min(len(r["limits"]), len(ex["limits"]))
        rel = np.abs(r["limits"][:n].astype(np.float64) - ex["limits"][:n]) / np.maximum(ex["limits"][:n].astype(np.float64), 1e-30) if n else np.zeros(0)
        out[label] = {
            "first_iteration_ids_changed": int(diff.sum()),
            "first_iteration_ids_changed_frac": float(diff.mean()),
            "of_which_same_distance_ties": int((diff & (r["d2"] == ex["d2"])).sum()),
            "lost_matches": int(((ex["ids"] >= 0) & (r["ids"] < 0)).sum()),
            "worst_distance_ratio": float(ratio.max()) if ratio.size else 1.0,
            "first_iteration_limit_rel_diff": float(rel[0]) if n else None,
            "worst_limit_rel_diff": float(rel.max()) if n else None,
            "first_iteration_kept": [int(ex["kept"][0]), int(r["kept"][0])] if n else None,
            "iterations": [int(ex["iters"]), int(r["iters"])],
            "final_pose_diff_m": float(np.linalg.norm(dt)),
            "final_pose_diff_rad": float(ang),
        }
    return out


def main():
    threads = int(os.environ.get("THREADS", "8"))
    yaml_cfg = orc.OracleConfig()  # open3d_slam_ros/param/icp.yaml
    cases = []
    g = np.load(os.path.join(ROOT, "tests", "golden", "car_clouds.npz"))
    ref, data = g["ref3D"], g["data3D"]
    car_cfg = orc.OracleConfig(matcher=0, max_dist=float("inf"), trim_ratio=0.85, max_normal_angle=-1, use_differential=True,
                               min_diff_rot=0.001, min_diff_trans=0.001, smooth_length=3, max_iters=40, counter_first=True)
    cases.append(run_case("car_cloud401 -> car_cloud400 (utest default chain)", car_cfg, ref[:, :3], ref[:, 3:6], data[:, :3].copy(), None,
                          np.eye(4), threads))
    c1 = syn.make_scan_pair(10000, 100000, 0.1, seed=0)
    cases.append(run_case("C1: 10k scan vs 100k map, icp.yaml chain", yaml_cfg, c1.map_xyz, c1.map_normals, c1.scan_xyz, c1.scan_normals,
                          c1.T_init, threads))
    if os.environ.get("FULL", "1") != "0":
        c2 = syn.make_scan_pair(100000, 2000000, 0.1, seed=0)
        cases.append(run_case("C2: 100k scan vs 2M map, icp.yaml chain", yaml_cfg, c2.map_xyz, c2.map_normals, c2.scan_xyz, c2.scan_normals,
                              c2.T_init, threads))
        # one C5-style sweep: a 64 x 2048 ray cast against a map built from the same world
        world = syn.make_world(1.25 * 400000 * 0.25 * 0.25)
        mp, mn = syn.make_map(world, 400000, 0.25)
        T_gt = syn.corridor_pose(world, 10)
        sx, sn = syn.make_lidar_scan(world, T_gt)[:2]
        keep = np.random.default_rng(5).permutation(len(sx))[: min(len(sx), 60000)]
        T0 = syn.perturb_pose(T_gt, 0.05, 0.5)
        cases.append(run_case("C5 sweep: 64x2048 ray cast (60k returns) vs 400k-pt 0.25 m map, icp.yaml chain", yaml_cfg, mp, mn,
                              sx[keep], None if sn is None else sn[keep], T0, threads))
    print(json.dumps({"what": "oracle exact (lowest-index ties) vs libnabo KDTREE_LINEAR_HEAP restated, same inputs", "cases": cases}, indent=1))


if __name__ == "__main__":
    main()
