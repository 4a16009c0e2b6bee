"""GPU-box helper: BASELINE config 5's per-scan loop driven by COMPILED host code — tests/cpp/mapper_loop.cpp over
cpp/o3s_mapper.hpp (Mapper::addRangeMeasurement) and cpp/o3s_submap_collection.hpp — instead of the Python tool: ray-cast
sweeps with analytic normals are written to a scenario file on the box, the driver (plain g++, links the C-ABI library only)
runs them with an odometry prior, and its own wall clock around every addRangeMeasurement is reported.  Prints one JSON line.
Environment: SCANS (300), STEP (0.25 m), GEN_PROCS (12), SUBMAP_RADIUS (1e9: one submap; 20: the reference's default),
LOOP=1: the closed-loop drive of tools/mapping_loop.py (syn.loop_pose) with loop-closure refinements between resident submaps
issued by the driver itself (O3S_DRIVER_LOOP_CLOSURES)."""
import json, os, struct, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open3d_slam_advanced_rss_2024_public_amd import synthetic as syn
from oracle import oracle as orc

n_scans = int(os.environ.get("SCANS", "300"))
step = float(os.environ.get("STEP", "0.25"))
radius = float(os.environ.get("SUBMAP_RADIUS", "1e9"))
loop = os.environ.get("LOOP", "0") == "1"
world = syn.make_world(60000.0, seed=11)


def make_one(k):
    T = syn.loop_pose(world, k, step) if loop else syn.corridor_pose(world, k, step)
    sp, sn = syn.make_lidar_scan(world, T, 64, 2048, max_range=60.0, sigma=0.01, seed=300 + k)
    return T, sp.astype(np.float64), sn.astype(np.float64)


procs = int(os.environ.get("GEN_PROCS", "12"))
if procs > 1:
    import multiprocessing as mp
    with mp.get_context("fork").Pool(procs) as pool:
        made = pool.map(make_one, range(n_scans), chunksize=8)
else:
    made = [make_one(k) for k in range(n_scans)]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pkg = os.path.join(root, "open3d_slam_advanced_rss_2024_public_amd")
tmp = os.environ.get("KEEP_DIR") or tempfile.mkdtemp(prefix="o3s_mapper_bench_")   # KEEP_DIR: scenario.bin and the driver stay there (tools/prof_mapper_cpp.sh)
os.makedirs(tmp, exist_ok=True)
cm = lambda T: np.ascontiguousarray(np.asarray(T, np.float64).T).tobytes()   # noqa: E731
rng = np.random.default_rng(3)
with open(os.path.join(tmp, "scenario.bin"), "wb") as f:
    ref_period = float(os.environ.get("REF_PERIOD", "0.0"))   # seconds between renewals of the ICP reference (sweeps are 0.1 s apart); 0: every sweep (the harshest), 2.0: the reference's tutorial parameter files
    f.write(struct.pack("<8d", 0.1, 0.1, 30.0, 25.0, ref_period, 0.0, 1.0, 2.0))      # scan / map voxel, wide / narrow radius, reference renewal period
    f.write(struct.pack("<d3q", radius, 5, int(os.environ.get("MAX_POINTS", "2000000")), 3))   # maxNumPoints_: never reached here; finite, so every submap's arrays are sized once
    f.write(struct.pack("<3q", n_scans, n_scans, -1))
    f.write(cm(np.eye(4)))
    f.write(cm(np.eye(4)))
    f.write(cm(np.eye(4)))   # calibration_: the odometry poses are the sensor's own
    for k, (T, sp, sn) in enumerate(made):
        odom = T @ syn.make_T(syn.rot_axis_angle([0, 0, 1], rng.normal(0, 0.001)), rng.normal(0, 0.01, 3))   # odometry: truth + 1 cm / 1 mrad noise
        f.write(struct.pack("<d", 0.1 * k))
        f.write(cm(odom))
        f.write(cm(T))
        f.write(struct.pack("<q", len(sp)))
        f.write(np.ascontiguousarray(sp).tobytes())
        f.write(np.ascontiguousarray(sn).tobytes())
exe = os.path.join(tmp, "mapper_loop")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(root, "include"), "-I" + os.path.join(pkg, "cpp"),
                       os.path.join(root, "tests", "cpp", "mapper_loop.cpp"), "-L" + pkg, "-lo3dslam_icp_hip" + ("_hooks" if os.environ.get("O3S_LIB_VARIANT") == "hooks" else ""),
                       "-Wl,-rpath," + pkg, "-o", exe])   # O3S_LIB_VARIANT=hooks: the build that reads the A/B environment switches
def run_driver(extra_env=None):
    """warm-up + timed run of the compiled driver on the scenario file; the timed run's files stay in tmp"""
    for run in ("warm-up", "timed"):
        env = dict(os.environ)
        env.update(extra_env or {})
        if loop:
            env["O3S_DRIVER_LOOP_CLOSURES"] = "1"
        if os.environ.get("PREFETCH", "0") in ("1", "2", "3"):   # sweep k + 1 read and staged in HBM (1) / pre-processed as well (2) by a second thread while sweep k is mapped; 3: one thread stages sweep k + 2, another pre-processes sweep k + 1
            env["O3S_DRIVER_PREFETCH"] = os.environ["PREFETCH"]
        if loop and os.environ.get("ASYNC_CLOSURES", "0") == "1":   # loop-closure refinements on a worker thread over snapshots of the two submaps
            env["O3S_DRIVER_ASYNC_CLOSURES"] = "1"
        if os.environ.get("ESTIMATE_NORMALS"):   # "radius,knn": the sweeps are handed over without normals, estimated on the device
            env["O3S_DRIVER_ESTIMATE_NORMALS"] = os.environ["ESTIMATE_NORMALS"]
        if os.environ.get("PINNED", "0") == "1":   # sweeps in page-locked host memory
            env["O3S_DRIVER_PINNED"] = "1"
        if os.environ.get("FETCH_DELAY_US"):   # experiment: the receiving thread starts its work that much later inside the mapping call
            env["O3S_DRIVER_FETCH_DELAY_US"] = os.environ["FETCH_DELAY_US"]
        if os.environ.get("PRELOAD", "0") == "1":   # the scenario file is read into memory before the clock starts
            env["O3S_DRIVER_PRELOAD"] = "1"
        r = subprocess.run([exe, os.path.join(tmp, "scenario.bin"), os.path.join(tmp, "out.txt"), os.path.join(tmp, "timing.txt")], capture_output=True, text=True, env=env)
        assert r.returncode == 0, (r.stdout, r.stderr)


def brief():
    """the few figures of the run just made that the second leg (ALSO_REF_PERIOD) reports"""
    tl_ = [ln.split() for ln in open(os.path.join(tmp, "timing.txt"))]
    prod_ = np.array([[float(w[2]), float(w[3])] for w in tl_ if w[0] == "producer"])
    period_ = np.array([float(w[2]) for w in tl_ if w[0] == "period"])
    rows = [w for w in tl_ if w[0] not in ("total", "producer", "second", "period", "switch", "closure_batch", "closure")]
    us_ = np.array([float(w[1]) for w in rows])[n_scans // 10:]
    st_ = np.array([[float(v) for v in w[2:6]] for w in rows if len(w) >= 6])[n_scans // 10:]
    out_lines = open(os.path.join(tmp, "out.txt")).read().strip().splitlines()
    errs_ = []
    for k in range(n_scans):
        w = out_lines[k].split()
        T = np.array([float.fromhex(v) for v in w[9:25]]).reshape(4, 4).T
        errs_.append(float(np.linalg.norm(orc.pose_error(made[k][0], T)[0])))
    return {"ms_per_scan_median": round(float(np.median(us_)) / 1e3, 3), "hz": round(1e6 / float(np.median(us_)), 1),
            "ms_per_scan_p90_p99_max": [round(float(np.percentile(us_, q)) / 1e3, 3) for q in (90, 99, 100)],
            "pipeline_hz_steady_state": round(1e6 / float(np.mean(period_[n_scans // 10:])), 1) if len(period_) else None,
            "producer_ms_median": round(float(np.median(prod_[:, 1])) / 1e3, 3) if len(prod_) else None,
            "mapping_thread_waits_for_producer_ms_median": round(float(np.median(prod_[:, 0])) / 1e3, 3) if len(prod_) else None,
            "mapper_stopwatches_ms_median": dict(zip(["auxiliary (pre-process)", "reference re-init", "scan2map registration", "scan insertion"],
                                                     [round(float(np.median(st_[:, c][st_[:, c] > 0])) / 1e3, 3) if (st_[:, c] > 0).any() else 0.0 for c in range(4)])) if len(st_) else None,
            "pose_error_m_max": round(max(errs_), 4), "pose_error_m_median": round(float(np.median(errs_)), 4)}


also = None
if os.environ.get("ALSO_REF_PERIOD"):   # a second leg on the same sweeps with another renewal period of the ICP reference (run first: the files of the main leg are parsed below)
    run_driver({"O3S_DRIVER_REF_PERIOD": os.environ["ALSO_REF_PERIOD"]})
    also = dict(reference_renewal_period_s=float(os.environ["ALSO_REF_PERIOD"]), **brief())
run_driver()
tl = [ln.split() for ln in open(os.path.join(tmp, "timing.txt"))]
total = [w for w in tl if w[0] == "total"]
prod = np.array([[float(w[2]), float(w[3])] for w in tl if w[0] == "producer"])
second = np.array([[float(w[2]), float(w[3])] for w in tl if w[0] == "second"])   # three stages (PREFETCH=3): the pre-processing thread
period = np.array([float(w[2]) for w in tl if w[0] == "period"])
switches = [dict(after_scan=int(w[1]), create_ms=float(w[2]), closing_insert_ms=float(w[3]), centre_ms=float(w[4]), retire_ms=float(w[5]), buffered_scans_ms=float(w[6]))
            for w in tl if w[0] == "switch"]
closure_batches = [dict(after_scan=int(w[1]), pairs=int(w[2]), ms=float(w[3])) for w in tl if w[0] == "closure_batch"]
tl = [w for w in tl if w[0] not in ("total", "producer", "second", "period", "switch", "closure_batch")]
us = np.array([float(w[1]) for w in tl if w[0] != "closure"])
stages = np.array([[float(v) for v in w[2:6]] for w in tl if w[0] != "closure" and len(w) >= 6])   # the Mapper's four stopwatches, us
closures = [dict(after_scan=int(w[1]), source=int(w[2]), target=int(w[3]), rc=int(w[4]), ms=float(w[5]), overlap_points=[int(w[6]), int(w[7])], updates=int(w[8]),
                 fitness=float(w[9]), offset_m=round(float(np.linalg.norm([float(w[10]), float(w[11]), float(w[12])])), 4)) for w in tl if w[0] == "closure"]
lines = open(os.path.join(tmp, "out.txt")).read().strip().splitlines()
errs, iters, subs = [], [], 0
for k in range(n_scans):
    w = lines[k].split()
    T = np.array([float.fromhex(v) for v in w[9:25]]).reshape(4, 4).T
    dt, _ = orc.pose_error(made[k][0], T)
    errs.append(float(np.linalg.norm(dt)))
    iters.append(int(w[5]))
    subs = max(subs, int(w[7]))
steady = us[n_scans // 10:]
# CPU_SCANS=n: the same steps through the oracle's host loops on the first n sweeps (crop, Open3D voxel grid, narrow crop, patch crop,
# double -> float copy, initReference, ICP with the icp.yaml chain, transform, voxelizeWithinCroppingVolume) — single-threaded like the
# reference's mapping thread except the matcher (OpenMP over queries, as libnabo can be built): BASELINE config 5's "end-to-end Hz vs CPU"
cpu = None
n_cpu = int(os.environ.get("CPU_SCANS", "0"))
if n_cpu > 1:
    import time
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    o = orc.OracleIcp(orc.OracleConfig(), threads=threads)
    mp_, mn_, T_prev, lat = None, None, None, []
    for k in range(min(n_cpu, n_scans)):
        T_gt, sp, sn = made[k]
        t0 = time.perf_counter()
        m = orc.crop_mask(orc.make_cropper("MaxRadius", 30.0), sp)
        p, nn, _ = orc.voxel_downsample_o3d(0.1, sp[m], sn[m])
        m2 = orc.crop_mask(orc.make_cropper("MaxRadius", 25.0), p)
        if k == 0:
            T = T_gt
        else:
            mask = orc.crop_mask(orc.make_cropper("MaxRadius", 30.0, centre=T_prev[:3, 3]), mp_)
            xyzw, n32 = orc.o3d_to_pm(mp_[mask], mn_[mask])
            o.init_reference(xyzw[:, :3], n32)
            q, qn = orc.o3d_to_pm(p[m2], nn[m2])
            T, _code = o.compute(q[:, :3], qn, T_prev, raise_on_error=False)
        tp, tn = orc.transform_cloud(np.asarray(T, np.float64), p, nn)
        allp = tp if mp_ is None else np.concatenate([mp_, tp])
        alln = tn if mn_ is None else np.concatenate([mn_, tn])
        mp_, mn_, _ = orc.voxelize_within_crop(orc.make_cropper("MaxRadius", 30.0, centre=np.asarray(T)[:3, 3]), 0.1, allp, alln)
        lat.append(time.perf_counter() - t0)
        T_prev = np.asarray(T, np.float64)
    cpu = {"sweeps": len(lat), "ms_per_scan_median": round(1e3 * float(np.median(lat[1:])), 1), "hz": round(1.0 / float(np.median(lat[1:])), 2),
           "cores": threads, "kind": "port", "what": "the oracle's host loops over the same sweeps: single thread like the reference's mapping thread, except the "
                                                     f"kd-tree matcher (OpenMP over queries on {threads} threads)"}
print(json.dumps({"driver": "tests/cpp/mapper_loop.cpp over cpp/o3s_mapper.hpp (compiled, g++ -O2)", "scans": n_scans, "raw_points_per_scan": int(np.mean([len(m[1]) for m in made])),
                  "scan_model": "64x2048 ray cast, " + ("normals estimated on the device (radius, knn = %s)" % os.environ["ESTIMATE_NORMALS"] if os.environ.get("ESTIMATE_NORMALS") else "analytic normals"), "prior": "odometry (truth + 1 cm / 1 mrad noise per scan)", "submap_radius_m": radius, "reference_renewal_period_s": float(os.environ.get("REF_PERIOD", "0.0")), "submaps": subs,
                  "ms_per_scan_median": round(float(np.median(steady)) / 1e3, 3), "hz": round(1e6 / float(np.median(steady)), 1),
                  "ms_per_scan_mean": round(float(np.mean(steady)) / 1e3, 3),
                  "ms_per_scan_p90_p99_max": [round(float(np.percentile(steady, 90)) / 1e3, 3), round(float(np.percentile(steady, 99)) / 1e3, 3), round(float(steady.max()) / 1e3, 3)],
                  "slowest_calls_scan_ms_stages_ms": [[int(i + n_scans // 10), round(float(steady[i]) / 1e3, 3)] + [round(float(v) / 1e3, 3) for v in stages[i + n_scans // 10]]
                                                      for i in np.argsort(steady)[::-1][:12]] if len(stages) == len(us) else None,
                  "prefetch_thread": {"0": False, "1": "stages the raw sweep", "2": "stages and pre-processes the sweep", "3": "one thread stages sweep k + 2, a second pre-processes sweep k + 1"}[os.environ.get("PREFETCH", "0")],
                  "second_stage_ms_median": round(float(np.median(second[:, 1])) / 1e3, 3) if len(second) else None,
                  "mapping_thread_waits_for_second_stage_ms_median": round(float(np.median(second[:, 0])) / 1e3, 3) if len(second) else None,
                  "end_to_end_hz": round(float(total[0][2]) / float(total[0][1]), 1) if total else None,
                  "pipeline_hz_steady_state": round(1e6 / float(np.mean(period[n_scans // 10:])), 1) if len(period) else None,   # mean period of a sweep on the mapping thread (call + output lines + wait for the next sweep), first tenth left out like the medians
                  "end_to_end_includes_reading_the_scenario_file": os.environ.get("PRELOAD", "0") != "1", "sweeps_in_pinned_host_memory": os.environ.get("PINNED", "0") == "1", "loop_closures_on_a_worker_thread": loop and os.environ.get("ASYNC_CLOSURES", "0") == "1",
                  "producer_ms_median": round(float(np.median(prod[:, 1])) / 1e3, 3) if len(prod) else None,
                  "mapping_thread_waits_for_producer_ms_median": round(float(np.median(prod[:, 0])) / 1e3, 3) if len(prod) else None,
                  "mapper_stopwatches_ms_median": dict(zip(["auxiliary (pre-process)", "reference re-init", "scan2map registration", "scan insertion"],
                                                           [round(float(np.median(stages[n_scans // 10:, c][stages[n_scans // 10:, c] > 0])) / 1e3, 3)
                                                            if (stages[n_scans // 10:, c] > 0).any() else 0.0 for c in range(4)])) if len(stages) else None, "icp_iterations_median": int(np.median(iters[1:])),
                  "pose_error_m_max": round(max(errs), 4), "pose_error_m_median": round(float(np.median(errs)), 4),
                  "cpu_host_loop": cpu, "also_with_reference_renewal_period": also, "submap_switches": switches, "closure_batches": closure_batches,
                  "loop_closures": closures}))
