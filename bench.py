#!/usr/bin/env python3
"""bench.py — frames/s of the ArUco EKF-SLAM hot path (detect + pose + EKF) on MI355X.

Headline workload (BASELINE.json configs[1], "cfg2"): one 1280x720 gray camera stream per GPU, 20 markers in view per
frame, a 200-landmark map (built before timing by driving one lap through the reference's own augment path), one
addEncoder + one addImage per frame.  A "step" = one pass of the hot path over one batch of `--batch` consecutive frames of
the stream, frames already resident in HBM: detection + pose run batched over the step's frames, the EKF steps run in
stream order.  N > 1 (`--gpus N`): one independent stream per rank (weak scaling) plus one RCCL all-gather of the landmark
map per step; when started without torchrun's environment the script launches the N ranks itself.

Prints ONE JSON line (task contract) with
  roofline      SURVEY §8(d): ALG_BYTES(frame) x frames per launch / average launch time of the dominant kernel (HIP events on
                the library's streams) / 8 TB/s; `traffic` = that kernel's HBM bytes per launch from the committed PMC passes
                (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH_SIZE doubled as MI355X_MICROARCH.md §HBM prescribes);
  cpu_baseline  the CPU oracle (port of the reference algorithm; OpenCV / Eigen / ROS do not exist on the box) on a bounded
                sample of the same frames: 1 thread, all host cores over independent streams, and the literal O(N^3) EKF;
  extra         (rank 0, N = 1) BASELINE configs[2] "cfg3" and configs[4] "cfg5" with their own roofline / cpu_baseline,
                the single-frame aslam_add_image latency (bgr8 from host), the PCIe-inclusive host-fed stream rate, and the
                headline sizes off the fast path: `cfg2_reference_defaults` (the headline scene with the detector exactly as
                the reference runs it, aruco_slam.cpp:313 - nothing forced, fused counts reported), `cfg2_sliding` (a ring
                world whose visible set changes every 2.5 frames) and `cfg2_sliding_reference_defaults`; each with the
                share of frames the EKF fused inside windows.
Every timed frame is checked afterwards: M markers detected and M corrections fused (no observation lost to the gates).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


# ---------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without torchrun's environment starts the N ranks itself, BEFORE anything touches
# the GPU (no torch / library import in this process), relays rank 0's JSON line and exits with the job's status.
def launch_ranks(args, argv):
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline worker: a fresh interpreter that never touches the GPU (numpy + the oracle only).
def cpu_worker(spec_path):
    import numpy as np
    from oracle import pyoracle as orc
    spec = json.load(open(spec_path))
    frames = np.load(spec["frames"], mmap_mode="r")
    K = np.array(spec["K"]).reshape(3, 3)
    D = np.zeros(5)
    enc = spec["enc"]
    if spec.get("detector"):
        orc.set_detector_params(**spec["detector"])
    o = None
    dense = None
    if spec["mode"] == "dense":
        # the dense N x N x N products exactly as the reference forms them (aruco_slam.cpp:73, 146, 204), numpy / BLAS standing in
        # for Eigen; the C++ oracle's `literal` mode skips the exact zeros of Hx and (I - K Gx), which Eigen does not
        from oracle.ekf_literal import LiteralSlam
        st = np.load(spec["state"])
        dense = LiteralSlam()
        dense.K, dense.D = K, D
        dense.mu, dense.sigma = st["mu"].copy(), st["sigma"].copy()
        dense.id_map = {int(i): k for k, i in enumerate(st["ids"])}
        dense.is_init, dense.last_time = True, 0.0
    elif spec["mode"] != "detect":
        o = orc.Slam(literal=(spec["mode"] == "literal"))
        o.set_camera(K, D)
        if spec.get("detector"):
            o.set_detector_params(**spec["detector"])
        st = np.load(spec["state"])
        o.set_state(st["mu"], st["sigma"], st["ids"])
        o.add_encoder(0.0, 0.0, 0.0)                           # arms the filter clock (aruco_slam.cpp:24-29)
    t0 = time.perf_counter()
    t_now, done = 0.0, 0
    for i in range(len(frames)):
        if done >= spec.get("min_frames", 1) and time.perf_counter() - t0 > spec["budget_s"]:
            break
        img = np.ascontiguousarray(frames[i])
        wl, wr, dt = enc[i]
        t_now += dt
        if dense is not None:
            ids_o, c_o = orc.detect(img)
            poses = [orc.solve_pnp(c, spec["marker_length"], K, D) for c in c_o]
            dense.add_encoder(wl, wr, t_now)
            dense.add_poses(list(ids_o), list(c_o), [p_[0] for p_ in poses], [p_[1] for p_ in poses])
        elif o is not None:
            o.add_encoder(wl, wr, t_now)
            o.add_image(img)
        else:
            ids_o, c_o = orc.detect(img)
            for c in c_o:
                orc.solve_pnp(c, spec["marker_length"], K, D)
        done += 1
    el = time.perf_counter() - t0
    n_land = int((o.get_state()[0].size - 3) // 3) if o is not None else (int((dense.mu.size - 3) // 3) if dense is not None else 0)
    print(json.dumps({"frames": done, "seconds": el, "landmarks": n_land}), flush=True)


def run_cpu_legs(tmp, base_spec, legs):
    """legs: list of (label, mode, workers, budget_s, min_frames); the workers of one leg run concurrently."""
    out = {}
    for label, mode, workers, budget, min_frames in legs:
        spec = dict(base_spec, mode=mode, budget_s=budget, min_frames=min_frames)
        sp = os.path.join(tmp, f"spec_{label}.json")
        json.dump(spec, open(sp, "w"))
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", sp], stdout=subprocess.PIPE,
                                  stderr=subprocess.DEVNULL, text=True) for _ in range(workers)]
        res = []
        for p in procs:
            so, _ = p.communicate()
            lines = [ln for ln in so.splitlines() if ln.startswith("{")]
            if p.returncode == 0 and lines:
                res.append(json.loads(lines[-1]))
        if res:
            out[label] = {"fps": sum(r["frames"] / r["seconds"] for r in res), "frames": sum(r["frames"] for r in res),
                          "seconds": max(r["seconds"] for r in res), "workers": len(res), "landmarks": res[0]["landmarks"]}
    return out


# ---------------------------------------------------------------------------------------------------------------------
def load_pmc(cfg_name, with_ekf):
    """newest committed PMC summary taken on this config (scripts/profile_round.sh + scripts/pmc_summary.py): per kernel, HBM
    bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts wide coalesced reads at 1/2,
    MI355X_MICROARCH.md §HBM), averaged over the launches of the profiled command (the same launch shapes as the timed steps)"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True)
    # (second pass: a detect + pose only run of a config takes the detection kernels' figures from the config's full profile)
    for f, need_exact in [(f, True) for f in files] + [(f, False) for f in files]:
        js = json.load(open(f))
        if js.get("config", "cfg2") == cfg_name and (js.get("ekf", True) == with_ekf or (not need_exact and not with_ekf)):
            per_launch = {k: int((2.0 * v.get("FETCH_SIZE_KB_per_launch", 0) + v.get("WRITE_SIZE_KB_per_launch", 0)) * 1024)
                          for k, v in js["kernels"].items() if k.startswith("k_") and k != "k_render"}      # (k_render is the input generator)
            return os.path.basename(f), per_launch
    return None, {}


def scene_setup(np, capi, synth, cfg_name, rank, local_rank, args, with_ekf, download, detector="scene", qualify=True):
    """context + one lap of the stream staged in HBM.  qualify: every frame is checked (M markers detected, all past both gates,
    none "stationary") and re-rendered with another noise seed otherwise; detector: "scene" = the scene's own profile
    (synth.CONFIGS), "reference" = cv::aruco::DetectorParameters defaults as the reference runs them (aruco_slam.cpp:313)"""
    cfg = synth.CONFIGS[cfg_name]
    world = synth.make_world(cfg)
    lap = world.lap_length()
    copies = 2                                  # the lap is staged twice: a step of one whole lap alternates between the two slot sets
    ctx = capi.Context(device_id=local_rank, max_rows=cfg.rows, max_cols=cfg.cols, max_batch=copies * lap, max_landmarks=world.L + 8,
                       persistent_waves=args.waves, ekf_reserved_cus_per_xcd=args.reserve,
                       max_updates_per_frame=24 if world.M <= 24 else 64)
    ctx.set_camera(world.K, np.zeros(5))
    if detector == "scene":
        synth.apply_detector(cfg, ctx)
    frames = [world.frame(i) for i in range(lap)]
    seeds = [1000 * rank + i for i in range(lap)]
    for i, fr in enumerate(frames):
        ctx.synth_render(i, cfg.rows, cfg.cols, world.K, fr.ids, fr.poses, noise_amp=2, seed=seeds[i], download=False)
    # "the generator must be tuned (and the count asserted) or the EKF silently sees fewer than M updates":
    # frames whose noise realisation costs a marker (detection or a gate) are re-rendered with another seed
    for attempt in range(10 if qualify else 0):
        ctx.run_staged(0, lap, with_ekf=False)
        ctx.sync()
        bad, prev = [], None
        for i in range(lap):
            ids, valid, xyth, _ = ctx.get_slot_raw_observations(i)
            cur = {int(a): z for a, z in zip(ids, xyth)}
            if len(ids) != world.M or int(valid.sum()) != world.M or sorted(ids.tolist()) != sorted(frames[i].ids.tolist()):
                bad.append(i)
            elif prev is not None and any(a in prev and np.linalg.norm(prev[a] - z) < 0.0101 for a, z in cur.items()):
                bad.append(i)              # would take the reference's "stationary" no-op branch (aruco_slam.cpp:192-198, < 0.01)
            prev = cur
        if not bad:
            break
        for i in bad:
            seeds[i] += 100003
            ctx.synth_render(i, cfg.rows, cfg.cols, world.K, frames[i].ids, frames[i].poses, noise_amp=2, seed=seeds[i], download=False)
    else:
        if qualify:
            raise SystemExit(f"{cfg_name}: could not qualify frames {bad[:8]} (markers lost to detection, the gates or the stationary branch)")
    host = None
    if download:
        host = np.stack([ctx.synth_render(i, cfg.rows, cfg.cols, world.K, frames[i].ids, frames[i].poses, noise_amp=2, seed=seeds[i])
                         for i in range(min(download, lap))])
    for i in range(lap):                        # second copy of the qualified lap
        ctx.synth_render(lap + i, cfg.rows, cfg.cols, world.K, frames[i].ids, frames[i].poses, noise_amp=2, seed=seeds[i], download=False)
    enc = [(f.wl, f.wr, f.dt) for f in frames]
    ctx.stage_encoders([e[0] for e in enc] * copies, [e[1] for e in enc] * copies, [e[2] for e in enc] * copies)
    return cfg, world, lap, ctx, frames, host


def run_config(mods, args, cfg_name, steps, warmup, batch, with_ekf, rank, local_rank, world_size, cpu_frames, want_gather,
               detector="scene", qualify=True):
    np, torch, dist, capi, synth, MapGather = mods
    cfg, world, lap, ctx, frames, host = scene_setup(np, capi, synth, cfg_name, rank, local_rank, args, with_ekf,
                                                     cpu_frames if rank == 0 else 0, detector=detector, qualify=qualify)
    B = min(batch, lap)
    turn = world.frame(lap)          # the first frame of every later lap is preceded by the turn that closes the polygon
    state = None
    if with_ekf:
        ctx.run_staged(0, lap, with_ekf=True)          # build the map: one full lap through the augment path (untimed)
        ctx.sync()
        st = ctx.get_slot_ekf_stats(0, lap)
        if qualify:
            assert int(st[:, 1].sum()) == world.L, f"map has {int(st[:, 1].sum())} landmarks, expected {world.L}"
            assert (st[:, 0] == world.M).all(), "a frame of the map-building lap lost a marker"
        ctx.stage_encoders([turn.wl], [turn.wr], [turn.dt], slot0=0)
        ctx.stage_encoders([turn.wl], [turn.wr], [turn.dt], slot0=lap)
        if rank == 0:
            state = (*ctx.get_state(), ctx.get_landmark_ids())
    gather = MapGather(ctx, device=f"cuda:{local_rank}") if want_gather else None
    pos = [0]

    def step():
        # frames in stream order; `pos` runs over the two staged copies of the lap (slot s and slot lap + s hold the same frame)
        first = pos[0]
        n = B
        while n > 0:
            take = min(n, 2 * lap - first, lap - first % lap)
            ctx.run_staged(first, take, with_ekf=with_ekf)
            first = (first + take) % (2 * lap)
            n -= take
        pos[0] = first
        if gather is not None:
            gather.gather_pipelined()              # export behind this step's EKF chain, all-gather of the previous step's map

    def barrier():
        ctx.sync()
        if gather is not None:
            gather.flush()
        torch.cuda.synchronize()
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(warmup, 1)):
        step()
    # ---- timed region
    barrier()
    ctx.profile_reset()                               # (plan statistics of the timed steps only; the event spans are off)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world_size > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- every slot's last pass: M detections, M corrections fused, nothing appended, no stationary no-op
    plan = ctx.plan_stats() if with_ekf else None
    per_frame = None
    if with_ekf:
        st = ctx.get_slot_ekf_stats(0, 2 * lap)
        if (warmup + steps) * B < 2 * lap:
            st = st[:lap]                        # the second copy of the lap was never reached
        if qualify:
            ok = (st[:, 0] == world.M) & (st[:, 1] == 0) & (st[:, 2] == world.M) & (st[:, 3] == 0)
            assert ok.all(), f"frames {np.nonzero(~ok)[0][:8].tolist()} did not fuse {world.M} updates: {st[~ok][:4].tolist()}"
        per_frame = {"markers_detected_mean": round(float(st[:, 0].mean()), 3), "corrections_fused_mean": round(float(st[:, 2].mean()), 3),
                     "corrections_fused_min": int(st[:, 2].min()), "frames_fusing_all": int((st[:, 2] == world.M).sum()), "frames": int(len(st)),
                     "new_landmarks": int(st[:, 1].sum()), "stationary_no_ops": int(st[:, 3].sum())}
    elif not qualify:
        pass
    else:
        for i in range(0, lap, max(1, lap // 16)):
            assert len(ctx.get_slot_detections(i)[0]) == world.M, f"frame {i}: marker lost"

    # ---- roofline: a second, identical pass with HIP events around every kernel family on the library's streams
    ctx.profile_enable(True)
    ctx.profile_reset()
    for _ in range(steps):
        step()
    ctx.sync()
    prof = ctx.profile_get()
    ctx.profile_enable(False)
    dominant = max(prof, key=lambda k: prof[k][1])
    calls, total_ms = prof[dominant]
    N = int(ctx.get_state()[0].size) if with_ekf else 0
    alg_frame = cfg.rows * cfg.cols + ((16 * N * N + 16 * N) if with_ekf else 0) + 84 * world.M      # SURVEY §8(d) ALG_BYTES
    launches = max(calls, 1)
    fpl = steps * B / launches                                       # frames one launch of the dominant kernel processes
    avg_s = total_ms / 1e3 / launches
    achieved = alg_frame * fpl / avg_s if avg_s > 0 else 0.0
    pmc_file, traffic_all = load_pmc(cfg_name, with_ekf)
    traffic = traffic_all.get(dominant)
    fps = world_size * steps * B / elapsed
    roofline = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved / 1e9, 3), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK, 6), "traffic": traffic,
                "avg_launch_us": round(avg_s * 1e6, 2), "frames_per_launch": round(fpl, 2), "alg_bytes_per_frame": int(alg_frame),
                "path_achieved": round(alg_frame * (fps / world_size) / 1e9, 3), "path_frac": round(alg_frame * (fps / world_size) / HBM_PEAK, 6),
                "traffic_per_launch_by_kernel": traffic_all or None, "traffic_source": pmc_file if traffic_all else None,
                "kernel_ms_per_step": {k: round(v[1] / max(steps, 1), 4) for k, v in prof.items() if v[0] > 0}}
    det_used = cfg.detector if detector == "scene" else {}
    res = {"value": round(fps, 2), "ms_per_step": round(elapsed / steps * 1e3, 4), "frames_per_step": B, "N": N,
           "workload": f"{cfg_name}: {cfg.cols}x{cfg.rows} gray stream per GPU, {world.M} markers/frame in view"
                       + (", visible set slides (one marker leaves / enters every 2.5 frames)" if cfg.kind == "ring" else "") + ", "
                       + (f"{world.L}-landmark EKF (N={N})" if with_ekf else "detect + pose only") + ", frames resident in HBM"
                       + (f", detector {det_used}" if det_used else ", detector = cv::aruco::DetectorParameters defaults (aruco_slam.cpp:313)"),
           "asserted": (f"every frame: {world.M} markers detected" + (f", {world.M} corrections fused" if with_ekf else "")) if qualify
                       else "nothing forced: whatever the default detector finds and the gates (aruco_slam.cpp:327-333, 367-368) let through is fused",
           "roofline": roofline}
    if per_frame is not None:
        res["per_frame"] = per_frame
    if plan is not None:
        tot = max(plan["frames_in_windows"] + plan["frames_per_frame_chain"], 1)
        res["ekf_schedule"] = dict(plan, share_of_frames_in_windows=round(plan["frames_in_windows"] / tot, 4),
                                   mean_window_frames=round(plan["frames_in_windows"] / max(plan["windows"], 1), 2))
    return res, dict(cfg=cfg, world=world, lap=lap, ctx=ctx, frames=frames, host=host, state=state, turn=turn, step=step, B=B)


def cpu_baseline(np, synth, run, cfg_name, with_ekf, legs_wanted, budget):
    """the oracle timed on this box's host cores on a bounded sample: the second lap of the same stream, started from the map
    the first lap built (the state the timed GPU steps start from)"""
    cfg, world, host, frames, turn = run["cfg"], run["world"], run["host"], run["frames"], run["turn"]
    if host is None or len(host) == 0:
        return None
    tmp = tempfile.mkdtemp(prefix="aslam_bench_")
    try:
        np.save(os.path.join(tmp, "frames.npy"), host)
        enc = [(turn.wl, turn.wr, turn.dt)] + [(f.wl, f.wr, f.dt) for f in frames[1:len(host)]]
        spec = {"frames": os.path.join(tmp, "frames.npy"), "K": world.K.reshape(-1).tolist(), "enc": enc,
                "marker_length": cfg.marker_length, "detector": cfg.detector}
        if with_ekf:
            mu, sigma, ids = run["state"]
            np.savez(os.path.join(tmp, "state.npz"), mu=mu, sigma=sigma, ids=ids)
            spec["state"] = os.path.join(tmp, "state.npz")
        ncores = min(len(os.sched_getaffinity(0)), 16)         # the box's CPU share for one GPU
        mode = "rank3" if with_ekf else "detect"
        legs = [("one", mode, 1, budget, 5)]
        if "all" in legs_wanted:
            legs.append(("all", mode, ncores, budget, 5))
        if "literal" in legs_wanted and with_ekf:
            legs.append(("literal", "literal", 1, budget, 1))
            legs.append(("dense", "dense", 1, budget, 2))
        r = run_cpu_legs(tmp, spec, legs)
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    if "one" not in r:
        return None
    what = f"detect+PnP+rank-3 EKF on the {r['one']['landmarks']}-landmark map" if with_ekf else "detect+PnP"
    out = {"value": round(r["one"]["fps"], 2), "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": f"{r['one']['frames']} frames of the same stream through oracle/ ({what}), {r['one']['seconds']:.1f} s, "
                     f"g++ -O2 scalar restatement of the reference algorithm (OpenCV / Eigen unavailable)",
           "nproc": ncores}
    if "all" in r:
        out["all_cores"] = {"value": round(r["all"]["fps"], 2), "cores": r["all"]["workers"],
                            "sample": f"{r['all']['workers']} independent streams (one oracle process each) x ~{r['all']['frames'] // max(r['all']['workers'], 1)} frames, {r['all']['seconds']:.1f} s"}
    if "literal" in r:
        out["literal_ekf"] = {"value": round(r["literal"]["fps"], 3), "cores": 1,
                              "sample": f"{r['literal']['frames']} frames with the reference's formulas as written (aruco_slam.cpp:73,146,204), "
                                        f"C++ loops that skip the exact zeros of Hx and I - K Gx, {r['literal']['seconds']:.1f} s"}
    if "dense" in r:
        out["literal_dense_ekf"] = {"value": round(r["dense"]["fps"], 3), "cores": ncores,
                                    "sample": f"{r['dense']['frames']} frames with the dense N x N x N products the reference executes "
                                              f"(aruco_slam.cpp:73,146,204; numpy / OpenBLAS on all cores standing in for Eigen), {r['dense']['seconds']:.1f} s"}
    return out


def add_image_latency(np, capi, run, n=120):
    """the drop-in single-frame call (ArucoSlam::addImage, aruco_slam_node.cpp:96): bgr8 frame from host memory, blocking"""
    cfg, world, host, frames, turn = run["cfg"], run["world"], run["host"], run["frames"], run["turn"]
    n = min(n, len(host))
    ctx = capi.Context(device_id=0, max_rows=cfg.rows, max_cols=cfg.cols, max_batch=1, max_landmarks=world.L + 8)
    ctx.set_camera(world.K, np.zeros(5))
    mu, sigma, ids = run["state"]
    ctx.set_state(mu, sigma, ids)
    ctx.add_encoder(0.0, 0.0, 0.0)
    bgr = [np.ascontiguousarray(np.repeat(host[i][:, :, None], 3, axis=2)) for i in range(n)]
    lat_img, lat_both, parts = [], [], []
    t_now = 0.0
    for i in range(n):
        fr = turn if i == 0 else frames[i]
        t_now += fr.dt
        if i == n - 24:                                  # the last frames run with the per-kernel HIP-event spans on (not part of the percentiles)
            ctx.profile_enable(True); ctx.profile_reset()
        t0 = time.perf_counter()
        ctx.add_encoder(fr.wl, fr.wr, t_now)
        t1 = time.perf_counter()
        ctx.add_image(bgr[i])
        t2 = time.perf_counter()
        if i < n - 24:
            lat_img.append(t2 - t1)
            lat_both.append(t2 - t0)
            parts.append(ctx.last_timing())
    prof = ctx.profile_get()
    ctx.profile_enable(False)
    st = ctx.get_observations()
    assert int((st[2] == 1).sum()) == world.M
    ctx.close()
    a = np.array(lat_img[10:]) * 1e6
    b = np.array(lat_both[10:]) * 1e6
    return {"unit": "us", "frames": len(a), "input": f"{cfg.cols}x{cfg.rows} bgr8 from pageable host memory, {world.L}-landmark map",
            "add_image_p50": round(float(np.percentile(a, 50)), 1), "add_image_p99": round(float(np.percentile(a, 99)), 1),
            "encoder_plus_image_p50": round(float(np.percentile(b, 50)), 1), "encoder_plus_image_p99": round(float(np.percentile(b, 99)), 1),
            "host_phases_p50": {k: round(float(np.percentile([p_[k] for p_ in parts[10:]], 50)), 1) for k in parts[0]},
            "device_kernels_us_per_frame": {k: round(v[1] / max(v[0], 1) * 1e3, 1) for k, v in prof.items() if v[0]}}


def host_fed_rate(np, run, H=100):
    """PCIe-inclusive rate of the host-fed stream API (pinned ring, asynchronous upload); never part of `value`"""
    cfg, world, lap, ctx, frames, turn, host = run["cfg"], run["world"], run["lap"], run["ctx"], run["frames"], run["turn"], run["host"]
    ctx.sync()
    enc = [(turn.wl, turn.wr, turn.dt)] + [(f.wl, f.wr, f.dt) for f in frames[1:]]
    ctx.set_state(*run["state"])                    # the filter as it stood at the start of a lap
    ctx.stream_open(cfg.rows, cfg.cols, 1, H)
    out = {}
    for mode in ("push", "pinned"):
        best = 0.0
        for rep in range(2):
            t0 = time.perf_counter()
            for i in range(lap):
                if mode == "push":
                    ctx.stream_push(host[i % len(host)] if i < len(host) else host[-1], *enc[i])
                else:
                    ctx.stream_slot(cfg.rows, cfg.cols)
                    ctx.stream_commit(*enc[i])
            ctx.stream_flush()
            best = max(best, lap / (time.perf_counter() - t0))
        out[mode] = {"frames_per_s": round(best, 1), "GB_per_s": round(best * cfg.rows * cfg.cols / 1e9, 2)}
    return out


def cfg5_run(np, capi, synth, args, steps):
    """BASELINE configs[4]: 64 x 640x480 frames per step, 4 markers each, detect + PnP only"""
    rows, cols, f, n = 480, 640, 450.0, 64
    K = synth.camera_matrix(rows, cols, f)
    ctx = capi.Context(device_id=0, max_rows=rows, max_cols=cols, max_batch=2 * n, max_landmarks=16, persistent_waves=args.waves)
    ctx.set_camera(K, np.zeros(5))
    host = []
    for i in range(2 * n):
        ids, poses, _ = synth.simple_scene(rows, cols, f, 4, seed=i, tz=(1.0, 2.0))
        img = ctx.synth_render(i, rows, cols, K, ids, poses, noise_amp=2, seed=i, download=(i < 48))
        if img is not None:
            host.append(img)
    for _ in range(3):
        ctx.run_staged(0, n, with_ekf=False)
    ctx.sync()
    t0 = time.perf_counter()
    for s in range(steps):
        ctx.run_staged((s & 1) * n, n, with_ekf=False)
    ctx.sync()
    el = time.perf_counter() - t0
    found = sum(len(ctx.get_slot_detections(i)[0]) for i in range(2 * n))
    assert found == 2 * n * 4, f"cfg5: {found} of {2 * n * 4} markers found"
    ctx.profile_enable(True)
    ctx.profile_reset()
    for s in range(steps):
        ctx.run_staged((s & 1) * n, n, with_ekf=False)
    ctx.sync()
    prof = ctx.profile_get()
    ctx.profile_enable(False)
    dominant = max(prof, key=lambda k: prof[k][1])
    avg_s = prof[dominant][1] / 1e3 / max(prof[dominant][0], 1)
    alg_frame = rows * cols + 84 * 4
    achieved = alg_frame * n / avg_s
    pmc_file, traffic_all = load_pmc("cfg5", False)
    res = {"value": round(steps * n / el, 1), "unit": "frames/s", "ms_per_step": round(el / steps * 1e3, 4), "frames_per_step": n,
           "workload": "cfg5: 64 x 640x480 gray frames per step, 4 markers each, detect + PnP only (one context)",
           "asserted": "every frame: 4 markers detected",
           "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(achieved / 1e9, 3), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK, 6), "traffic": traffic_all.get(dominant), "avg_launch_us": round(avg_s * 1e6, 2),
                        "frames_per_launch": n, "alg_bytes_per_frame": alg_frame,
                        "traffic_per_launch_by_kernel": traffic_all or None, "traffic_source": pmc_file if traffic_all else None,
                        "path_frac": round(alg_frame * steps * n / el / HBM_PEAK, 6),
                        "kernel_ms_per_step": {k: round(v[1] / steps, 4) for k, v in prof.items() if v[0] > 0}}}
    ctx.close()
    return res, np.stack(host), K


def emu_check(mods, args, rank, world):
    """CPU rehearsal of the N-rank path for the test suite (gloo + the emulation build named by ARUCO_SLAM_LIB): the same
    launcher, rank environment, barrier / max-over-ranks timing and map gather as the GPU run, on a tiny scene"""
    np, torch, dist, capi, synth, MapGather = mods
    assert "emu" in os.path.basename(capi.lib_path()), "--emu-check is a test mode for the CPU emulation build"
    dist.init_process_group(backend="gloo")
    cfg = synth.SceneConfig(rows=240, cols=320, f=225.0, grid=(2, 2), n_panels=3, col_spacing=0.9, row_spacing=0.7, step=0.05,
                            tz_far=2.4, tz_near=1.9, seed=1 + rank)
    w = synth.PanelWorld(cfg)
    n = 3
    ctx = capi.Context(max_rows=cfg.rows, max_cols=cfg.cols, max_batch=n, persistent_waves=4, max_landmarks=16)
    ctx.set_camera(w.K, np.zeros(5))
    frs = [w.frame(i) for i in range(n)]
    for i, f in enumerate(frs):
        ctx.synth_render(i, cfg.rows, cfg.cols, w.K, f.ids, f.poses, noise_amp=1, seed=10 * rank + i, download=False)
    ctx.stage_encoders([f.wl for f in frs], [f.wr for f in frs], [f.dt for f in frs])
    g = MapGather(ctx)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.run_staged(0, n, with_ekf=True)
        ctx.sync()
        g.gather()
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    rec = g.records()
    assert rec.shape[0] == world and all((rec[r]["id"] >= 0).sum() > 0 for r in range(world))
    if rank == 0:
        print(json.dumps({"metric": "emu-check", "value": round(world * args.steps * n / float(t.item()), 2), "unit": "frames/s",
                          "n_gpus": world, "steps": args.steps, "warmup": 0, "emu_check": True,
                          "landmarks_per_rank": [int((rec[r]["id"] >= 0).sum()) for r in range(world)]}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=320, help="frames per step (one lap of the headline scene; the lap is staged twice, so consecutive steps use disjoint slots)")
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--cpu-sample", type=int, default=240, help="frames handed to the CPU oracle (0 = skip the CPU baseline)")
    ap.add_argument("--cpu-budget", type=float, default=10.0, help="seconds of CPU work per baseline leg")
    ap.add_argument("--no-ekf", action="store_true", help="detect + pose only")
    ap.add_argument("--no-extra", action="store_true", help="skip the cfg3 / cfg5 / latency / host-fed extras")
    ap.add_argument("--waves", type=int, default=0, help="wavefronts of the work-queue kernels (0 = library default)")
    ap.add_argument("--force-gather", action="store_true", help="run the (pipelined) map gather even with one rank (development check)")
    ap.add_argument("--reserve", type=int, default=0, help="CUs per XCD kept free of detection beside the EKF chain (0 = library default 16, <0 = off)")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--emu-check", action="store_true", help=argparse.SUPPRESS)   # tests/test_dist_gloo.py: launcher + rank plumbing on CPU
    args = ap.parse_args()

    if args.cpu_worker:
        return cpu_worker(args.cpu_worker)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    import torch.distributed as dist
    import numpy as np
    from aruco_slam_amd import capi, synth
    from aruco_slam_amd.dist import MapGather
    mods = (np, torch, dist, capi, synth, MapGather)

    if args.emu_check:
        return emu_check(mods, args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1 or (args.force_gather and "RANK" in os.environ):
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if args.config == "cfg5":                      # BASELINE configs[4] on its own (profiling runs): detect + PnP only
        r5, _, _ = cfg5_run(np, capi, synth, args, steps=args.steps)
        print(json.dumps({"metric": "frames/s (detect+pose) cfg5", "value": r5["value"], "unit": "frames/s", "n_gpus": 1, "steps": args.steps,
                          "warmup": 3, "ms_per_step": r5["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "u8/f64", "data": "synthetic", "config": {"workload": r5["workload"], "asserted": r5["asserted"]},
                          "roofline": r5["roofline"], "cpu_baseline": None}), flush=True)
        return
    with_ekf = not args.no_ekf
    extras_on = rank == 0 and world == 1 and not args.no_extra and args.config == "cfg2" and with_ekf

    head, run = run_config(mods, args, args.config, args.steps, args.warmup, args.batch, with_ekf, rank, local_rank, world,
                           (max(args.cpu_sample, 400 if extras_on else 0) if args.cpu_sample > 0 else 0),
                           want_gather=(world > 1 or args.force_gather))
    cpu = None
    extra = None
    if rank == 0 and args.cpu_sample > 0:
        sample_run = dict(run, host=run["host"][:args.cpu_sample])
        cpu = cpu_baseline(np, synth, sample_run, args.config, with_ekf, ("all", "literal"), args.cpu_budget)
    if extras_on:
        extra = {}
        try:
            extra["add_image_latency"] = add_image_latency(np, capi, run)
        except Exception as e:                                           # an extra must never cost the headline line
            extra["add_image_latency"] = {"error": repr(e)}
        try:
            extra["host_fed_stream"] = host_fed_rate(np, run)
        except Exception as e:
            extra["host_fed_stream"] = {"error": repr(e)}
        run["ctx"].close()
        run = None
        # the same sizes off the fast path: the reference's own detector configuration on the headline scene, and a
        # world whose visible set slides (with the scene's detector profile and with the defaults)
        # ... and the headline frames through detection + pose alone (the EKF side bounds the headline; this is what the detector does)
        for label, cname, det, qual, ekf_on in (("cfg2_reference_defaults", "cfg2", "reference", False, True),
                                                ("cfg2_sliding", "cfg2_sliding", "scene", True, True),
                                                ("cfg2_sliding_reference_defaults", "cfg2_sliding", "reference", False, True),
                                                ("cfg2_detect_pose_only", "cfg2", "scene", True, False)):
            try:
                lapn = synth.make_world(synth.CONFIGS[cname]).lap_length()
                rx, runx = run_config(mods, args, cname, 8, 2, lapn, ekf_on, 0, local_rank, 1, 0, want_gather=False, detector=det, qualify=qual)
                rx["unit"] = "frames/s"
                runx["ctx"].close()
                extra[label] = rx
            except (Exception, SystemExit) as e:
                extra[label] = {"error": repr(e)}
        try:
            r5, host5, K5 = cfg5_run(np, capi, synth, args, steps=20)
            tmp = tempfile.mkdtemp(prefix="aslam_bench5_")
            np.save(os.path.join(tmp, "frames.npy"), host5)
            spec = {"frames": os.path.join(tmp, "frames.npy"), "K": K5.reshape(-1).tolist(), "enc": [(0, 0, 0)] * len(host5),
                    "marker_length": 0.27, "detector": {}}
            c5 = run_cpu_legs(tmp, spec, [("one", "detect", 1, 5.0, 5)])
            import shutil
            shutil.rmtree(tmp, ignore_errors=True)
            if "one" in c5:
                r5["cpu_baseline"] = {"value": round(c5["one"]["fps"], 2), "unit": "frames/s", "cores": 1, "kind": "port",
                                      "sample": f"{c5['one']['frames']} of the same frames through oracle/ (detect+PnP), {c5['one']['seconds']:.1f} s"}
            extra["cfg5"] = r5
        except Exception as e:
            extra["cfg5"] = {"error": repr(e)}
        try:
            r3, run3 = run_config(mods, args, "cfg3", 8, 2, 100, True, 0, local_rank, 1, 16, want_gather=False)    # (8 steps: the two-stage pipeline detection | EKF needs a few steps to show its steady rate)
            r3["unit"] = "frames/s"
            r3["cpu_baseline"] = cpu_baseline(np, synth, run3, "cfg3", True, (), 8.0)
            run3["ctx"].close()
            extra["cfg3"] = r3
        except Exception as e:
            extra["cfg3"] = {"error": repr(e)}

    if rank == 0:
        out = {
            "metric": "frames/s (detect+pose+EKF) at 1280x720, 20 markers, 200 landmarks" if args.config == "cfg2" and with_ekf
                      else f"frames/s ({'detect+pose+EKF' if with_ekf else 'detect+pose'}) {args.config}",
            "value": head["value"], "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/f64", "data": "synthetic",
            "config": {"workload": head["workload"], "frames_per_step": head["frames_per_step"], "streams": world, "ekf": with_ekf,
                       "asserted": head["asserted"],
                       "map_gather": "rccl all_gather per step, pipelined one step behind" if world > 1 else "none"},
            "roofline": head["roofline"], "cpu_baseline": cpu,
        }
        if extra is not None:
            out["extra"] = extra
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
