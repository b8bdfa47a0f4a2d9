"""The drop-in boundary as a ROS node would use it (SURVEY §8 b, f1): include/aruco_slam/aruco_slam.h keeps the reference's
class surface signature for signature, so

  * tests/cpp/node_harness.cpp (our own node-shaped driver) and
  * the reference's OWN node source, /root/reference/src/aruco_slam_node.cpp, compiled IN PLACE and unchanged (build
    container only: the reference does not travel to the GPU box and is never copied)

are compiled against that header plus the scripted single-process ROS stand-in of tests/ros_stubs, linked with the library
(the gfx950 build on a GPU box, the CPU emulation build here), run over a synthetic scenario, and everything they publish
(pose, detected_markers, detected_map, real_map, marked image) is compared with the ctypes path on the same inputs."""
import os
import subprocess

import numpy as np
import pytest

from aruco_slam_amd import capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_NODE = "/root/reference/src/aruco_slam_node.cpp"
ROWS, COLS, F = 240, 320, 225.0
R2C = (0.18, -0.1, 0.05, 0.5, -0.5, 0.5, -0.5)          # base_link <- camera_optical: translation, quaternion (x, y, z, w)


def build(src, out):
    lib = capi.lib_path()
    cmd = ["g++", "-std=c++11", "-O1", "-Wall", "-Wno-unused-but-set-variable", f"-DARUCO_SLAM_MAX_ROWS={ROWS}", f"-DARUCO_SLAM_MAX_COLS={COLS}",
           "-DARUCO_SLAM_MAX_LANDMARKS=16", "-I", os.path.join(ROOT, "tests", "ros_stubs"), "-I", os.path.join(ROOT, "include"),
           src, "-o", out, lib, f"-Wl,-rpath,{os.path.dirname(lib)}", "-pthread"]
    if os.path.isdir("/opt/rocm/lib"):
        cmd += ["-Wl,-rpath-link,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]


def make_scenario(d, n_frames=5):
    """bgr8 frames + interleaved encoder samples + parameter server + TF + a ground-truth map file"""
    cfg = synth.SceneConfig(rows=ROWS, cols=COLS, f=F, grid=(2, 2), n_panels=3, col_spacing=0.9, row_spacing=0.7, step=0.05,
                            tz_far=2.4, tz_near=1.9, r2c=(R2C[0], R2C[1]))
    w = synth.PanelWorld(cfg)
    ctx = capi.Context(max_rows=ROWS, max_cols=COLS, max_batch=1, persistent_waves=4, max_landmarks=16,
                       r2c_t=R2C[:3], r2c_q=R2C[3:], max_updates_per_frame=64)
    ctx.set_camera(w.K, np.zeros(5))
    with open(os.path.join(d, "map.txt"), "w") as f:
        f.write("# id length x y z roll pitch yaw\n0 0.27 5.1 0 0.3 0 -1.5708 0\n3 0.27 4 0.6 0.3\n")
    with open(os.path.join(d, "params.txt"), "w") as f:
        for k, v in (("odom/kl", 0.05), ("odom/kr", 0.05), ("odom/b", 0.09), ("covariance/Q_k", 0.01), ("covariance/R_x", 100),
                     ("covariance/R_y", 100), ("covariance/R_theta", 10), ("aruco/markers_dictionary", 16), ("aruco/marker_length", 0.27),
                     ("frame/robot_frame_base", "base_link"), ("frame/camera_frame_optical", "camera_optical"), ("frame/world_frame", "world"),
                     ("topic/image", "/camera/image_raw"), ("topic/encoder", "/encoder"), ("const/USEFUL_DISTANCE_THRESHOLD", 4),
                     ("map/map_file", os.path.join(d, "map.txt"))):
            f.write(f"/aruco_slam_node/{k} {v}\n")
        f.write("tf base_link camera_optical " + " ".join(str(x) for x in R2C) + "\n")
    events, expect = [], []
    events.append("caminfo %r %r %r %r 5 0 0 0 0 0" % (float(w.K[0, 0]), float(w.K[1, 1]), float(w.K[0, 2]), float(w.K[1, 2])))
    expect.append(("real_map", ctx.load_map_txt(os.path.join(d, "map.txt"))))
    t = 100.0
    # an image BEFORE the first encoder message must be ignored (aruco_slam.cpp:84-85)
    fr0 = w.frame(0)
    g0 = ctx.synth_render(0, ROWS, COLS, w.K, fr0.ids, fr0.poses, noise_amp=1, seed=0)
    bgr0 = np.repeat(g0[:, :, None], 3, axis=2)
    bgr0.tofile(os.path.join(d, "pre.raw"))
    events.append(f"img {t!r} pre.raw {ROWS} {COLS} 3")
    ctx.add_image(bgr0)
    expect.append(("image_untouched", None))
    expect.append(("markers", []))                      # detected_markers: nothing yet
    expect.append(("markers", []))                      # detected_map: nothing yet
    for i in range(n_frames):
        fr = w.frame(i)
        t += fr.dt
        events.append(f"enc {t!r} {float(np.float32(fr.wl))!r} {float(np.float32(fr.wr))!r}")          # the topic carries float32
        ctx.add_encoder(float(np.float32(fr.wl)), float(np.float32(fr.wr)), t)
        expect.append(("pose", ctx.pose_msg()))
        gray = ctx.synth_render(0, ROWS, COLS, w.K, fr.ids, fr.poses, noise_amp=1, seed=i)
        bgr = np.stack([gray, np.roll(gray, 1, 1), gray], -1)
        bgr.tofile(os.path.join(d, f"f{i}.raw"))
        events.append(f"img {t!r} f{i}.raw {ROWS} {COLS} 3")
        ctx.add_image(bgr)
        expect.append(("image", ctx.draw_detected_markers(bgr)))
        expect.append(("markers", ctx.detected_markers()))
        expect.append(("markers", ctx.map_markers()))
    with open(os.path.join(d, "events.txt"), "w") as f:
        f.write("\n".join(events) + "\n")
    mu, _ = ctx.get_state()
    assert mu.size > 3, "the scenario must have mapped landmarks"
    return expect


def fnv1a(a):
    h = 1469598103934665603
    for b in np.ascontiguousarray(a).tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def check_output(d, expect, label):
    lines = open(os.path.join(d, "out.txt")).read().splitlines()
    assert len(lines) == len(expect), f"{label}: {len(lines)} messages published, {len(expect)} expected"
    for ln, (kind, want) in zip(lines, expect):
        tok = ln.split()
        if kind == "pose":
            pos, q, cov = want
            assert tok[0] == "pose" and tok[1] == "aruco_slam_node/pose" and tok[2] == "world"
            got = np.array(tok[3:], float)
            assert np.array_equal(got[:3], pos) and np.array_equal(got[3:7], q) and np.array_equal(got[7:], cov.reshape(-1)), f"{label}: pose differs"
        elif kind in ("markers", "real_map"):
            assert tok[0] == "markers" and int(tok[2]) == len(want), f"{label}: {ln[:80]} vs {len(want)} markers"
            recs = ln.split(" | ")[1:]
            for rec, m in zip(recs, want):
                f = rec.split()
                assert int(f[0]) == m["id"] and int(f[2]) == 1            # CUBE
                vals = np.array(f[3:], float)
                ref = np.concatenate([m["scale"], m["color"], m["position"], m["orientation"], [m["lifetime"]]])
                assert np.array_equal(vals, ref), f"{label}: marker {m['id']} differs"
        elif kind == "image":
            assert tok[0] == "image" and tok[1] == "aruco_slam_node/image" and (int(tok[2]), int(tok[3])) == want.shape[:2] and tok[4] == "bgr8"
            assert int(tok[5]) == fnv1a(want), f"{label}: marked image differs"
        elif kind == "image_untouched":
            assert tok[0] == "image" and int(tok[2]) == 0 and int(tok[3]) == 0      # getMarkedImg() before the filter is armed: empty Mat


def run_node(exe, d):
    env = dict(os.environ, ASLAM_STUB_SCENARIO=d, ASLAM_PERSISTENT_WAVES="4" if "emu" in os.path.basename(capi.lib_path()) else "0")
    r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


def _dropin(tmp_path, src, label):
    d = str(tmp_path)
    expect = make_scenario(d)
    exe = os.path.join(d, "node")
    build(src, exe)
    run_node(exe, d)
    check_output(d, expect, label)


def test_harness_node_publishes_what_the_ctypes_path_computes(tmp_path):
    _dropin(tmp_path, os.path.join(ROOT, "tests", "cpp", "node_harness.cpp"), "harness")


@pytest.mark.gpu
def test_harness_node_on_the_device(tmp_path):
    """the same C++ program linked with -laruco_slam_hip on the MI355X box"""
    assert capi.lib_path().endswith("libaruco_slam_hip.so")
    _dropin(tmp_path, os.path.join(ROOT, "tests", "cpp", "node_harness.cpp"), "harness/gpu")


@pytest.mark.skipif(not os.path.exists(REF_NODE), reason="the reference checkout exists only in the build container")
def test_reference_node_source_compiles_unchanged_and_runs(tmp_path):
    """src/aruco_slam_node.cpp of the reference, in place and unmodified, against include/aruco_slam/{aruco_slam,map_loader}.h"""
    _dropin(tmp_path, REF_NODE, "reference node")

