"""The C++ oracle (oracle/ekf.cpp, both the literal dense form and the rank-3 form) replayed on the committed golden
vectors produced by the independently written numpy transcription (oracle/ekf_literal.py via oracle/make_golden.py)."""
import glob
import os

import numpy as np
import pytest

from oracle import pyoracle as orc

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ekf_literal_*.npz")))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
@pytest.mark.parametrize("literal", [True, False])
def test_cpp_oracle_matches_numpy_literal(path, literal):
    g = np.load(path)
    s = orc.Slam(r2c_tx=float(g["r2c"][0]), r2c_ty=float(g["r2c"][1]), literal=literal)
    s.set_camera(g["K"], g["D"])
    seen = np.zeros(3, int)
    for f in range(int(g["n_frames"])):
        s.add_encoder(float(g[f"in{f}_wl"]), float(g[f"in{f}_wr"]), float(g[f"in{f}_t"]))
        s.add_poses(g[f"in{f}_ids"], g[f"in{f}_corners"], g[f"in{f}_rvecs"], g[f"in{f}_tvecs"])
        ids, idx, act, xyth, R = s.log_observations()
        log = g[f"out{f}_log"]
        assert np.array_equal(np.stack([ids, idx, act], 1).reshape(-1, 3), log)      # pop order, indices, branch taken
        mu, S = s.get_state()
        assert mu.shape == g[f"out{f}_mu"].shape
        assert np.allclose(mu, g[f"out{f}_mu"], rtol=1e-11, atol=1e-12)
        assert np.allclose(S, g[f"out{f}_sigma"], rtol=1e-9, atol=1e-13)
        seen += np.bincount(act, minlength=3)[:3]
    assert seen[0] > 0 and seen[1] > 0 and seen[2] > 0       # augment, update and stationary branches all exercised


def test_golden_files_present():
    assert len(GOLDEN) == 3
