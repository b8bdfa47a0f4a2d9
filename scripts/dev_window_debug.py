"""development aid: the window test cases on the device with per-slot stats and the kernel profile printed"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import test_ekf_window as tw
for name in sorted(tw.CASES):
    seed, groups, n_land = tw.CASES[name]
    frames, exp = tw.make_case(seed, groups, n_land)
    try:
        (mu, S), prof, worst = tw.run_device(frames, exp, batch=len(frames))
        print(name, "ok worst", worst, {k: v for k, v in prof.items() if v[0]})
    except AssertionError as e:
        print(name, "FAILED", str(e)[:300])
