"""Condense gpurun_out/prof_<tag>/ (scripts/profile_round.sh) into profiles/: kernel stats CSV, PMC traffic JSON, bench JSON."""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

tag = sys.argv[1]
src = os.path.join("gpurun_out", "prof_" + tag)
os.makedirs("profiles", exist_ok=True)
st = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if st:
    shutil.copy(st[0], os.path.join("profiles", f"{tag}_bench_kernel_stats.csv"))
kern = defaultdict(lambda: defaultdict(float))
launch = defaultdict(int)
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    d = "pmc_fetch" if name == "FETCH_SIZE" else "pmc_write"
    for f in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].split("<")[0].replace("aslam::", "").replace("void ", "").strip()
            if row["Counter_Name"] != name:
                continue
            kern[k][name] += float(row["Counter_Value"])
            if name == "FETCH_SIZE":
                launch[k] += 1
FRAMES_PER_STEP = 320          # bench.py default: one lap of cfg2 per step, every detection launch covers one step
FRAMES_PER_LAUNCH = {"k_ekf_win_chain": 8, "k_ekf_win_scan": 8, "k_ekf_win_flush": 32}     # chain pieces of 8 frames, runs of 32
out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of `python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-extra`; "
               "KB per launch, uncorrected (gfx950: FETCH_SIZE counts 1/2 of wide coalesced reads, MI355X_MICROARCH.md HBM section: "
               "bench.py doubles it); frames_per_launch = frames one launch of the kernel covers in that command",
       "config": "cfg2", "ekf": True, "frames_per_step": FRAMES_PER_STEP,
       "kernels": {}}
for k in kern:
    n = max(launch[k], 1)
    fpl = FRAMES_PER_LAUNCH.get(k, 1 if k.startswith("k_ekf") else FRAMES_PER_STEP)
    out["kernels"][k] = {"FETCH_SIZE_KB_per_launch": round(kern[k]["FETCH_SIZE"] / n, 2), "launches": launch[k],
                         "WRITE_SIZE_KB_per_launch": round(kern[k]["WRITE_SIZE"] / n, 2), "frames_per_launch": fpl}
json.dump(out, open(os.path.join("profiles", f"{tag}_pmc_traffic.json"), "w"), indent=1)
b = os.path.join(src, "bench_default.json")
if os.path.exists(b):
    shutil.copy(b, os.path.join("profiles", f"{tag}_bench_default.json"))
print("kernels:", {k: v for k, v in out["kernels"].items() if k.startswith("k_")})
