"""Condense gpurun_out/prof_<tag>_<config>/ (scripts/profile_round.sh) into profiles/: kernel stats CSV, PMC traffic JSON.
usage: python scripts/pmc_summary.py <tag> [config=cfg2]"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

tag = sys.argv[1]
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
src = os.path.join("gpurun_out", f"prof_{tag}_{cfg}")
os.makedirs("profiles", exist_ok=True)
st = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if st:
    shutil.copy(st[0], os.path.join("profiles", f"{tag}_{cfg}_kernel_stats.csv"))
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
out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of `python3 bench.py --config <config> --steps 3 --warmup 1 "
               "--cpu-sample 0 --no-extra`; KB per launch averaged over every launch of the kernel in that command, uncorrected (gfx950: "
               "FETCH_SIZE counts 1/2 of wide coalesced reads, MI355X_MICROARCH.md HBM section: bench.py doubles it)",
       "config": cfg, "ekf": cfg != "cfg5", "kernels": {}}
for k in kern:
    n = max(launch[k], 1)
    out["kernels"][k] = {"FETCH_SIZE_KB_per_launch": round(kern[k]["FETCH_SIZE"] / n, 2), "launches": launch[k],
                         "WRITE_SIZE_KB_per_launch": round(kern[k]["WRITE_SIZE"] / n, 2)}
json.dump(out, open(os.path.join("profiles", f"{tag}_{cfg}_pmc_traffic.json"), "w"), indent=1)
for nm in ("bench_stats.json",):
    b = os.path.join(src, nm)
    if os.path.exists(b) and os.path.getsize(b) > 0:
        shutil.copy(b, os.path.join("profiles", f"{tag}_{cfg}_bench_under_rocprof.json"))
print("kernels:", {k: v for k, v in out["kernels"].items() if k.startswith("k_")})
