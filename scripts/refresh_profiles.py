"""Copy the rocprofv3 summaries of a bench run from gpurun_out/ into profiles/<round>/ and
recompute profiles/traffic.json (usage: refresh_profiles.py <prefix> <round>, e.g. prof5 r01)."""
import csv, glob, json, shutil, sys
pre, rnd = sys.argv[1], sys.argv[2]
shutil.copy(glob.glob("gpurun_out/%s_kt/*/*kernel_stats.csv" % pre)[0], "profiles/%s/bench_kernel_stats.csv" % rnd)
shutil.copy("gpurun_out/%s_kt_bench.json" % pre, "profiles/%s/bench_under_rocprof.json" % rnd)
out = {}
for name, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    agg = {}
    for r in csv.DictReader(open(glob.glob("gpurun_out/%s_%s/*/*counter_collection.csv" % (pre, d))[0])):
        agg.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
    out[name] = {k: {"dispatches": len(v), "mean_KiB": sum(v) / len(v)} for k, v in agg.items() if "hz::" in k}
json.dump(out, open("profiles/%s/pmc_fetch_write_summary.json" % rnd, "w"), indent=1)
k = [x for x in out["FETCH_SIZE"] if "k_horizon<2, false, true, false" in x][0]
f, w = out["FETCH_SIZE"][k]["mean_KiB"], out["WRITE_SIZE"][k]["mean_KiB"]
cal_r = out["FETCH_SIZE"]["hz::k_bounds"]["mean_KiB"] * 1024 / (12 * 3601 * 3601)
cal_w = out["WRITE_SIZE"]["hz::k_emit_prims"]["mean_KiB"] * 1024 / (48 * 3600 * 3600)
b = json.load(open("gpurun_out/%s_kt_bench.json" % pre))
rows = int(b["config"]["cells_per_step"]) // 3569
t = {"tile": 3601, "azim": 360, "rows_per_step": rows, "hbm_bytes_per_launch": 2 * f * 1024 + w * 1024,
     "fetch_bytes": 2 * f * 1024, "write_bytes": w * 1024,
     "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (profiles/%s/pmc_fetch_write_summary.json); "
             "KiB units; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2); calibration on this run: "
             "k_bounds read ratio %.3f, k_emit_prims write ratio %.3f; kernel %s" % (rnd, cal_r, cal_w, k)}
json.dump(t, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps(t))
