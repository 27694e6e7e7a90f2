"""Copy the rocprofv3 summaries of a bench run from gpurun_out/ into profiles/<round>/ and recompute
profiles/traffic.json (HBM bytes of k_horizon per launch) and profiles/valu_model.json (wave-level VALU
instructions per wave iteration, calibrated on SQ_INSTS_VALU), both stamped with the hash of the kernels' device assembly
(scripts/kernel_asm.py) they were measured for -- bench.py refuses them when the machine code changed.  "The traversal" is the
production instantiation of k_horizon PLUS its follow-up launch (LEFT) of the same step: their counters are added up.
usage: refresh_profiles.py <prefix> <round>, e.g. prof2 r02"""
import csv, glob, json, os, shutil, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pre, rnd = sys.argv[1], sys.argv[2]
os.makedirs("profiles/%s" % rnd, exist_ok=True)
sha = bench.kernel_asm_sha()
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)       # gpurun_out/ keeps the files of earlier calls
shutil.copy(newest("gpurun_out/%s_kt/*/*kernel_stats.csv" % pre), "profiles/%s/bench_kernel_stats.csv" % rnd)
shutil.copy("gpurun_out/%s_kt_bench.json" % pre, "profiles/%s/bench_under_rocprof.json" % rnd)
try:      # the default command (with the untimed drop-in call: 5 more, shorter launches of the same kernel)
    shutil.copy(newest("gpurun_out/%s_kte/*/*kernel_stats.csv" % pre), "profiles/%s/bench_kernel_stats_default_command.csv" % rnd)
    shutil.copy("gpurun_out/%s_kte_bench.json" % pre, "profiles/%s/bench_under_rocprof_default_command.json" % rnd)
except (ValueError, OSError):
    pass
try:      # config 4
    shutil.copy(newest("gpurun_out/%s_c4kt/*/*kernel_stats.csv" % pre), "profiles/%s/bench_c4_kernel_stats.csv" % rnd)
    shutil.copy("gpurun_out/%s_c4kt_bench.json" % pre, "profiles/%s/bench_c4_under_rocprof.json" % rnd)
except (ValueError, OSError):
    pass
KERNEL = "k_horizon<2, false, true, false, false, false>"      # the production instantiation
LEFT = "k_horizon<2, false, true, false, false, true>"         # its follow-up launch (the cells handed over), fast stack


def per_kernel(d, names):
    agg = {}
    for f in [newest("gpurun_out/%s_%s/*/*counter_collection.csv" % (pre, d))]:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] in names:
                agg.setdefault((r["Kernel_Name"].split("(")[0], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    return agg


out = {}
for name, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    agg = per_kernel(d, (name,))
    out[name] = {k[0]: {"dispatches": len(v), "mean_KiB": sum(v) / len(v)} for k, v in agg.items() if "hz::" in k[0]}
json.dump(out, open("profiles/%s/pmc_fetch_write_summary.json" % rnd, "w"), indent=1)
k = [x for x in out["FETCH_SIZE"] if KERNEL in x][0]
kl = [x for x in out["FETCH_SIZE"] if LEFT in x]
f, w = out["FETCH_SIZE"][k]["mean_KiB"], out["WRITE_SIZE"][k]["mean_KiB"]
f_prod, w_prod = f, w
if kl:      # one follow-up launch per production launch
    f += out["FETCH_SIZE"][kl[0]]["mean_KiB"]; w += out["WRITE_SIZE"][kl[0]]["mean_KiB"]
cal_r = out["FETCH_SIZE"]["hz::k_bounds"]["mean_KiB"] * 1024 / (12 * 3601 * 3601)
cal_w = out["WRITE_SIZE"]["hz::k_morton"]["mean_KiB"] * 1024 / (8 * 3600 * 3600)     # key + primitive id per quad
b = json.loads(open("gpurun_out/%s_kt_bench.json" % pre).read().strip().splitlines()[-1])
t = {"tile": 3601, "azim": 360, "rows_per_step": b["config"]["rows_per_step"], "kernel_asm_sha": sha,
     "device": b["config"].get("device"), "rocm": b["config"].get("rocm"),
     "hbm_bytes_per_launch": 2 * f * 1024 + w * 1024, "fetch_bytes": 2 * f * 1024, "write_bytes": w * 1024,
     "production_launch": {"fetch_bytes": 2 * f_prod * 1024, "write_bytes": w_prod * 1024},
     "follow_up_launch": {"fetch_bytes": 2 * (f - f_prod) * 1024, "write_bytes": (w - w_prod) * 1024},
     "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (profiles/%s/pmc_fetch_write_summary.json), mean "
             "over the launches of 2 steps; KiB units; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2); calibration "
             "on this run: k_bounds read ratio %.3f, k_morton write ratio %.3f; kernel %s + its follow-up launch" % (rnd, cal_r, cal_w, k)}
json.dump(t, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps(t))
# ---- VALU model: scale the per-iteration constants so that they reproduce SQ_INSTS_VALU ------------------------
sq = per_kernel("sq", ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVES", "SQ_INSTS_SALU",
                       "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS"))
kk = [x for x in sq if KERNEL in x[0]]
if kk:
    m = {c: sum(v) / len(v) for (kn, c), v in sq.items() if KERNEL in kn}
    m_left = {c: sum(v) / len(v) for (kn, c), v in sq.items() if LEFT in kn}
    m_prod = dict(m)
    for c, v in m_left.items():
        m[c] = m.get(c, 0.0) + v
    bs = json.loads(open("gpurun_out/%s_sq_bench.json" % pre).read().strip().splitlines()[-1])
    model_winst = bs["roofline"].get("valu_winst_per_launch")
    d = bs["roofline"].get("valu_model_constants") or dict(bench.VALU_MODEL_DEFAULT)   # the constants that run used
    fac = m["SQ_INSTS_VALU"] / model_winst if model_winst else None
    vm = {"kernel_asm_sha": sha, "device": bs["config"].get("device"), "rocm": bs["config"].get("rocm"),
          "sq_counters_per_launch": m, "sq_counters_production_launch": m_prod, "sq_counters_follow_up_launch": m_left,
          "lane_utilisation_valu_production_launch": m_prod["SQ_THREAD_CYCLES_VALU"] / (64.0 * m_prod["SQ_INSTS_VALU"]) if m_prod.get("SQ_INSTS_VALU") else None,
          "lane_utilisation_valu_follow_up_launch": m_left["SQ_THREAD_CYCLES_VALU"] / (64.0 * m_left["SQ_INSTS_VALU"]) if m_left.get("SQ_INSTS_VALU") else None,
          "model_winst_before": model_winst,
          "scale": fac, "lane_utilisation_valu": m["SQ_THREAD_CYCLES_VALU"] / (64.0 * m["SQ_INSTS_VALU"]) if m.get("SQ_INSTS_VALU") else None,
          "note": "per-iteration constants of that bench run scaled by SQ_INSTS_VALU / (its model) on the launches of 2 bench steps "
                  "(the counter pass of the same run supplies the wave-iteration counts)"}
    # engine cycles of the launch (own PMC pass) -> the clock the kernel ran at and the share of issue slots it used
    gr = per_kernel("grbm", ("GRBM_GUI_ACTIVE",))
    gk = [v for (kn, c), v in gr.items() if KERNEL in kn]
    gl = [v for (kn, c), v in gr.items() if LEFT in kn]
    if gk:
        cyc = sum(gk[0]) / len(gk[0]) / 8.0          # rocprofv3 sums the counter over the 8 XCDs
        if gl:
            cyc += sum(gl[0]) / len(gl[0]) / 8.0
        rows = list(csv.DictReader(open("profiles/%s/bench_kernel_stats.csv" % rnd)))
        dur = sum(float(r["AverageNs"]) for r in rows if KERNEL in r["Name"] or LEFT in r["Name"]) * 1e-9
        simds = bs["roofline"].get("simds") or 1024
        vm.update({"grbm_gui_active_cycles_per_launch": cyc, "engine_clock_ghz_during_kernel": cyc / dur / 1e9,
                   "valu_issue_slots_used": 4.0 * m["SQ_INSTS_VALU"] / (simds * cyc)})
    if fac:
        for key in ("node_iter", "leaf_iter", "refill_iter"):
            vm[key] = d[key] * fac
    json.dump(vm, open("profiles/valu_model.json", "w"), indent=1)
    json.dump(vm, open("profiles/%s/valu_model.json" % rnd, "w"), indent=1)
    print(json.dumps(vm))

# ---- config 4: SQ counters of the shadow kernel -------------------------------------------------------------------
try:
    agg = {}
    for d in ("c4sq", "c4grbm"):
        f = newest("gpurun_out/%s_%s/*/*counter_collection.csv" % (pre, d))
        for r in csv.DictReader(open(f)):
            if "k_shadow" in r["Kernel_Name"]:
                agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in agg.items()}
    b4 = json.loads(open("gpurun_out/%s_c4kt_bench.json" % pre).read().strip().splitlines()[-1])
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0
    rf = b4["roofline"]
    out4 = {"kernel": "hz::k_shadow_refill<false>, 144 sun positions per launch", "counters_per_launch": m,
            "lane_utilisation_valu": m["SQ_THREAD_CYCLES_VALU"] / (64.0 * m["SQ_INSTS_VALU"]),
            "engine_cycles_per_launch": cyc, "simd_cycles_per_valu_instruction": 1024.0 * cyc / m["SQ_INSTS_VALU"],
            "note": "1024 SIMDs x engine cycles / SQ_INSTS_VALU: below 4 means the instructions cannot all have taken the 4 cycles "
                    "of the round-2 model -- the fast class issues every ~2.4 cycles (profiles/%s/inst_rates.json)" % rnd}
    if rf.get("wave_node_iters"):
        try:
            vmj = json.load(open("profiles/valu_model.json"))
            ni, li = vmj.get("node_iter", 147.0), vmj.get("leaf_iter", 218.0)
        except Exception:
            ni, li = 147.0, 218.0
        out4["node_iter_leaf_iter_used"] = [ni, li]
        rest = m["SQ_INSTS_VALU"] - ni * rf["wave_node_iters"] - li * rf["wave_leaf_iters"]
        out4["setup_winst_per_64_cells"] = rest / (144 * 3569 * 3569 / 64.0)
        # the bench line of config 4 prices the per-cell set-up with this (stamped with the kernel sources like the rest)
        try:
            vmj["shadow_setup_winst_per_64_cells"] = out4["setup_winst_per_64_cells"]
            json.dump(vmj, open("profiles/valu_model.json", "w"), indent=1)
        except Exception:
            pass
    json.dump(out4, open("profiles/%s/pmc_shadow_refill.json" % rnd, "w"), indent=1)
    print(json.dumps(out4))
except (ValueError, OSError, KeyError) as e:
    print("no config-4 counters:", e)


# ---- wait / busy split of the wave cycles: two PMC passes over one bench step and over config 4 ---------------
try:
    res = {"command": "bench.py --steps 1 --warmup 1 (c3) / --workload c4 --steps 1; rocprofv3 --pmc in two passes each; sums over the dispatches of "
                      "each kernel and all XCDs", "kernels": {}}
    for d in ("wb1", "wb2", "c4wb1", "c4wb2"):
        try:
            f = newest("gpurun_out/%s_%s/*/*counter_collection.csv" % (pre, d))
        except ValueError:
            continue
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"].split("(")[0]
            if "hz::" in kn:
                res["kernels"].setdefault(kn, {}).setdefault(r["Counter_Name"], 0.0)
                res["kernels"][kn][r["Counter_Name"]] += float(r["Counter_Value"])
    for kn, c in res["kernels"].items():
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            c["frac_wave_cycles"] = {"wait_any(parked: s_waitcnt)": round(c.get("SQ_WAIT_ANY", 0) / wc, 3),
                                     "wait_inst_any(issue stall)": round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
                                     "active_inst_any": round(c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3),
                                     "active_inst_valu": round(c.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3),
                                     "active_inst_sca": round(c.get("SQ_ACTIVE_INST_SCA", 0) / wc, 3),
                                     "active_inst_lds": round(c.get("SQ_ACTIVE_INST_LDS", 0) / wc, 3)}
    if res["kernels"]:
        json.dump(res, open("profiles/%s/pmc_sq_wait_busy.json" % rnd, "w"), indent=1)
        for kn in res["kernels"]:
            if "frac_wave_cycles" in res["kernels"][kn] and ("k_horizon" in kn or "k_shadow" in kn or "k_near" in kn or "k_topo" in kn):
                print(kn[:70], res["kernels"][kn]["frac_wave_cycles"])
except Exception as e:
    print("no wait / busy passes:", e)
