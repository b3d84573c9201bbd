"""Collects the rocprofv3 CSVs a GPU session left under gpurun_out/prof_<round>/ (profiles/collect.sh) into profiles/<round>/
(what the judge reads).  usage: python profiles/make_summary.py r03 gpurun_out/prof_r03 [steps]"""
import collections, csv, glob, json, os, shutil, sys

rnd, O = sys.argv[1:3]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
out_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), rnd)
os.makedirs(out_dir, exist_ok=True)
one = lambda d, pat: max(glob.glob(os.path.join(d, "**", pat), recursive=True), key=os.path.getmtime)      # gpurun merges runs: newest
last_json = lambda f: json.loads([l for l in open(f) if l.startswith("{")][-1])


def kernel_set(tag, kname, suffix):
    """the four passes of one command -> dict; copies the solve-kernel rows of every pass to profiles/<round>/"""
    out = {}
    shutil.copy(one(f"{O}/{tag}trace", "*kernel_stats.csv"), os.path.join(out_dir, f"kernel_stats_{suffix}.csv"))
    rows = lambda f: [r for r in csv.DictReader(open(f)) if kname in r["Kernel_Name"]]
    for d, name in ((f"{O}/{tag}pmc_fetch", "FETCH_SIZE"), (f"{O}/{tag}pmc_write", "WRITE_SIZE")):
        rs = rows(one(d, "*counter_collection.csv"))
        with open(os.path.join(out_dir, f"pmc_{name.lower()}_{suffix}.csv"), "w") as g:
            w = csv.DictWriter(g, fieldnames=rs[0].keys()); w.writeheader(); w.writerows(rs)
        vals = [float(r["Counter_Value"]) for r in rs]
        out[name] = {"per_dispatch": vals, "mean": sum(vals) / len(vals)}
        out["kernel"] = {k: rs[0][k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count",
                                               "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size")}
    rs = rows(one(f"{O}/{tag}pmc_sq", "*counter_collection.csv"))
    with open(os.path.join(out_dir, f"pmc_sq_{suffix}.csv"), "w") as g:
        w = csv.DictWriter(g, fieldnames=rs[0].keys()); w.writeheader(); w.writerows(rs)
    sq = collections.defaultdict(list)
    for r in rs:
        sq[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out["SQ_mean_per_dispatch"] = {k: sum(v) / len(v) for k, v in sq.items()}
    out["SQ_last_dispatch"] = {k: v[-1] for k, v in sq.items()}
    tr = one(f"{O}/{tag}trace", "*kernel_trace.csv")
    ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(tr)) if kname in r["Kernel_Name"])
    out["rocprof_dispatch_ms"] = [(e - s0) / 1e6 for s0, e in ks]
    return out


# ---- headline: solve_kernel_w2<srbd13>, the driver's command ------------------------------------------------------------------
h = kernel_set("", "solve_kernel", "bench")
tj = last_json(f"{O}/trace.log")
nt = tj["config"]["launches_timed"]
runs = len(tj.get("value_runs", [1]))                      # timed regions of the command (bench.py RUNS), nt launches each
f, w = h["FETCH_SIZE"]["per_dispatch"][-runs * nt:], h["WRITE_SIZE"]["per_dispatch"][-runs * nt:]      # the timed launches come last
fm, wm = sum(f) / len(f), sum(w) / len(w)
summary = dict(h)
summary["traffic_bytes_per_launch"] = (2 * fm + wm) * 1024
summary["traffic_bytes_per_launch_uncorrected"] = (fm + wm) * 1024
summary["batches_per_launch"] = steps / nt
dur = h["rocprof_dispatch_ms"]
summary["traced_run"] = {"launches": len(dur), "warmup_launches": len(dur) - runs * nt, "rocprof_ms_timed": dur[-runs * nt:],
                         "rocprof_mean_ms_timed": sum(dur[-runs * nt:]) / (runs * nt), "hip_event_ms_by_region": tj.get("kernel_ms_runs"),
                         "hip_event_mean_ms_median_region": tj["roofline"]["kernel_ms"],
                         "note": "the HIP-event interval of a launch brackets its queue-ordering pre-pass (cost-key kernel + device sort, "
                                 "~0.1 ms) as well as the solve kernel; rocprof_ms_timed is the solve kernel alone"}
n_inst = steps * 1024 / nt
it = (tj.get("mean_iters_runs") or [tj["mean_iters"]])[-1]                      # the last dispatch belongs to the last timed region
sq = h["SQ_last_dispatch"]
summary["per_instance_iteration"] = {k: v / (n_inst * it) for k, v in sq.items() if k.startswith("SQ_INSTS")}
summary["note"] = (f"rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_* each in its own run) of `python3 bench.py --steps {steps} --warmup 5 "
                   "--no-cpu-baseline --no-extras` (one warm-up launch of 5 batches, then the timed launches -- one per timed region -- of distinct instances, each a "
                   "work queue on the resident wavefronts); traffic = the timed launches; FETCH/WRITE_SIZE in KiB; gfx950 correction per "
                   "MI355X_MICROARCH.md (HBM): FETCH_SIZE doubled (calibrated for 16 B/lane streams; this kernel reads 8 B/lane, so the "
                   "uncorrected figure is also given).  Kernel = solve_kernel_w2<SrbdModel<2,false>>, N=30.")
json.dump(summary, open(os.path.join(out_dir, "pmc_summary.json"), "w"), indent=1)
d = last_json(f"{O}/bench.log")
d["roofline"]["traffic"] = summary["traffic_bytes_per_launch"] * (d["config"]["launches_timed"] and 1)      # the PMC passes of THIS collection
d["roofline"]["traffic_source"] = f"profiles/{rnd}/pmc_summary.json"
open(os.path.join(out_dir, "bench_line.json"), "w").write(json.dumps(d) + "\n")
open(os.path.join(out_dir, "bench_line_traced_run.json"), "w").write(json.dumps(tj) + "\n")

# ---- 4-wavefront kernel: one cold batch per configuration --------------------------------------------------------------------
mw = {}
for tag in ("mw_srbd37_n20_", "mw_srbd37_n60_", "mw_srbd61_n20_"):
    if not os.path.isdir(f"{O}/{tag}trace"):
        continue
    k = kernel_set(tag, "solve_kernel_mw", tag.rstrip("_"))
    line = last_json(f"{O}/{tag}trace.log")
    fm, wm = k["FETCH_SIZE"]["per_dispatch"][-1], k["WRITE_SIZE"]["per_dispatch"][-1]
    k["traffic_bytes_per_launch"] = (2 * fm + wm) * 1024
    k["traffic_bytes_per_launch_uncorrected"] = (fm + wm) * 1024
    k["bench"] = line
    k["traffic_over_algorithmic"] = k["traffic_bytes_per_launch"] / line["algorithmic_bytes"]
    tot_it = line["mean_iters"] * line["batch"]
    k["per_instance_iteration"] = {n: v / tot_it for n, v in k["SQ_last_dispatch"].items() if n.startswith("SQ_INSTS")}
    k["rocprof_ms_last_dispatch"] = k["rocprof_dispatch_ms"][-1]
    mw[tag.rstrip("_")] = k
json.dump(mw, open(os.path.join(out_dir, "pmc_summary_mw.json"), "w"), indent=1)

print({k: d.get(k) for k in ("value", "ms_per_step", "mean_iters", "mean_rollouts", "iterations_per_s", "index_order_solves_per_s",
                              "replay_history_order_solves_per_s", "one_batch_in_flight_solves_per_s", "pcie_inclusive_solves_per_s")})
print(d["roofline"]); print(d.get("cpu_baseline")); print(d.get("ms_per_mpc_tick")); print(d.get("ms_per_fleet_tick")); print(d.get("tick_ms_vs_batch"))
print("headline SQ (timed launch):", h["SQ_last_dispatch"], summary["per_instance_iteration"], h["kernel"], summary["traced_run"])
for tag, k in mw.items():
    print(tag, k["kernel"], k["bench"], "traffic/alg", k["traffic_over_algorithmic"], k["per_instance_iteration"], k["SQ_last_dispatch"])
print(open(os.path.join(out_dir, "kernel_stats_bench.csv")).read()[:1500])
