#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of tools/collect_profiles.sh (gpurun_out/prof/) into the tracked summaries:

  profiles/<tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats table (our kernels + everything else)
  profiles/<tag>_pmc_hbm_traffic.csv  FETCH_SIZE / WRITE_SIZE per kernel launch (separate passes), bytes
  profiles/<tag>_pmc_sq.csv           SQ wave-cycle breakdown per kernel
  profiles/<tag>_bench_n1.json        the bench line of the same box
  profiles/traffic.json               HBM bytes per launch of the dominant kernel group (read by bench.py)

FETCH_SIZE / WRITE_SIZE units: KiB (x1024 -> bytes).  FETCH_SIZE tallies every 128-byte line at 64 bytes on gfx950
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section); calibrated on k_fast_main's own staging pattern
(profiles/r03_fetch_calibration.txt): the factor is exactly 2 for every request shape this path uses, so the HBM read
bytes written below are 2 x FETCH_SIZE x 1024.  WRITE_SIZE is exact for 16-byte-per-lane stores.
"""
FETCH_CORRECTION = 2.0
import csv, collections, glob, json, os, sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = os.path.join(root, "gpurun_out", "prof")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = os.path.join(root, "profiles")


def short(name):
    if name.startswith("void "):               # template instantiations are printed with their return type
        name = name[5:]
    if name.startswith("(anonymous namespace)::"):
        name = name[len("(anonymous namespace)::"):]
    return name.split("(")[0].split("<")[0][:60]


def find(sub, pat):
    g = glob.glob(os.path.join(src, sub, "**", pat), recursive=True)
    return g[0] if g else None


# 1. kernel stats
f = find("stats", "*kernel_stats.csv")
if f:
    with open(f) as fi, open(os.path.join(out, tag + "_kernel_stats.csv"), "w") as fo:
        fo.write(fi.read())

# 1b. the stream workload (bench.py --config 3)
f = find("stats3", "*kernel_stats.csv")
if f:
    with open(f) as fi, open(os.path.join(out, tag + "_kernel_stats_cfg3.csv"), "w") as fo:
        fo.write(fi.read())

# 2. PMC passes
def per_kernel(sub, counter):
    f = find(sub, "*counter_collection.csv")
    acc = collections.defaultdict(list)
    if not f:
        return acc
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc

fetch = per_kernel("fetch", "FETCH_SIZE")
write = per_kernel("write", "WRITE_SIZE")
rd32 = per_kernel("rdreq", "TCC_EA0_RDREQ_32B_sum")
rd64 = per_kernel("rdreq", "TCC_EA0_RDREQ_64B_sum")
rd128 = per_kernel("rdreq", "TCC_EA0_RDREQ_128B_sum")
rows = []
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("k_"):
        continue
    fb = 1024.0 * sum(fetch.get(k, [0])) / max(len(fetch.get(k, [1])), 1)
    wb = 1024.0 * sum(write.get(k, [0])) / max(len(write.get(k, [1])), 1)
    n_rd = max(len(rd128.get(k, [])), 1)
    exact = (32.0 * sum(rd32.get(k, [])) + 64.0 * sum(rd64.get(k, [])) + 128.0 * sum(rd128.get(k, []))) / n_rd if k in rd128 else float("nan")
    rows.append((k, len(fetch.get(k, [])), fb, wb, exact))
with open(os.path.join(out, tag + "_pmc_hbm_traffic.csv"), "w") as fo:
    fo.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes): python3 bench.py --steps 2 --warmup 1 --cpu-pairs 0\n")
    fo.write("# workload: 1024 independent 1280x720 BGR pairs per launch (2048 frames), ORB 500; average bytes per kernel launch = counter x 1024\n")
    fo.write("# fetch_counter_bytes = FETCH_SIZE x 1024 as reported; hbm_read_bytes = %.1f x that (128-byte lines tallied at 64 bytes: profiles/r03_fetch_calibration.txt)\n" % FETCH_CORRECTION)
    fo.write("# rdreq_bytes = 32*TCC_EA0_RDREQ_32B + 64*_64B + 128*_128B of a separate pass (nan = pass not collected): the direct count of the same reads\n")
    fo.write("kernel,launches,fetch_counter_bytes_per_launch,hbm_read_bytes_per_launch,write_bytes_per_launch,rdreq_bytes_per_launch\n")
    for k, n, fb, wb, ex in rows:
        fo.write("%s,%d,%.0f,%.0f,%.0f,%.0f\n" % (k, n, fb, FETCH_CORRECTION * fb, wb, ex))

# traffic.json: dominant group = FAST (sample + main [+ redo]) per step
steps = max(len(fetch.get("k_fast_main", [])), len(fetch.get("k_fast", [])), 1)   # k_fast: the dense kernel of the reference key-point order
fast_fetch = fast_write = 0.0
for k in ("k_fast_sample", "k_fast_main", "k_fast_redo", "k_fast_thr", "k_fast_verify", "k_fast", "k_fast_hint"):
    fast_fetch += 1024.0 * sum(fetch.get(k, [])) / steps
    fast_write += 1024.0 * sum(write.get(k, [])) / steps
# every kernel of a step: the rocprof-measured HBM bytes the whole path moves per step (corrected reads + writes)
step_total = 0.0
for k in set(fetch) | set(write):
    if k.startswith("k_"):
        step_total += (FETCH_CORRECTION * 1024.0 * sum(fetch.get(k, [])) + 1024.0 * sum(write.get(k, []))) / steps
if fast_fetch > 0:
    try:
        prev = json.load(open(os.path.join(out, "traffic.json")))
    except Exception:
        prev = {}
    json.dump({**prev, "step@1280x720x1024_n500_c3": {"hbm_bytes_per_step": int(step_total), "source": "profiles/%s_pmc_hbm_traffic.csv (all k_* kernels of a step)" % tag},
               "fast@1280x720x1024_n500_c3": {"fetch_counter_bytes": int(fast_fetch), "write_bytes": int(fast_write),
                                              "fetch_correction": FETCH_CORRECTION,
                                              "hbm_bytes": int(FETCH_CORRECTION * fast_fetch + fast_write),
                                              "source": "profiles/%s_pmc_hbm_traffic.csv" % tag}},
              open(os.path.join(out, "traffic.json"), "w"))

# 2b. stream workloads (bench.py --config 3 / 5): bytes per step of the scan kernel group and of every kernel, 3 launches of a
#     step group per pass (1 warm-up + 2 steps)
STREAMS = {3: (1280, 720, 48, 2000), 5: (3840, 2160, 16, 4000)}
try:
    tj_all = json.load(open(os.path.join(out, "traffic.json")))
except Exception:
    tj_all = {}
for cfg_id, (sw, sh, sb, sn) in STREAMS.items():
    fe = per_kernel("fetch%d" % cfg_id, "FETCH_SIZE")
    wr = per_kernel("write%d" % cfg_id, "WRITE_SIZE")
    if not fe or not wr:
        continue
    scan = [k for k in set(fe) | set(wr) if k.startswith("k_ransac_final")]
    nstep = max([len(fe.get(k, [])) for k in scan] + [1])
    sf = sum(1024.0 * sum(fe.get(k, [])) for k in scan) / nstep
    sw_ = sum(1024.0 * sum(wr.get(k, [])) for k in scan) / nstep
    tot = sum(FETCH_CORRECTION * 1024.0 * sum(fe.get(k, [])) + 1024.0 * sum(wr.get(k, [])) for k in set(fe) | set(wr) if k.startswith("k_")) / nstep
    with open(os.path.join(out, tag + "_pmc_hbm_traffic_cfg%d.csv" % cfg_id), "w") as fo:
        fo.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes): python3 bench.py --config %d --steps 2 --warmup 1 --cpu-pairs 0\n" % cfg_id)
        fo.write("# stream of %d pairs per step at %dx%d, ORB %d; average bytes per kernel launch = counter x 1024; hbm_read = %.1f x fetch counter (profiles/r03_fetch_calibration.txt)\n" % (sb, sw, sh, sn, FETCH_CORRECTION))
        fo.write("kernel,launches,fetch_counter_bytes_per_launch,hbm_read_bytes_per_launch,write_bytes_per_launch\n")
        for k in sorted(set(fe) | set(wr)):
            if k.startswith("k_"):
                n = max(len(fe.get(k, [])), 1)
                fo.write("%s,%d,%.0f,%.0f,%.0f\n" % (k, len(fe.get(k, [])), 1024.0 * sum(fe.get(k, [])) / n, FETCH_CORRECTION * 1024.0 * sum(fe.get(k, [])) / n,
                                                   1024.0 * sum(wr.get(k, [])) / max(len(wr.get(k, [])), 1)))
    tj_all["ransac_final@%dx%dx%d_n%d_c3" % (sw, sh, sb, sn)] = {
        "fetch_counter_bytes": int(sf), "write_bytes": int(sw_), "fetch_correction": FETCH_CORRECTION, "hbm_bytes": int(FETCH_CORRECTION * sf + sw_),
        "source": "profiles/%s_pmc_hbm_traffic_cfg%d.csv" % (tag, cfg_id)}
    tj_all["step@%dx%dx%d_n%d_c3" % (sw, sh, sb, sn)] = {"hbm_bytes_per_step": int(tot), "source": "profiles/%s_pmc_hbm_traffic_cfg%d.csv (all k_* kernels of a step)" % (tag, cfg_id)}
if tj_all:
    json.dump(tj_all, open(os.path.join(out, "traffic.json"), "w"))

# 3. SQ breakdown
f = find("sq", "*counter_collection.csv")
if f:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[short(r["Kernel_Name"])].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
    names = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
             "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_INSTS_VALU"]
    with open(os.path.join(out, tag + "_pmc_sq.csv"), "w") as fo:
        fo.write("# rocprofv3 --kernel-trace --pmc " + " ".join(names) + " : sums over the launches of the run (1 warm-up + 2 steps; python3 bench.py --steps 2 --warmup 1 --cpu-pairs 0 --skip-no-temporal) of 1024 pairs\n")
        fo.write("# fractions are of SQ_WAVE_CYCLES (WAIT_ANY = parked on s_waitcnt/barrier, WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY = issuing)\n")
        fo.write("kernel,wave_cycles,wait_any,wait_inst_any,active_any,active_valu,active_lds,wait_inst_lds,insts_valu\n")
        for k, v in sorted(agg.items()):
            if not k.startswith("k_"):
                continue
            wc = v["SQ_WAVE_CYCLES"] or 1.0
            fo.write("%s,%.4g,%.3f,%.3f,%.3f,%.3f,%.3f,%.3f,%.4g\n" % (
                k, wc, v["SQ_WAIT_ANY"] / wc, v["SQ_WAIT_INST_ANY"] / wc, v["SQ_ACTIVE_INST_ANY"] / wc,
                v["SQ_ACTIVE_INST_VALU"] / wc, v["SQ_ACTIVE_INST_LDS"] / wc, v["SQ_WAIT_INST_LDS"] / wc, v["SQ_INSTS_VALU"]))

    # VALU issue of the dominant (FAST) group per launch, for bench.py's informational "valu_issue" entry
    fast = [k for k in ("k_fast_sample", "k_fast_main", "k_fast_redo", "k_fast") if k in agg]
    launches = float(max(len(disp.get("k_fast_main", ())), len(disp.get("k_fast", ())), 1))
    insts = sum(agg[k]["SQ_INSTS_VALU"] for k in fast) / launches
    if insts > 0:
        json.dump({"fast@1280x720x1024_n500_c3": {"valu_wave_insts_per_launch": int(insts),
                                                  "source": "profiles/%s_pmc_sq.csv (SQ_INSTS_VALU)" % tag}},
                  open(os.path.join(out, "valu.json"), "w"))

# 4. bench line
b = os.path.join(src, "bench.json")
if os.path.exists(b):
    line = open(b).read().strip().splitlines()[-1]
    d = json.loads(line)
    # the bench ran before this summary existed: put the traffic / VALU figures of THIS collection into its line
    key = "fast@1280x720x1024_n500_c3"
    try:
        rf = d["roofline"]
        tj = json.load(open(os.path.join(out, "traffic.json")))
        if rf.get("kernel") == "fast" and key in tj:
            rf["traffic"] = tj[key]["hbm_bytes"]
            rf["traffic_over_algorithmic"] = round(rf["traffic"] / max(rf["algorithmic_bytes_per_launch"], 1), 3)
            if "traffic_correction" in rf:
                rf["traffic_correction"]["fetch_counter_bytes"] = tj[key]["fetch_counter_bytes"]
                rf["traffic_correction"]["write_bytes"] = tj[key]["write_bytes"]
            rf["traffic_source"] = "profiles/%s_pmc_hbm_traffic.csv: the PMC passes of the same collection run as this bench line (tools/collect_profiles.sh)" % tag
        skey = "step@1280x720x1024_n500_c3"
        if skey in tj and d.get("ms_per_step"):
            gbs = tj[skey]["hbm_bytes_per_step"] / (d["ms_per_step"] * 1e-3) / 1e9
            rf["hbm_traffic"] = {"bytes_per_step": tj[skey]["hbm_bytes_per_step"], "GBps": round(gbs, 1),
                                 "frac_of_peak": round(gbs / 8000.0, 4), "source": tj[skey]["source"] + "; step time of this run"}
        vj = json.load(open(os.path.join(out, "valu.json"))).get(key)
        if vj and "valu_issue" in rf and rf.get("avg_launch_ms"):
            rf["valu_issue"]["wave_insts_per_launch"] = vj["valu_wave_insts_per_launch"]
            rf["valu_issue"]["per_clk_per_cu_at_2.4GHz"] = round(
                vj["valu_wave_insts_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / (256 * 2.4e9), 3)
    except Exception:
        pass
    open(os.path.join(out, tag + "_bench_n1.json"), "w").write(json.dumps(d) + "\n")
# 5. the other configs and the probes of the same collection run
import shutil
for name, dst in [("bench.json", "_bench_cfg1.json"), ("bench_cfg2.json", "_bench_cfg2.json"), ("bench_cfg3.json", "_bench_cfg3.json"),
                  ("bench_cfg4.json", "_bench_cfg4.json"), ("bench_cfg5.json", "_bench_cfg5.json"),
                  ("bench_cfg3_forced.json", "_bench_cfg3_forced.json"), ("stream_probe.json", "_stream_probe.json"),
                  ("stream_probe_4k.json", "_stream_probe_4k.json"), ("multi_stream_probe.json", "_multi_stream_probe.json"),
                  ("e2e_probe.json", "_e2e_probe.json"), ("types_probe.json", "_types_probe.json"),
                  ("types_probe_720p.json", "_types_probe_720p.json")]:
    f = os.path.join(src, name)
    if os.path.exists(f) and os.path.getsize(f) > 2:
        txt = open(f).read().strip()
        if name.startswith("bench"):
            txt = txt.splitlines()[-1]          # the JSON line (bench.py prints nothing else on stdout)
        open(os.path.join(out, tag + dst), "w").write(txt + "\n")
f = find("types_stats", "*kernel_stats.csv")
if f:
    with open(f) as fi, open(os.path.join(out, tag + "_types_kernel_stats.csv"), "w") as fo:
        fo.write("# rocprofv3 --kernel-trace --stats -- python3 tools/types_probe.py 400x224 (one 33-frame stream through ORB, SIFT, SURF and SURF+SIFT+ORB, 4 calls each)\n")
        fo.write(fi.read())
print("profiles/ updated with tag", tag)
