#!/usr/bin/env python3
"""Round profile on the GPU box: rocprofv3 kernel-trace statistics of the bench command (configs 2, 4, 5) and hardware
counters (separate --pmc passes, never combined with tracing) for the bench command and for the other BASELINE shapes.
Writes text / json summaries under gpurun_out/prof_<tag>/ that are then copied into profiles/.
usage (GPU box, repo root): python tools/profile_round.py <tag>"""
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = dict(os.environ, TMPDIR="/tmp")


def trace(out, name, cmd):
    d = os.path.join(out, "trace_" + name)
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "t", "--"] + cmd, cwd="/tmp", env=ENV,
                       capture_output=True, text=True)
    lines = ["# rocprofv3 --kernel-trace --stats -- " + " ".join(os.path.relpath(c, ROOT) if c.startswith(ROOT) else c for c in cmd),
             "# taken at commit " + os.environ.get("GACQ_EVIDENCE_HEAD", "unknown"),
             "# (--lanes 1 = one step in flight, as in bench.py's profiling pass: overlapped launches have no duration of their own)", ""]
    for path in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(path)))
        lines.append("%-100s %6s %12s %12s %7s %10s %10s" % ("kernel", "calls", "total_us", "avg_us", "%", "min_us", "max_us"))
        for row in rows[:16]:
            nm = re.sub(r"\(.*", "", row["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", ""))[:100]
            lines.append("%-100s %6s %12.1f %12.2f %7.2f %10.2f %10.2f" % (nm, row["Calls"], float(row["TotalDurationNs"]) / 1e3, float(row["AverageNs"]) / 1e3,
                                                                           float(row["Percentage"]), float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3))
    bench = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if bench:
        j = json.loads(bench[-1])
        lines += ["", "bench line of the traced run (tracing overhead included): value %.4g %s, %.4f ms/step, roofline %s %.3f (avg kernel %.4f ms by HIP events), sustained %s"
                  % (j["value"], j["unit"], j["ms_per_step"], j["roofline"]["bound"], j["roofline"]["frac"], j["roofline"]["avg_kernel_ms"],
                     json.dumps(j.get("sustained")))]
        open(os.path.join(out, "bench_line_%s_traced.json" % name), "w").write(bench[-1] + "\n")
    else:
        lines += ["", "(no bench JSON line)", r.stderr[-1500:]]
    open(os.path.join(out, "rocprofv3_kernel_stats_%s.txt" % name), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


def pmc(out, name, cmd, filt):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_profile.py"), os.path.join(out, "pmc_" + name), "--groups",
                        "sq_time,sq_inst,grbm,fetch,write,tcc_hit", "--filter", filt, "--"] + cmd, capture_output=True, text=True, env=ENV)
    head = "# rocprofv3 --pmc, one pass per counter group (tools/pmc_profile.py), command: %s\n# taken at commit %s\n" % (
        " ".join(os.path.relpath(c, ROOT) if c.startswith(ROOT) else c for c in cmd), os.environ.get("GACQ_EVIDENCE_HEAD", "unknown"))
    open(os.path.join(out, "pmc_counters_%s.txt" % name), "w").write(head + r.stdout + ("\n" + r.stderr[-2000:] if r.returncode else ""))
    print(head + r.stdout[-6000:])
    return r.stdout


def main():
    tag = sys.argv[1]
    out = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    os.makedirs(out, exist_ok=True)
    py = sys.executable
    # --lanes 1: ONE step in flight.  The bench's default keeps two (overlapped on two hardware queues); overlapped launches share the
    # device, so their durations in a trace say nothing about the kernel -- the roofline's kernel time comes from the bench's
    # one-step-in-flight profiling pass, and that is what these traces must agree with
    bench = [py, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-latency", "--no-pmc", "--no-others", "--lanes", "1"]
    for cfg in (2, 3, 4, 5):
        trace(out, "config%d" % cfg, bench + ["--config", str(cfg)])
    txt = pmc(out, "config2", bench + ["--steps", "2", "--warmup", "1", "--preroll-s", "0", "--sustained-s", "0"], "fused4k|lds_correlate|lds_forward|best_doppler")
    # measured HBM traffic of the dominant kernel of the bench command -> profiles/traffic_latest.json (bench.py measures this live; the file is its labelled fall-back)
    m = re.search(r"== (lds_fused4k_kernel[^\n]*)\n(.*?)(?=\n== |\Z)", txt, re.S)
    if m:
        blk = m.group(2)
        f = float(re.search(r"FETCH_SIZE\s+([\d.e+]+)", blk).group(1))
        w = float(re.search(r"WRITE_SIZE\s+([\d.e+]+)", blk).group(1))
        busy = re.search(r"VALU pipes busy ([\d.]+) %", blk)
        json.dump({"config": 2, "kernel_stage": "lds_correlate", "kernel": m.group(1).split("   ")[0], "epochs": 1024,
                   "hbm_bytes_per_launch": (2 * f + w) * 1024, "fetch_size_kib_raw": f, "write_size_kib": w,
                   "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; bytes=(2*FETCH_SIZE+WRITE_SIZE)*1024 (gfx950 FETCH_SIZE "
                             "counts half of a wide coalesced read, MI355X_MICROARCH.md HBM section)",
                   "source": "profiles/%s_pmc_counters_config2.txt" % tag, "valu_pipe_busy": float(busy.group(1)) / 100 if busy else None},
                  open(os.path.join(out, "traffic_latest.json"), "w"), indent=1)
    # the headline workload in the reference's arithmetic type (engine 5: one fused complex128 kernel per search), its own trace + counters
    c128 = bench + ["--engine", "5", "--steps", "3", "--warmup", "1", "--preroll-s", "0.2", "--sustained-s", "1.0"]
    trace(out, "config2_complex128", c128)
    pmc(out, "config2_complex128", c128 + ["--steps", "2", "--sustained-s", "0", "--preroll-s", "0"], "fused4k_c128|best_doppler64")
    cfgs = [py, os.path.join(ROOT, "tools", "bench_configs.py"), "--reps", "2", "cfg3_e1b", "cfg4_l5i", "cfg4_b2ad_b1", "cfg5_b1i", "cfg5_glonass", "cfg5_e1b", "gps_l1_ms10"]
    pmc(out, "configs345", cfgs, "inner_corr|outer_inverse|lds16k|r32_|lds_inner|outer_forward|inner_forward|lds_correlate|lds_forward")
    # the radix-16 form of the N = 16384 transform (option lds_variant = 16) next to the default radix-32 form, and engine 5's split form
    pmc(out, "n16384_radix16", [py, os.path.join(ROOT, "tools", "bench_configs.py"), "--reps", "2", "--option", "lds_variant=16", "cfg5_b1i", "cfg5_glonass"], "lds16k")
    pmc(out, "complex128_split", [py, os.path.join(ROOT, "tools", "bench_configs.py"), "--reps", "2", "--engine", "5", "cfg3_e1b", "cfg5_b1i"], "c128_split|mix64")
    r = subprocess.run([py, os.path.join(ROOT, "tools", "bench_configs.py"), "--stages"], capture_output=True, text=True)
    open(os.path.join(out, "all_configs_stage_times.log"), "w").write("\n".join(l for l in r.stdout.splitlines() if "amdgpu.ids" not in l) + "\n")


if __name__ == "__main__":
    main()
