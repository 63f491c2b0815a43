#!/usr/bin/env python3
"""Hardware counters per kernel for an arbitrary command, one rocprofv3 --pmc pass per counter group (the SQ block has 8 slots,
TCC 4; FETCH_SIZE / WRITE_SIZE need their own passes -- MI355X_MICROARCH.md "rocprofv3 PMC slots").  Counter passes are never
combined with tracing.  usage (GPU box): tools/pmc_profile.py <outdir> [--groups a,b,..] [--filter REGEX] -- <command ...>
Prints, per kernel matching REGEX, the average per launch of every counter and a few derived ratios."""
import csv
import glob
import os
import re
import subprocess
import sys
from collections import defaultdict

GROUPS = {
    "sq_time": "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS",
    "sq_inst": "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_SALU",
    "grbm": "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS",
    "mfma": "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_BUSY_CYCLES",
    "fetch": "FETCH_SIZE",
    "write": "WRITE_SIZE",
    "tcc": "TCC_EA0_WRREQ_STALL TCC_TOO_MANY_EA_WRREQS_STALL TCC_TAG_STALL TCC_BUSY",
    "tcc_hit": "TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_EA0_WRREQ",
    "tcp": "TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES",
}


def main():
    argv = sys.argv[1:]
    cut = argv.index("--")
    opts, cmd = argv[:cut], argv[cut + 1:]
    out = os.path.abspath(opts[0])
    groups = list(GROUPS)
    filt = ".*"
    i = 1
    while i < len(opts):
        if opts[i] == "--groups":
            groups = opts[i + 1].split(",")
        elif opts[i] == "--filter":
            filt = opts[i + 1]
        i += 2
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
    for g in groups:
        d = os.path.join(out, g)
        r = subprocess.run(["rocprofv3", "--pmc"] + GROUPS[g].split() + ["--output-format", "csv", "-d", d, "-o", "p", "--"] + cmd,
                           cwd="/tmp", env=env, capture_output=True, text=True)
        open(os.path.join(out, g + ".log"), "w").write(r.stdout[-4000:] + "\n" + r.stderr[-4000:])
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(path)):
                acc[row["Kernel_Name"]][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"] or 0)
    rx = re.compile(filt)
    for k, cs in sorted(acc.items()):
        if not rx.search(k):
            continue
        short = re.sub(r"\(.*", "", k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", ""))[:110]
        avg = {c: sum(v.values()) / len(v) for c, v in cs.items()}
        n = max(len(v) for v in cs.values())
        print("== %s   (%d launches per pass)" % (short, n))
        for c in sorted(avg):
            print("   %-32s %.5g" % (c, avg[c]))
        g = avg.get
        if g("SQ_WAVE_CYCLES"):
            print("   -> wave time: issuing %.0f %% / issue-stalled %.0f %% / parked (s_waitcnt, barrier) %.0f %%" % (
                100 * (g("SQ_WAVE_CYCLES") - g("SQ_WAIT_ANY", 0) - g("SQ_WAIT_INST_ANY", 0)) / g("SQ_WAVE_CYCLES"),
                100 * g("SQ_WAIT_INST_ANY", 0) / g("SQ_WAVE_CYCLES"), 100 * g("SQ_WAIT_ANY", 0) / g("SQ_WAVE_CYCLES")))
        if g("SQ_BUSY_CYCLES") and g("SQ_ACTIVE_INST_VALU"):
            # SQ_BUSY_CYCLES counts per SQ (one per shader engine x XCD, 32 in all) in quad-cycles? calibrate with GRBM_GUI_ACTIVE instead
            pass
        if g("GRBM_GUI_ACTIVE") and g("SQ_ACTIVE_INST_VALU"):
            dur = g("GRBM_GUI_ACTIVE") / 8.0                  # summed over 8 XCDs
            print("   -> duration %.0f cycles per XCD; VALU pipes busy %.1f %%; LDS instr busy %.1f %%; VMEM instr busy %.1f %%" % (
                dur, 100 * 4 * g("SQ_ACTIVE_INST_VALU") / (dur * 1024), 100 * 4 * g("SQ_ACTIVE_INST_LDS", 0) / (dur * 1024),
                100 * 4 * g("SQ_ACTIVE_INST_VMEM", 0) / (dur * 1024)))
        if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
            print("   -> HBM bytes per launch: fetch %.4g (2 x FETCH_SIZE KiB, gfx950 correction) + write %.4g = %.4g" % (
                2 * g("FETCH_SIZE") * 1024, g("WRITE_SIZE") * 1024, (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024))
        if g("TCC_HIT") is not None:
            print("   -> L2 hit rate %.1f %%" % (100 * g("TCC_HIT") / max(1.0, g("TCC_HIT") + g("TCC_MISS", 0))))


if __name__ == "__main__":
    main()
