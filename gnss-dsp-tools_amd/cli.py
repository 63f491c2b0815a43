"""Command-line shim with the argument surface and output lines of the reference's acquire scripts:

    python -m gnss_dsp_tools_amd.cli gps-l1 [--prn 1-32] [--doppler-search MIN,MAX,INCR] [--time MS] FILE FS COFFSET
    python -m gnss_dsp_tools_amd.cli glonass-l1 [--channel -7:7] ... FILE FS COFFSET

== acquire-gps-l1.py:46-111 (options, positionals, one result line per item in input order).  The search itself
runs on the GPU (acquire.Engine.search_all replaces the multiprocessing.Pool over PRNs, :105-108)."""
import argparse
import sys

LONGCODE = {"gps-l2cl": 40, "glonass-l1-p": 80, "glonass-l2-p": 80}        # default --time (acquire-gps-l2cl.py:57)
# Run as a program, the FFT searches need no device tensor (gacq_acquire_int8): the package is then loaded without torch, whose import
# alone (~0.8 s) would be most of the run (_native.py).  The long-code commands hand device tensors around and keep it.
GACQ_NO_TORCH = __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] in LONGCODE)

from . import acquire, codes, frontend, signals  # noqa: E402


def build_parser(sig):
    ap = argparse.ArgumentParser(prog="acquire-%s" % sig.name, description="Acquire %s signals on MI355X" % sig.name)
    ap.add_argument(sig.item_opt, dest="items", default=sig.default_items,
                    help="items to search, e.g. 1,3,7%s14 (default %%(default)s)" % sig.item_sep)
    ap.add_argument("--doppler-search", metavar="MIN,MAX,INCR", default=",".join("%g" % v for v in sig.default_doppler),
                    help="Doppler search grid: min,max,increment (default %(default)s)")
    ap.add_argument("--time", type=int, default=80, help="integration time in milliseconds (default %(default)s)")
    ap.add_argument("--device", type=int, default=0, help="GPU index")
    ap.add_argument("input_filename")
    ap.add_argument("sample_rate", type=float)
    ap.add_argument("carrier_offset", type=float)
    return ap


_VALUE_OPTS = ("--prn", "--channel", "--doppler-search", "--time", "--device")


def _join_option_values(argv):
    """optparse (the reference) accepts '--doppler-search -7000,7000,200' and '--channel -7:7'; argparse would take
    the value for an option because it starts with '-'.  Rewrite 'opt value' as 'opt=value'."""
    out, i = [], 0
    while i < len(argv):
        if argv[i] in _VALUE_OPTS and i + 1 < len(argv):
            out.append(argv[i] + "=" + argv[i + 1])
            i += 2
        else:
            out.append(argv[i])
            i += 1
    return out


def run(name, argv, out=sys.stdout):
    sig = signals.get(name)
    args = build_parser(sig).parse_args(_join_option_values(list(argv)))
    if args.items:
        items = acquire.parse_list_ranges(args.items, sep=sig.item_sep)
    else:
        items = codes.prns(sig.code)            # acquire-beidou-b2bi.py:70: default = every PRN in the table
    doppler_search = acquire.parse_list_floats(args.doppler_search)
    ms = args.time
    ms_pad = ms + 5                             # acquire-gps-l1.py:80
    n = int(args.sample_rate * 0.001 * ms_pad)
    with open(args.input_filename, "rb") as fp:
        raw = fp.read(2 * n)
    if len(raw) != 2 * n:
        raise SystemExit("input file too short: need %d complex int8 samples" % n)     # gnsstools/io.py:5-6 returns None here
    import numpy as np
    eng = acquire.Engine(args.device)
    try:
        # file bytes -> GPU once; front-end and search stay device-resident (gacq_acquire_int8: nothing on this path imports torch,
        # which alone would be most of the run's wall time)
        iq = np.frombuffer(raw, dtype=np.int8)
        dop = acquire.doppler_grid(doppler_search)
        results = eng.acquire_int8(sig, iq, args.sample_rate, args.carrier_offset, ms_pad, items, dop, max(sig.blocks(ms), 0))
    finally:
        eng.close()
    lines = [acquire.format_result(sig, it, r) for it, r in zip(items, results)]
    for line in lines:
        print(line, file=out)
    return lines


def run_longcode(name, argv, out=sys.stdout):
    """acquire-gps-l2cl.py / acquire-glonass-l{1,2}-p.py: FILE FS COFFSET ITEM DOPPLER CODE_PHASE [--time MS]."""
    from . import longcode
    ap = argparse.ArgumentParser(prog="acquire-%s" % name)
    ap.add_argument("--time", type=int, default=LONGCODE[name])
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("input_filename")
    ap.add_argument("sample_rate", type=float)
    ap.add_argument("carrier_offset", type=float)
    ap.add_argument("item", type=int)
    ap.add_argument("doppler", type=float)
    ap.add_argument("code_phase", type=float)
    a = ap.parse_args(_join_option_values(list(argv)))
    ms_pad = a.time + 5
    n = int(a.sample_rate * 0.001 * ms_pad)
    with open(a.input_filename, "rb") as fp:
        x = frontend.read_iq_int8(fp, n)
    if x is None:
        raise SystemExit("input file too short: need %d complex int8 samples" % n)
    eng = acquire.Engine(a.device)
    try:
        # the raw int8 samples go to the GPU; nco.mix(x,-coffset/fs,0) (acquire-gps-l2cl.py:72) runs there too
        if name == "gps-l2cl":
            metric, k = longcode.search_l2cl(x, a.item, a.doppler, a.code_phase, a.time, a.sample_rate, engine=eng, coffset=a.carrier_offset)
            line = '%f %f' % (10230 * k + a.code_phase, metric)                  # acquire-gps-l2cl.py:76
        else:
            metric, k = longcode.search_glonass_p(x, a.item, a.doppler, a.code_phase, a.time, a.sample_rate,
                                                  band=name.split("-")[1], engine=eng, coffset=a.carrier_offset)
            line = '%f %f' % (5110 * k + 10 * a.code_phase, metric)              # acquire-glonass-l1-p.py:79
    finally:
        eng.close()
    print(line, file=out)
    return [line]


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help"):
        print(__doc__)
        print("signals:", ", ".join(sorted(signals.SIGNALS)), "| long-code:", ", ".join(sorted(LONGCODE)))
        return 0
    if argv[0] in LONGCODE:
        run_longcode(argv[0], argv[1:])
    else:
        run(argv[0], argv[1:])
    return 0


if __name__ == "__main__":
    sys.exit(main())
