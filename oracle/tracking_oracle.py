"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Sequential restatement of the reference's tracking correlators
(gnsstools/gps/ca.py:120-128 plain; gps/l1cd.py:101-112 BOC(1,1); galileo/e1b.py:45-58 CBOC; gps/l1cp.py:210-228 TMBOC;
gps/l2cm.py:81-92 and gps/l2cl.py RZ), phases advanced by repeated fp64 addition exactly like the reference.
Pinned by tests/golden/tracking_cases.json (tools/make_goldens_tracking.py ran the reference functions)."""
from . import codes_oracle

BOC11 = (1.0, -1.0)
TMBOC = (1, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0)
KIND = {"gps.l1cd": 1, "beidou.b1cd": 1, "beidou.b1cp": 1, "galileo.e1b": 2, "galileo.e1c": 2, "gps.l1cp": 3, "gps.l2cm": 4,
        "gps.l2cl": 5}


def correlate(code, x, prn, chips, frac, incr):
    c = codes_oracle.chips(code, prn)
    L = len(c)
    kind = KIND.get(code, 0)
    p = 0.0j
    cp = (chips + frac) % L
    bp = (2 * (chips + frac)) % 2
    bp6 = (12 * (chips + frac)) % 2
    for i in range(len(x)):
        w = 1.0 - 2.0 * c[int(cp)]
        if kind == 1:
            w *= BOC11[int(bp)]
        elif kind == 2:
            w *= 0.953463 * BOC11[int(bp)] + 0.301511 * BOC11[int(bp6)]
        elif kind == 3:
            w *= BOC11[int(bp6)] if TMBOC[int(cp % 33)] else BOC11[int(bp)]
        elif kind == 4:
            w *= (1.0, 0.0)[int(bp)]
        elif kind == 5:
            w *= (0.0, 1.0)[int(bp)]
        p += x[i] * w
        cp = (cp + incr) % L
        bp = (bp + 2 * incr) % 2
        bp6 = (bp6 + 12 * incr) % 2
    return p
