"""numpy model of the prime-factor (Good-Thomas) form of the split engine for N = 31 * M, M = 1980 = 11*20*9 / 990 = 11*10*9
(gacq_pfa.hip): the 4-D transform 31 x Na x Nb x Nc has no twiddle factors at any level.  The model replays the index algebra of the four
kernels -- forward outer (rotated gather + DFT-31), forward inner (three in-place passes in LDS), inverse inner (C conj(X) + three passes),
inverse outer (DFT-31 + lag labels) -- against numpy.fft, and checks every LDS access pattern of the inner kernels for bank conflicts
(ds_write_b64: 16-lane groups, 8-byte slot mod 16; ds_read_b64: 32-lane groups, slot mod 32; MI355X guide, LDS table).
Run: python tools/model_pfa.py"""
import itertools

import numpy as np

R = 31


def egcd_inv(a, m):
    return pow(a % m, -1, m)


class Shape:
    """N = 31 * M, M = Na*Nb*Nc pairwise coprime.  Time index n <-> (n mod 31, n mod Na, n mod Nb, n mod Nc) (CRT)."""

    def __init__(self, M, dims, seg):
        self.M, self.N = M, R * M
        self.Na, self.Nb, self.Nc = dims
        assert self.Na * self.Nb * self.Nc == M
        self.seg = seg                      # pitch of one c-segment of a Z' row (>= Na*Nb, whole 128-byte lines)
        self.Mp = self.Nc * seg             # Z' row pitch
        self.Mm = M % R
        self.Minv = egcd_inv(M, R)
        # idempotents of the CRT inside M
        self.ea = (M // self.Na) * egcd_inv(M // self.Na, self.Na)
        self.eb = (M // self.Nb) * egcd_inv(M // self.Nb, self.Nb)
        self.ec = (M // self.Nc) * egcd_inv(M // self.Nc, self.Nc)

    def t_of(self, a, b, c):
        return (a * self.ea + b * self.eb + c * self.ec) % self.M

    def lz(self, a, b, c, seg=None):        # column layout (A rows, Z' rows)
        return c * (self.seg if seg is None else seg) + b * self.Na + a

    def lg(self, a, b, c):                  # spectrum layout (X rows, C rows)
        return a * (self.Nb * self.Nc) + self.Nb * c + b

    def q0_of(self, t):
        return ((R - t % R) * self.Minv) % R


SHAPES = {1980: Shape(1980, (11, 20, 9), 224), 990: Shape(990, (11, 10, 9), 112)}

def dft(v, inv, axis):
    n = v.shape[axis]
    k = np.arange(n)
    Wm = np.exp((2j if inv else -2j) * np.pi * np.outer(k, k) / n)
    return np.moveaxis(np.tensordot(Wm, np.moveaxis(v, axis, 0), axes=(1, 0)), 0, axis)


def forward_outer(sh, x):
    """x[N] natural order -> A[31][Lz unpadded], row k1, via rotated gather + standard DFT-31 + row permutation."""
    A = np.zeros((R, sh.M), complex)
    for a, b, c in itertools.product(range(sh.Na), range(sh.Nb), range(sh.Nc)):
        t = sh.t_of(a, b, c)
        q0 = sh.q0_of(t)
        xs = np.array([x[t + sh.M * ((q0 + u) % R)] for u in range(R)])       # slot u: time coordinate n1 = Mm u mod 31
        assert all((t + sh.M * ((q0 + u) % R)) % R == (sh.Mm * u) % R for u in range(R))
        Y = np.fft.fft(xs)
        pos = sh.lz(a, b, c, sh.Na * sh.Nb)
        for kp in range(R):
            A[(sh.Minv * kp) % R, pos] = Y[kp]
    return A


def small_dft(v, inv):
    """v[threads, R] -> DFT along axis 1 (the register butterfly of one pass)."""
    n = v.shape[1]
    k = np.arange(n)
    return v @ np.exp((2j if inv else -2j) * np.pi * np.outer(k, k) / n)


def inner(sh, rows, inv, seg):
    """The three in-place passes of pfa_inner_forward_kernel (inv False: column layout with segment pitch seg in, spectrum layout out;
    passes c, b, a) / pfa_inner_corr_kernel (inv True: spectrum layout in, column layout with pitch seg out; passes a, b, c), with the
    kernels' own thread -> address expressions on an explicit LDS array L(a,b,c) = a + 11 b + AB c."""
    Na, Nb, Nc = sh.Na, sh.Nb, sh.Nc
    AB, BC = Na * Nb, Nb * Nc
    pb = np.array(pass_b_lanes(sh))
    pb = pb[pb >= 0]
    j = np.arange(BC)                       # pass a threads: spectrum position j = Nb kc + kb
    i = np.arange(AB)                       # pass c threads: column position i = 11 b + a
    out = np.zeros((rows.shape[0], Nc * seg if inv else sh.M), complex)
    for r in range(rows.shape[0]):
        lds = np.zeros(sh.M, complex)

        def pass_a():
            if inv:
                v = np.stack([rows[r, t * BC + j] for t in range(Na)], 1)
            else:
                v = np.stack([lds[Na * j + t] for t in range(Na)], 1)
            v = small_dft(v, inv)
            for t in range(Na):
                if inv:
                    lds[Na * j + t] = v[:, t]
                else:
                    out[r, t * BC + j] = v[:, t]

        def pass_b():
            v = np.stack([lds[pb + Na * t] for t in range(Nb)], 1)
            v = small_dft(v, inv)
            for t in range(Nb):
                lds[pb + Na * t] = v[:, t]

        def pass_c():
            if inv:
                v = np.stack([lds[i + AB * t] for t in range(Nc)], 1)
            else:
                v = np.stack([rows[r, t * seg + i] for t in range(Nc)], 1)
            v = small_dft(v, inv)
            for t in range(Nc):
                if inv:
                    out[r, t * seg + i] = v[:, t]
                else:
                    lds[i + AB * t] = v[:, t]

        for p in ((pass_a, pass_b, pass_c) if inv else (pass_c, pass_b, pass_a)):
            p()
    return out


def inverse_outer(sh, Z):
    """Z[31][Mp] (Lz padded) -> z[N] natural order."""
    z = np.zeros(sh.N, complex)
    for a, b, c in itertools.product(range(sh.Na), range(sh.Nb), range(sh.Nc)):
        t = sh.t_of(a, b, c)
        q0 = sh.q0_of(t)
        pos = sh.lz(a, b, c)
        v = np.array([Z[(sh.Minv * kp) % R, pos] for kp in range(R)])
        zz = np.fft.ifft(v) * R
        for u in range(R):
            z[t + sh.M * ((q0 + u) % R)] = zz[u]
    return z


def correlate(sh, x, c):
    """ifft(fft(c) * conj(fft(x))) * N through the four model kernels."""
    segu = sh.Na * sh.Nb
    X = inner(sh, forward_outer(sh, x), False, segu)
    C = inner(sh, forward_outer(sh, c), False, segu)
    Z = inner(sh, C * np.conj(X), True, sh.seg)
    return inverse_outer(sh, Z)


# ---- LDS bank conflicts of the inner kernels (buffer L(a,b,c) = a + 11 b + 11 Nb c) ------------------------------------------------
def group_extra(addr, width):
    """addr: 8-byte slots of consecutive lanes; extra LDS cycles = sum over lane groups of (largest bank multiplicity - 1)."""
    extra = 0
    for g0 in range(0, len(addr), width):
        a = [x % width for x in addr[g0:g0 + width] if x >= 0]
        if a:
            extra += max(a.count(v) for v in set(a)) - 1
    return extra


def pass_b_lanes(sh):
    """pfa_pass_b_base of gacq_pfa.hip: lane l of 32-lane group g takes the g-th butterfly (a, c) with (a + AB c) mod 32 == l."""
    AB = sh.Na * sh.Nb
    out = []
    for tid in range(128):
        l, g, cnt, pb = tid & 31, tid >> 5, 0, -1
        for c in range(sh.Nc):
            a = (l - (AB % 32) * c) & 31
            if a < sh.Na:
                if cnt == g:
                    pb = a + AB * c
                cnt += 1
        out.append(pb)
    return out


def conflicts(sh):
    AB, BC = sh.Na * sh.Nb, sh.Nb * sh.Nc
    res = {}
    # pass a: thread j < BC, slot 11 j + a   (inverse kernel writes, forward kernel reads)
    res['a_write'] = sum(group_extra([sh.Na * j + a for j in range(BC)], 16) for a in range(sh.Na))
    res['a_read'] = sum(group_extra([sh.Na * j + a for j in range(BC)], 32) for a in range(sh.Na))
    # pass c: thread i < AB, slot i + AB c    (forward kernel writes, inverse kernel reads)
    res['c_write'] = sum(group_extra([i + AB * c for i in range(AB)], 16) for c in range(sh.Nc))
    res['c_read'] = sum(group_extra([i + AB * c for i in range(AB)], 32) for c in range(sh.Nc))
    # pass b: table lanes, slot pb + 11 b
    pb = pass_b_lanes(sh)
    covered = sorted(x for x in pb if x >= 0)
    want = sorted(a + AB * c for a in range(sh.Na) for c in range(sh.Nc))
    res['b_cover'] = 0 if covered == want else 1
    res['b_read'] = sum(group_extra([x + sh.Na * b if x >= 0 else -1 for x in pb], 32) for b in range(sh.Nb))
    res['b_write'] = sum(group_extra([x + sh.Na * b if x >= 0 else -1 for x in pb], 16) for b in range(sh.Nb))
    return res


def run():
    rng = np.random.default_rng(5)
    errs, conf = {}, {}
    for M, sh in SHAPES.items():
        x = rng.standard_normal(sh.N) + 1j * rng.standard_normal(sh.N)
        c = np.where(rng.random(sh.N) < 0.5, 1.0, -1.0) + 0j
        ref = np.fft.ifft(np.fft.fft(c) * np.conj(np.fft.fft(x))) * sh.N
        got = correlate(sh, x, c)
        errs[M] = float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
        conf[M] = conflicts(sh)
    return errs, conf


if __name__ == "__main__":
    print(run())
