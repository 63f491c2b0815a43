"""numpy model of the radix-32 form of the N = 16384 transforms (gacq_lds16k.hip): one 512-thread workgroup, 32 points per lane,
16384 = 32 (across the waves) x 32 x 16 (inside a 16-lane group).  Checks the index algebra end to end against numpy.fft and every
LDS access pattern for bank conflicts (ds_write_b64: 16-lane groups, 8-byte slot mod 16; ds_read_b64: 32-lane groups, slot mod 32;
ds_read_b128 of the staged row: 16-byte slots).  Run: python tools/model_fft16k_r32.py"""
import numpy as np

N = 16384
T = 512                 # threads
S = 560                 # region stride (complex elements): >= 544 (pitch-17 transpose) and = 16 (mod 32)
PITCH = 17
rng = np.random.default_rng(1)
W = lambda n, e: np.exp(-2j * np.pi * (np.asarray(e) % n) / n)
t = np.arange(T)
w = t >> 6
l = t & 63
g = l >> 4              # 16-lane group of the wave
lam = l & 15            # lane within the group
ka_own = 4 * w + g      # the 512-point sub-transform this group owns

conf = {}


def check_write(name, addr):
    a = addr.reshape(-1, 16) % 16
    conf[name] = max(conf.get(name, 0), max(len(r) - len(set(r)) for r in a.tolist()))


def check_read(name, addr):
    a = addr.reshape(-1, 32) % 32
    conf[name] = max(conf.get(name, 0), max(len(r) - len(set(r)) for r in a.tolist()))


def dft(v, n, inv):      # v [T, n] natural in -> natural out
    k = np.arange(n)
    M = np.exp((2j if inv else -2j) * np.pi * np.outer(k, k) / n)
    return v @ M.T


def fwd(x):
    """v[t, j] = x[t + 512 j]  ->  out[t', r] = X[(t' >> 4) + 32 (t' & 15) + 512 r]"""
    lds = np.zeros(32 * S, complex)
    v = x[t[:, None] + T * np.arange(32)[None, :]]
    v = dft(v, 32, False)                                      # over j -> ka
    v = v * W(N, t[:, None] * np.arange(32)[None, :])          # table A
    for ka in range(32):
        a = ka * S + t
        check_write('x0w', a)
        lds[a] = v[:, ka]
    base = ka_own * S
    u = np.empty_like(v)
    for jp in range(32):
        a = base + lam + 16 * jp
        check_read('x0r', a)
        u[:, jp] = lds[a]
    v = dft(u, 32, False)                                      # over j' -> k0
    v = v * W(512, lam[:, None] * np.arange(32)[None, :])      # table B
    for k0 in range(32):
        a = base + PITCH * k0 + lam
        check_write('t1w', a)
        lds[a] = v[:, k0]
    out = np.empty_like(v)
    for h in range(2):
        vv = np.empty((T, 16), complex)
        for lp in range(16):
            a = base + PITCH * (lam + 16 * h) + lp              # lane mu = lam holds k0 = mu + 16 h
            check_read('t1r', a)
            vv[:, lp] = lds[a]
        vv = dft(vv, 16, False)                                # over lambda -> k1
        for k1 in range(16):
            out[:, h + 2 * k1] = vv[:, k1]
    return out


def inv(Yp):
    """Yp[t', r] = Y[(t' >> 4) + 32 (t' & 15) + 512 r]  ->  v[t, j] = N y[t + 512 j]"""
    lds = np.zeros(32 * S, complex)
    base = ka_own * S
    for h in range(2):
        vv = np.stack([Yp[:, h + 2 * k1] for k1 in range(16)], 1)
        vv = dft(vv, 16, True)                                 # over k1 -> lambda
        for lp in range(16):
            a = base + PITCH * (lam + 16 * h) + lp
            check_write('t1iw', a)
            lds[a] = vv[:, lp]
    v = np.empty((T, 32), complex)
    for k0 in range(32):
        a = base + PITCH * k0 + lam
        check_read('t1ir', a)
        v[:, k0] = lds[a]
    v = v * np.conj(W(512, lam[:, None] * np.arange(32)[None, :]))
    v = dft(v, 32, True)                                       # over k0 -> j'
    for jp in range(32):
        a = base + lam + 16 * jp
        check_write('x0iw', a)
        lds[a] = v[:, jp]
    vv = np.empty_like(v)
    for ka in range(32):
        a = ka * S + t
        check_read('x0ir', a)
        vv[:, ka] = lds[a]
    vv = vv * np.conj(W(N, t[:, None] * np.arange(32)[None, :]))
    return dft(vv, 32, True)


def staged_row_conflicts():
    """LDS-DMA of a spectrum row: piece p (16 bytes per lane) of wave w lands at byte 4 w S 8 + 1024 p + 16 l; read back with ds_read_b128
    (16-lane groups of the guide's table are contiguous 256-byte runs here: conflict-free by construction).  Returns the highest byte used."""
    return 4 * 7 * S * 8 + 1024 * 15 + 16 * 63 + 16


def run():
    conf.clear()
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    X = np.fft.fft(x)
    Xp = fwd(x)
    idx = (t[:, None] >> 4) + 32 * (t[:, None] & 15) + 512 * np.arange(32)[None, :]
    ferr = np.abs(Xp - X[idx]).max() / np.abs(X).max()
    y = inv(Xp) / N
    ierr = np.abs(y - x[t[:, None] + T * np.arange(32)[None, :]]).max()
    assert sorted(idx.ravel().tolist()) == list(range(N))
    return ferr, ierr, dict(conf)


if __name__ == "__main__":
    ferr, ierr, c = run()
    print('fwd err', ferr)
    print('inv err', ierr)
    print('conflicts (extra lanes on a slot per group):', c)
    print('staged row: highest LDS byte', staged_row_conflicts(), 'of', 32 * S * 8)
