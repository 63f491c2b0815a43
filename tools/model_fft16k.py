"""numpy model of the N = 16384 transforms of gacq_ldsfft.hip (forward DIF, inverse DIT, 16 wave-private 1024-point transforms + one
cross-wave radix-16 pass): checks the index algebra end to end against numpy.fft and every LDS access pattern for bank conflicts
(ds_write_b64: 16-lane groups, 8-byte slot mod 16; ds_read_b64: 32-lane groups, slot mod 32).  Run: python tools/model_fft16k.py"""
# numpy model of the wave-private 16384-point transforms (forward DIF, inverse DIT): checks the index algebra and LDS bank conflicts
import numpy as np
N=16384
rng=np.random.default_rng(1)
def rev16(k): return 4*(k&3)+(k>>2)
W=lambda n,e: np.exp(-2j*np.pi*(np.asarray(e)%n)/n)
t=np.arange(1024); w=t>>6; l=t&63

conf={}
def check_write(name, addr):   # addr: [1024] element (8-byte) slot within LDS for one register; groups of 16 contiguous lanes
    a=addr.reshape(-1,16)%16
    worst=max(len(r)-len(set(r)) for r in a.tolist())
    conf[name]=max(conf.get(name,0),worst)
def check_read(name, addr):    # groups of 32 lanes, slot mod 32
    a=addr.reshape(-1,32)%32
    worst=max(len(r)-len(set(r)) for r in a.tolist())
    conf[name]=max(conf.get(name,0),worst)

def dft16(v, inv):   # v [1024,16] natural in -> natural out (model; register order handled by caller)
    k=np.arange(16); M=np.exp((2j if inv else -2j)*np.pi*np.outer(k,k)/16)
    return v@M.T
def dft4cols(v, inv):  # on regs (a, a+4, a+8, a+12) for a<4; in place, natural
    out=np.empty_like(v)
    s=1j if inv else -1j
    for a in range(4):
        x=[v[:,a+4*i] for i in range(4)]
        for k in range(4):
            out[:,a+4*k]=sum(x[i]*(s**(i*k)) for i in range(4))
    return out

def fwd(x):
    lds=np.zeros(16*1056,complex)
    v=x[t[:,None]+1024*np.arange(16)[None,:]]               # v[t,j]
    v=dft16(v,False)                                        # -> ka
    v=v*W(N, t[:,None]*np.arange(16)[None,:])
    for ka in range(16):
        a=ka*1056+t; check_write('x0w',a); lds[a]=v[:,ka]
    v=np.stack([lds[w*1056+l+64*j] for j in range(16)],1)
    for j in range(16): check_read('x0r', w*1056+l+64*j)
    v=dft16(v,False)                                        # over j' -> k0
    v=v*W(1024, l[:,None]*np.arange(16)[None,:])
    base=w*1056
    for k0 in range(16):
        a=base+66*k0+l; check_write('t1w',a); lds[a]=v[:,k0]
    k0p=l&15; llo=l>>4
    vv=np.empty_like(v)
    for lhi in range(16):
        a=base+66*k0p+llo+4*lhi; check_read('t1r',a); vv[:,lhi]=lds[a]
    v=dft16(vv,False)                                       # over l_hi -> k1
    v=v*W(64, llo[:,None]*np.arange(16)[None,:])
    for k1 in range(16):
        a=base+256*llo+16*k1+k0p; check_write('t2w',a); lds[a]=v[:,k1]
    k1lo=l>>4                                               # lane mu = k0'' + 16 k1lo
    vv=np.empty_like(v)
    for k1hi in range(4):
        for lo in range(4):
            k1=k1lo+4*k1hi
            a=base+256*lo+16*k1+k0p; check_read('t2r',a); vv[:,k1hi+4*lo]=lds[a]
    v=dft4cols(vv,False)
    return v   # v[t,r] = X[w + 16 mu + 1024 r]

def inv(Yp):  # Yp[t,r] = Y[w+16mu+1024r]
    lds=np.zeros(16*1056,complex)
    base=w*1056
    k0=l&15; k1lo=l>>4
    v=dft4cols(Yp,True)                                     # reg k1hi + 4 l_lo
    def pi(k0,lo): return 16*((lo>>1)+2*(lo&1))+((k0+8*(lo&1))&15)
    for k1hi in range(4):
        for lo in range(4):
            k1=k1lo+4*k1hi
            a=base+64*k1+16*lo+k0; check_write('t2iw',a); lds[a]=v[:,k1hi+4*lo]
    llo=l>>4; k0p=l&15                                       # lane lambda = l_lo + 4 k0
    vv=np.empty_like(v)
    for k1 in range(16):
        a=base+64*k1+16*llo+k0p; check_read('t2ir',a); vv[:,k1]=lds[a]
    vv=vv*np.conj(W(64, llo[:,None]*np.arange(16)[None,:]))
    v=dft16(vv,True)                                        # -> l_hi
    for lhi in range(16):
        ll=llo+4*lhi
        a=base+65*k0p+ll; check_write('t1iw',a); lds[a]=v[:,lhi]
    vv=np.empty_like(v)
    for k in range(16):
        a=base+65*k+l; check_read('t1ir',a); vv[:,k]=lds[a]
    vv=vv*np.conj(W(1024, l[:,None]*np.arange(16)[None,:]))
    v=dft16(vv,True)                                        # -> j'
    for j in range(16):
        a=base+l+64*j; check_write('x0iw',a); lds[a]=v[:,j]
    vv=np.stack([lds[ka*1056+t] for ka in range(16)],1)
    for ka in range(16): check_read('x0ir',ka*1056+t)
    vv=vv*np.conj(W(N, t[:,None]*np.arange(16)[None,:]))
    v=dft16(vv,True)
    return v   # v[t,j] = y[t+1024j] * N

def run():
    """(forward error, inverse error, {access: worst extra lanes on one bank slot}) for one random row"""
    conf.clear()
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    X = np.fft.fft(x)
    Xp = fwd(x)
    idx = (w[:, None] + 16 * l[:, None] + 1024 * np.arange(16)[None, :])          # register r of lane (w, mu) holds X[w + 16 mu + 1024 r]
    ferr = np.abs(Xp - X[idx]).max() / np.abs(X).max()
    y = inv(Xp) / N
    ierr = np.abs(y - x[t[:, None] + 1024 * np.arange(16)[None, :]]).max()       # register j of lane t holds y[t + 1024 j]
    return ferr, ierr, dict(conf)


if __name__ == "__main__":
    ferr, ierr, c = run()
    print('fwd err', ferr)
    print('inv err', ierr)
    print('conflicts (extra lanes on a slot per group):', c)
