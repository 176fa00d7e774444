import numpy as np
from scipy.special import erfc, log_ndtr
from numpy.polynomial import chebyshev as C
ZMAX=6.0
z=np.linspace(0,ZMAX,20001)
q=log_ndtr(-z)/np.log(2.0)          # log2(Phi(-z))
phi=np.exp(log_ndtr(-z))
w=z*phi*np.log(2)+1e-9
def fit(n, iters=60):
    # iteratively reweighted LS -> approx minimax of abs error in z*2^q
    ww=w.copy()
    for it in range(iters):
        V=np.vander(z,n+1,increasing=True)
        c,*_=np.linalg.lstsq(V*ww[:,None], q*ww, rcond=None)
        err=np.abs(z*(2.0**(V@c))-z*phi)
        ww=ww*(1+ 2*err/err.max())**0.5
        ww/=ww.max()
    return c
def evalf32(c,g):
    g=g.astype(np.float32)
    zz=np.minimum(np.abs(g),np.float32(ZMAX))
    acc=np.full_like(zz,np.float32(c[-1]))
    for k in range(len(c)-2,-1,-1):
        acc=(acc.astype(np.float64)*zz+np.float64(np.float32(c[k]))).astype(np.float32)   # fma-ish
    E=np.exp2(acc.astype(np.float64)).astype(np.float32)
    relu=np.maximum(g,np.float32(0))
    return (relu.astype(np.float64)-zz.astype(np.float64)*E).astype(np.float32)
g=np.linspace(-12,12,2400001)
from scipy.special import ndtr
ref=g*ndtr(g)
for n in range(4,10):
    c=fit(n)
    out=evalf32(c,g)
    e=np.abs(out-ref)
    rel=e/np.maximum(np.abs(ref),1e-3)
    print(n,'max abs',e.max(),'at',g[e.argmax()],'max rel(>1e-3)',rel.max(), 'lead',c[-1])
    if n in(6,7): print(repr(c.astype(np.float32)))
print()
c=fit(6,iters=200)
print(', '.join('%.9gf'%np.float32(x) for x in c))
out=evalf32(c,g); e=np.abs(out-ref); print('deg6 max abs',e.max(), 'rel', (e/np.maximum(np.abs(ref),1e-3)).max())
# old formula error for comparison
def old(x):
    x=x.astype(np.float32); z=np.abs(x)*np.float32(0.70710678); t=1/(1+np.float32(0.3275911)*z)
    poly=((((np.float32(1.061405429)*t-np.float32(1.453152027))*t+np.float32(1.421413741))*t-np.float32(0.284496736))*t+np.float32(0.254829592))*t
    er=1-poly*np.exp(-z*z); return 0.5*x*(1+np.copysign(er,x))
e=np.abs(old(g)-ref); print('old max abs',e.max(),'rel',(e/np.maximum(np.abs(ref),1e-3)).max())
