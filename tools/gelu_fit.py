import numpy as np
from scipy.special import erf
from numpy.polynomial import chebyshev as C, polynomial as P
def Phi(x): return 0.5*(1+erf(x/np.sqrt(2)))
def pdf(x): return np.exp(-x*x/2)/np.sqrt(2*np.pi)
def fit(fun_odd_over_x, X0, nterms):
    n=8000
    u = (np.cos(np.pi*(np.arange(n)+0.5)/n)+1)/2*X0**2
    x = np.sqrt(u); t = 2*u/X0**2-1
    V = C.chebvander(t, nterms-1)
    coef = np.linalg.lstsq(V*x[:,None], fun_odd_over_x(x)*x, rcond=None)[0]
    return C.cheb2poly(coef)   # monomial in t
def horner32(a, t):
    r = np.full_like(t, np.float32(a[-1]))
    for c in a[-2::-1]:
        r = (r*t + np.float32(c)).astype(np.float32)   # not fused but close
    return r
for X0 in (4.0, 4.25, 4.5):
  for nt in (9,10,11,12):
    aF = fit(lambda x:(Phi(x)-0.5)/x, X0, nt)
    aG = fit(lambda x:(Phi(x)-0.5+x*pdf(x))/x, X0, nt)
    xs = np.linspace(-X0, X0, 400001).astype(np.float32)
    t = (xs*xs*np.float32(2/X0**2) - np.float32(1)).astype(np.float32)
    qF = horner32(aF.astype(np.float32), t); qG = horner32(aG.astype(np.float32), t)
    phi = (np.float32(0.5) + xs*qF).astype(np.float32)
    gp = (np.float32(0.5) + xs*qG).astype(np.float32)
    x64 = xs.astype(np.float64)
    eF = np.abs(phi - Phi(x64)).max(); eG = np.abs(gp - (Phi(x64)+x64*pdf(x64))).max()
    eg = np.abs(xs*phi - x64*Phi(x64)).max()
    print(f"X0={X0} nt={nt}: dPhi={eF:.2e} dgelu={eg:.2e} dgrad={eG:.2e}  min phi={phi.min():.2e} max={phi.max():.8f} maxcoef={np.abs(aF).max():.2f}/{np.abs(aG).max():.2f}")
print("----")
X0, nt = 4.5, 12
aF = fit(lambda x:(Phi(x)-0.5)/x, X0, nt).astype(np.float32)
aG = fit(lambda x:(Phi(x)-0.5+x*pdf(x))/x, X0, nt).astype(np.float32)
print("F:", ", ".join(f"{float(c):.9e}f" for c in aF))
print("G:", ", ".join(f"{float(c):.9e}f" for c in aG))
print("scale", repr(np.float32(2/X0**2)))
