"""Dev tool: the polynomial of gelu_erf (pio_gemm_common.h).  log2(0.5 erfc(a / sqrt 2)) on [0, 6] by weighted minimax
(Lawson iteration on a Chebyshev basis; weight a * E = d gelu / d log2 E), printed per degree with the fp32 error of
gelu(x) = max(x, 0) - |x| 2^p(min(|x|, 6)) against the float64 erf form."""
import numpy as np
from scipy.special import erfc, erf
np.set_printoptions(precision=17)
A=6.0
a=np.linspace(0,A,200001)
Ep=0.5*erfc(a/np.sqrt(2))
f=np.log2(Ep)
w0=np.maximum(a*Ep*np.log(2), 1e-9)   # d y / d P
def fit(deg, iters=60):
    w=w0.copy()
    V=np.polynomial.chebyshev.chebvander(2*a/A-1, deg)
    for it in range(iters):
        c,*_=np.linalg.lstsq(V*w[:,None], f*w, rcond=None)
        err=np.abs((V@c-f)*w0)
        w=w*(0.5+err/err.max())   # Lawson-like
        w/=w.max()/w0.max()
    return c, err.max()
for deg in range(5,13):
    c,e=fit(deg)
    # convert to monomial in a
    p=np.polynomial.chebyshev.Chebyshev(c, domain=[0,A]).convert(kind=np.polynomial.Polynomial)
    coef=p.coef
    # fp32 evaluation
    x=np.linspace(-8,8,400001).astype(np.float32)
    ax=np.minimum(np.abs(x),np.float32(A)).astype(np.float32)
    P=np.full_like(ax, np.float32(coef[-1]))
    for k in range(deg-1,-1,-1):
        P=(P*ax+np.float32(coef[k])).astype(np.float32)
    E=np.exp2(P.astype(np.float32)).astype(np.float32)
    y=(np.maximum(x,0)-np.abs(x)*E).astype(np.float32)
    xd=x.astype(np.float64)
    yref=0.5*xd*(1+erf(xd/np.sqrt(2)))
    print(deg, 'weighted fit err %.2e'%e, 'gelu max abs err fp32 %.3e'%np.abs(y-yref).max(), 'at', x[np.abs(y-yref).argmax()])
    if deg in (8,9,10):
        print('  coef', [float(np.float32(v)) for v in coef])
