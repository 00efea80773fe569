import mpmath as mp, random, struct
mp.mp.prec = 240
ln2 = mp.log(2)
a = ln2/2 * mp.mpf('1.000001')
def h(r):
    r = mp.mpf(r)
    if abs(r) < mp.mpf(10)**-12: return mp.mpf(1)/6 + r/24 + r*r/120
    return (mp.exp(r) - 1 - r - r*r/2)/(r**3)
def fma(a,b,c): return float(mp.mpf(a)*mp.mpf(b)+mp.mpf(c))
for NH in (8, 9):
    coef, err = mp.chebyfit(h, [-a, a], NH, error=True)
    c = [float(x) for x in coef][::-1]   # c[0] = c3
    print("exp degree", NH+2, "fit err h", mp.nstr(err,5), " -> exp abs err", mp.nstr(err*a**3,5))
    random.seed(2); worst=0; wr=0
    for i in range(40000):
        r = random.uniform(-float(a), float(a))
        p = c[-1]
        for k in range(len(c)-2, -1, -1): p = fma(p, r, c[k])
        p = fma(p, r, 0.5); p = fma(p, r, 1.0); p = fma(p, r, 1.0)
        ex = mp.exp(mp.mpf(r)); ulp = mp.mpf(2)**(mp.floor(mp.log(ex,2))-52)
        e = abs(mp.mpf(p)-ex)/ulp
        if e > worst: worst=e; wr=r
    print("  worst ulp", mp.nstr(worst,5), "at r", wr)
    for i,x in enumerate(c): print("  c%d = %.17g  (%s)" % (i+3, x, x.hex()))
