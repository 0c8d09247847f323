"""Exact emulation (Decimal arithmetic, correctly rounded fma) of the two exp(-t) kernels of csrc/exp_neg.h against a
60-digit reference: largest error in ulp of the result over random t in [0, 40] plus the neighbourhood of the table's
break points.  usage: python tools/exp_neg_error.py [samples]"""
import random, struct, sys
from decimal import Decimal, getcontext
getcontext().prec = 80
LN2 = Decimal(2).ln()
MAGIC = 6755399441055744.0

def fma(a, b, c):
    return float(Decimal(a) * Decimal(b) + Decimal(c))

def ulp(x):
    b = struct.unpack("<q", struct.pack("<d", x))[0]
    return struct.unpack("<d", struct.pack("<q", b + 1))[0] - x

def make(entries, degree):
    tab = [float((LN2 * Decimal(j) / Decimal(entries)).exp()) for j in range(entries)]
    c1 = float(Decimal(entries) / LN2)
    hi = float(LN2 / Decimal(entries))
    lo = float(LN2 / Decimal(entries) - Decimal(hi))
    shift = entries.bit_length() - 1
    coef = {6: [1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5, 1.0], 5: [1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5, 1.0],
            4: [1.0 / 24.0, 1.0 / 6.0, 0.5, 1.0]}[degree]

    def f(t):
        u = -t
        z = fma(u, c1, MAGIC)
        ni = struct.unpack("<q", struct.pack("<d", z))[0] & 0xFFFFFFFF
        if ni >= 1 << 31:
            ni -= 1 << 32
        fn = z - MAGIC
        r = fma(fn, -hi, u)
        r = fma(fn, -lo, r)
        T = tab[ni & (entries - 1)]
        q = coef[0]
        for c in coef[1:]:
            q = fma(r, q, c)
        v = fma(float(Decimal(T) * Decimal(r)), q, T)
        return v * 2.0 ** (ni >> shift)
    return f

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = random.Random(1)
ts = [rng.uniform(0, 40) for _ in range(n)] + [rng.uniform(0, 1e-3) for _ in range(n // 10)]
for entries, degree in ((32, 6), (256, 4), (256, 5), (128, 5), (64, 5)):
    f = make(entries, degree)
    step = float(LN2 / Decimal(entries))
    pts = ts + [k * step / 2 + e for k in range(1, 400) for e in (-1e-12, 0.0, 1e-12)]
    worst = 0.0
    for t in pts:
        if t < 0:
            continue
        got = f(t)
        ref = (-Decimal(t)).exp()
        err = abs(Decimal(got) - ref) / Decimal(ulp(float(ref)))
        worst = max(worst, float(err))
    print(f"table 2^(n/{entries}), degree {degree}: max error {worst:.3f} ulp over {len(pts)} arguments")
