"""Predicted time of a launch plan of the fused factorisation under the packer's cost model (no GPU):
python tools/ci_plan_predict.py Np "win,far_k,far_kind,defer+1" [-v]"""
import heapq, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from cholinv_sim import get_plan, SMALL, BIG, BIG256

def cost(t):
    k, K = int(t[0]), int(t[2])
    return 4 + 0.03 * K if k == SMALL else (14 + 0.122 * K if k == BIG else 21 + 0.244 * K)

def predict(Np, opt, verbose=False, ncu=256):
    L, T = get_plan(Np, opt)
    tot = work = 0.0
    for l in L:
        h = [0.0] * ncu
        w = 0.0
        for _ in range(l[1]):
            heapq.heapreplace(h, h[0] + 37); w += 37
        g = int(l[4])
        tl = T[l[2]:l[2] + l[3]]
        for i in range(0, len(tl), g):
            c = sum(cost(t) - (5.0 if j else 0.0) for j, t in enumerate(tl[i:i + g])); heapq.heapreplace(h, h[0] + c); w += c
        mk = max(h) + 1.5
        tot += mk; work += w
        if verbose: print(list(l), f"pred {mk:7.1f} us  util {w / (mk * ncu):.2f}")
    return tot, work / ncu, len(L), len(T)

if __name__ == "__main__":
    Np = int(sys.argv[1])
    opts = [[int(x) for x in a.split(",")] for a in sys.argv[2:] if not a.startswith("-")] or [None]
    for o in opts:
        tot, ideal, nl, nt = predict(Np, o, "-v" in sys.argv)
        print(f"Np={Np} opt={o}: predicted {tot/1e3:.3f} ms, work/ncu {ideal/1e3:.3f} ms, {nl} launches, {nt} tiles")
