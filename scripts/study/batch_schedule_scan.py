"""Scan of two-phase batch schedules (CPU, numpy): G1 equal steps over the rows presented while the radius is >= 1
(5/6 of a pass), G2 steps over the BMU-only tail, tail batch sizes equal or geometric.  See batch_rule_study.py."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(__file__))
from batch_rule_study import *  # noqa


def two_phase(G1, G2, tail_ratio=1.0, head_ratio=1.0, split=5.0 / 6):
    head = np.geomspace(1.0, head_ratio, G1)
    head = head / head.sum() * split
    tail = np.geomspace(1.0, tail_ratio, G2)
    tail = tail / tail.sum() * (1 - split)
    return np.r_[head, tail]


def main():
    n = 1 << 20
    x = np.concatenate([synth.make_fov_numpy(n // 8, C, seed=1000 + i, dtype=np.float32) for i in range(8)]).astype(np.float64)
    rr = default_radius_range(XD, YD)
    seeds = (42, 43, 44)
    runs = []
    for s in seeds:
        rs = np.random.RandomState(s)
        w0 = x[rs.choice(n, K, replace=False)].copy()
        ev = x[rs.choice(n, 200_000, replace=False)]
        order = np.random.RandomState(7 + s).randint(0, n, size=n).astype(np.int64)
        w_on = ob.som_online(x, w0, XD, YD, 1, (0.05, 0.01), rr, order)
        runs.append((w0, ev, qe(ev, w_on), np.random.RandomState(3 + s).permutation(n)))
    print("online QE:", [round(r[2], 5) for r in runs], flush=True)

    def report(name, mk):
        vals = []
        for w0, ev, q_on, perm in runs:
            b, f = mk(perm)
            vals.append((qe(ev, train(x, w0, b, f, 0)) / q_on - 1) * 100)
        print("%-52s QE %+.2f %%  (%s)" % (name, np.mean(vals), " ".join("%+.2f" % v for v in vals)), flush=True)

    for G in (32, 64, 128):
        report("strided G=%d" % G, lambda perm, G=G: strided(n, G))
    for G1, G2, tr, hr in [(16, 16, 1, 1), (12, 20, 1, 1), (8, 24, 1, 1), (8, 16, 1, 1), (6, 18, 1, 1), (12, 12, 1, 1),
                           (8, 24, 4, 1), (8, 24, 0.25, 1), (8, 16, 4, 1), (12, 20, 4, 1), (8, 24, 16, 1),
                           (8, 24, 1, 0.25), (8, 24, 1, 4), (6, 26, 4, 1), (4, 28, 4, 1), (10, 30, 4, 1), (8, 40, 4, 1), (8, 56, 4, 1)]:
        report("two-phase G1=%d G2=%d tail x%g head x%g" % (G1, G2, tr, hr),
               lambda perm, a=G1, b=G2, t=tr, h=hr: sized(n, two_phase(a, b, t, h), perm))


if __name__ == "__main__":
    main()
