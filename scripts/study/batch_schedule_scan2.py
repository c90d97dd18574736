"""Second scan: candidates with 20-32 dependent launches, six seeds each (see batch_schedule_scan.py)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(__file__))
from batch_rule_study import *  # noqa
from batch_schedule_scan import two_phase


def main():
    n = 1 << 20
    x = np.concatenate([synth.make_fov_numpy(n // 8, C, seed=1000 + i, dtype=np.float32) for i in range(8)]).astype(np.float64)
    rr = default_radius_range(XD, YD)
    runs = []
    for s in range(50, 56):
        rs = np.random.RandomState(s)
        w0 = x[rs.choice(n, K, replace=False)].copy()
        ev = x[rs.choice(n, 200_000, replace=False)]
        order = np.random.RandomState(7 + s).randint(0, n, size=n).astype(np.int64)
        w_on = ob.som_online(x, w0, XD, YD, 1, (0.05, 0.01), rr, order)
        runs.append((w0, ev, qe(ev, w_on), np.random.RandomState(3 + s).permutation(n)))

    def report(name, mk):
        vals = []
        for w0, ev, q_on, perm in runs:
            b, f = mk(perm)
            vals.append((qe(ev, train(x, w0, b, f, 0)) / q_on - 1) * 100)
        print("%-52s QE %+.2f %% +- %.2f (%s)" % (name, np.mean(vals), np.std(vals) / np.sqrt(len(vals)), " ".join("%+.2f" % v for v in vals)), flush=True)

    for G in (32, 64, 128):
        report("strided G=%d" % G, lambda perm, G=G: strided(n, G))
    for G1, G2, tr, hr in [(8, 24, 1, 0.25), (8, 24, 1, 1), (8, 16, 1, 0.25), (8, 16, 4, 0.25), (8, 16, 4, 1), (6, 18, 1, 0.25),
                           (6, 14, 1, 0.25), (6, 14, 4, 0.25), (12, 20, 1, 0.25), (10, 22, 1, 0.5), (8, 24, 2, 0.25), (8, 24, 1, 0.1),
                           (5, 15, 2, 0.25), (4, 12, 2, 0.25)]:
        report("two-phase G1=%d G2=%d tail x%g head x%g" % (G1, G2, tr, hr),
               lambda perm, a=G1, b=G2, t=tr, h=hr: sized(n, two_phase(a, b, t, h), perm))


if __name__ == "__main__":
    main()
