"""Fourth scan: where the tail breaks -- fewer tail steps than the 6 + 20 default (eight seeds; see batch_schedule_scan3.py)."""
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
    for s in range(50, 58):
        rs = np.random.RandomState(s)
        w0 = x[rs.choice(n, K, replace=False)].copy()
        ev = x[rs.choice(n, 200_000, replace=False)]
        order = np.random.RandomState(7 + s).randint(0, n, size=n).astype(np.int64)
        w_on = ob.som_online(x, w0, XD, YD, 1, (0.05, 0.01), rr, order)
        runs.append((w0, ev, qe(ev, w_on), np.random.RandomState(3 + s).permutation(n)))
    sets = [(6, 16, 0.25), (6, 14, 0.25), (6, 12, 0.25), (6, 10, 0.25), (6, 8, 0.25), (5, 14, 0.25), (5, 12, 0.25), (4, 12, 0.25), (6, 20, 0.25)]
    if len(sys.argv) > 1:
        sets = [tuple(float(v) if "." in v else int(v) for v in a.split(",")) for a in sys.argv[1:]]
    for G1, G2, hr in sets:
        vals = []
        for w0, ev, q_on, perm in runs:
            b, f = sized(n, two_phase(G1, G2, 1, hr), perm)
            vals.append((qe(ev, train(x, w0, b, f, 0)) / q_on - 1) * 100)
        print("two-phase G1=%d G2=%d head x%g: QE %+.2f %% +- %.2f (%s)" % (G1, G2, hr, np.mean(vals), np.std(vals) / np.sqrt(len(vals)),
                                                                          " ".join("%+.2f" % v for v in vals)), flush=True)


if __name__ == "__main__":
    main()
