"""Fifth scan: a tail whose LAST step is larger (the final codebook carries the sampling noise of the last mini-batch:
87 rows per node at 20 equal tail steps) -- k small steps + one step of m small-step sizes, the tail's row budget fixed
(eight seeds; see batch_schedule_scan3.py)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(__file__))
from batch_rule_study import *  # noqa


def sched(G1, k, m, hr=0.25, split=5.0 / 6):
    head = np.geomspace(1.0, hr, G1)
    head = head / head.sum() * split
    tail = np.r_[np.ones(k), [float(m)]] if m > 0 else np.ones(k)
    tail = tail / tail.sum() * (1 - split)
    return np.r_[head, tail]


def main():
    n = 1 << 20
    x = np.concatenate([synth.make_fov_numpy(n // 8, C, seed=1000 + i, dtype=np.float32) for i in range(8)]).astype(np.float64)
    rr = default_radius_range(XD, YD)
    runs = []
    for s in range(int(os.environ.get('SCAN_SEED0', '50')), int(os.environ.get('SCAN_SEED0', '50')) + 8):
        rs = np.random.RandomState(s)
        w0 = x[rs.choice(n, K, replace=False)].copy()
        ev = x[rs.choice(n, 200_000, replace=False)]
        order = np.random.RandomState(7 + s).randint(0, n, size=n).astype(np.int64)
        w_on = ob.som_online(x, w0, XD, YD, 1, (0.05, 0.01), rr, order)
        runs.append((w0, ev, qe(ev, w_on), np.random.RandomState(3 + s).permutation(n)))
    sets = [(6, 15, 5), (6, 13, 4), (6, 11, 4), (6, 16, 4), (6, 9, 3), (6, 12, 8)]
    if len(sys.argv) > 1:
        sets = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
    for G1, k, m in sets:
        vals = []
        for w0, ev, q_on, perm in runs:
            b, f = sized(n, sched(G1, k, m), perm)
            vals.append((qe(ev, train(x, w0, b, f, 0)) / q_on - 1) * 100)
        print("head %d + tail %d small + 1 of %dx (%d launches): QE %+.2f %% +- %.2f (%s)" % (G1, k, m, G1 + k + 1, np.mean(vals), np.std(vals) / np.sqrt(len(vals)),
                                                                                             " ".join("%+.2f" % v for v in vals)), flush=True)


if __name__ == "__main__":
    main()
