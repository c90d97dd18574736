"""Does a one-step delay hurt in the BMU-only tail?  (two-phase schedule; delay applied from step `from_step` on)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(__file__))
from batch_rule_study import *  # noqa
from batch_schedule_scan import two_phase


def train_mixed(x, w0, batches, fracs, from_step, delay):
    w = w0.copy()
    pend = []
    for g, rows in enumerate(batches):
        d = delay if g >= from_step else 0
        while pend and pend[0][0] <= g - 1 - d:
            _, S, cnt, fr = pend.pop(0)
            w = update(w, S, cnt, *sched(fr))
        S, cnt = stats(x[rows], w)
        pend.append((g, S, cnt, fracs[g]))
    for _, S, cnt, fr in pend:
        w = update(w, S, cnt, *sched(fr))
    return w


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
    for name, G1, G2, from_step, delay in [("8+24 no delay", 8, 24, 99, 0), ("8+24 tail delay 1", 8, 24, 9, 1), ("8+32 tail delay 1", 8, 32, 9, 1),
                                           ("8+48 tail delay 1", 8, 48, 9, 1), ("8+24 tail delay 2", 8, 24, 10, 2), ("8+48 tail delay 3", 8, 48, 11, 3)]:
        vals = []
        for w0, ev, q_on, perm in runs:
            b, f = sized(n, two_phase(G1, G2, 1, 0.25), perm)
            vals.append((qe(ev, train_mixed(x, w0, b, f, from_step, delay)) / q_on - 1) * 100)
        print("%-24s QE %+.2f %% +- %.2f (%s)" % (name, np.mean(vals), np.std(vals) / np.sqrt(len(vals)), " ".join("%+.2f" % v for v in vals)), flush=True)


if __name__ == "__main__":
    main()
