"""Quality study of batch-rule variants on the bench workload (CPU, numpy; not a parity tool).

Mean quantisation error of the trained codebook over an evaluation sample, relative to the codebook the ONLINE
oracle (FlowSOM C_SOM restatement) reaches from the same initial nodes.  Variants: number of mini-batch steps G,
delay D (step g searches with a codebook that has absorbed the statistics up to step g-1-D), batch-size schedules.
    python scripts/study/batch_rule_study.py
"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from ark_analysis_amd import synth
from ark_analysis_amd.flowsom import default_radius_range
from tests import oracle_binding as ob

XD = YD = 10
K = 100
C = 22


def grid_dist():
    g = np.array([(x, y) for x in range(XD) for y in range(YD)])
    return np.abs(g[:, None, :] - g[None, :, :]).max(axis=2)


NH = grid_dist()


def bmu(x, w):
    d = (w * w).sum(1)[None, :] - 2.0 * (x @ w.T)
    return d.argmin(1)


def qe(x, w):
    b = bmu(x, w)
    return float(np.sqrt(((x - w[b]) ** 2).sum(1)).mean())


def stats(x, w):
    b = bmu(x, w)
    S = np.zeros((K, C))
    np.add.at(S, b, x)
    cnt = np.bincount(b, minlength=K).astype(np.float64)
    return S, cnt


def update(w, S, cnt, thr, alpha):
    m = (NH <= thr).astype(np.float64)
    num, den = m @ S, m @ cnt
    gain = 1.0 - (1.0 - alpha) ** den   # (the kernels and the oracle form the power by binary exponentiation: batch_gain)
    ok = den > 0
    out = w.copy()
    out[ok] = w[ok] + gain[ok, None] * (num[ok] / den[ok, None] - w[ok])
    return out


def sched(frac, a0=0.05, a1=0.01, r0=6.0, r1=0.0):
    thr = r0 - (r0 - r1) * frac
    if thr < 1.0:
        thr = 0.5
    return thr, a0 - (a0 - a1) * frac


def train(x, w0, batches, fracs, delay=0):
    """batches: list of row-index arrays; fracs[g]: schedule position of the update that absorbs batch g.
    Step g searches with W_g = W_0 + updates of batches 0 .. g-1-delay."""
    w = w0.copy()
    pend = []
    for g, rows in enumerate(batches):
        # apply updates that are due: statistics of step g-1-delay
        while pend and pend[0][0] <= g - 1 - delay:
            _, S, cnt, fr = pend.pop(0)
            w = update(w, S, cnt, *sched(fr))
        S, cnt = stats(x[rows], w)
        pend.append((g, S, cnt, fracs[g]))
    for _, S, cnt, fr in pend:
        w = update(w, S, cnt, *sched(fr))
    return w


def strided(n, G):
    return [np.arange(t, n, G) for t in range(G)], [g / G for g in range(G)]


def sized(n, sizes, perm):
    """consecutive slices of a fixed permutation with the given relative sizes; schedule position = fraction of rows
    presented before the batch"""
    sizes = np.asarray(sizes, dtype=np.float64)
    edges = np.concatenate([[0], np.round(np.cumsum(sizes) / sizes.sum() * n)]).astype(np.int64)
    return [perm[edges[i]:edges[i + 1]] for i in range(len(sizes))], [edges[i] / n for i in range(len(sizes))]


def main():
    n = 1 << 20
    parts = [synth.make_fov_numpy(n // 8, C, seed=1000 + i, dtype=np.float32) for i in range(8)]
    x = np.concatenate(parts).astype(np.float64)
    rs = np.random.RandomState(42)
    w0 = x[rs.choice(n, K, replace=False)].copy()
    ev = x[rs.choice(n, 200_000, replace=False)]
    rr = default_radius_range(XD, YD)
    order = np.random.RandomState(7).randint(0, n, size=n).astype(np.int64)
    t = time.time()
    w_on = ob.som_online(x, w0, XD, YD, 1, (0.05, 0.01), rr, order)
    q_on = qe(ev, w_on)
    print("online oracle: QE %.6f (%.1f s)" % (q_on, time.time() - t))
    perm = np.random.RandomState(3).permutation(n)

    def report(name, w, launches):
        print("%-58s launches %3d  QE %+.2f %%" % (name, launches, (qe(ev, w) / q_on - 1) * 100), flush=True)

    for G in (16, 32, 48, 64, 128, 256):
        b, f = strided(n, G)
        report("strided G=%d delay 0" % G, train(x, w0, b, f, 0), G)
    for G in (64, 96, 128):
        b, f = strided(n, G)
        report("strided G=%d delay 1 (pairs)" % G, train(x, w0, b, f, 1), G // 2)
    for G in (64, 128):
        b, f = strided(n, G)
        report("strided G=%d delay 3" % G, train(x, w0, b, f, 3), G // 4)
    # growing / shrinking batch sizes at a fixed number of steps
    for G in (24, 32, 40, 48):
        for name, sizes in (("equal", np.ones(G)),
                            ("geometric x8 growth", np.geomspace(1, 8, G)),
                            ("geometric x8 shrink", np.geomspace(8, 1, G)),
                            ("tail-heavy steps (5/6 rows in G/2, 1/6 in G/2)", np.r_[np.full(G // 2, 5.0 / 6 / (G // 2)), np.full(G - G // 2, 1.0 / 6 / (G - G // 2))]),
                            ("tail-light steps (5/6 rows in 3G/4+, tail in G/8)", np.r_[np.full(G - G // 8, 5.0 / 6 / (G - G // 8)), np.full(G // 8, 1.0 / 6 / (G // 8))])):
            b, f = sized(n, sizes, perm)
            report("G=%d %s" % (G, name), train(x, w0, b, f, 0), G)


if __name__ == "__main__":
    main()
