"""Reading the g9_chaos fixtures: the REAL reference run on x0 and on 32 copies of x0 perturbed by 1e-14 .. 5e-13
(tests/golden/make_golden.py g9), for items the GPU decides differently from the CPU oracle.

A solve's decision trace is (accepted alpha index per iteration ..., END).  An implementation's trace is walked decision by
decision against the reference members that share its prefix so far ("alive"); each of its decisions falls into one class:
    witnessed     an alive member takes the same decision
    undetermined  no alive member does, but the alive members disagree among themselves: the reference does not determine
                  this decision at fp64 resolution (the line search is a lottery there: up to seven different accepted
                  step sizes among 25 members on cfg2 seed 132, iteration 8)
    violation     >= MIN_UNANIMOUS alive members all take ONE decision and the implementation another
    inconclusive  fewer than MIN_UNANIMOUS alive members, unanimous: too few samples to call the decision determined
    exhausted     no member shares the prefix any more: the fixture has nothing to say (the oracle-replay envelope does)
Unanimity of n samples bounds the probability of another decision only by about 3 / n, so a lone violation among hundreds
of decisions is within what a faithful implementation shows (the C oracle: 1 of 687); the tests bound the RATE."""
import numpy as np

MIN_UNANIMOUS = 8
FAMILIES = {"cfg2": (0, 5, 4, 2, 2), "uni8": (3, 8, 4, 2, 2), "quad10": (4, 10, 6, 3, 3)}    # model, k, n_s, n_c, n_d


def trace_of(n_bwd, acc):
    return tuple(int(v) for v in np.asarray(acc)[:int(n_bwd)]) + ("END",)


def member_traces(z, a):
    return [trace_of(z["n_bwd"][a, b], z["acc_trace"][a, b]) for b in range(z["n_bwd"].shape[1])]


def classify(trace, members):
    """Counts per class for one item's trace, and the list of (iteration, alive members) of its violations."""
    cat = dict(witnessed=0, undetermined=0, violation=0, inconclusive=0, exhausted=0)
    where = []
    alive = list(range(len(members)))
    for j, dec in enumerate(trace):
        if not alive:
            cat["exhausted"] += 1
            break
        decs = [members[b][j] if len(members[b]) > j else None for b in alive]
        nxt = [b for b, d in zip(alive, decs) if d == dec]
        if nxt:
            cat["witnessed"] += 1
        elif len(set(decs)) >= 2:
            cat["undetermined"] += 1
        elif len(alive) >= MIN_UNANIMOUS:
            cat["violation"] += 1; where.append((j, len(alive)))
        else:
            cat["inconclusive"] += 1
        alive = nxt
    return cat, where


def unanimous_prefix(members):
    """Number of leading decisions on which ALL members agree."""
    j = 0
    while all(len(m) > j for m in members) and len({m[j] for m in members}) == 1:
        j += 1
    return j


def problem_inputs(z):
    """Oracle / ProblemBatch inputs of a family's items: (model list, n_dims list, x0, xf, U0, Q, R, Qf, T)."""
    k, T = int(z["k"]), int(z["T"])
    model = {"DoubleIntDynamics4D": 0, "UnicycleDynamics4D": 3, "QuadcopterDynamics6D": 4}[str(z["model"])]
    ns, nc = (6, 3) if model == 4 else (4, 2)
    nd = 3 if ns == 6 else 2
    Q, R = (50.0 * np.eye(6), np.eye(3)) if ns == 6 else (np.diag([1.0, 1, 0, 0]), np.eye(2))
    U0 = np.zeros((len(z["seeds"]), T, k * nc))
    if bool(z["hover"]):
        U0[:, :, 0::3] = 9.80665
    return [model] * k, [nd] * k, z["x0"], z["xf"], U0, Q, R, 1000.0 * np.eye(ns), T
