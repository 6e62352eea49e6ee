"""ctypes door onto oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module.  The product package (dpilqr_amd) never does.

Also holds the (small, pure-Python) restatement of the reference's dispatch
layer -- define_inter_graph_threshold / split / solve_distributed
(distributed.py:25-103,224-247, util.py:102-117, problem.py:36-64) -- which
calls the C solve for each sub-problem.
"""
import ctypes as C
import itertools
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
_LIB = None
_REF = None

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)


class _Problem(C.Structure):
    _fields_ = [("k", C.c_int), ("n_s", C.c_int), ("n_c", C.c_int), ("T", C.c_int), ("dt", C.c_double),
                ("model", c_ip), ("n_dims", c_ip), ("xf", c_dp), ("Q", c_dp), ("R", c_dp), ("Qf", c_dp),
                ("radius", C.c_double), ("w_ref", C.c_double), ("w_prox", C.c_double)]


def build(force=False):
    so = HERE / "liboracle.so"
    if force or not so.exists() or so.stat().st_mtime < (HERE / "ilqr_oracle.c").stat().st_mtime:
        subprocess.run(["make", "-C", str(HERE)], check=True, stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = build()
        L = C.CDLL(str(so))
        L.oracle_cost.restype = C.c_double
        L.oracle_prox_cost.restype = C.c_double
        L.oracle_rollout.restype = C.c_double
        L.oracle_forward_pass.restype = C.c_double
        L.oracle_forward_pass.argtypes = [C.c_void_p, c_dp, c_dp, c_dp, c_dp, C.c_double, c_dp, c_dp]
        L.oracle_backward_pass.argtypes = [C.c_void_p, c_dp, c_dp, C.c_double, c_dp, c_dp]
        L.oracle_backward_pass_tiles.argtypes = [C.c_int, C.c_int, C.c_int] + [c_dp] * 7 + [C.c_double, c_dp, c_dp]
        L.oracle_model_integrate.argtypes = [C.c_int, c_dp, c_dp, C.c_double, c_dp]
        L.oracle_model_linearize.argtypes = [C.c_int, c_dp, c_dp, C.c_double, c_dp, c_dp]
        L.oracle_quadraticize_distance.argtypes = [c_dp, c_dp, C.c_double, C.c_int, c_dp, c_dp]
        L.oracle_solve.argtypes = [C.c_void_p, c_dp, c_dp, C.c_int, C.c_double, c_dp, c_dp, c_dp, c_ip, c_ip]
        L.oracle_solve_batch.argtypes = [C.c_void_p, C.c_int, c_dp, c_dp, c_dp, C.c_int, C.c_double, c_dp, c_dp,
                                         c_ip, c_ip, c_ip, C.c_int]
        _LIB = L
    return _LIB


def ref_lib():
    """The REAL reference dynamics (oracle/_ref), or None when it was never built."""
    global _REF
    if _REF is None:
        so = HERE / "_ref" / "libbbdynamics_ref.so"
        if not so.exists():
            return None
        R = C.CDLL(str(so))
        R.ref_model_integrate.argtypes = [C.c_int, c_dp, c_dp, C.c_double, c_dp]
        R.ref_model_linearize.argtypes = [C.c_int, c_dp, c_dp, C.c_double, c_dp, c_dp]
        _REF = R
    return _REF


def _p(a):
    return a.ctypes.data_as(c_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


MODEL_DIMS = {0: (4, 2), 1: (6, 3), 2: (3, 2), 3: (4, 2), 4: (6, 3), 5: (6, 3), 6: (6, 3), 7: (12, 4), 8: (12, 4)}


def model_f(model, x, u, ref=False):
    x, u = _f64(x), _f64(u)
    o = np.zeros(MODEL_DIMS[model][0])
    L = ref_lib() if ref else lib()
    fn = L.ref_model_f if ref else L.oracle_model_f
    fn.argtypes = [C.c_int, c_dp, c_dp, c_dp]
    assert fn(model, _p(x), _p(u), _p(o)) == 0
    return o


def model_integrate(model, x, u, dt, ref=False):
    x, u = _f64(x), _f64(u)
    o = np.zeros(MODEL_DIMS[model][0])
    fn = ref_lib().ref_model_integrate if ref else lib().oracle_model_integrate
    assert fn(model, _p(x), _p(u), float(dt), _p(o)) == 0
    return o


def model_linearize(model, x, u, dt, ref=False):
    x, u = _f64(x), _f64(u)
    ns, nc = MODEL_DIMS[model]
    A = np.zeros((ns, ns)); B = np.zeros((ns, nc))
    fn = ref_lib().ref_model_linearize if ref else lib().oracle_model_linearize
    assert fn(model, _p(x), _p(u), float(dt), _p(A), _p(B)) == 0
    return A, B


def quadraticize_distance(pa, pb, radius, n_d):
    pa = _f64(np.r_[pa, 0, 0, 0][:3]); pb = _f64(np.r_[pb, 0, 0, 0][:3])
    g = np.zeros(3); H = np.zeros((3, 3))
    lib().oracle_quadraticize_distance(_p(pa), _p(pb), float(radius), int(n_d), _p(g), _p(H))
    return g, H


def alphas():
    a = np.zeros(10)
    lib().oracle_alphas(_p(a))
    return a


class Problem:
    """One centralised (sub)problem; holds the arrays the C struct points into."""

    def __init__(self, model, n_dims, xf, Q, R, Qf, radius, dt, T, w_ref=1.0, w_prox=200.0):
        self.model = np.ascontiguousarray(model, dtype=np.int32)
        self.k = int(self.model.size)
        self.n_s, self.n_c = MODEL_DIMS[int(self.model[0])]
        self.n_dims = np.ascontiguousarray(n_dims, dtype=np.int32)
        self.xf = _f64(xf).reshape(-1)
        bc = lambda M, n: _f64(np.broadcast_to(np.asarray(M, dtype=np.float64), (self.k, n, n)))
        self.Q, self.R, self.Qf = bc(Q, self.n_s), bc(R, self.n_c), bc(Qf, self.n_s)
        self.radius, self.dt, self.T = float(radius), float(dt), int(T)
        self.w_ref, self.w_prox = float(w_ref), float(w_prox)
        self.n_x, self.n_u = self.k * self.n_s, self.k * self.n_c
        self._s = _Problem(self.k, self.n_s, self.n_c, self.T, self.dt,
                           self.model.ctypes.data_as(c_ip), self.n_dims.ctypes.data_as(c_ip),
                           _p(self.xf), _p(self.Q), _p(self.R), _p(self.Qf), self.radius, self.w_ref, self.w_prox)

    @property
    def ptr(self):
        return C.addressof(self._s)

    def subproblem(self, idx):
        """Sub-problem over agents idx (original order kept: dynamics.py:194, cost.py:254-257)."""
        idx = list(idx)
        return Problem(self.model[idx], self.n_dims[idx], self.xf.reshape(self.k, -1)[idx], self.Q[idx],
                       self.R[idx], self.Qf[idx], self.radius, self.dt, self.T, self.w_ref, self.w_prox)

    def with_T(self, T):
        return Problem(self.model, self.n_dims, self.xf, self.Q, self.R, self.Qf, self.radius, self.dt, T,
                       self.w_ref, self.w_prox)

    # --- cost / dynamics
    def cost(self, x, u, terminal=False):
        x, u = _f64(x), _f64(u)
        L = lib(); L.oracle_cost.argtypes = [C.c_void_p, c_dp, c_dp, C.c_int]
        return L.oracle_cost(self.ptr, _p(x), _p(u), int(terminal))

    def prox_cost(self, x):
        x = _f64(x); L = lib(); L.oracle_prox_cost.argtypes = [C.c_void_p, c_dp]
        return L.oracle_prox_cost(self.ptr, _p(x))

    def prox_quadraticize(self, x):
        x = _f64(x); Lx = np.zeros(self.n_x); Lxx = np.zeros((self.n_x, self.n_x))
        L = lib(); L.oracle_prox_quadraticize.argtypes = [C.c_void_p, c_dp, c_dp, c_dp]
        L.oracle_prox_quadraticize(self.ptr, _p(x), _p(Lx), _p(Lxx))
        return Lx, Lxx

    def quadraticize(self, x, u, terminal=False):
        x, u = _f64(x), _f64(u); n, m = self.n_x, self.n_u
        Lx, Lu, Lxx, Luu, Lux = np.zeros(n), np.zeros(m), np.zeros((n, n)), np.zeros((m, m)), np.zeros((m, n))
        L = lib(); L.oracle_quadraticize.argtypes = [C.c_void_p, c_dp, c_dp, C.c_int] + [c_dp] * 5
        L.oracle_quadraticize(self.ptr, _p(x), _p(u), int(terminal), _p(Lx), _p(Lu), _p(Lxx), _p(Luu), _p(Lux))
        return Lx, Lu, Lxx, Luu, Lux

    def step(self, x, u):
        x, u = _f64(x), _f64(u); xn = np.zeros(self.n_x)
        L = lib(); L.oracle_step.argtypes = [C.c_void_p, c_dp, c_dp, c_dp]
        L.oracle_step(self.ptr, _p(x), _p(u), _p(xn))
        return xn

    def linearize(self, x, u):
        x, u = _f64(x), _f64(u); A = np.zeros((self.n_x, self.n_x)); B = np.zeros((self.n_x, self.n_u))
        L = lib(); L.oracle_linearize.argtypes = [C.c_void_p, c_dp, c_dp, c_dp, c_dp]
        L.oracle_linearize(self.ptr, _p(x), _p(u), _p(A), _p(B))
        return A, B

    # --- passes
    def rollout(self, x0, U):
        x0, U = _f64(x0).reshape(-1), _f64(U)
        X = np.zeros((self.T + 1, self.n_x))
        L = lib(); L.oracle_rollout.argtypes = [C.c_void_p, c_dp, c_dp, c_dp]
        J = L.oracle_rollout(self.ptr, _p(x0), _p(U), _p(X))
        return X, J

    def backward_pass(self, X, U, mu):
        X, U = _f64(X), _f64(U)
        K = np.zeros((self.T, self.n_u, self.n_x)); d = np.zeros((self.T, self.n_u))
        rc = lib().oracle_backward_pass(self.ptr, _p(X), _p(U), float(mu), _p(K), _p(d))
        assert rc == 0
        return K, d

    def forward_pass(self, X, U, K, d, alpha):
        X, U, K, d = _f64(X), _f64(U), _f64(K), _f64(d)
        Xn = np.zeros_like(X); Un = np.zeros_like(U)
        J = lib().oracle_forward_pass(self.ptr, _p(X), _p(U), _p(K), _p(d), float(alpha), _p(Xn), _p(Un))
        return Xn, Un, J

    def solve(self, x0, U0, n_lqr_iter=50, tol=1e-3):
        x0 = _f64(x0).reshape(-1); U = _f64(U0).copy()
        X = np.zeros((self.T + 1, self.n_x)); J = C.c_double(0.0)
        trace = np.full((max(n_lqr_iter, 1), 5), np.nan)
        nb, nf = C.c_int(0), C.c_int(0)
        st = lib().oracle_solve(self.ptr, _p(x0), _p(U), n_lqr_iter, tol, _p(X), C.byref(J), _p(trace),
                                C.byref(nb), C.byref(nf))
        return dict(X=X, U=U, J=J.value, status=st, n_bwd=nb.value, n_fwd=nf.value, trace=trace[:nb.value])


def backward_pass_tiles(A, B, Lx, Lu, Lxx, Luu, Lux, mu):
    A, B, Lx, Lu, Lxx, Luu, Lux = map(_f64, (A, B, Lx, Lu, Lxx, Luu, Lux))
    T, n, m = B.shape
    K = np.zeros((T, m, n)); d = np.zeros((T, m))
    rc = lib().oracle_backward_pass_tiles(n, m, T, _p(A), _p(B), _p(Lx), _p(Lu), _p(Lxx), _p(Luu), _p(Lux),
                                          float(mu), _p(K), _p(d))
    assert rc == 0
    return K, d


def solve_batch(proto, x0, xf, U0, n_lqr_iter=50, tol=1e-3, n_threads=0, trace=False):
    """B independent problems that differ in (x0, xf) only; OpenMP over the batch.
    trace=True adds the decision trace (B, n_lqr_iter, 5) = (mu_before, accepted alpha index or -1, J_last, J_star_after,
    n_forward_passes), NaN beyond each item's last iteration."""
    x0, xf = _f64(x0), _f64(xf); U = _f64(U0).copy()
    Bn = x0.shape[0]
    X = np.zeros((Bn, proto.T + 1, proto.n_x)); J = np.zeros(Bn)
    st = np.zeros(Bn, dtype=np.int32); nb = np.zeros(Bn, dtype=np.int32); nf = np.zeros(Bn, dtype=np.int32)
    ip = lambda a: a.ctypes.data_as(c_ip)
    tr = np.full((Bn, max(n_lqr_iter, 1), 5), np.nan) if trace else None
    L = lib()
    L.oracle_solve_batch_trace.argtypes = [C.c_void_p, C.c_int, c_dp, c_dp, c_dp, C.c_int, C.c_double, c_dp, c_dp,
                                           c_ip, c_ip, c_ip, C.c_int, c_dp]
    L.oracle_solve_batch_trace(proto.ptr, Bn, _p(x0), _p(xf), _p(U), n_lqr_iter, tol, _p(X), _p(J), ip(st), ip(nb),
                               ip(nf), _threads(n_threads), _p(tr) if trace else None)
    out = dict(X=X, U=U, J=J, status=st, n_bwd=nb, n_fwd=nf)
    if trace:
        for i in range(Bn):
            tr[i, nb[i]:] = np.nan
        out["trace"] = tr
    return out


def replay_batch(proto, x0, xf, U0, forced, n_lqr_iter=50, tol=1e-3, n_threads=0):
    """The oracle made to FOLLOW the decisions of an implementation under test (oracle_solve_replay): `forced` is that
    implementation's result dict (trace (B, n_lqr_iter, 5), n_bwd, status).  Returns X, U, J and rtrace (B, n_lqr_iter, 8) =
    (mu_before, the oracle's own accepted index, J_last, J_star_after, accept margin, convergence margin, J_star_before,
    the oracle's own converged flag), NaN beyond each item's last iteration."""
    x0, xf = _f64(x0), _f64(xf); U = _f64(U0).copy()
    Bn = x0.shape[0]
    rows = max(n_lqr_iter, 1)
    ft = np.full((Bn, rows, 5), -1.0)
    t = np.asarray(forced["trace"], dtype=np.float64)
    ft[:, :min(rows, t.shape[1])] = t[:, :rows]
    ft = _f64(np.nan_to_num(ft, nan=-1.0))
    nfo = np.ascontiguousarray(forced["n_bwd"], dtype=np.int32); fst = np.ascontiguousarray(forced["status"], dtype=np.int32)
    X = np.zeros((Bn, proto.T + 1, proto.n_x)); J = np.zeros(Bn); st = np.zeros(Bn, dtype=np.int32)
    rt = np.full((Bn, rows, 8), np.nan)
    ip = lambda a: a.ctypes.data_as(c_ip)
    L = lib()
    L.oracle_replay_batch.argtypes = [C.c_void_p, C.c_int, c_dp, c_dp, c_dp, C.c_int, C.c_double, c_dp, c_ip, c_ip,
                                      c_dp, c_dp, c_ip, c_dp, C.c_int]
    L.oracle_replay_batch(proto.ptr, Bn, _p(x0), _p(xf), _p(U), n_lqr_iter, tol, _p(ft), ip(nfo), ip(fst), _p(X), _p(J),
                          ip(st), _p(rt), _threads(n_threads))
    for i in range(Bn):
        rt[i, nfo[i]:] = np.nan
    return dict(X=X, U=U, J=J, status=st, n_bwd=nfo, rtrace=rt)


def max_threads():
    return lib().oracle_max_threads()


def usable_cores():
    """The CPUs this process may actually keep busy: its affinity mask, capped by the cgroup's CPU quota (the GPU boxes
    show 256 hardware threads behind `cpu.max = 1600000 100000`, i.e. 16 CPUs: 256 OpenMP threads then spend their time
    throttled -- 2.3 k sub-problems/s against 5.2 k with 16 threads, profiles/r03_cpu_probe.txt)."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = Path(path).read_text().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
    try:
        q = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text()); per = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
        if q > 0:
            n = min(n, max(1, q // per))
    except (OSError, ValueError):
        pass
    return n


def _threads(n_threads):
    return int(n_threads) if n_threads and n_threads > 0 else usable_cores()


# ---------------------------------------------------------------- dispatch layer
def pairwise_planar_distance(X, k, n_s):
    """util.compute_pairwise_distance (util.py:48-61) with its default n_d=2."""
    X = np.atleast_2d(X)
    pairs = list(itertools.combinations(range(k), 2))
    out = np.zeros((X.shape[0], len(pairs)))
    for c, (i, j) in enumerate(pairs):
        dxy = X[:, i * n_s:i * n_s + 2] - X[:, j * n_s:j * n_s + 2]
        out[:, c] = np.sqrt(dxy[:, 0] ** 2 + dxy[:, 1] ** 2)
    return out


def define_inter_graph_threshold(X, radius, k, n_s):
    """distributed.py:224-247 on agent INDICES 0..k-1; returns {i: sorted neighbourhood incl. i}."""
    X = np.atleast_2d(X)
    rel = pairwise_planar_distance(X, k, n_s)
    N = X.shape[0]
    step = max(N // 10, 1)
    rows = slice(0, N + 1, step)
    graph = {i: [i] for i in range(k)}
    for c, (i, j) in enumerate(itertools.combinations(range(k), 2)):
        if np.any(rel[rows, c] < 2 * radius):
            graph[i].append(j); graph[j].append(i)
    return {i: sorted(v) for i, v in graph.items()}


def solve_distributed(prob, X, U, radius, n_lqr_iter=50, tol=1e-3):
    """distributed.py:25-103 (serial branch, ignore_ids=[]): one sub-problem per AGENT."""
    X = np.atleast_2d(X); k, ns, nc, T = prob.k, prob.n_s, prob.n_c, U.shape[0]
    prob = prob.with_T(T)
    graph = define_inter_graph_threshold(X, radius, k, ns)
    X_dec = np.zeros((T + 1, k * ns)); U_dec = np.zeros((T, k * nc))
    for i in range(k):
        idx = graph[i]
        sub = prob.subproblem(idx)
        x0i = np.concatenate([X[0, a * ns:(a + 1) * ns] for a in idx])
        Ui = np.concatenate([U[:, a * nc:(a + 1) * nc] for a in idx], axis=1)
        r = sub.solve(x0i, Ui, n_lqr_iter, tol)
        pos = idx.index(i)
        X_dec[:, i * ns:(i + 1) * ns] = r["X"][:, pos * ns:(pos + 1) * ns]
        U_dec[:, i * nc:(i + 1) * nc] = r["U"][:, pos * nc:(pos + 1) * nc]
    _, J_full = prob.rollout(X[0], U_dec)
    return X_dec, U_dec, J_full, graph


def selfish_warmstart(prob, x0, N):
    """problem.py:66-91: every agent solves alone from U = 0 with the solver's defaults; U_warm stacks the controls."""
    x0 = np.asarray(x0, dtype=np.float64).reshape(-1)
    cols = []
    for i in range(prob.k):
        sub = prob.with_T(N).subproblem([i])
        r = sub.solve(x0[i * prob.n_s:(i + 1) * prob.n_s], np.zeros((N, prob.n_c)))
        cols.append(r["U"])
    return np.concatenate(cols, axis=1)


def solve_rhc(prob, x0, N, U_warm, radius=None, centralized=True, n_d=2, step_size=1, dist_converge=None,
              t_diverge=None, n_lqr_iter=50, tol=1e-3):
    """distributed.py:106-221 (dist_converge form; the J_converge form raises NameError in the reference).
    U_warm is the random warm start the reference draws at :152.  Returns X_full, U_full, J_full and the per-round
    log fields [(t, J, converged, subgraphs as index lists, distance_left)] the CSV rows are printed from."""
    k, ns = prob.k, prob.n_s
    prob = prob.with_T(N)
    xf = prob.xf

    def left(x):
        return np.linalg.norm((x - xf).reshape(k, ns)[:, :n_d], axis=1)

    xi = np.asarray(x0, dtype=np.float64).reshape(1, -1)
    X, U = xi.copy(), np.array(U_warm, dtype=np.float64)
    t, J, converged = 0, np.inf, True
    X_full = np.zeros((0, prob.n_x)); U_full = np.zeros((0, prob.n_u))
    rounds = []
    while np.any(left(xi.reshape(-1)) > dist_converge):
        if centralized:
            r = prob.solve(xi.reshape(-1), U, n_lqr_iter, tol)
            X, U, J = r["X"], r["U"], r["J"]
            graphs = [list(range(k))] * k
        else:
            X, U, J, g = solve_distributed(prob, X, U, radius, n_lqr_iter, tol)
            graphs = [g[i] for i in range(k)]
        xi = X[step_size]
        X_full = np.r_[X_full, X[:step_size]]; U_full = np.r_[U_full, U[:step_size]]
        X = np.r_[X[step_size:], np.tile(X[-1], (step_size, 1))]
        U = np.r_[U[step_size:], np.zeros((step_size, prob.n_u))]
        rounds.append((t, J, converged, graphs, left(xi).tolist()))
        if t_diverge and t >= t_diverge:
            converged = False
            break
        t += step_size * prob.dt
    if not X_full.size and not U_full.size:
        X_full = np.asarray(x0, dtype=np.float64).copy(); U_full = np.zeros((1, prob.n_u))
    _, J_full = prob.with_T(U_full.shape[0]).rollout(np.asarray(x0, dtype=np.float64).reshape(-1), U_full)
    return X_full, U_full, J_full, rounds, converged
