"""oracle/libdpilqr_cpu_twin.so: the C ABI of include/dpilqr_hip.h over the CPU oracle, host pointers (SURVEY 8(b),
"identical-ABI CPU build").  Test infrastructure: it checks, without a GPU, that the ABI's data layouts (batch
descriptor and its strides, tile records, item lists, trace rows, status codes) carry the reference's golden vectors
through, using the very ctypes signatures the product binds (dpilqr_amd._lib.SIGNATURES) -- and that the product can
never run on it by accident (dpilqr_device_info says "no GPU", which _lib.require_gpu() turns into an exception)."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

from tests.golden_util import relerr
from tests.test_abi import declared_symbols

ROOT = Path(__file__).resolve().parent.parent
PASS_CASES = ["cfg2_di4d_k5", "car3d_k2", "mixed_q6h6", "quad12d_k2"]


@pytest.fixture(scope="module")
def twin():
    subprocess.run(["make", "-s", "-C", str(ROOT / "oracle"), str(ROOT / "oracle" / "libdpilqr_cpu_twin.so")], check=True)
    from dpilqr_amd import _lib
    L = C.CDLL(str(ROOT / "oracle" / "libdpilqr_cpu_twin.so"))
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    return L, _lib


def hp(a):
    return a.ctypes.data


class Desc:
    """host-pointer batch descriptor for B copies of one golden problem (per-item xf, shared weights)"""

    def __init__(self, _lib, z, pre="", B=1, T=None, xf=None):
        g = lambda k: z[pre + k]
        self.k, self.ns, self.nc = int(g("k")), int(g("n_s")), int(g("n_c"))
        self.T = int(T if T is not None else g("T"))
        self.n, self.m, self.B = self.k * self.ns, self.k * self.nc, B
        self.model = np.ascontiguousarray(g("model"), dtype=np.int32); self.n_dims = np.ascontiguousarray(g("n_dims"), dtype=np.int32)
        self.xf = np.ascontiguousarray(np.broadcast_to(g("xf"), (B, self.n)) if xf is None else xf, dtype=np.float64)
        self.Q, self.R, self.Qf = (np.ascontiguousarray(g(nm), dtype=np.float64) for nm in ("Q", "R", "Qf"))
        self.radius = np.array([float(g("radius"))])
        self.d = _lib.BatchDesc(B, self.k, self.ns, self.nc, self.T, 0, float(g("dt")), 1.0, 200.0, hp(self.model), 0, hp(self.n_dims), 0,
                                hp(self.xf), self.n, hp(self.Q), 0, hp(self.R), 0, hp(self.Qf), 0, hp(self.radius), 0)

    def ref(self):
        return C.byref(self.d)


def test_same_symbols_and_never_a_device(twin):
    L, _lib = twin
    for name in declared_symbols():
        assert hasattr(L, name), name
    assert L.dpilqr_abi_version() == 4
    arch = C.create_string_buffer(32)
    assert L.dpilqr_device_info(0, None, None, arch, 32) == _lib.ENOGPU and arch.value == b"cpu-twin"
    assert L.dpilqr_solve_enqueue(*([None] * 3), 0, 0.0, 0.0, 0, None, 0, *([None] * 8), 0, 0, None) == _lib.EUNSUPPORTED
    assert L.dpilqr_random_setup(0, 0, 1, 4, 2, 1.0, 1.0, None, None, None) == _lib.EUNSUPPORTED
    # nothing of the product names the twin
    for f in (ROOT / "dpilqr_amd").rglob("*.py"):
        assert "cpu_twin" not in f.read_text(), f


def test_model_ffi_through_the_abi(twin, golden):
    L, _ = twin
    z = golden("g1_models")
    names = [k[:-len("_enum")] for k in z if k.endswith("_enum")]
    assert len(names) >= 8
    for nm in names:
        x, u = np.ascontiguousarray(z[nm + "_x"]), np.ascontiguousarray(z[nm + "_u"])
        n, ns = x.shape
        mdl = np.full(n, int(z[nm + "_enum"]), dtype=np.int32)
        for i in range(n):                                  # dt varies per sample in the fixture; the ABI takes one dt per call
            dt = float(z[nm + "_dt"][i])
            f = np.empty(ns); xn = np.empty(ns); A = np.empty((ns, ns)); Bm = np.empty((ns, u.shape[1]))
            assert L.dpilqr_model_f(1, ns, hp(mdl[i:]), hp(x[i]), hp(u[i]), hp(f), None) == 0
            assert L.dpilqr_model_integrate(1, ns, hp(mdl[i:]), hp(x[i]), hp(u[i]), dt, hp(xn), None) == 0
            assert L.dpilqr_model_linearize(1, ns, hp(mdl[i:]), hp(x[i]), hp(u[i]), dt, hp(A), hp(Bm), None) == 0
            assert relerr(f, z[nm + "_f"][i]) < 1e-13 and relerr(xn, z[nm + "_xn"][i]) < 1e-13
            assert relerr(A, z[nm + "_A"][i]) < 1e-12 and relerr(Bm, z[nm + "_B"][i]) < 1e-12


@pytest.mark.parametrize("case", PASS_CASES)
def test_passes_through_the_abi(twin, golden, case):
    L, _lib = twin
    z = golden(f"g3_passes_{case}")
    B = 3
    D = Desc(_lib, z, B=B)
    n, m, T = D.n, D.m, D.T
    x0 = np.ascontiguousarray(np.broadcast_to(z["x0"], (B, n))); U0 = np.ascontiguousarray(np.broadcast_to(z["U0"], (B, T, m)))
    X = np.empty((B, T + 1, n)); J = np.empty(B)
    assert L.dpilqr_rollout(D.ref(), hp(x0), hp(U0), hp(X), hp(J), None) == 0
    assert relerr(X[2], z["X_roll"]) < 1e-12 and abs(J[1] - z["J_roll"]) < 1e-11 * abs(z["J_roll"])
    # tile records: the layout dpilqr_tile_layout describes holds the reference's linearize / quadraticize
    off, ld, stride = (C.c_int64 * 7)(), (C.c_int64 * 7)(), C.c_int64()
    assert L.dpilqr_tile_layout(n, m, C.byref(off), C.byref(ld), C.byref(stride)) == 0
    lay, lstride = _lib.tile_layout(n, m)                  # the product's own (host-only) answer
    assert stride.value == lstride and [(off[i], ld[i]) for i in range(7)] == [lay[p] for p in _lib.TILE_PARTS]
    assert L.dpilqr_tiles_bytes(B, T, n, m) == 8 * B * (T + 1) * stride.value
    Xo = np.ascontiguousarray(np.broadcast_to(z["X"], (B, T + 1, n))); Uo = np.ascontiguousarray(np.broadcast_to(z["U"], (B, T, m)))
    items = np.array([2, 0], dtype=np.int32); cnt = np.array([2], dtype=np.int32)
    tiles = np.full((2, T + 1, stride.value), np.nan)
    assert L.dpilqr_make_tiles(D.ref(), hp(Xo), hp(Uo), hp(tiles), hp(items), hp(cnt), None) == 0
    shapes = dict(A=(n, n), B=(n, m), Lxx=(n, n), Lux=(m, n), Luu=(m, m), Lx=(1, n), Lu=(1, m))

    def part(rec, p):
        o, l = lay[p]
        r, c = shapes[p]
        return np.stack([rec[o + i * l:o + i * l + c] for i in range(r)])
    for t in (0, T // 2, T - 1):
        for p in _lib.TILE_PARTS:
            ref = z[f"tile_{p}"][t]
            assert np.allclose(part(tiles[1, t], p), ref.reshape(shapes[p]), rtol=1e-11, atol=1e-9), (p, t)
    assert np.allclose(part(tiles[0, T], "Lxx"), z["tile_Lxx"][T], rtol=1e-11, atol=1e-9)
    # the sweep on records (listed items: K, d by position; mu by item id) and on (X, U)
    mu = np.array([99.0, 99.0, float(z["mu"])]); mu[0] = float(z["mu"])
    K = np.empty((2, T, m, n)); d = np.empty((2, T, m)); sing = np.zeros(B, dtype=np.int32)
    assert L.dpilqr_backward_pass_tiles_blocks(B, T, n, m, D.ns, D.nc, hp(tiles), hp(mu), hp(K), hp(d), hp(sing), hp(items), hp(cnt), None) == 0
    for slot in range(2):
        assert relerr(K[slot], z["K"]) < 1e-9 and relerr(d[slot], z["d"]) < 1e-9
    mu3 = np.full(B, float(z["mu"])); K3 = np.empty((B, T, m, n)); d3 = np.empty((B, T, m))
    assert L.dpilqr_backward_pass(D.ref(), hp(Xo), hp(Uo), hp(mu3), hp(K3), hp(d3), None, None) == 0
    assert np.array_equal(K3[1], K[0]) and np.array_equal(d3[2], d[1])
    # the ten line-search candidates
    al = (C.c_double * 10)()
    assert L.dpilqr_alphas(C.byref(al)) == 0 and np.array_equal(np.array(al), z["alphas"])
    Kg = np.ascontiguousarray(np.broadcast_to(z["K"], (B, T, m, n))); dg = np.ascontiguousarray(np.broadcast_to(z["d"], (B, T, m)))
    Xn = np.empty((B, 10, T + 1, n)); Un = np.empty((B, 10, T, m)); Jn = np.empty((B, 10))
    assert L.dpilqr_forward_pass(D.ref(), hp(Xo), hp(Uo), hp(Kg), hp(dg), hp(np.array(al)), 10, hp(Xn), hp(Un), hp(Jn), None) == 0
    assert relerr(Xn[1], z["X_fwd"]) < 1e-9 and relerr(Un[2], z["U_fwd"]) < 1e-9
    with np.errstate(invalid="ignore"):
        assert np.all((np.abs(Jn[0] - z["J_fwd"]) < 1e-10 * np.abs(z["J_fwd"])) | (np.isnan(Jn[0]) & np.isnan(z["J_fwd"])))
    # the joint cost at the operating point
    c = np.empty((B, T))
    assert L.dpilqr_cost_eval(D.ref(), T, hp(np.ascontiguousarray(Xo[:, :T])), hp(Uo), 0, hp(c), None) == 0
    cT = np.empty((B, 1)); uz = np.zeros((B, 1, m))
    assert L.dpilqr_cost_eval(D.ref(), 1, hp(np.ascontiguousarray(Xo[:, T:])), hp(uz), 1, hp(cT), None) == 0
    Jsum = c[0].sum() + cT[0, 0]
    X1 = np.empty((B, T + 1, n)); J1 = np.empty(B)
    L.dpilqr_rollout(D.ref(), hp(np.ascontiguousarray(Xo[:, 0])), hp(Uo), hp(X1), hp(J1), None)
    assert abs(Jsum - J1[0]) < 1e-12 * abs(J1[0])


def test_solve_batch_through_the_abi(twin, golden):
    """dpilqr_solve_batch over a batch of four cfg2 seeds: trace rows, statuses and counters as the header documents them"""
    L, _lib = twin
    z = golden("g4_solves_cfg2"); zp = golden("g3_passes_cfg2_di4d_k5")
    seeds = [0, 1, 2, 3]
    B, n_iter = len(seeds), 50
    D = Desc(_lib, zp, B=B, xf=np.stack([z[f"s{s}_xf"] for s in seeds]))
    x0 = np.stack([z[f"s{s}_x0"] for s in seeds]); U = np.zeros((B, D.T, D.m))
    X = np.empty((B, D.T + 1, D.n)); J = np.empty(B); st = np.zeros(B, dtype=np.int32); nb = np.zeros(B, dtype=np.int32); nf = np.zeros(B, dtype=np.int32)
    trace = np.zeros((B, n_iter, 5))
    sv = C.c_void_p()
    assert L.dpilqr_solver_create(C.byref(sv)) == 0
    seen = []
    cb = _lib.PROGRESS_FN(lambda user, done, total: seen.append((done, total)))
    assert L.dpilqr_solver_set_progress(sv, C.cast(cb, C.c_void_p), None) == 0
    assert L.dpilqr_solve_batch(sv, D.ref(), hp(x0), hp(U), n_iter, 1e-3, 0.0, 0, None, 0, hp(X), hp(J), hp(st), hp(nb), hp(nf), hp(trace), None, None, None) == 0
    assert seen and seen[-1] == (B, B)         # the header: "a last time with n_finished = n_items"
    assert L.dpilqr_solver_set_progress(sv, None, None) == 0
    assert L.dpilqr_solver_destroy(sv) == 0
    for i, s in enumerate(seeds):
        pre = f"s{s}_"
        k = len(z[pre + "mu_trace"])
        assert nb[i] == k and nf[i] == z[pre + "nfwd_trace"].sum()
        np.testing.assert_array_equal(trace[i, :k, 0], z[pre + "mu_trace"])
        np.testing.assert_array_equal(trace[i, :k, 1].astype(int), z[pre + "acc_trace"])
        np.testing.assert_array_equal(trace[i, :k, 4].astype(int), z[pre + "nfwd_trace"])
        assert relerr(X[i], z[pre + "X"]) < 1e-7 and relerr(U[i], z[pre + "U"]) < 1e-7
        assert abs(J[i] - z[pre + "J"]) < 1e-7 * abs(z[pre + "J"])
        assert st[i] == (_lib.STATUS_LINESEARCH_FAILED if z[pre + "acc_trace"][-1] < 0 else _lib.STATUS_CONVERGED)


def test_pairwise_graph_through_the_abi(twin, golden):
    L, _ = twin
    z = golden("g5_dispatch")
    for tag in ("uni5", "quad10", "uni8"):
        k, ns = int(z[tag + "_k"]), int(z[tag + "_n_s"])
        rad = np.array([0.5])
        for Xs, ref in ((z[tag + "_x0"].reshape(1, -1), z[tag + "_adj_x0"]), (z[tag + "_X_dec"], z[tag + "_adj_traj"])):
            Xs = np.ascontiguousarray(Xs, dtype=np.float64)
            adj = np.empty((1, k, k), dtype=np.int32)
            assert L.dpilqr_pairwise_graph(1, Xs.shape[0], k, ns, hp(Xs), hp(rad), hp(adj), None) == 0
            np.testing.assert_array_equal(adj[0], ref)
