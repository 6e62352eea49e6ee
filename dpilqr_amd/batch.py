"""Batched device solver: the lowering target of ilqrProblem and the one-launch-per-pass API.

A `ProblemBatch` is B sub-problems of one shape (k agents x (n_s, n_c), horizon T) held as device
tensors behind a `dpilqr_batch_desc` (include/dpilqr_hip.h).  It replaces the reference's per-problem
Python objects on the hot path:

    ilqrSolver._rollout        control.py:80-93    -> ProblemBatch.rollout
    ilqrSolver._backward_pass  control.py:116-148  -> ProblemBatch.make_tiles + backward_pass_tiles
    ilqrSolver._forward_pass   control.py:95-114   -> ProblemBatch.forward_pass
    ilqrSolver.solve           control.py:150-225  -> ProblemBatch.solve
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from .device import device, empty, ptr, stream_handle, to_dev, zeros

MODEL_DIMS = {0: (4, 2), 1: (6, 3), 2: (3, 2), 3: (4, 2), 4: (6, 3), 5: (6, 3), 6: (6, 3), 7: (12, 4), 8: (12, 4)}


def _shared_or_batched(a, B, per_item_shape, dtype):
    """Returns (device tensor, batch stride in elements): stride 0 when one copy serves the batch."""
    if isinstance(a, torch.Tensor):           # already on the device (the dispatch front end builds batches there)
        per = int(np.prod(per_item_shape))
        t = a.to(device=device(), dtype=dtype).contiguous()
        if t.numel() == per:
            return t, 0
        if t.numel() == B * per:
            return t, per
        raise ValueError(f"expected {per_item_shape} or {(B,) + tuple(per_item_shape)}, got {tuple(t.shape)}")
    a = np.asarray(a)
    per = int(np.prod(per_item_shape))
    if a.size == per:
        return to_dev(a.reshape(per_item_shape), dtype), 0
    if a.size == B * per:
        return to_dev(a.reshape((B,) + tuple(per_item_shape)), dtype), per
    raise ValueError(f"expected {per_item_shape} or {(B,) + tuple(per_item_shape)}, got {a.shape}")


class _WorkspacePool:
    """Solve workspaces (GBs each: tile records, gains, line-search candidates of a window of items), shared by all
    ProblemBatch objects and host threads of the process.  The dispatch front end solves one bucket per cluster size,
    concurrently, with a different workspace size each and different sizes again on the next call; handing every one
    to the caching allocator as a fresh odd-sized block made it hoard > 200 GB and then stall for seconds while it
    gave them back.  Buffers are rounded up to 256 MiB, the smallest free one that fits is reused, a solve returns its
    buffer when dpilqr_solve_batch has synchronised.  The pool is bounded by COUNT and by BYTES (a quarter of the
    device's memory): the buffers are live tensors, which the allocator's own out-of-memory recovery cannot reclaim."""
    GRANULE = 1 << 28
    MAX_FREE = 24
    MAX_FRACTION = 0.25

    def __init__(self):
        import threading
        self._lock = threading.Lock()
        self._free = []

    def _budget(self, dev):
        return int(torch.cuda.get_device_properties(dev).total_memory * self.MAX_FRACTION)

    def acquire(self, nbytes):
        dev = device()
        with self._lock:
            self._free = [t for t in self._free if t.device == dev]      # buffers of another device are dropped
            fit = [i for i, t in enumerate(self._free) if t.device == dev and t.numel() >= nbytes]
            if fit:
                return self._free.pop(min(fit, key=lambda i: self._free[i].numel()))
        size = max(1, -(-int(nbytes) // self.GRANULE)) * self.GRANULE
        try:
            return torch.empty(size, dtype=torch.uint8, device=dev)
        except getattr(torch, "OutOfMemoryError", torch.cuda.OutOfMemoryError):
            self.clear()                       # the pooled buffers are the likeliest reason: give them back and retry once
            torch.cuda.empty_cache()
            return torch.empty(size, dtype=torch.uint8, device=dev)

    def release(self, buf):
        with self._lock:
            self._free.append(buf)
            budget = self._budget(buf.device)
            # keep the large ones (they serve every request) within the count and byte bounds
            while len(self._free) > self.MAX_FREE or (len(self._free) > 1 and sum(t.numel() for t in self._free) > budget):
                self._free.pop(min(range(len(self._free)), key=lambda i: self._free[i].numel()))

    def clear(self):
        with self._lock:
            self._free.clear()


_workspace_pool = _WorkspacePool()


def release_workspaces():
    """Hand the pooled solve workspaces back to the allocator (they are kept between solves otherwise)."""
    _workspace_pool.clear()


class ProblemBatch:
    """B independent sub-problems with identical (k, n_s, n_c, T), resident in HBM.

    model, n_dims : (k,) shared or (B,k) per item     xf : (B, k*n_s)
    Q, Qf : (n_s,n_s) | (k,n_s,n_s) | (B,k,n_s,n_s)   R likewise with n_c     radius : scalar | (B,)
    """

    def __init__(self, model, n_dims, xf, Q, R, Qf, radius, dt, T, w_ref=1.0, w_prox=200.0, B=None, hints=None):
        """hints: (k, n_s, n_c, uniform_model word) for batches whose model / n_dims / weight arrays are device tensors
        (the dispatch front end): the constructor then neither reads them back nor derives the kernel hints from them."""
        lib = _lib.load()
        if hints is not None:
            self._init_from_device(lib, model, n_dims, xf, Q, R, Qf, radius, dt, T, w_ref, w_prox, B, hints)
            return
        xf = np.asarray(xf, dtype=np.float64) if not isinstance(xf, torch.Tensor) else xf
        model = np.asarray(model, dtype=np.int32)
        self.k = int(model.shape[-1])
        m0 = int(model.reshape(-1)[0])
        self.n_s, self.n_c = MODEL_DIMS[m0]
        if any(MODEL_DIMS[int(v)] != (self.n_s, self.n_c) for v in np.unique(model)):
            raise ValueError("all agents of a batch must share (n_s, n_c) -- the reference assumes it too "
                             "(dynamics.py:165-166)")
        self.n_x, self.n_u = self.k * self.n_s, self.k * self.n_c
        xf = xf.reshape(-1, self.n_x)
        self.B = int(B if B is not None else xf.shape[0])
        self.T, self.dt = int(T), float(dt)
        self.w_ref, self.w_prox = float(w_ref), float(w_prox)
        B_, k, ns, nc = self.B, self.k, self.n_s, self.n_c

        def expand(M, n):  # (n,n) -> (k,n,n)
            M = np.asarray(M, dtype=np.float64)
            return np.broadcast_to(M, (k, n, n)) if M.ndim == 2 else M

        self._model, ms = _shared_or_batched(model, B_, (k,), torch.int32)
        self._n_dims, ds = _shared_or_batched(np.asarray(n_dims, dtype=np.int32), B_, (k,), torch.int32)
        self._xf, xs = _shared_or_batched(xf, B_, (self.n_x,), torch.float64)
        if xf.shape[0] == 1 and B_ > 1:
            xs = 0
        self._Q, qs = _shared_or_batched(expand(Q, ns), B_, (k, ns, ns), torch.float64)
        self._R, rs = _shared_or_batched(expand(R, nc), B_, (k, nc, nc), torch.float64)
        self._Qf, fs = _shared_or_batched(expand(Qf, ns), B_, (k, ns, ns), torch.float64)
        self._radius, ras = _shared_or_batched(np.asarray(radius, dtype=np.float64), B_, (1,), torch.float64)
        uniform = 1 + m0 if bool((model == m0).all()) else 0      # hints for model-specialised kernels (dpilqr_hip.h)
        nd_all = np.asarray(n_dims, dtype=np.int32)
        if nd_all.size and bool((nd_all == nd_all.reshape(-1)[0]).all()):
            uniform |= (1 + int(nd_all.reshape(-1)[0])) << 8
        Qk, Rk, Qfk = (np.asarray(expand(M_, n_)) for M_, n_ in ((Q, ns), (R, nc), (Qf, ns)))
        if qs == 0 and rs == 0 and fs == 0 and all(M_.ndim == 3 and bool((M_ == M_[0]).all()) for M_ in (Qk, Rk, Qfk)):
            uniform |= 1 << 16                                       # one Q, R, Q_f for every agent of every item
        if bool(np.isin(np.asarray(model), (0, 3)).all()):
            uniform |= 1 << 17                                       # DoubleIntDynamics4D / UnicycleDynamics4D agents only
        self.desc = _lib.BatchDesc(B_, k, ns, nc, self.T, uniform, self.dt, self.w_ref, self.w_prox,
                                   ptr(self._model), ms, ptr(self._n_dims), ds, ptr(self._xf), xs,
                                   ptr(self._Q), qs, ptr(self._R), rs, ptr(self._Qf), fs, ptr(self._radius), ras)
        self._lib = lib
        self.tile_offsets, self.tile_stride = _lib.tile_layout(self.n_x, self.n_u)

    def _init_from_device(self, lib, model, n_dims, xf, Q, R, Qf, radius, dt, T, w_ref, w_prox, B, hints):
        self.k, self.n_s, self.n_c, uniform = (int(v) for v in hints)
        k, ns, nc = self.k, self.n_s, self.n_c
        self.n_x, self.n_u = k * ns, k * nc
        self.B = B_ = int(B)
        self.T, self.dt = int(T), float(dt)
        self.w_ref, self.w_prox = float(w_ref), float(w_prox)
        self._model, ms = _shared_or_batched(model, B_, (k,), torch.int32)
        self._n_dims, ds = _shared_or_batched(n_dims, B_, (k,), torch.int32)
        self._xf, xs = _shared_or_batched(xf, B_, (self.n_x,), torch.float64)
        self._Q, qs = _shared_or_batched(Q, B_, (k, ns, ns), torch.float64)
        self._R, rs = _shared_or_batched(R, B_, (k, nc, nc), torch.float64)
        self._Qf, fs = _shared_or_batched(Qf, B_, (k, ns, ns), torch.float64)
        rad = radius if isinstance(radius, torch.Tensor) else np.asarray(radius, dtype=np.float64)
        self._radius, ras = _shared_or_batched(rad, B_, (1,), torch.float64)
        if qs or rs or fs:
            uniform &= ~(1 << 16)
        self.desc = _lib.BatchDesc(B_, k, ns, nc, self.T, uniform, self.dt, self.w_ref, self.w_prox,
                                   ptr(self._model), ms, ptr(self._n_dims), ds, ptr(self._xf), xs,
                                   ptr(self._Q), qs, ptr(self._R), rs, ptr(self._Qf), fs, ptr(self._radius), ras)
        self._lib = lib
        self.tile_offsets, self.tile_stride = _lib.tile_layout(self.n_x, self.n_u)

    @staticmethod
    def hint_word(model, n_dims, Q, R, Qf):
        """The uniform_model hints (include/dpilqr_hip.h) of a k-agent problem whose host arrays are given."""
        model = np.asarray(model, dtype=np.int32); nd = np.asarray(n_dims, dtype=np.int32)
        w = (1 + int(model.reshape(-1)[0])) if bool((model == model.reshape(-1)[0]).all()) else 0
        if nd.size and bool((nd == nd.reshape(-1)[0]).all()):
            w |= (1 + int(nd.reshape(-1)[0])) << 8
        if all(np.asarray(M).ndim == 3 and bool((np.asarray(M) == np.asarray(M)[0]).all()) for M in (Q, R, Qf)):
            w |= 1 << 16
        if model.size and bool(np.isin(model, (0, 3)).all()):
            w |= 1 << 17                                             # DoubleIntDynamics4D / UnicycleDynamics4D agents only
        return w

    # ------------------------------------------------------------------ helpers
    @property
    def _d(self):
        return C.byref(self.desc)

    def tiles_buffer(self):
        return empty((self.B, self.T + 1, self.tile_stride))

    # ------------------------------------------------------------------ passes
    # dtype=torch.float32 selects the fp32 arm of BASELINE config 5's tolerance study (the *_f32 entry points): the
    # same passes with trajectories, gains and every intermediate in float; costs and the solver state stay double.
    @property
    def fused_sweep(self):
        """Clusters with n_x > 60 take the large-cluster path: linearize / quadraticize are evaluated inside the sweep,
        no tile records exist (one would be 1.28 MB at n_x = 240)."""
        return self.n_x > 60

    def _in(self, a, shape, dtype=torch.float64):
        t = to_dev(a, dtype)
        if tuple(t.shape) != tuple(shape):
            t = t.reshape(shape)
        return t.contiguous()

    def rollout(self, x0, U, dtype=torch.float64):
        """control.py:80-93 for every item: returns X (B,T+1,n_x), J (B,) device tensors."""
        x0 = self._in(x0, (self.B, self.n_x), dtype); U = self._in(U, (self.B, self.T, self.n_u), dtype)
        X = empty((self.B, self.T + 1, self.n_x), dtype); J = empty((self.B,))
        fn = self._lib.dpilqr_rollout if dtype == torch.float64 else self._lib.dpilqr_rollout_f32
        _lib.check(fn(self._d, ptr(x0), ptr(U), ptr(X), ptr(J), stream_handle()))
        return X, J

    def make_tiles(self, X, U, tiles=None):
        """linearize + quadraticize at every (X[t],U[t]) -> packed tile records (B,T+1,stride)."""
        X = self._in(X, (self.B, self.T + 1, self.n_x)); U = self._in(U, (self.B, self.T, self.n_u))
        tiles = self.tiles_buffer() if tiles is None else tiles
        _lib.check(self._lib.dpilqr_make_tiles(self._d, ptr(X), ptr(U), ptr(tiles), None, None, stream_handle()))
        return tiles

    def unpack_tiles(self, tiles):
        """Tile records -> dict of host arrays shaped like the plugin returns (per item, per step)."""
        t = tiles.cpu().numpy(); n, m = self.n_x, self.n_u
        shapes = dict(A=(n, n), B=(n, m), Lxx=(n, n), Lux=(m, n), Luu=(m, m), Lx=(1, n), Lu=(1, m))
        out = {}
        for key, (rows, cols) in shapes.items():
            off, ld = self.tile_offsets[key]
            idx = off + (np.arange(rows) * ld)[:, None] + np.arange(cols)[None, :]
            a = t[:, :, idx]
            out[key] = a[:, :, 0, :] if key in ("Lx", "Lu") else a
        return out

    def backward_pass(self, X, U, mu, tiles=None, dtype=torch.float64):
        """control.py:116-148: K (B,T,n_u,n_x), d (B,T,n_u)."""
        if dtype == torch.float64 and not self.fused_sweep and not _lib.route_flag("DPILQR_FORCE_BIG"):
            tiles = self.make_tiles(X, U, tiles)
            return backward_pass_tiles(tiles, self.B, self.T, self.n_x, self.n_u, mu, blocks=(self.n_s, self.n_c))
        B, T, n, m = self.B, self.T, self.n_x, self.n_u
        X = self._in(X, (B, T + 1, n), dtype); U = self._in(U, (B, T, m), dtype)
        mu_t = to_dev(np.broadcast_to(np.asarray(mu, dtype=np.float64), (B,))) if not isinstance(mu, torch.Tensor) else mu
        K = empty((B, T, m, n), dtype); d = empty((B, T, m), dtype)
        nbytes = self._lib.dpilqr_backward_pass_workspace_bytes(self._d, 8 if dtype == torch.float64 else 4)
        _lib.check(nbytes)
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device())
        fn = self._lib.dpilqr_backward_pass if dtype == torch.float64 else self._lib.dpilqr_backward_pass_f32
        _lib.check(fn(self._d, ptr(X), ptr(U), ptr(mu_t), ptr(K), ptr(d), ptr(ws), stream_handle()))
        return K, d

    def backward_pass_fused(self, X, U, mu):
        """The backward pass without tile records (dpilqr_backward_pass_fused); raises DpilqrError(EUNSUPPORTED) for batches
        the fused sweep does not serve."""
        B, T, n, m = self.B, self.T, self.n_x, self.n_u
        X = self._in(X, (B, T + 1, n)); U = self._in(U, (B, T, m))
        mu_t = to_dev(np.broadcast_to(np.asarray(mu, dtype=np.float64), (B,))) if not isinstance(mu, torch.Tensor) else mu
        K = empty((B, T, m, n)); d = empty((B, T, m))
        _lib.check(self._lib.dpilqr_backward_pass_fused(self._d, ptr(X), ptr(U), ptr(mu_t), ptr(K), ptr(d), None, stream_handle()))
        return K, d

    def forward_pass(self, X, U, K, d, alphas, dtype=torch.float64):
        """control.py:95-114 for all alphas: Xn (B,A,T+1,n_x), Un (B,A,T,n_u), Jn (B,A)."""
        X = self._in(X, (self.B, self.T + 1, self.n_x), dtype); U = self._in(U, (self.B, self.T, self.n_u), dtype)
        K = self._in(K, (self.B, self.T, self.n_u, self.n_x), dtype); d = self._in(d, (self.B, self.T, self.n_u), dtype)
        al = to_dev(np.asarray(alphas, dtype=np.float64)); A = int(al.numel())
        Xn = empty((self.B, A, self.T + 1, self.n_x), dtype); Un = empty((self.B, A, self.T, self.n_u), dtype); Jn = empty((self.B, A))
        fn = self._lib.dpilqr_forward_pass if dtype == torch.float64 else self._lib.dpilqr_forward_pass_f32
        _lib.check(fn(self._d, ptr(X), ptr(U), ptr(K), ptr(d), ptr(al), A, ptr(Xn), ptr(Un), ptr(Jn), stream_handle()))
        return Xn, Un, Jn

    def cost(self, x, u, terminal=False):
        """GameCost.__call__ at n_pts points per item: x (B,n_pts,n_x), u (B,n_pts,n_u) -> (B,n_pts)."""
        x = to_dev(x).reshape(self.B, -1, self.n_x).contiguous()
        n_pts = x.shape[1]
        u = to_dev(u).reshape(self.B, n_pts, self.n_u).contiguous()
        out = empty((self.B, n_pts))
        _lib.check(self._lib.dpilqr_cost_eval(self._d, n_pts, ptr(x), ptr(u), int(bool(terminal)), ptr(out),
                                              stream_handle()))
        return out

    # ------------------------------------------------------------------ whole solve
    def workspace_bytes(self, window, gains_in_ws, dtype=torch.float64):
        fn = self._lib.dpilqr_solve_workspace_bytes if dtype == torch.float64 else self._lib.dpilqr_solve_workspace_bytes_f32
        nbytes = fn(self._d, int(window), int(bool(gains_in_ws)))
        _lib.check(nbytes)
        return int(nbytes)

    def default_window(self, dtype=torch.float64):
        """Items in flight when the caller does not say: 6144 (two rounds of three sweep wavefronts per SIMD on an
        MI355X) for small clusters, fewer where the per-item buffers (gains, line-search candidates; tile records where a
        record-fed sweep serves the batch) are large, so that one solve's workspace stays near 4 GB: several cluster sizes
        are solved concurrently (dispatch.py), and workspaces of 10 GB each did not fit the pool of reusable buffers --
        the fresh multi-GB allocations that followed cost a many-scenario call up to half a second, at random.  The
        workgroup-per-item sweeps hold 2-4 sub-problems per CU, so a thousand in flight fill the device: never below
        1024, except on the large-cluster path (one workgroup per item: 256 = one per CU)."""
        per_item = self.workspace_bytes(2, True, dtype) - self.workspace_bytes(1, True, dtype)
        floor = 256 if (self.fused_sweep or dtype != torch.float64) else 1024
        w = max(floor, (4 << 30) // max(per_item, 1))
        if self.n_x <= 24 and dtype == torch.float64:
            # the wavefront sweeps deal items in layers of 1024 (one wavefront per SIMD of every CU): a window of 2450 items
            # would run every sweep as a full round plus a fifth of one
            w = max(1024, (w // 1024) * 1024)
        return int(min(self.B, 6144, w))

    def solve(self, x0, U0, n_lqr_iter=50, tol=1e-3, trace=False, gains=False, window=None, dtype=torch.float64, out=None,
              progress=None, t_kill=None):
        """ilqrSolver.solve (control.py:150-225) for all B items.

        window: most items in flight at once (default: default_window()); finished items are retired on the device and
        replaced by not-yet-started ones, so launches stay full and memory is bounded.
        out: dict of pre-allocated device tensors X (B,T+1,n_x), U (B,T,n_u), J (B,), status, n_bwd, n_fwd (B,) int32 to
        write the results into (sharding.ResultBuffers: the solve then writes straight into the collective's send buffer).
        progress: callable(n_finished, n_items), called from inside the solve as a PREFIX of the batch finishes (X, U,
        status, n_bwd, n_fwd of items below n_finished are final; see dpilqr_solver_set_progress).
        t_kill: seconds of solve time each item may use (control.py:213-218, every item's clock starts when it is admitted
        to the window); an item whose time is up after an accepted, unconverged step ends with status STATUS_KILLED and
        that step's iterate.  None / 0: no limit.
        Returns a dict of device tensors: X, U, J, status, n_bwd, n_fwd (+ trace, K, d on request).
        """
        B, T, n, m = self.B, self.T, self.n_x, self.n_u
        window = self.default_window(dtype) if window is None else int(window)
        x0 = self._in(x0, (B, n), dtype)
        if out is not None:
            U = out["U"]; U.copy_(self._in(U0, (B, T, m), dtype))
            X, J, status, n_bwd, n_fwd = out["X"], out["J"], out["status"], out["n_bwd"], out["n_fwd"]
            for t_, shp, dt_ in ((X, (B, T + 1, n), dtype), (U, (B, T, m), dtype), (J, (B,), torch.float64),
                                 (status, (B,), torch.int32), (n_bwd, (B,), torch.int32), (n_fwd, (B,), torch.int32)):
                if tuple(t_.shape) != shp or t_.dtype != dt_ or not t_.is_contiguous():
                    raise ValueError(f"out tensor of shape {tuple(t_.shape)} / {t_.dtype}: expected contiguous {shp} {dt_}")
        else:
            U = self._in(U0, (B, T, m), dtype).clone()
            X = empty((B, T + 1, n), dtype); J = empty((B,))
            status = empty((B,), torch.int32); n_bwd = empty((B,), torch.int32); n_fwd = empty((B,), torch.int32)
        tr = torch.full((B, max(n_lqr_iter, 1), 5), float("nan"), dtype=torch.float64, device=device()) if trace else None
        K = empty((B, T, m, n), dtype) if gains else None
        d = empty((B, T, m), dtype) if gains else None
        ws = _workspace_pool.acquire(self.workspace_bytes(window, not gains, dtype))
        fn = self._lib.dpilqr_solve_batch if dtype == torch.float64 else self._lib.dpilqr_solve_batch_f32
        out = dict(X=X, U=U, J=J, status=status, n_bwd=n_bwd, n_fwd=n_fwd)
        if trace:
            out["trace"] = tr
        if gains:
            out["K"], out["d"] = K, d
        ok = False
        try:
            with _lib.progress_callback(progress):
                _lib.check(fn(_lib.solver(), self._d, ptr(x0), ptr(U), int(n_lqr_iter), float(tol), float(t_kill or 0.0), window,
                              ptr(ws), ws.numel(),
                              ptr(X), ptr(J), ptr(status), ptr(n_bwd), ptr(n_fwd), ptr(tr), ptr(K), ptr(d), stream_handle()))
            ok = True
        except _lib.DpilqrError as e:
            # DPILQR_EHIP because the device gave items up (STATUS_FAULT): the solve ran to its end, the other items' results
            # are valid -- they travel with the exception
            e.results = out
            raise
        finally:
            if ok:
                _workspace_pool.release(ws)   # solve_batch has synchronised its stream: the buffer is idle
            # a failed solve's buffer is dropped, not pooled: the allocator frees it in stream order
        return out

    def solve_enqueue(self, x0, U0, n_global_iter, n_lqr_iter=None, tol=None, window=None, state=None, t_kill=None):
        """The same solve as pure enqueue on torch's current stream (dpilqr_solve_enqueue): nothing is waited for.
        Returns (results dict, state); pass `state` back to continue with another n_global_iter iterations.  An item is
        finished when its status is no longer 0 (STATUS_ACTIVE).
        n_lqr_iter (default 50), tol (1e-3) and t_kill (none) are fixed by the FIRST call and kept in `state`: a continuing call
        need not repeat them and may not change them (the per-item admission clocks t_kill reads are stamped only by a solve that
        started with a limit)."""
        B, T, n, m = self.B, self.T, self.n_x, self.n_u
        if state is None:
            window = self.default_window() if window is None else int(window)
            x0 = self._in(x0, (B, n)); U = self._in(U0, (B, T, m)).clone()
            r = dict(X=empty((B, T + 1, n)), U=U, J=empty((B,)), status=empty((B,), torch.int32),
                     n_bwd=empty((B,), torch.int32), n_fwd=empty((B,), torch.int32))
            ws = torch.empty(self.workspace_bytes(window, True), dtype=torch.uint8, device=device())
            state = dict(r=r, ws=ws, x0=x0, window=window, resume=0, n_lqr_iter=int(50 if n_lqr_iter is None else n_lqr_iter),
                         tol=float(1e-3 if tol is None else tol), t_kill=float(t_kill or 0.0))
        else:
            for name, given in (("n_lqr_iter", n_lqr_iter), ("tol", tol), ("t_kill", t_kill)):
                if given is not None and float(given) != float(state[name]):
                    raise ValueError(f"solve_enqueue: {name}={given} on a continuing call, the solve was started with {state[name]}")
        r, ws = state["r"], state["ws"]
        _lib.check(self._lib.dpilqr_solve_enqueue(self._d, ptr(state["x0"]), ptr(r["U"]), state["n_lqr_iter"], state["tol"],
                                                  state["t_kill"], state["window"], ptr(ws), ws.numel(), ptr(r["X"]), ptr(r["J"]),
                                                  ptr(r["status"]), ptr(r["n_bwd"]), ptr(r["n_fwd"]), None, None, None,
                                                  int(n_global_iter), state["resume"], stream_handle()))
        state["resume"] = 1
        return r, state

    def iterations_bound(self, n_lqr_iter=50, window=None):
        window = self.default_window() if window is None else int(window)
        return int(self._lib.dpilqr_solve_iterations_bound(self._d, window, int(n_lqr_iter)))


def backward_pass_tiles(tiles, B, T, n_x, n_u, mu, singular=None, blocks=None):
    """The Riccati sweep on explicit tile records -- the plugin contract (any linearize/quadraticize).

    blocks=(n_s, n_c): the caller's promise that A, B of every record are block diagonal with per-agent
    blocks of that size (MultiDynamicalModel.linearize, dynamics.py:173-186); same gains, fewer products."""
    lib = _lib.load()
    mu_t = to_dev(np.broadcast_to(np.asarray(mu, dtype=np.float64), (B,))) if not isinstance(mu, torch.Tensor) else mu
    K = empty((B, T, n_u, n_x)); d = empty((B, T, n_u))
    ns, nc = blocks if blocks else (0, 0)
    _lib.check(lib.dpilqr_backward_pass_tiles_blocks(B, T, n_x, n_u, ns, nc, ptr(tiles), ptr(mu_t), ptr(K), ptr(d),
                                                     ptr(singular), None, None, stream_handle()))
    return K, d


def pack_tiles(A, Bm, Lx, Lu, Lxx, Luu, Lux):
    """Host plugin outputs -> device tile records.

    A (B,T,n,n), Bm (B,T,n,m); Lx (B,T+1,n), Lu (B,T+1,m), Lxx (B,T+1,n,n), Luu (B,T+1,m,m),
    Lux (B,T+1,m,n), entry T being the terminal quadraticisation.
    """
    A = np.asarray(A, dtype=np.float64); Bm = np.asarray(Bm, dtype=np.float64)
    Bn, T, n, m = Bm.shape
    lay, stride = _lib.tile_layout(n, m)
    rec = np.zeros((Bn, T + 1, stride))

    def put(key, arr, rows, cols, steps):
        off, ld = lay[key]
        idx = off + (np.arange(rows) * ld)[:, None] + np.arange(cols)[None, :]
        rec[:, :steps, idx] = np.asarray(arr, dtype=np.float64).reshape(Bn, steps, rows, cols)

    put("A", A, n, n, T); put("B", Bm, n, m, T)
    put("Lxx", Lxx, n, n, T + 1); put("Lux", Lux, m, n, T + 1); put("Luu", Luu, m, m, T + 1)
    put("Lx", Lx, 1, n, T + 1); put("Lu", Lu, 1, m, T + 1)
    return to_dev(rec)
