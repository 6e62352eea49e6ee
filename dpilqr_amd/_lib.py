"""ctypes binding of libdpilqr_hip.so (include/dpilqr_hip.h).

The HIP library is the product: there is NO CPU fallback.  If the shared object is missing, or no
gfx950 device is visible when a compute entry point is called, this module raises.
"""
import ctypes as C
import os
from pathlib import Path

HERE = Path(__file__).resolve().parent


def route_env(name):
    """The library's A/B route switches (csrc/launch.hpp: route_env) and this binding's DPILQR_LIB override are for experiments
    and tests: they count only in a process that sets DPILQR_DEBUG_ROUTES=1.  A production process ignores them all."""
    return os.environ.get(name) if os.environ.get("DPILQR_DEBUG_ROUTES") == "1" else None


def route_flag(name):
    return route_env(name) is not None


LIB_PATH = Path(route_env("DPILQR_LIB")) if route_env("DPILQR_LIB") else HERE / "libdpilqr_hip.so"   # override: A/B builds only

i32, i64, f64, vp = C.c_int32, C.c_int64, C.c_double, C.c_void_p

OK, EINVAL, EUNSUPPORTED, EHIP, ENOGPU, EWORKSPACE = 0, -1, -2, -3, -4, -5
N_ALPHA = 10
ABI_VERSION = 4          # DPILQR_ABI_VERSION of include/dpilqr_hip.h
STATUS_ACTIVE, STATUS_CONVERGED, STATUS_LINESEARCH_FAILED, STATUS_MAX_ITER, STATUS_SINGULAR, STATUS_KILLED, STATUS_FAULT = 0, 1, 2, 3, 4, 5, 6


class BatchDesc(C.Structure):
    """struct dpilqr_batch_desc"""
    _fields_ = [("B", i32), ("k", i32), ("n_s", i32), ("n_c", i32), ("T", i32), ("uniform_model", i32),
                ("dt", f64), ("w_ref", f64), ("w_prox", f64),
                ("model", vp), ("model_bstride", i64), ("n_dims", vp), ("n_dims_bstride", i64),
                ("xf", vp), ("xf_bstride", i64), ("Q", vp), ("Q_bstride", i64), ("R", vp), ("R_bstride", i64),
                ("Qf", vp), ("Qf_bstride", i64), ("radius", vp), ("radius_bstride", i64)]


class DpilqrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libdpilqr_hip error {code}: {msg}")
        self.code = code


# name -> (restype, argtypes); every symbol include/dpilqr_hip.h declares
_DP = C.POINTER(BatchDesc)
SIGNATURES = {
    "dpilqr_abi_version": (i32, []),
    "dpilqr_last_error": (C.c_char_p, []),
    "dpilqr_device_info": (i32, [i32, C.POINTER(i32), C.POINTER(i32), C.c_char_p, i32]),
    "dpilqr_model_dims": (i32, [i32, C.POINTER(i32), C.POINTER(i32)]),
    "dpilqr_model_f": (i32, [i32, i32, vp, vp, vp, vp, vp]),
    "dpilqr_model_integrate": (i32, [i32, i32, vp, vp, vp, f64, vp, vp]),
    "dpilqr_model_linearize": (i32, [i32, i32, vp, vp, vp, f64, vp, vp, vp]),
    "dpilqr_cost_eval": (i32, [_DP, i32, vp, vp, i32, vp, vp]),
    "dpilqr_tile_layout": (i32, [i32, i32, C.POINTER(i64 * 7), C.POINTER(i64 * 7), C.POINTER(i64)]),
    "dpilqr_make_tiles": (i32, [_DP, vp, vp, vp, vp, vp, vp]),
    "dpilqr_rollout": (i32, [_DP, vp, vp, vp, vp, vp]),
    "dpilqr_backward_pass_tiles": (i32, [i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dpilqr_backward_pass_tiles_blocks": (i32, [i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dpilqr_tiles_bytes": (i64, [i32, i32, i32, i32]),
    "dpilqr_backward_pass": (i32, [_DP, vp, vp, vp, vp, vp, vp, vp]),
    "dpilqr_backward_pass_fused": (i32, [_DP, vp, vp, vp, vp, vp, vp, vp]),
    "dpilqr_forward_pass": (i32, [_DP, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp]),
    "dpilqr_alphas": (i32, [C.POINTER(f64 * N_ALPHA)]),
    "dpilqr_solve_workspace_bytes": (i64, [_DP, i32, i32]),
    "dpilqr_solver_create": (i32, [C.POINTER(vp)]),
    "dpilqr_solver_destroy": (i32, [vp]),
    "dpilqr_solver_set_progress": (i32, [vp, vp, vp]),
    "dpilqr_solve_batch": (i32, [vp, _DP, vp, vp, i32, f64, f64, i32, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dpilqr_solve_enqueue": (i32, [_DP, vp, vp, i32, f64, f64, i32, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "dpilqr_solve_iterations_bound": (i64, [_DP, i32, i32]),
    "dpilqr_rollout_f32": (i32, [_DP, vp, vp, vp, vp, vp]),
    "dpilqr_backward_pass_workspace_bytes": (i64, [_DP, i32]),
    "dpilqr_backward_pass_f32": (i32, [_DP, vp, vp, vp, vp, vp, vp, vp]),
    "dpilqr_forward_pass_f32": (i32, [_DP, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp]),
    "dpilqr_solve_workspace_bytes_f32": (i64, [_DP, i32, i32]),
    "dpilqr_solve_batch_f32": (i32, [vp, _DP, vp, vp, i32, f64, f64, i32, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dpilqr_debug_stamps": (i32, [vp]),
    "dpilqr_profile_enable": (i32, [vp, i32]),
    "dpilqr_profile_read": (i32, [vp, C.POINTER(f64 * 4), C.POINTER(i64 * 4), C.POINTER(i64 * 4), i32]),
    "dpilqr_profile_read_sweep": (i32, [vp, i32, C.POINTER(f64), C.POINTER(i64), C.POINTER(i64), i32]),
    "dpilqr_pairwise_graph": (i32, [i32, i32, i32, i32, vp, vp, vp, vp]),
    "dpilqr_dispatch_graph": (i32, [i32, i32, i32, i32, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dpilqr_dispatch_gather": (i32, [i32, i32, i32, i32, i32, i32, vp, i32, i32, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp]),
    "dpilqr_dispatch_gather_params": (i32, [i32, i32, i32, i32, vp, vp, vp, vp]),
    "dpilqr_dispatch_stitch": (i32, [i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dpilqr_dispatch_pack_rows": (i32, [i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp]),
    "dpilqr_dispatch_scatter_rows": (i32, [i64, i32, i32, i32, i32, vp, i64, vp, vp, vp]),
    "dpilqr_random_setup": (i32, [i32, i64, i32, i32, i32, f64, f64, vp, vp, vp]),
}

MAX_AGENTS = 64


class BucketResults(C.Structure):
    """struct dpilqr_bucket_results"""
    _fields_ = [("X", vp * (MAX_AGENTS + 1)), ("U", vp * (MAX_AGENTS + 1)), ("first", i32 * (MAX_AGENTS + 1)),
                ("count", i32 * (MAX_AGENTS + 1))]

_lib = None


def load():
    """Load libdpilqr_hip.so; raises if it has not been built (run `python __graft_entry__.py`)."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950).  dpilqr_amd has no CPU fallback.")
        lib = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if lib.dpilqr_abi_version() != ABI_VERSION:
            raise ImportError(f"{LIB_PATH}: ABI version {lib.dpilqr_abi_version()}, this binding is for {ABI_VERSION}")
        _lib = lib
    return _lib


def check(rc):
    if rc < 0:
        raise DpilqrError(rc, load().dpilqr_last_error().decode())
    return rc


def alphas():
    a = (f64 * N_ALPHA)()
    check(load().dpilqr_alphas(C.byref(a)))
    return list(a)


TILE_PARTS = ["A", "B", "Lxx", "Lux", "Luu", "Lx", "Lu"]


def tile_layout(n_x, n_u):
    """-> ({part: (offset, row_stride)}, record_stride), all in doubles."""
    off, ld = (i64 * 7)(), (i64 * 7)()
    stride = i64(0)
    check(load().dpilqr_tile_layout(n_x, n_u, C.byref(off), C.byref(ld), C.byref(stride)))
    return {p: (off[i], ld[i]) for i, p in enumerate(TILE_PARTS)}, stride.value


PROFILE_CLASSES = {"tiles": 1, "riccati": 2, "forward": 4, "rollout": 8}


def profile_enable(on=True, classes=None):
    """classes: iterable of PROFILE_CLASSES names to bracket with events (None = all)."""
    mask = 0 if not classes else sum(PROFILE_CLASSES[c] for c in classes)
    return load().dpilqr_profile_enable(solver(), int(bool(on)) | (mask << 1))


def profile_read(reset=True):
    """Per kernel class (tiles, riccati, forward, rollout): total ms, launches, sub-problems processed."""
    ms, ln, it = (f64 * 4)(), (i64 * 4)(), (i64 * 4)()
    check(load().dpilqr_profile_read(solver(), C.byref(ms), C.byref(ln), C.byref(it), int(bool(reset))))
    names = ["tiles", "riccati", "forward", "rollout"]
    return {n: dict(ms=ms[i], launches=ln[i], items=it[i]) for i, n in enumerate(names)}


def profile_read_sweep(waves, reset=True):
    """The wavefront sweep's launches of one variant (waves = 4, 8 or 12 wavefronts per workgroup): ms, launches, items."""
    ms, ln, it = f64(), i64(), i64()
    check(load().dpilqr_profile_read_sweep(solver(), int(waves), C.byref(ms), C.byref(ln), C.byref(it), int(bool(reset))))
    return dict(ms=ms.value, launches=ln.value, items=it.value)


import threading

_solvers = threading.local()


def solver():
    """The calling thread's dpilqr_solver for the current device (the synchronous solve's pinned mailbox and events):
    created once, explicitly, so that no solve allocates behind the caller's back."""
    import torch
    dev = torch.cuda.current_device()
    table = getattr(_solvers, "table", None)
    if table is None:
        table = _solvers.table = {}
        import atexit
        atexit.register(_destroy_solvers, table)     # the pinned mailbox and events of this thread's solvers
    if dev not in table:
        h = vp()
        check(load().dpilqr_solver_create(C.byref(h)))
        table[dev] = h
    return table[dev]


PROGRESS_FN = C.CFUNCTYPE(None, vp, i32, i32)


class progress_callback:
    """with progress_callback(fn): ... -- fn(n_finished, n_items) is called from inside this thread's synchronous solves as
    a prefix of the batch finishes (dpilqr_solver_set_progress).

    ctypes swallows an exception raised inside a callback (it only prints it), and ranks whose callbacks issue collectives
    would then have issued different numbers of them: the first exception is kept, later calls of the callback do nothing,
    and leaving the `with` block re-raises it -- after the solve has returned, so no kernel is left queued behind the error."""

    def __init__(self, fn):
        self.error = None

        def call(user, done, total):
            if self.error is not None:
                return
            try:
                fn(int(done), int(total))
            except BaseException as e:          # noqa: BLE001 -- must not escape into the C caller
                self.error = e

        self._c = PROGRESS_FN(call) if fn is not None else None

    def __enter__(self):
        if self._c is not None:
            check(load().dpilqr_solver_set_progress(solver(), C.cast(self._c, vp), None))
        return self

    def __exit__(self, *exc):
        if self._c is not None:
            check(load().dpilqr_solver_set_progress(solver(), None, None))
        if self.error is not None and exc[0] is None:
            raise self.error
        return False


def _destroy_solvers(table):
    lib = _lib
    if lib is not None:
        for h in table.values():
            lib.dpilqr_solver_destroy(h)
    table.clear()


_device_ok = None


def require_gpu():
    """Raise unless a gfx950 device is usable through both torch and the HIP library."""
    global _device_ok
    if _device_ok is None:
        import torch
        if not torch.cuda.is_available():
            raise DpilqrError(ENOGPU, "torch.cuda.is_available() is False: dpilqr_amd needs an MI355X (no CPU fallback)")
        check(load().dpilqr_device_info(torch.cuda.current_device(), None, None, None, 0))
        _device_ok = True
    return True
