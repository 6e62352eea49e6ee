"""Device-backed counterpart of the reference's Cython module `dpilqr.bbdynamicswrap`
(bbdynamicswrap.pyx:8-16,61-164): the `Model` enum and the three per-agent operations
f / integrate / linearize, here evaluated by the HIP library (dpilqr_model_* in dpilqr_hip.h).

Single calls move a handful of doubles to the GPU and back -- this module exists so that code written
against the reference's FFI keeps working; the solver itself never calls it per time step.
"""
from enum import Enum

import numpy as np
import torch

from . import _lib
from .device import empty, ptr, stream_handle, to_dev


class Model(Enum):
    # same values as bbdynamicswrap.pyx:8-16 (declaration order)
    DoubleInt4D = 0
    DoubleInt6D = 1
    Car3D = 2
    Unicycle4D = 3
    Quadcopter6D = 4
    Human6D = 5
    HumanLin6D = 6
    Quadcopter12D = 7
    HumanPad12D = 8      # this library's own: HumanDynamics6D zero-padded to 12 states / 4 controls (dpilqr_hip.h)


MODEL_DIMS = {Model.DoubleInt4D: (4, 2), Model.DoubleInt6D: (6, 3), Model.Car3D: (3, 2), Model.Unicycle4D: (4, 2),
              Model.Quadcopter6D: (6, 3), Model.Human6D: (6, 3), Model.HumanLin6D: (6, 3), Model.Quadcopter12D: (12, 4),
              Model.HumanPad12D: (12, 4)}


def _check(model, x, u):
    if not isinstance(model, Model):
        raise ValueError("model must be a Model enum member")       # pyx:52-54
    x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
    u = np.ascontiguousarray(u, dtype=np.float64).reshape(-1)
    n_s, n_c = MODEL_DIMS[model]
    if x.size != n_s or u.size != n_c:
        raise ValueError(f"{model.name} expects x of {n_s} and u of {n_c} entries")
    return x, u, n_s, n_c


def _run(op, model, x, u, dt=0.0):
    x, u, n_s, n_c = _check(model, x, u)
    lib = _lib.load(); _lib.require_gpu()
    mid = to_dev(np.array([model.value]), torch.int32); xd, ud = to_dev(x[None]), to_dev(u[None])
    if op == "linearize":
        A, B = empty((1, n_s, n_s)), empty((1, n_s, n_c))
        _lib.check(lib.dpilqr_model_linearize(1, n_s, ptr(mid), ptr(xd), ptr(ud), float(dt), ptr(A), ptr(B), stream_handle()))
        return A[0].cpu().numpy(), B[0].cpu().numpy()
    out = empty((1, n_s))
    if op == "f":
        _lib.check(lib.dpilqr_model_f(1, n_s, ptr(mid), ptr(xd), ptr(ud), ptr(out), stream_handle()))
    else:
        _lib.check(lib.dpilqr_model_integrate(1, n_s, ptr(mid), ptr(xd), ptr(ud), float(dt), ptr(out), stream_handle()))
    return out[0].cpu().numpy()


def f(x, u, model):
    """Continuous dynamics x_dot = f(x, u) (pyx:61-90)."""
    return _run("f", model, x, u)


def integrate(x, u, dt, model):
    """One zero-order-hold step: classical RK4 with 5 sub-steps (pyx:93-123, bbdynamics.cpp:39-93)."""
    return _run("integrate", model, x, u, dt)


def linearize(x, u, dt, model):
    """Forward-Euler discretised Jacobians (A, B) (pyx:125-164, bbdynamics.cpp:95-106)."""
    return _run("linearize", model, x, u, dt)
