"""Recognise the reference's plugin objects and lower them to device descriptors.

An ilqrProblem whose dynamics are the eight known models (alone or stacked in a MultiDynamicalModel) and
whose cost is ReferenceCost / GameCost(ReferenceCost..., ProximityCost) is described completely by a few
small arrays; `lower_problems` turns any number of such problems of ONE shape into a ProblemBatch.
Anything else (user subclasses with host code) is not lowerable and takes the host-plugin path of
control.ilqrSolver, which still runs the Riccati sweep on the GPU.
"""
import numpy as np

from .batch import ProblemBatch
from .cost import GameCost, ProximityCost, ReferenceCost, is_device_cost
from .dynamics import MultiDynamicalModel, is_device_model


def _submodels(dynamics):
    return list(dynamics.submodels) if isinstance(dynamics, MultiDynamicalModel) else [dynamics]


def is_lowerable(problem):
    dyn, cost = problem.dynamics, problem.game_cost
    if type(dyn) is not MultiDynamicalModel and not is_device_model(dyn):
        return False
    subs = _submodels(dyn)
    if not all(is_device_model(m) for m in subs) or not is_device_cost(cost):
        return False
    dims = {(m.n_x, m.n_u) for m in subs}
    refs = cost.ref_costs if isinstance(cost, GameCost) else [cost]
    # one dt for the whole stack: the device descriptor holds a single step size, while the reference's
    # MultiDynamicalModel calls every submodel with its own (dynamics.py:159-186) -- mixed step sizes take the host loop
    if len({float(m.dt) for m in subs}) != 1:
        return False
    return len(dims) == 1 and len(refs) == len(subs) and all((r.x_dim, r.u_dim) in dims for r in refs)


def describe(problem):
    """Per-problem parameter arrays (host) in the layout ProblemBatch expects."""
    subs = _submodels(problem.dynamics)
    cost = problem.game_cost
    refs = cost.ref_costs if isinstance(cost, GameCost) else [cost]
    prox = cost.prox_cost if isinstance(cost, GameCost) and isinstance(cost.prox_cost, ProximityCost) else None
    k = len(subs)
    for r, sub in zip(refs, subs):
        if np.size(r.xf) != sub.n_x:
            raise ValueError(f"ReferenceCost.xf has {np.size(r.xf)} entries for a {sub.n_x}-state agent (id {getattr(r, 'id', None)})")
    return dict(model=np.array([m.model.value for m in subs], dtype=np.int32),
                n_dims=np.array(prox.n_dims if prox is not None else [2] * k, dtype=np.int32),
                xf=np.concatenate([r.xf for r in refs]),
                Q=np.stack([r.Q for r in refs]), R=np.stack([r.R for r in refs]), Qf=np.stack([r.Qf for r in refs]),
                radius=float(prox.radius) if prox is not None else 0.0,
                w_ref=float(getattr(cost, "REF_WEIGHT", 1.0)),
                w_prox=float(cost.PROX_WEIGHT) if prox is not None else 0.0,
                dt=float(subs[0].dt), k=k)


def lower_problems(problems, T):
    """B lowerable problems of identical shape (k, n_s, n_c, dt, weights) -> one ProblemBatch."""
    ds = [describe(p) for p in problems]
    d0 = ds[0]
    for d in ds[1:]:
        if (d["k"], d["dt"], d["w_ref"], d["w_prox"]) != (d0["k"], d0["dt"], d0["w_ref"], d0["w_prox"]):
            raise ValueError("problems of one batch must share k, dt and the cost weights")
    stack = lambda key: np.stack([d[key] for d in ds])
    return ProblemBatch(stack("model"), stack("n_dims"), stack("xf"), stack("Q"), stack("R"), stack("Qf"),
                        np.array([d["radius"] for d in ds]), d0["dt"], T, w_ref=d0["w_ref"], w_prox=d0["w_prox"],
                        B=len(ds))
