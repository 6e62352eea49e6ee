"""Cost plugins: same names and signatures as the reference's dpilqr/cost.py
(Cost :19-34, ReferenceCost :37-107, ProximityCost :110-171, GameCost :174-266, quadraticize_distance :269-315).

The recognised cost types hold parameters only; values and derivatives are computed on the GPU
(dpilqr_cost_eval, dpilqr_make_tiles).  A user subclass of Cost with host code is honoured by the solver
through the tile contract, exactly like a host dynamics plugin.
"""
import abc

import numpy as np

from . import batch as _batch
from .util import Point


class Cost(abc.ABC):
    @abc.abstractmethod
    def __call__(self, *args):
        """cost at (x, u)"""

    @abc.abstractmethod
    def quadraticize(self):
        """(L_x, L_u, L_xx, L_uu, L_ux) at (x, u)"""


def _family_model(n_s, n_c):
    """Any model tag with these per-agent dimensions (cost kernels only need the dimensions)."""
    for tag, dims in _batch.MODEL_DIMS.items():
        if dims == (n_s, n_c):
            return tag
    raise ValueError(f"no device model family with (n_x, n_u) = ({n_s}, {n_c}) per agent; "
                     "supported: (3,2), (4,2), (6,3), (12,4)")


def _point_batch(game, x, terminal):
    """ProblemBatch of ONE item / ONE time step holding this cost's parameters."""
    refs = game.ref_costs
    k, n_s, n_c = len(refs), refs[0].x_dim, refs[0].u_dim
    prox = game.prox_cost if isinstance(game.prox_cost, ProximityCost) else None
    n_dims = list(prox.n_dims) if prox is not None else [2] * k
    radius = prox.radius if prox is not None else 0.0
    w_prox = game.PROX_WEIGHT if prox is not None else 0.0
    return _batch.ProblemBatch([_family_model(n_s, n_c)] * k, n_dims, np.concatenate([r.xf for r in refs])[None],
                               np.stack([r.Q for r in refs]), np.stack([r.R for r in refs]),
                               np.stack([r.Qf for r in refs]), radius, 1.0, 1, w_ref=game.REF_WEIGHT, w_prox=w_prox)


def _device_quadraticize(game, x, u, terminal):
    x = np.asarray(x, dtype=np.float64).reshape(-1); u = np.asarray(u, dtype=np.float64).reshape(-1)
    pb = _point_batch(game, x, terminal)
    t = pb.unpack_tiles(pb.make_tiles(np.stack([x, x])[None], u[None, None]))
    i = 1 if terminal else 0            # record 0: stage quadraticisation, record T=1: terminal one
    n_u = pb.n_u
    if terminal:                        # reference: L_u = 0, L_uu = 0, L_ux = 0 at the terminal step
        return t["Lx"][0, 1], np.zeros(n_u), t["Lxx"][0, 1], np.zeros((n_u, n_u)), np.zeros((n_u, pb.n_x))
    return t["Lx"][0, i], t["Lu"][0, i], t["Lxx"][0, i], t["Luu"][0, i], t["Lux"][0, i]


def _device_cost(game, x, u, terminal):
    x = np.asarray(x, dtype=np.float64).reshape(-1); u = np.asarray(u, dtype=np.float64).reshape(-1)
    pb = _point_batch(game, x, terminal)
    return float(pb.cost(x[None, None], u[None, None], terminal).item())


class ReferenceCost(Cost):
    """(x - xf)^T Q (x - xf) + u^T R u, terminal (x - xf)^T Qf (x - xf); no 1/2 factors."""

    _id = 0

    def __init__(self, xf, Q, R, Qf=None, id=None):
        Q = np.asarray(Q, dtype=np.float64); R = np.asarray(R, dtype=np.float64)
        if Qf is None:
            Qf = np.eye(Q.shape[0])
        if not id:
            id = ReferenceCost._id
            ReferenceCost._id += 1
        self.xf = np.asarray(xf, dtype=np.float64).flatten()
        self.Q, self.R, self.Qf, self.id = Q, R, np.asarray(Qf, dtype=np.float64), id
        self.nx, self.nu = Q.shape[0], R.shape[0]

    @property
    def x_dim(self):
        return self.nx

    @property
    def u_dim(self):
        return self.nu

    @classmethod
    def _reset_ids(cls):
        cls._id = 0

    def _as_game(self):
        return GameCost([self])

    def __call__(self, x, u, terminal=False):
        return _device_cost(self._as_game(), x, u, terminal)

    def quadraticize(self, x, u, terminal=False):
        return _device_quadraticize(self._as_game(), x, u, terminal)

    def __repr__(self):
        return f"ReferenceCost(\n\tQ: {self.Q},\n\tR: {self.R},\n\tQf: {self.Qf},\n\tid: {self.id}\n)"


class ProximityCost(Cost):
    """sum over agent pairs of min(0, d_ij - radius)^2 (distances over min(n_dims_i, n_dims_j) coordinates)."""

    def __init__(self, x_dims, radius, n_dims):
        self.x_dims, self.radius, self.n_dims = list(x_dims), radius, list(n_dims)
        self.n_agents = len(self.x_dims)

    def _as_game(self):
        n_s = self.x_dims[0]
        n_c = {3: 2, 4: 2, 6: 3, 12: 4}[n_s]
        refs = [ReferenceCost(np.zeros(n_s), np.zeros((n_s, n_s)), np.zeros((n_c, n_c)), np.zeros((n_s, n_s)), id=-1 - i)
                for i in range(self.n_agents)]
        g = GameCost(refs, self)
        g.PROX_WEIGHT = 1.0
        return g

    def __call__(self, x):
        if self.n_agents == 1:
            return 0.0
        g = self._as_game()
        return _device_cost(g, x, np.zeros(sum(g.u_dims)), False)

    def quadraticize(self, x):
        g = self._as_game()
        L_x, _, L_xx, _, _ = _device_quadraticize(g, x, np.zeros(sum(g.u_dims)), False)
        return L_x, L_xx


class GameCost(Cost):
    """PROX_WEIGHT * proximity + REF_WEIGHT * sum of the agents' reference costs."""

    def __init__(self, reference_costs, proximity_cost=None):
        self.ref_costs = list(reference_costs)
        self.prox_cost = proximity_cost if proximity_cost else (lambda _x: 0.0)
        self.REF_WEIGHT, self.PROX_WEIGHT = 1.0, 200.0
        self.x_dims = [r.x_dim for r in self.ref_costs]
        self.u_dims = [r.u_dim for r in self.ref_costs]
        self.ids = [r.id for r in self.ref_costs]
        self.n_agents = len(self.ref_costs)

    @property
    def xf(self):
        return np.concatenate([r.xf for r in self.ref_costs])

    def __call__(self, x, u, terminal=False):
        return _device_cost(self, x, u, terminal)

    def quadraticize(self, x, u, terminal=False):
        return _device_quadraticize(self, x, u, terminal)

    def split(self, graph):
        """One GameCost per sub-problem of the interaction graph (agents in their original order)."""
        n_s, radius, n_dims = self.ref_costs[0].x_dim, self.prox_cost.radius, self.prox_cost.n_dims
        out = []
        for members in graph.values():
            keep = [(r, nd) for r, nd in zip(self.ref_costs, n_dims) if r.id in members]
            prox = ProximityCost([n_s] * len(members), radius, [nd for _, nd in keep])
            out.append(GameCost([r for r, _ in keep], prox))
        return out

    def __repr__(self):
        return f"GameCost(\n\tids: {self.ids},\n\tprox_cost: {self.prox_cost}\n)"


def is_device_cost(c):
    """True when the cost is a GameCost / ReferenceCost made only of the recognised types."""
    if type(c) is ReferenceCost:
        return True
    if type(c) is not GameCost or not all(type(r) is ReferenceCost for r in c.ref_costs):
        return False
    return type(c.prox_cost) is ProximityCost or (c.n_agents == 1 and not isinstance(c.prox_cost, Cost))


def quadraticize_distance(point_a, point_b, radius, n_d):
    """Gradient (n_d,) and Hessian (n_d, n_d) of min(0, |a - b| - radius)^2 w.r.t. a (cost.py:269-315)."""
    a = np.array([point_a.x, point_a.y, point_a.z], dtype=np.float64)[:n_d]
    b = np.array([point_b.x, point_b.y, point_b.z], dtype=np.float64)[:n_d]
    n_s = 6 if n_d == 3 else 4
    x = np.zeros(2 * n_s); x[:n_d] = a; x[n_s:n_s + n_d] = b
    L_x, L_xx = ProximityCost([n_s, n_s], radius, [n_d, n_d]).quadraticize(x)
    return L_x[:n_d], L_xx[:n_d, :n_d]
