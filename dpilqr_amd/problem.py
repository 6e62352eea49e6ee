"""ilqrProblem and the sub-problem worker: same surface as the reference's dpilqr/problem.py (:15-116)."""
from time import perf_counter as pc

import numpy as np

from .control import ilqrSolver
from .cost import GameCost, ReferenceCost
from .dynamics import DynamicalModel, MultiDynamicalModel
from .util import split_agents_gen


class ilqrProblem:
    """Dynamics + cost of one (sub)problem."""

    def __init__(self, dynamics, cost):
        self.dynamics = dynamics
        self.game_cost = cost
        self.n_agents = len(cost.ref_costs) if isinstance(cost, GameCost) else 1

    @property
    def ids(self):
        if not isinstance(self.dynamics, MultiDynamicalModel):
            raise NotImplementedError("Only MultiDynamicalModel's have an 'ids' attribute")
        if self.dynamics.ids != self.game_cost.ids:
            raise ValueError(f"Dynamics and cost have inconsistent ID's: {self}")
        return list(self.dynamics.ids)

    def split(self, graph):
        """One sub-problem per entry of the interaction graph (problem.py:36-47)."""
        return [ilqrProblem(dyn, cost) for dyn, cost in zip(self.dynamics.split(graph), self.game_cost.split(graph))]

    def extract(self, X, U, id_):
        """The columns of agent `id_` inside this problem's joint trajectory (problem.py:49-64)."""
        ids = self.ids
        if id_ not in ids:
            raise IndexError(f"Index {id_} not in ids: {ids}.")
        pos = ids.index(id_)
        ns, nc = self.game_cost.x_dims[0], self.game_cost.u_dims[0]
        return X[:, pos * ns:(pos + 1) * ns], U[:, pos * nc:(pos + 1) * nc]

    def selfish_warmstart(self, x0, N, verbose=False):
        """Every agent solves alone (k = 1 sub-problems); on the device this is ONE batch (problem.py:66-91)."""
        from .dispatch import solve_problem_list
        x0 = np.asarray(x0, dtype=np.float64).reshape(-1)
        ids = self.ids
        subs = self.split({id_: [id_] for id_ in ids})
        t0 = pc()
        x0s = list(split_agents_gen(x0, self.game_cost.x_dims))
        res = solve_problem_list(subs, x0s, [np.zeros((N, p.dynamics.n_u)) for p in subs], tol=1e-3)
        U_warm = np.concatenate([r[1] for r in res], axis=1)
        if verbose:
            print(f"selfish warm start of {ids}: {pc() - t0:.3g} s")
        return U_warm

    def __repr__(self):
        return f"ilqrProblem(\n\t{self.dynamics},\n\t{self.game_cost}\n)"


def solve_subproblem(args, **kwargs):
    """(subproblem, x0, U, id_, verbose) -> (X_agent, U_agent, id_), as problem.py:97-105."""
    subproblem, x0, U, id_, verbose = args
    Xi, Ui, _ = ilqrSolver(subproblem, U.shape[0]).solve(x0, U, verbose=verbose, **kwargs)
    return (*subproblem.extract(Xi, Ui, id_), id_)


def solve_subproblem_starmap(subproblem, x0, U, id_):
    return solve_subproblem((subproblem, x0, U, id_, False))


def _reset_ids():
    DynamicalModel._reset_ids()
    ReferenceCost._reset_ids()
