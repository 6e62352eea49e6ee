"""Dynamics plugins: same class names, constructors and methods as the reference's dpilqr/dynamics.py
(DynamicalModel :54-92, CppModel :117-130, MultiDynamicalModel :133-202, the eight models :205-250).

The recognised models carry only their `Model` tag; every evaluation happens on the GPU.  A user may
still subclass DynamicalModel with host code (the reference's BikeDynamics5D is such a plugin): the
solver then calls that code itself and feeds the GPU sweep with the tiles it returns.
"""
import abc

import numpy as np

from . import bbdynamics
from .bbdynamics import Model
from .util import split_agents_gen, uniform_block_diag


class DynamicalModel(abc.ABC):
    """Plugin surface of a discrete-time model: n_x, n_u, dt, id, __call__(x,u), f(x,u), linearize(x,u)."""

    _id = 0

    def __init__(self, n_x, n_u, dt, id=None):
        if not id:  # like the reference, a falsy id (None or 0) takes the class-wide running counter
            id = DynamicalModel._id
            DynamicalModel._id += 1
        self.n_x, self.n_u, self.dt, self.id = n_x, n_u, dt, id

    def __call__(self, x, u):
        """x[t+1]: classical RK4 of self.f over dt (a host plugin may override this)."""
        k0 = self.f(x, u)
        k1 = self.f(x + 0.5 * self.dt * k0, u)
        k2 = self.f(x + 0.5 * self.dt * k1, u)
        k3 = self.f(x + self.dt * k2, u)
        return x + self.dt * (k0 + 2.0 * k1 + 2.0 * k2 + k3) / 6.0

    @staticmethod
    def f():
        pass

    @abc.abstractmethod
    def linearize(self):
        """(A, B) of the DISCRETE dynamics at (x, u): shapes (n_x, n_x), (n_x, n_u)."""

    @classmethod
    def _reset_ids(cls):
        cls._id = 0

    def __repr__(self):
        return f"{type(self).__name__}(n_x: {self.n_x}, n_u: {self.n_u}, id: {self.id})"


class CppModel(DynamicalModel):
    """A model the HIP library implements (the reference's C++-backed models); `self.model` is its tag."""

    model = None

    def __init__(self, dt, *args, **kwargs):
        n_x, n_u = bbdynamics.MODEL_DIMS[type(self).model]
        super().__init__(n_x, n_u, dt, *args, **kwargs)
        self.model = type(self).model

    def __call__(self, x, u):
        return bbdynamics.integrate(x, u, self.dt, self.model)

    def f(self, x, u):
        return bbdynamics.f(x, u, self.model)

    def linearize(self, x, u):
        return bbdynamics.linearize(x, u, self.dt, self.model)


def _device_model(name, tag):
    return type(name, (CppModel,), {"model": tag, "__doc__": f"{tag.name}: see models.hpp / bbdynamics.cpp"})


DoubleIntDynamics4D = _device_model("DoubleIntDynamics4D", Model.DoubleInt4D)
DoubleIntDynamics6D = _device_model("DoubleIntDynamics6D", Model.DoubleInt6D)
CarDynamics3D = _device_model("CarDynamics3D", Model.Car3D)
UnicycleDynamics4D = _device_model("UnicycleDynamics4D", Model.Unicycle4D)
QuadcopterDynamics6D = _device_model("QuadcopterDynamics6D", Model.Quadcopter6D)
QuadcopterDynamics12D = _device_model("QuadcopterDynamics12D", Model.Quadcopter12D)
HumanDynamics6D = _device_model("HumanDynamics6D", Model.Human6D)
HumanDynamicsLin6D = _device_model("HumanDynamicsLin6D", Model.HumanLin6D)
# Not in the reference: BASELINE config 5's "zero-padded" human, so that humans can be stacked with
# QuadcopterDynamics12D (a MultiDynamicalModel needs uniform dims, dynamics.py:165-170).  First 6 states / 3 controls:
# HumanDynamics6D; the other 6 states never move, the fourth control does nothing.
HumanDynamics6DPadded12 = _device_model("HumanDynamics6DPadded12", Model.HumanPad12D)

DEVICE_MODEL_CLASSES = (DoubleIntDynamics4D, DoubleIntDynamics6D, CarDynamics3D, UnicycleDynamics4D,
                        QuadcopterDynamics6D, QuadcopterDynamics12D, HumanDynamics6D, HumanDynamicsLin6D,
                        HumanDynamics6DPadded12)


def is_device_model(m):
    """True when `m` is one of the recognised models with none of its methods overridden."""
    return type(m) in DEVICE_MODEL_CLASSES


class MultiDynamicalModel(DynamicalModel):
    """Several agents stacked into one model: joint x = [x_1; ...; x_k] (dynamics.py:133-202).
    Like the reference, every agent is sliced with the FIRST agent's dimensions."""

    def __init__(self, submodels):
        self.submodels = list(submodels)
        self.n_players = len(self.submodels)
        self.x_dims = [m.n_x for m in self.submodels]
        self.u_dims = [m.n_u for m in self.submodels]
        self.ids = [m.id for m in self.submodels]
        super().__init__(sum(self.x_dims), sum(self.u_dims), self.submodels[0].dt, -1)

    def _per_agent(self, method, x, u):
        x = np.asarray(x, dtype=np.float64).reshape(-1); u = np.asarray(u, dtype=np.float64).reshape(-1)
        return [getattr(m, method)(xi, ui) for m, xi, ui in
                zip(self.submodels, split_agents_gen(x, self.x_dims), split_agents_gen(u, self.u_dims))]

    def f(self, x, u):
        return np.concatenate(self._per_agent("f", x, u))

    def __call__(self, x, u):
        return np.concatenate(self._per_agent("__call__", x, u))

    def linearize(self, x, u):
        blocks = self._per_agent("linearize", x, u)
        return uniform_block_diag(*[b[0] for b in blocks]), uniform_block_diag(*[b[1] for b in blocks])

    def split(self, graph):
        """One MultiDynamicalModel per sub-problem of the interaction graph, agents kept in their original order."""
        return [MultiDynamicalModel([m for m in self.submodels if m.id in members]) for members in graph.values()]

    def __repr__(self):
        inner = ",\n\t".join(repr(m) for m in self.submodels)
        return f"MultiDynamicalModel(\n\t{inner}\n)"
