"""dpilqr_amd -- MI355X-native batched iLQR for DP-iLQR (hot path of labicon/dp-ilqr).

Host-side mirror of the reference's plugin / solver interface (dpilqr/__init__.py:1-58, hot-path names only)
over hand-written HIP kernels (libdpilqr_hip.so, C ABI in include/dpilqr_hip.h).  There is no CPU fallback:
every cost / dynamics / solver evaluation of the recognised plugin types runs on the GPU.
"""
from . import _lib  # noqa: F401
from .batch import ProblemBatch, backward_pass_tiles, pack_tiles, release_workspaces  # noqa: F401
from .bbdynamics import Model, f, integrate, linearize  # noqa: F401
from .control import ilqrSolver  # noqa: F401
from .cost import Cost, GameCost, ProximityCost, ReferenceCost, quadraticize_distance  # noqa: F401
from .dispatch import pairwise_graph, solve_problem_list, solve_scenarios_distributed  # noqa: F401
from .distributed import (define_inter_graph_threshold, solve_centralized, solve_distributed, solve_rhc,  # noqa: F401
                          solve_rhc_scenarios)
from .dynamics import (CarDynamics3D, CppModel, DoubleIntDynamics4D, DoubleIntDynamics6D, DynamicalModel,  # noqa: F401
                       HumanDynamics6D, HumanDynamics6DPadded12, HumanDynamicsLin6D, MultiDynamicalModel, QuadcopterDynamics6D,
                       QuadcopterDynamics12D, UnicycleDynamics4D)
from .problem import _reset_ids, ilqrProblem, solve_subproblem  # noqa: F401
from .util import (Point, compute_energy, compute_pairwise_distance, compute_pairwise_distance_nd, distance_to_goal,  # noqa: F401
                   normalize_energy, perturb_state, pos_mask, random_setup, randomize_locs, split_agents,
                   split_agents_gen, split_graph, uniform_block_diag, π)
