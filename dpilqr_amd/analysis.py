"""The Monte-Carlo study of the reference (scripts/analysis.py:35-107 multi_agent_run, :110-123 setup_logger,
:126-174 monte_carlo_analysis) over the batched device path: centralized vs distributed receding-horizon control over
models x team sizes x trials, logged as the reference's CSV.

The reference runs one trial at a time: random_setup, solve_rhc(centralized=True), solve_rhc(centralized=False), each a
loop of single solves.  Here all trials of one (model, team size) run in lock step -- every receding-horizon round of
every trial is ONE batched solve (distributed.solve_rhc_scenarios) -- and the rows come out in the reference's order
(trial by trial, the centralized branch's rows, then the distributed branch's).

Randomness.  The reference never seeds: a trial consumes NumPy's global stream for random_setup and then for the warm
start of each branch's solve_rhc (distributed.py:152).  Here trial i of a (model, team size) cell is seeded
(np.random.seed(seed_of(...))) and consumes the stream in exactly that order, so a trial is reproducible and a
per-trial run of solve_rhc on the same seed sees the same inputs (tests/test_gpu_analysis.py compares the rows).

limit_solve_time (analysis 2, the reference's default, analysis.py:179): t_kill = dt reaches every solve -- inside the
batched device solve (include/dpilqr_hip.h: every item's own clock from its admission) -- and t_diverge = N dt.
What that clock measures differs in kind from the reference's: there a solve has a CPU core to itself and perf_counter() runs
over its own work; here an item shares every launch with the rest of the window, so its elapsed time includes the work the GPU
does for up to `window` other items.  Which solves are killed therefore depends on the batch size, the window and whatever else
the GPU runs, and varies from run to run: analysis-2 rows answer "what does a real-time budget of dt buy on THIS device at THIS
load", not the reference's question, and cannot be compared row by row with its CSV.  For a reproducible budget pass
n_lqr_iter (a cap on iterations per solve, deterministic) instead of, or beside, limit_solve_time.
"""
import logging
from os import getpid
from pathlib import Path
from time import strftime

import numpy as np

from .cost import GameCost, ProximityCost, ReferenceCost
from .distributed import solve_rhc_scenarios
from .dynamics import DoubleIntDynamics4D, MultiDynamicalModel, QuadcopterDynamics6D, UnicycleDynamics4D
from .problem import ilqrProblem
from .util import random_setup, split_agents_gen

HEADER = "dynamics,n_agents,trial,centralized,last,t,J,horizon,dt,converged,ids,times,subgraphs,dist_left"   # analysis.py:120-123
STEP_SIZE = 3                                                                                                # analysis.py:43
MODELS = (DoubleIntDynamics4D, UnicycleDynamics4D, QuadcopterDynamics6D)                                     # analysis.py:133-137


def seed_of(model, n_agents, i_trial, seed0=0):
    """The seed of one trial: distinct per (model, team size, trial), stable across runs."""
    return int(seed0) + 100003 * MODELS.index(model) + 1009 * int(n_agents) + int(i_trial)


def trial_inputs(n_agents, n_states, n_u, N, energy, n_d, seed):
    """What one trial of multi_agent_run draws, in the order the reference draws it: (x0, xf) from random_setup
    (analysis.py:45-54), then the warm start of the centralized and of the distributed solve_rhc (distributed.py:152)."""
    np.random.seed(seed)
    x0, xf = random_setup(n_agents, n_states, is_rotation=False, rel_dist=n_agents, var=n_agents / 2, n_d=n_d, random=True,
                          energy=energy)
    U_c = np.random.rand(N, n_u) * 0.01
    U_d = np.random.rand(N, n_u) * 0.01
    return x0, xf, U_c, U_d


def weights_of(model, n_states):
    """analysis.py:62-69"""
    if model in (DoubleIntDynamics4D, UnicycleDynamics4D):
        Q = 1.0 * np.diag([1, 1] + [0] * (n_states - 2)); R = np.eye(2)
    elif model is QuadcopterDynamics6D:
        Q = np.eye(n_states) * 50; R = np.eye(3)
    else:
        raise ValueError(f"no weights defined for {model.__name__} (analysis.py:62-67 knows three models)")
    return Q, R, 1000.0 * np.eye(n_states)


def build_problem(model, n_agents, dt, radius, xf, n_d):
    """analysis.py:56-78: ids 100.., one ReferenceCost per agent, the proximity cost, the game cost, the problem."""
    n_states = model(-1).n_x
    x_dims, n_dims = [n_states] * n_agents, [n_d] * n_agents
    ids = [100 + i for i in range(n_agents)]
    dynamics = MultiDynamicalModel([model(dt, id_) for id_ in ids])
    Q, R, Qf = weights_of(model, n_states)
    goal_costs = [ReferenceCost(xf_i, Q.copy(), R.copy(), Qf.copy(), id_)
                  for xf_i, id_ in zip(split_agents_gen(np.asarray(xf).reshape(-1, 1), x_dims), ids)]
    return ilqrProblem(dynamics, GameCost(goal_costs, ProximityCost(x_dims, radius, n_dims)))


def multi_agent_run(model, x_dims, dt, N, radius, energy=10.0, n_d=2, trials=(0,), seed0=0, emit=logging.info, window=None,
                    **kwargs):
    """multi_agent_run (analysis.py:35-107) for all `trials` of one (model, team size) at once: both branches, every
    receding-horizon round of every trial one batched solve.  kwargs as the reference passes them: t_kill, dist_converge,
    t_diverge (i_trial is replaced by `trials`; energy is consumed here).  Emits the rows in the reference's order and
    returns {trial: ((Xc, Uc, Jc, converged), (Xd, Ud, Jd, converged))}."""
    if not len(set(x_dims)) == 1:
        raise ValueError("Dynamics dimensions must be consistent")
    kwargs.pop("i_trial", None); kwargs.pop("verbose", None)
    n_agents, n_states = len(x_dims), x_dims[0]
    trials = list(trials)
    n_u = n_agents * model(-1).n_u
    drawn = [trial_inputs(n_agents, n_states, n_u, N, energy, n_d, seed_of(model, n_agents, i, seed0)) for i in trials]
    x0 = np.stack([d[0].ravel() for d in drawn]); xf = np.stack([d[1].ravel() for d in drawn])
    problem = build_problem(model, n_agents, dt, radius, xf[0], n_d)     # per-trial goals go in as `xf` below
    out, rows = {}, {}
    for centralized, col in ((True, 2), (False, 3)):
        r = []
        res = solve_rhc_scenarios(problem, x0, N, radius, xf=xf, U0=np.stack([d[col] for d in drawn]), centralized=centralized,
                                  n_d=n_d, step_size=STEP_SIZE, i_trial=trials, rows=r, window=window, **kwargs)
        rows[centralized] = r
        for j, i in enumerate(trials):
            out.setdefault(i, []).append(res[j])
    for j, i in enumerate(trials):
        for centralized in (True, False):
            for row in rows[centralized][j]:
                emit(row)
    return {i: tuple(v) for i, v in out.items()}


LOGGER_NAME = "dpilqr.analysis"


def setup_logger(limit_solve_time, log_file=None):
    """setup_logger (analysis.py:110-123): logs/dec-mc-{1|2}_<date>_<pid>.csv beside the repository's scripts unless a
    file is named; the header row first.  Returns the path.  The reference configures the ROOT logger of its own script; a
    library must not (the embedding application's handlers are its own): the rows go through the logger `dpilqr.analysis`, which
    gets the file handler and does not propagate."""
    analysis = "1" if not limit_solve_time else "2"
    if log_file is None:
        log_path = Path(__file__).resolve().parent.parent / "logs"
        log_path.mkdir(exist_ok=True)
        log_file = log_path / strftime(f"dec-mc-{analysis}_%m-%d-%y_%H.%M.%S_{getpid()}.csv")
    print(f"Logging results to {log_file}")
    logger = logging.getLogger(LOGGER_NAME)
    for h in list(logger.handlers):            # a previous study's file
        logger.removeHandler(h); h.close()
    handler = logging.FileHandler(log_file)
    handler.setFormatter(logging.Formatter("%(message)s"))
    logger.addHandler(handler)
    logger.setLevel(logging.INFO)
    logger.propagate = False
    logger.info(HEADER)
    return Path(log_file)


def monte_carlo_analysis(limit_solve_time=False, n_trials=2, n_agents_iter=(3, 4, 5, 6, 7), models=MODELS, dt=0.1, N=50,
                         energy=10.0, radius=0.50, seed0=0, log_file=None, emit=None, window=None):
    """monte_carlo_analysis (analysis.py:126-174): the same loops, constants and stopping rules; the innermost loop (the
    trials) is the batch.  n_trials may be thousands -- that is what the batched path is for."""
    if emit is None:
        setup_logger(limit_solve_time, log_file)
        emit = logging.getLogger(LOGGER_NAME).info
    if limit_solve_time:
        t_kill, t_diverge = dt, N * dt
    else:
        t_kill, t_diverge = None, 4 * N * dt
    results = {}
    for model in models:
        print(f"{model.__name__}")
        for n_agents in n_agents_iter:
            print(f"\tn_agents: {n_agents}")
            n_d = 3 if model is QuadcopterDynamics6D else 2
            x_dims = [model(-1).n_x] * n_agents
            results[(model.__name__, n_agents)] = multi_agent_run(
                model, x_dims, dt, N, radius, n_d=n_d, t_kill=t_kill, dist_converge=0.1, t_diverge=t_diverge, energy=energy,
                trials=range(n_trials), seed0=seed0, emit=emit, window=window, verbose=False)
    return results
