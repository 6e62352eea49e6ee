"""Host helpers around the hot path: agent slicing, graph splitting, scenario generation.

Mirrors the names of the reference's dpilqr/util.py that the hot path and its callers use
(util.py:20-117,125-236).  These are bookkeeping on small host arrays (shapes, ids, RNG draws);
no solver arithmetic happens here.
"""
import itertools
from dataclasses import dataclass

import numpy as np

π = np.pi


@dataclass
class Point:
    """3-D point with z defaulting to 0 (util.py:20-45)."""
    x: float
    y: float
    z: float = 0

    @property
    def ndim(self):
        return 3 if self.z != 0 else 2

    def _zip(self, other, op):
        return Point(op(self.x, other.x), op(self.y, other.y), op(self.z, other.z))

    def __add__(self, other):
        return self._zip(other, lambda a, b: a + b)

    def __sub__(self, other):
        return self._zip(other, lambda a, b: a - b)

    def __mul__(self, other):
        return self._zip(other, lambda a, b: a * b)

    def hypot2(self):
        return self.x ** 2 + self.y ** 2 + self.z ** 2

    def __repr__(self):
        return str((self.x, self.y, self.z))


def _pairs(n):
    return list(itertools.combinations(range(n), 2))


def compute_pairwise_distance(X, x_dims, n_d=2):
    """Distances between every pair of agents over the first n_d coordinates (util.py:48-61).
    X: (N, n_x) or (n_x,); returns (N, n_pairs) in itertools.combinations order."""
    if len(set(x_dims)) != 1:
        raise AssertionError("agents must share one state dimension")
    k, ns = len(x_dims), x_dims[0]
    if k == 1:
        raise ValueError("Can't compute pairwise distance for one agent.")
    P = np.asarray(X, dtype=float).reshape(-1, k, ns)[:, :, :n_d]
    ij = np.array(_pairs(k))
    diff = P[:, ij[:, 0], :] - P[:, ij[:, 1], :]
    return np.sqrt(np.sum(diff * diff, axis=2))


def compute_pairwise_distance_nd(X, x_dims, n_dims, dec_ind=None):
    """Like compute_pairwise_distance with min(n_dims_i, n_dims_j) coordinates per pair (util.py:64-87)."""
    X = np.atleast_2d(np.asarray(X, dtype=float))
    ns, k = x_dims[0], len(x_dims)
    cols = []
    for i, j in _pairs(k):
        if dec_ind is not None and dec_ind not in (i, j):
            continue
        nd = min(n_dims[i], n_dims[j])
        diff = X[:, i * ns:i * ns + nd] - X[:, j * ns:j * ns + nd]
        cols.append(np.sqrt(np.sum(diff * diff, axis=1)))
    return np.stack(cols, axis=1) if cols else np.zeros((X.shape[0], 0))


def split_agents(Z, z_dims):
    """Column-split a joint state/control into per-agent pieces (util.py:90-92)."""
    return np.split(np.atleast_2d(Z), np.cumsum(z_dims[:-1]), axis=1)


def split_agents_gen(z, z_dims):
    """Generator over per-agent slices; uses z_dims[0] for every agent like the reference (util.py:95-99)."""
    w = z_dims[0]
    for i in range(len(z_dims)):
        yield z[i * w:(i + 1) * w]


def split_graph(Z, z_dims, graph):
    """Gather, for every sub-problem of `graph`, the columns of its member agents (util.py:102-117)."""
    if len(set(z_dims)) != 1:
        raise AssertionError("agents must share one dimension")
    w = z_dims[0]
    where = {id_: pos for pos, id_ in enumerate(graph)}
    return [np.concatenate([Z[:, where[i] * w:(where[i] + 1) * w] for i in members], axis=1)
            for members in graph.values()]


def pos_mask(x_dims, n_d=2):
    return np.array([(i % x_dims[0]) < n_d for i in range(sum(x_dims))])


def uniform_block_diag(*arrs):
    """Block-diagonal matrix of equally shaped blocks (util.py:229-236)."""
    r, c = arrs[0].shape
    out = np.zeros((len(arrs) * r, len(arrs) * c))
    for i, blk in enumerate(arrs):
        out[i * r:(i + 1) * r, i * c:(i + 1) * c] = blk
    return out


def distance_to_goal(x, x_goal, n_agents, n_states, n_d):
    return np.linalg.norm((x - x_goal).reshape(n_agents, n_states)[:, :n_d], axis=1)


# ------------------------------------------------------------------ scenario generation
# The Monte-Carlo scenario distribution of scripts/analysis.py:45-54 draws from NumPy's legacy global
# RNG; the calls below are made in the same order so that np.random.seed(s) reproduces its scenarios.

def randomize_locs(n_pts, random=False, rel_dist=3.0, var=3.0, n_d=2):
    """util.py:125-149: uniform points in [-var, var]^n_d; unless `random`, pushed apart until every
    pair is farther than rel_dist."""
    x = var * np.random.uniform(-1, 1, (n_pts, n_d))
    if random:
        return x
    push = 0.1 * n_pts
    ij = np.array(_pairs(n_pts))
    while True:
        centre = x.mean(axis=0)
        dist = compute_pairwise_distance(x.flatten(), [n_d] * n_pts, n_d=2).T
        close = ij[dist.flatten() <= rel_dist]
        if not close.size:
            return x
        x[close] += push * (x[close] - centre)


def compute_energy(x, x_dims, n_d=2):
    return np.linalg.norm(x[pos_mask(x_dims, n_d)].reshape(-1, n_d), axis=1).sum()


def normalize_energy(x, x_dims, energy=10.0, n_d=2):
    """Centre the positions and scale them so the summed distance from the origin is `energy` (util.py:203-217)."""
    x = x.copy()
    mask = pos_mask(x_dims, n_d)
    centre = x[mask].reshape(-1, n_d).mean(0)
    x[mask] -= np.tile(centre, len(x_dims)).reshape(-1, 1)
    x[mask] *= energy / compute_energy(x, x_dims, n_d)
    return x


def random_setup(n_agents, n_states, is_rotation=False, n_d=2, energy=None, do_face=False, **kwargs):
    """Random start/goal pair (util.py:165-195).  Rotation/facing variants are not part of the hot path's
    scenario distribution and are not provided."""
    if is_rotation or do_face:
        raise NotImplementedError("only the is_rotation=False, do_face=False scenarios of scripts/analysis.py")
    starts = randomize_locs(n_agents, n_d=n_d, **kwargs)
    goals = randomize_locs(n_agents, n_d=n_d, **kwargs)
    pad = np.zeros((n_agents, n_states - n_d))
    x0 = np.c_[starts, pad].reshape(-1, 1)
    xf = np.c_[goals, pad].reshape(-1, 1)
    if energy:
        dims = [n_states] * n_agents
        x0 = normalize_energy(x0, dims, energy, n_d)
        xf = normalize_energy(xf, dims, energy, n_d)
    return x0, xf


def perturb_state(x, x_dims, n_d=2, var=0.5):
    x = x.copy()
    mask = pos_mask(x_dims, n_d)
    x[mask] += var * np.random.randn(*x[mask].shape)
    return x


def random_setup_batch(seeds, n_agents, n_states, var, n_d=2, energy=None):
    """(x0, xf) device tensors (S, n_agents * n_states) of the scenarios np.random.seed(s); random_setup(n_agents, n_states,
    is_rotation=False, var=var, n_d=n_d, random=True, energy=energy) for s in range(seeds[0], seeds[0] + S) -- generated on
    the device, bit for bit what the host loop gives (dpilqr_random_setup).  seeds: (first seed, count) or a range."""
    from . import _lib
    from .device import empty, ptr, stream_handle
    seed0, S = (seeds.start, len(seeds)) if isinstance(seeds, range) else (int(seeds[0]), int(seeds[1]))
    if isinstance(seeds, range) and seeds.step != 1:
        raise ValueError("consecutive seeds only")
    x0 = empty((S, n_agents * n_states)); xf = empty((S, n_agents * n_states))
    _lib.check(_lib.load().dpilqr_random_setup(S, seed0, n_agents, n_states, n_d, float(var), float(energy or 0.0), ptr(x0), ptr(xf),
                                               stream_handle()))
    return x0, xf
