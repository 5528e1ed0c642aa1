"""Multi-GPU side of the hot path: one process per GPU (``torch.distributed``, backend
``nccl`` = RCCL over xGMI on the GPU box, ``gloo`` in the CPU tests).

The reference is single-process; this layer is new.  Nodes of a frontier are
independent QPs, so they are dealt to the ranks round-robin with no data-path collective.  The
only exchange the path needs is the incumbent: once per round every rank contributes its best
upper bound and its number of open candidates to ONE small all-reduce -- MIN over the two
float64 (ub, -open): the first entry is the global upper bound, minus the second the LARGEST
number of open candidates on any rank, zero exactly when every rank is done -- so that every
rank prunes against the global best and all ranks stop together.  The same exchange is offered
to non-Python callers of the library as ``hmpc_allreduce_incumbent`` (include/hmpc.h, RCCL).

A rank whose search fails (a node that ends MAXITER / NUMERICAL raises, as in the reference a
Gurobi status other than optimal / infeasible does, ``bounded_qp.py:216-228``) still owes the
other ranks this round's all-reduce: it contributes an upper bound of -inf before re-raising,
which every other rank reads as "a peer failed" and raises ``PeerFailure`` -- no rank is left
waiting in a collective.
"""
import numpy as np


def shard_indices(count, rank, world):
    """Indices of the nodes of a ``count``-node frontier owned by ``rank`` (round-robin: node k
    goes to rank k % world, which balances random-depth nodes)."""
    return np.arange(rank, count, world)


class PeerFailure(RuntimeError):
    """Another rank's search raised; this rank's result would be incomplete."""


class IncumbentExchange(object):
    """Callable for ``branch_and_bound(incumbent_exchange=...)``."""

    def __init__(self, group=None, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.device = device if device is not None else 'cpu'
        self.rounds = 0
        # Do the peers wait for a round from this rank?  Yes before the first round and after every round that ended with
        # open candidates somewhere; no while a round is in progress (an error raised BY the exchange must not be answered
        # with another collective) and after the round in which every rank reported none (all ranks have left the loop).
        self.owed = True

    def __call__(self, ub, n_candidates):
        torch, dist = self.torch, self.dist
        self.owed = False
        pair = torch.tensor([ub, -float(n_candidates)], dtype=torch.float64, device=self.device)
        dist.all_reduce(pair, op=dist.ReduceOp.MIN, group=self.group)
        self.rounds += 1
        ub_all, open_max = pair.tolist()          # (the search only asks whether ANY rank still has candidates)
        if ub_all == -np.inf:
            raise PeerFailure('the branch and bound of another rank failed')
        self.owed = int(round(-open_max)) > 0
        return float(ub_all), int(round(-open_max))

    def abort(self):
        """This rank's contribution to the round the others are waiting in, after its own search failed -- only if they
        are: an error raised by the exchange itself, or after the last round, is not answered with a collective the peers
        would never match."""
        if not self.owed:
            return
        self.owed = False
        pair = self.torch.tensor([-np.inf, 0.], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(pair, op=self.dist.ReduceOp.MIN, group=self.group)


def solve_frontier_sharded(ctrl, fix, x0, group=None, device=None):
    """Solves this rank's shard of a frontier and returns (indices, local result dict, global
    incumbent upper bound).  ``fix`` is the WHOLE frontier (B, T*nub) on every rank."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    idx = shard_indices(fix.shape[0], rank, world)
    res = ctrl.qp.solve_batch(np.asarray(x0, dtype=np.float64), fix[idx]) if idx.size else None
    ub = np.inf
    if res is not None:
        full = (fix[idx] >= 0).all(axis=1) & (res['status'] == 0)
        if full.any():
            ub = float(res['obj'][full].min())
    t = torch.tensor([ub], dtype=torch.float64, device=device if device is not None else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return idx, res, float(t.item())


def feedforward_sharded(ctrl, x0, cover, group=None, device=None, **kwargs):
    """Branch and bound with the root cover dealt to the ranks.

    ``cover`` is a list of nodes that covers the binary cube (a warm start, or the children of
    a replicated partial expansion) and is identical on every rank; rank r owns
    ``cover[r::world]``.  Returns (objective, binary assignment (T, nub) or None, local leaves,
    local solves) -- objective and assignment are the same on every rank.
    """
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    mine = [n for k, n in enumerate(cover) if k % world == rank]
    exchange = IncumbentExchange(group, device)
    kwargs.setdefault('printing_period', None)
    if not mine:
        from .branch_and_bound import Node
        mine = [Node({}, lb=np.inf)]  # nothing to do, but keep taking part in the rounds
    try:
        sol, leaves, solves, _ = ctrl.feedforward(x0, warm_start=mine, incumbent_exchange=exchange, **kwargs)
    except PeerFailure:
        raise
    except Exception:
        exchange.abort()                      # the round the other ranks are waiting in
        raise
    local = np.inf if sol is None else sol.objective
    # the owner of the global incumbent (lowest rank on ties) publishes its assignment
    gathered = [None] * world
    dist.all_gather_object(gathered, (local, None if sol is None else np.array(sol.variables['ub'])), group=group)
    best = min(range(world), key=lambda r: (gathered[r][0], r))
    objective, assignment = gathered[best]
    if not np.isfinite(objective):
        return np.inf, None, leaves, solves
    return objective, assignment, leaves, solves
