"""Branch-and-bound driver for the hybrid-MPC MIQP.

Semantics follow the reference driver (``warm_start_hmpc/branch_and_bound.py:
408-563``, quirks listed in SURVEY.md Appendix B):

* a node is a candidate while ``lb < ub - tol``; the search stops when no
  leaf is a candidate;
* a solved node is pruned if ``lb >= cutoff``, becomes the incumbent if it is
  binary feasible, and is branched otherwise; only branched nodes leave the
  list of leaves, children are appended in the order the brancher gives them;
* ``best_first`` is ``argmin lb`` with the first index winning ties.

What is different is how nodes reach the QP solver.  The reference solves one
node per loop turn on one mutable Gurobi model.  Here a *frontier* of up to
``frontier_width`` candidates is handed to the batched GPU solver in one call
(``batch_solver``); the results are then consumed one by one, in selection
order, through exactly the prune / incumbent / branch rules above.  With
``frontier_width=1`` the node order of the reference is reproduced turn for
turn; wider frontiers solve some nodes a serial search would have pruned
(they are counted in ``solves``) but return the same incumbent, since every
node is still judged against the current upper bound when it is consumed.
"""
from time import time

import numpy as np


class Node(object):
    '''
    Node of the branch and bound tree.
    '''

    def __init__(self, identifier, lb=-np.inf, extra=None):
        self.identifier = identifier
        self.lb = lb
        self.extra = extra
        self.binary_feasible = None
        self.solve_time = None

    def solve(self, solver, cutoff=None):
        # same unpacking contract as branch_and_bound.py:38-55
        self.lb, self.binary_feasible, self.solve_time, self.extra = solver(self.identifier, cutoff, self.extra)

    def _take(self, result):
        self.lb, self.binary_feasible, self.solve_time, self.extra = result


class Printer(object):
    """Progress table on stdout (the reference prints a similar one,
    branch_and_bound.py:57-218): a line at the root, at every new incumbent
    and every ``printing_period`` seconds."""

    def __init__(self, printing_period):
        self.period = printing_period
        self.t0 = time()
        self.last = self.t0
        self.ub = np.inf
        self.solves = 0

    def initialize(self, warm_start, tol):
        if self.period is None:
            return
        print('|%12s|%12s|%12s|%12s|' % ('Updates', 'Time (s)', 'Solves', 'Upper bound'))
        if warm_start is not None:
            print(' Warm start with %d nodes, tolerance %.3e.' % (len(warm_start), tol))

    def update(self, leaves, ub, solves):
        if self.period is None:
            return
        now = time()
        tag = None
        if self.solves == 0:
            tag = 'Root node'
        elif ub < self.ub:
            tag = 'New incumbent'
        elif now - self.last > self.period:
            tag = ''
        self.solves, self.ub = solves, ub
        if tag is not None:
            self.last = now
            print('|%12s|%12.3f|%12d|%12.3e|' % (tag, now - self.t0, solves, ub))

    def finalize(self, solves, ub):
        if self.period is None:
            return
        print('|%12s|%12.3f|%12d|%12.3e|' % ('Solution', time() - self.t0, solves, ub))


def branch_and_bound(
        solver,
        candidate_selection,
        brancher,
        tol=0.,
        warm_start=None,
        printing_period=3.,
        draw_label=None,
        batch_solver=None,
        frontier_width=1,
        incumbent_exchange=None,
        speculation=None,
        stats=None,
        **kwargs
        ):
    '''
    Parameters
    ----------
    solver : function (identifier, cutoff, extra) -> (lb, binary_feasible, solve_time, extra)
        Solves one subproblem (``np.inf`` if infeasible).
    candidate_selection : function list of Node -> Node
    brancher : function Node -> list of Node
    tol : float
        Nonnegative tolerance on the convergence of the branch and bound.
    warm_start : list of Node or None
        Root nodes of the tree (a cover of the binary cube).
    printing_period : float or None
    draw_label : ignored (the reference renders the tree with graphviz, out of scope here).
    batch_solver : function (list of Node, cutoff) -> list of result tuples, optional
        Solves a whole frontier in one call to the batched GPU solver.
    frontier_width : int
        Maximum number of candidates solved per round (1 = reference node order).
    incumbent_exchange : function (ub, n_candidates) -> (ub, n_candidates), optional
        Multi-GPU hook, called once per round by every rank: returns the minimum of the upper
        bound over all ranks and the LARGEST number of open candidates on any rank
        (``distributed.py``).  The search ends when no rank has a candidate left.
    speculation : function identifier -> list of identifiers, optional
        Speculative multi-level expansion (needs ``batch_solver``): descendants of a node that are
        solved in the same launch as the node itself, before it is known whether the search will
        reach them.  Their results wait in a cache and are consumed, unchanged, when (and only
        when) the search selects that node -- the sequence of consumed results, hence incumbent,
        leaves and the returned number of solves, is exactly that of the search without
        speculation; only the number of kernel launches (rounds) drops.
    stats : dict, optional
        Filled with 'rounds' (batch_solver calls), 'launched' (nodes sent to the solver),
        'speculative' (of those, nodes solved ahead of time) and 'wasted' (never consumed).

    Returns
    -------
    incumbent (Node or None), leaves (list of Node), solves (int), solver_time (float)
    '''
    ub = np.inf
    incumbent = None
    leaves = [Node({})] if warm_start is None else warm_start
    solves = 0
    solver_time = 0.

    printer = Printer(printing_period)
    printer.initialize(warm_start, tol)
    width = max(1, int(frontier_width))
    cache = {}                      # identifier key -> result of a speculative solve
    rounds = launched = speculative = 0

    def key(identifier):
        return frozenset(identifier.items())

    while True:
        candidates = [l for l in leaves if l.lb < ub - tol]
        if incumbent_exchange is not None:
            # every rank takes part in every round, also one whose shard is exhausted
            ub_all, open_all = incumbent_exchange(ub, len(candidates))
            if ub_all < ub:
                ub = ub_all
                candidates = [l for l in candidates if l.lb < ub - tol]
            if open_all == 0:
                break
            if not candidates:
                continue
        elif not candidates:
            break
        cutoff = ub - tol

        # pick the frontier by repeated application of the selection rule
        if width == 1 or len(candidates) == 1:
            frontier = [candidate_selection(candidates)]
        else:
            pool = list(candidates)
            frontier = []
            while pool and len(frontier) < width:
                pick = candidate_selection(pool)
                frontier.append(pick)
                pool.remove(pick)

        if batch_solver is not None:
            todo = [node for node in frontier if key(node.identifier) not in cache]
            ahead = []
            if speculation is not None and todo:
                seen = set(key(node.identifier) for node in todo)
                for node in todo:
                    for identifier in speculation(node.identifier):
                        k = key(identifier)
                        if k not in cache and k not in seen:
                            seen.add(k)
                            ahead.append(Node(identifier))
            if todo:
                results = batch_solver(todo + ahead, cutoff)
                rounds += 1
                launched += len(todo) + len(ahead)
                speculative += len(ahead)
                for node, result in zip(todo, results):
                    node._take(result)
                for node, result in zip(ahead, results[len(todo):]):
                    cache[key(node.identifier)] = result
            for node in frontier:
                if node not in todo:
                    node._take(cache.pop(key(node.identifier)))
        else:
            for node in frontier:
                node.solve(solver, cutoff)

        for node in frontier:
            solves += 1
            solver_time += node.solve_time
            cutoff = ub - tol
            if node.lb >= cutoff:
                pass                                    # pruned, stays a leaf
            elif node.binary_feasible:
                incumbent, ub = node, node.lb           # new incumbent, stays a leaf
            else:
                children = brancher(node)
                leaves.remove(node)
                leaves.extend(children)

        printer.update(leaves, ub, solves)

    printer.finalize(solves, ub)
    if stats is not None:
        stats.update(rounds=rounds, launched=launched, speculative=speculative, wasted=len(cache))
    return incumbent, leaves, solves, solver_time


def breadth_first(candidate_nodes):
    '''FIFO selection.'''
    return candidate_nodes[0]


def depth_first(candidate_nodes):
    '''LIFO selection.'''
    return candidate_nodes[-1]


def best_first(candidate_nodes):
    '''Smallest lower bound; the first in the list wins ties.'''
    return candidate_nodes[int(np.argmin([l.lb for l in candidate_nodes]))]
