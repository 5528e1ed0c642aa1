"""Input container for the drop-in boundary: a discrete-time MLD system

    x(t+1) = A x(t) + B u(t),      F x(t) + G u(t) <= h,

with the last ``nub`` entries of ``u`` binary.

Mirrors the numeric part of the reference's ``MLDSystem``
(``warm_start_hmpc/mld_system.py:19-64``): same constructor signature, same
attribute names (``A B F G h nx nu nub nuc V``) and the same ``ValueError``s.
The symbolic constructors (``from_symbolic``, ``from_pwa``; ``mld_system.py:66-214``)
are offline modelling helpers and are out of scope here; an
instance of the reference's own class can be passed to the controller
unchanged because only these attributes are read.
"""
import numpy as np


class MLDSystem(object):

    def __init__(self, dynamics, constraints, nub):
        A, B = dynamics
        F, G, h = constraints
        self.A = np.atleast_2d(np.asarray(A, dtype=np.float64))
        self.B = np.atleast_2d(np.asarray(B, dtype=np.float64))
        self.F = np.atleast_2d(np.asarray(F, dtype=np.float64))
        self.G = np.atleast_2d(np.asarray(G, dtype=np.float64))
        self.h = np.asarray(h, dtype=np.float64).reshape(-1)

        self.nx = self.A.shape[1]
        self.nu = self.B.shape[1]
        self.nub = int(nub)
        self.nuc = self.nu - self.nub

        # binaries are the trailing block of u
        self.V = np.hstack((np.zeros((self.nub, self.nuc)), np.eye(self.nub)))

        if self.A.shape[0] != self.A.shape[1]:
            raise ValueError('Nonsquare A matrix.')
        if self.B.shape[0] != self.nx:
            raise ValueError('A and B matrices have incompatible size.')
        if self.F.shape != (self.h.size, self.nx):
            raise ValueError('Matrix F has incompatible size.')
        if self.G.shape != (self.h.size, self.nu):
            raise ValueError('Matrix G has incompatible size.')
        if not 0 <= self.nub <= self.nu:
            raise ValueError('Number of binaries exceeds number of inputs.')
