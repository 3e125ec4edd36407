"""Stand-in for `pygambit` used ONLY by tests/golden/make_golden.py.

The reference imports pygambit 16.0.2 at module top (reference environment/tree.py:5) and
calls it from `Tree._solve` (tree.py:199-234).  pygambit is a third-party dependency that is
not vendored under /root/reference and is not installed in this image (no network), so the
fixture generator injects this module into `sys.modules["pygambit"]` before importing the
reference.  It implements only the four names the reference touches:

    pygambit.Decimal, pygambit.Game.from_arrays, pygambit.nash.enummixed_solve,
    pygambit.nash.lcp_solve

`enummixed_solve` enumerates the extreme equilibria of the zero-sum game by the Shapley-Snow
kernel method (every extreme optimal strategy of a matrix game with positive value comes from a
square non-singular sub-matrix B: v = 1/(1'B^-1 1), x = v 1'B^-1, y = v B^-1 1).  The game VALUE
is unique, so `value_tensor / expected_value_tensor / root_value_tensor` of a generated tree do
not depend on which solver is used; the choice among several equilibria (`solution_tensor`) does
("parity unpinned" for that tensor).  The product's native solver
(r-nad_amd/csrc/tree_gen.cpp) walks sub-matrices in the same order, so both pick the same one.
"""
from itertools import combinations

import numpy as np

Decimal = float


class Game:
    def __init__(self, a):
        self.a = np.array(a, dtype=np.float64)

    @classmethod
    def from_arrays(cls, a, b):
        return cls(a)


def extreme_strategies(m, eps=1e-9):
    """All extreme optimal (x, y) of the zero-sum matrix game `m` (row maximises)."""
    m = np.asarray(m, dtype=np.float64)
    ra, ca = m.shape
    b = m + (1.0 - m.min())  # every entry >= 1  => value > 0
    xs, ys = [], []

    def add(lst, v):
        for w in lst:
            if np.max(np.abs(w - v)) < 1e-7:
                return
        lst.append(v)

    for k in range(1, min(ra, ca) + 1):
        for rows in combinations(range(ra), k):
            for cols in combinations(range(ca), k):
                sub = b[np.ix_(rows, cols)]
                if abs(np.linalg.det(sub)) < 1e-12:
                    continue
                ones = np.ones(k)
                yk = np.linalg.solve(sub, ones)
                xk = np.linalg.solve(sub.T, ones)
                if yk.min() < -eps or xk.min() < -eps:
                    continue
                sy, sx = yk.sum(), xk.sum()
                if sy <= eps or sx <= eps:
                    continue
                v = 1.0 / sy
                x = np.zeros(ra)
                y = np.zeros(ca)
                x[list(rows)] = xk / sx
                y[list(cols)] = yk / sy
                if (x @ b).min() < v - 1e-7 or (b @ y).max() > v + 1e-7:
                    continue
                x[np.abs(x) < 1e-12] = 0.0
                y[np.abs(y) < 1e-12] = 0.0
                x[np.abs(x - 1.0) < 1e-12] = 1.0
                y[np.abs(y - 1.0) < 1e-12] = 1.0
                add(xs, x)
                add(ys, y)
    return xs, ys


class nash:
    @staticmethod
    def enummixed_solve(g, rational=False):
        xs, ys = extreme_strategies(g.a)
        return [list(x) + list(y) for x in xs for y in ys]

    @staticmethod
    def lcp_solve(g, rational=False):
        return nash.enummixed_solve(g, rational)
