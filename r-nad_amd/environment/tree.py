"""Tree -- the tensor-encoded stochastic matrix tree, drop-in for reference environment/tree.py.

Same constructor, attributes and methods as the reference `Tree` (environment/tree.py:66-442): seven
per-state tensors keyed by state id (0 = absorbing, 1 = root), `generate()`, `assert_index_is_tree()`,
`save() / load()`, `to()`.  Differences, all behind the same API:

  * the kernels do not read these tensors: `handle()` re-packs them once into the HBM layout of
    librnad_hip.so (node rows + per-joint-action transition records) and caches the handle;
  * `generate()` is an iterative numpy builder instead of a recursion over torch ops, but consumes
    numpy's / python's global RNGs in exactly the reference's order (one Dirichlet draw per `Tree()`
    the reference would have constructed, lambdas called in `_init_child` order, one
    `random.choice` per terminal child), so a seeded reference tree and a seeded tree built here have
    identical `index / chance / legal` tensors;
  * each state's matrix game is solved by the native Shapley-Snow solver (rnad_solve_matrix) instead
    of pygambit (not installable here).  Zero-sum game VALUES are unique, so `value /
    expected_value / root_value` match any exact solver; which equilibrium lands in
    `solution_tensor` when there are several is solver-specific ("parity unpinned");
  * `generate_native(seed)` builds large regular trees (depth-6 ternary: 66 431 states) in under a
    second in C++ with its own seeded generator.
"""
import logging
import os
import random
import time
from typing import Dict

import numpy as np
import torch

import rnad_hip

_SAVED_KEYS_TENSORS = (
    "index_tensor", "value_tensor", "expected_value_tensor", "legal_tensor", "chance_tensor",
    "root_value_tensor", "solution_tensor",
)


def _save_root():
    return os.environ.get("RNAD_SAVE_DIR") or os.path.join(os.path.dirname(os.path.realpath(__file__)), "..")


class _Node:
    """Build-time record of one `Tree()` the reference would construct (tree.py:115-146)."""

    __slots__ = ("row_actions", "col_actions", "depth_bound", "max_actions", "max_transitions", "chance")

    def __init__(self, owner, row_actions, col_actions, depth_bound):
        A, Cc = owner.max_actions, owner.max_transitions
        self.max_actions, self.max_transitions = A, Cc
        self.row_actions, self.col_actions, self.depth_bound = row_actions, col_actions, depth_bound
        self.chance = owner._transition_probs(A, A, Cc, owner.transition_threshold)  # [C, A, A]
        self.chance[:, row_actions:, :] = 0.0  # chance_tensor *= legal_tensor (tree.py:137)
        self.chance[:, :, col_actions:] = 0.0


class Tree:
    def __init__(
        self,
        is_root=True,
        device=torch.device("cpu"),
        max_actions=3,
        max_transitions=1,
        row_actions=None,
        col_actions=None,
        depth_bound=1,
        row_actions_lambda=None,
        col_actions_lambda=None,
        depth_bound_lambda=None,
        transition_threshold=0,
        terminal_values=(-1, 1),
        desc="",
    ):
        if not (1 <= max_actions <= rnad_hip.MAX_ACTIONS and 1 <= max_transitions <= rnad_hip.MAX_TRANSITIONS):
            raise ValueError(f"max_actions / max_transitions must be in [1, {rnad_hip.MAX_ACTIONS}]")
        self.is_root = is_root
        self.device = device
        self.max_actions = max_actions
        self.max_transitions = max_transitions
        self.row_actions = row_actions if row_actions is not None else max_actions
        self.col_actions = col_actions if col_actions is not None else max_actions
        self.depth_bound = depth_bound
        self.transition_threshold = transition_threshold
        self.terminal_values = terminal_values

        A, Cc = max_actions, max_transitions
        self.index_tensor = torch.zeros((1, Cc, A, A), device=device, dtype=torch.long)
        self.value_tensor = torch.zeros((1, Cc, A, A), device=device, dtype=torch.float)
        self.expected_value_tensor = torch.zeros((1, 1, A, A), device=device, dtype=torch.float)
        self.legal_tensor = torch.zeros((1, 1, A, A), device=device, dtype=torch.float)
        self.legal_tensor[0, 0, : self.row_actions, : self.col_actions] = 1.0
        # the root's chance profile: first Dirichlet draw, as in the reference ctor (tree.py:134-137)
        self._root = _Node(self, self.row_actions, self.col_actions, depth_bound)
        self.chance_tensor = torch.from_numpy(self._root.chance.copy()).unsqueeze(0).to(device)
        self.root_value_tensor = torch.zeros((1, 1), device=device, dtype=torch.float)
        self.solution_tensor = torch.zeros((1, 2 * A), device=device, dtype=torch.float)
        self.desc = desc
        self.hash = 0

        self.saved_keys = [k for k in self.__dict__.keys() if not k.startswith("_")]
        # Only the keys above this are saved and reloaded (tree.py:145).

        self.row_actions_lambda = row_actions_lambda if row_actions_lambda is not None else (lambda tree: tree.row_actions)
        self.col_actions_lambda = col_actions_lambda if col_actions_lambda is not None else (lambda tree: tree.col_actions)
        self.depth_bound_lambda = depth_bound_lambda if depth_bound_lambda is not None else (lambda tree: tree.depth_bound - 1)
        self._handle = None

    # ------------------------------------------------------------------ tree.py:182-197
    def _transition_probs(self, rows, cols, n_trans, transition_threshold):
        """Dirichlet(1/C) per joint action -> zero entries below the threshold -> L1 renormalise.  Returns [C, rows, cols] fp32."""
        chance = np.random.dirichlet((1 / n_trans,) * n_trans, (1, rows, cols)).astype(np.float32)[0]
        chance = chance - np.where(chance < np.float32(transition_threshold), chance, np.float32(0))
        denom = np.maximum(np.abs(chance).sum(axis=2, keepdims=True, dtype=np.float32), np.float32(1e-12))
        chance = (chance / denom).astype(np.float32)
        return np.ascontiguousarray(np.moveaxis(chance, 2, 0))

    # ------------------------------------------------------------------ tree.py:199-234
    def _solve(self, M: torch.Tensor, max_actions=2):
        sol, _ = rnad_hip.solve_matrix(M, max_actions)
        return sol.unsqueeze(0).to(self.device)

    # ------------------------------------------------------------------ tree.py:164-180
    def _child_params(self, parent):
        A = self.max_actions
        row = min(A, max(1, self.row_actions_lambda(parent)))
        col = min(A, max(1, self.col_actions_lambda(parent)))
        depth = max(0, self.depth_bound_lambda(parent))
        return row, col, depth

    # ------------------------------------------------------------------ tree.py:236-366
    def generate(self):
        """Depth-first build in the reference's visiting order (row, col, chance), ids in DFS pre-order."""
        if not self.is_root:
            raise Exception("generate() builds whole trees; sub-trees are internal to the builder")
        A, Cc = self.max_actions, self.max_transitions
        AA = A * A
        index, value, chance, ev, legal, rootv, sol = [], [], [], [], [], [], []

        def new_state(node):
            sid = len(index)
            index.append(np.zeros((Cc, A, A), np.int64))
            value.append(np.zeros((Cc, A, A), np.float32))
            chance.append(node.chance)
            ev.append(np.zeros((A, A), np.float32))
            lg = np.zeros((A, A), np.float32)
            lg[: node.row_actions, : node.col_actions] = 1.0
            legal.append(lg)
            rootv.append(np.float32(0))
            sol.append(np.zeros((2 * A,), np.float32))
            return sid

        # absorbing state 0 is appended LAST by the reference (its ctor draws after the whole recursion,
        # tree.py:338-349) but sits at id 0; reserve the slot now and fill it at the end.
        index.append(None); value.append(None); chance.append(None); ev.append(None)  # noqa: E702
        legal.append(None); rootv.append(None); sol.append(None)  # noqa: E702

        def finish(node, sid):
            """tree.py:285-301: solve the expected-value matrix of a state whose children are done."""
            m = torch.from_numpy(ev[sid][: node.row_actions, : node.col_actions].copy())
            s, rv = rnad_hip.solve_matrix(m, A)
            sol[sid] = s.numpy()
            rootv[sid] = np.float32(rv)
            return np.float32(rv)

        # explicit stack of (node, state id, iterator position) instead of recursion
        root_sid = new_state(self._root)
        stack = [[self._root, root_sid, 0, None]]  # node, sid, next flat position, pending (chance, row, col)
        ret = None
        while stack:
            frame = stack[-1]
            node, sid, pos, pending = frame
            if pending is not None:  # a child just returned its NE payoff
                t, r, c = pending
                value[sid][t, r, c] = ret
                frame[3] = None
            advanced = False
            while pos < node.row_actions * node.col_actions * Cc:
                r, rem = divmod(pos, node.col_actions * Cc)
                c, t = divmod(rem, Cc)
                pos += 1
                if node.chance[t, r, c] > 0:
                    crow, ccol, cdepth = self._child_params(node)
                    child = _Node(self, crow, ccol, cdepth)  # the reference's child ctor draws its Dirichlet here
                    if cdepth > 0 and crow * ccol > 0:
                        csid = new_state(child)
                        index[sid][t, r, c] = csid
                        frame[2], frame[3] = pos, (t, r, c)
                        stack.append([child, csid, 0, None])
                        advanced = True
                        break
                    value[sid][t, r, c] = np.float32(random.choice(self.terminal_values))  # tree.py:273-275
            if advanced:
                continue
            # all children done: expected values (tree.py:280-282), solve, return
            for r in range(node.row_actions):
                for c in range(node.col_actions):
                    ev[sid][r, c] = (value[sid][:, r, c] * node.chance[:, r, c]).sum(dtype=np.float32)
            ret = finish(node, sid)
            stack.pop()

        absorbing = _Node(self, 1, 1, 0)  # tree.py:338-346: consumes one Dirichlet draw
        index[0] = np.zeros((Cc, A, A), np.int64)
        value[0] = np.zeros((Cc, A, A), np.float32)
        ch0 = np.zeros((Cc, A, A), np.float32)
        ch0[0, 0, 0] = 1.0  # tree.py:347-348
        del absorbing
        chance[0] = ch0
        ev[0] = np.zeros((A, A), np.float32)
        lg0 = np.zeros((A, A), np.float32)
        lg0[0, 0] = 1.0
        legal[0] = lg0
        rootv[0] = np.float32(0)
        sol[0] = np.zeros((2 * A,), np.float32)

        dev = self.device
        self.index_tensor = torch.from_numpy(np.stack(index)).to(dev)
        self.value_tensor = torch.from_numpy(np.stack(value)).to(dev)
        self.chance_tensor = torch.from_numpy(np.stack(chance)).to(dev)
        self.expected_value_tensor = torch.from_numpy(np.stack(ev)).unsqueeze(1).to(dev)
        self.legal_tensor = torch.from_numpy(np.stack(legal)).unsqueeze(1).to(dev)
        self.root_value_tensor = torch.from_numpy(np.array(rootv, np.float32)).unsqueeze(1).to(dev)
        self.solution_tensor = torch.from_numpy(np.stack(sol)).to(dev)
        self.hash = torch.randint(-(2**63), 2**63 - 1, size=(1,)).item()  # tree.py:366
        self._handle = None

    def generate_native(self, seed=0, prune=(0, 0)):
        """Regular tree (every state `max_actions` x `max_actions`, depth_bound - 1 below it) built by the C++
        generator with its own seeded RNG.  prune=(num, den): each child's depth is lowered by 2 more with
        probability num/den (the pruning of reference main.py:37)."""
        if self.row_actions != self.max_actions or self.col_actions != self.max_actions:
            raise ValueError("generate_native builds regular trees only")
        out = rnad_hip.tree_generate(self.max_actions, self.max_transitions, self.depth_bound, float(self.transition_threshold),
                                     tuple(float(v) for v in self.terminal_values), prune, seed)
        dev = self.device
        self.index_tensor = out["index"].to(dev)
        self.value_tensor = out["value"].to(dev)
        self.chance_tensor = out["chance"].to(dev)
        self.expected_value_tensor = out["expected_value"].to(dev)
        self.legal_tensor = out["legal"].to(dev)
        self.root_value_tensor = out["root_value"].to(dev)
        self.solution_tensor = out["solution"].to(dev)
        g = torch.Generator().manual_seed(int(seed))
        self.hash = torch.randint(-(2**63), 2**63 - 1, size=(1,), generator=g).item()
        self._handle = None

    # ------------------------------------------------------------------ tree.py:368-383
    def assert_index_is_tree(self):
        """The non-zero index entries are exactly [1 + is_root, S) once each, and every child id > its parent's."""
        idx = self.index_tensor.cpu()
        nz = idx[idx != 0]
        indices = torch.sort(nz).values
        expect = torch.arange(1 + int(self.is_root), 1 + int(self.is_root) + indices.numel())
        assert torch.equal(indices, expect)
        ids = torch.arange(idx.shape[0] - 1).view(-1, 1, 1, 1)
        sl = idx[1:]
        assert torch.all((sl > ids) | (sl == 0))

    # ------------------------------------------------------------------ tree.py:385-433
    def save(self, directory_name=None):
        """Save to <root>/saved_trees/<directory_name>/tree.tar and <root>/saved_trees/recent/tree.tar (torch.save of the
        same key -> value dict as the reference).  <root> is the package directory, or $RNAD_SAVE_DIR."""
        if not self.is_root:
            raise Exception("Attempting to save non-root tree")
        directory = os.path.join(_save_root(), "saved_trees")
        if directory_name is None:
            directory_name = str(int(time.time()))
        path = os.path.join(directory, directory_name)
        recent_path = os.path.join(directory, "recent")
        os.makedirs(path, exist_ok=True)
        os.makedirs(recent_path, exist_ok=True)
        saved_dict = {key: self.__dict__[key] for key in self.saved_keys}
        torch.save(saved_dict, os.path.join(recent_path, "tree.tar"))
        torch.save(saved_dict, os.path.join(path, "tree.tar"))
        logging.info("saving trees to '{}' and 'recent'".format(path))

    def load(self, directory_name="recent"):
        path = os.path.join(_save_root(), "saved_trees", directory_name, "tree.tar")
        logging.info("loading tree from '{}'".format(directory_name))
        saved: Dict = torch.load(path, weights_only=False)
        for key, value in saved.items():
            self.__dict__[key] = value
        self._handle = None
        logging.info("loaded tree has hash {}".format(self.hash))

    def to(self, device):
        self.device = device
        for key, value in self.__dict__.items():
            if torch.is_tensor(value):
                self.__dict__[key] = value.to(device)
        self._handle = None

    # ------------------------------------------------------------------ native tables
    def handle(self, device=None) -> "rnad_hip.TreeHandle":
        """The packed GPU tables of this tree (built on first use, rebuilt after to()/load()/generate())."""
        dev = torch.device(device if device is not None else self.device)
        if self._handle is None or self._handle.device != dev:
            self._handle = rnad_hip.TreeHandle(self.index_tensor, self.value_tensor, self.chance_tensor,
                                               self.expected_value_tensor, self.legal_tensor, dev)
        return self._handle
