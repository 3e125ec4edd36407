"""Tree construction (host side): seeded generation matches the reference bit for bit; native generator and solver."""
import random

import numpy as np
import pytest
import torch

from _util import TREES, assert_bits_equal, load_tree
from environment.tree import Tree
import rnad_hip


def _lambdas(meta):
    """Re-declaration of the lambdas tests/golden/make_golden.py passed to the reference."""
    kw = {}
    if meta.get("depth_lambda") == "prune":
        kw["depth_bound_lambda"] = lambda tree: tree.depth_bound - 1 - 2 * (random.random() < 0.5)
    if meta.get("row_lambda") == "ragged":
        kw["row_actions_lambda"] = lambda tree: random.randint(1, tree.max_actions)
    if meta.get("col_lambda") == "ragged":
        kw["col_actions_lambda"] = lambda tree: random.randint(1, tree.max_actions)
    return kw


def build_like_reference(name):
    g = load_tree(name)
    m = g["meta"]
    torch.manual_seed(m["seed"]); np.random.seed(m["seed"]); random.seed(m["seed"])  # noqa: E702
    t = Tree(**m["kw"], **_lambdas(m))
    t.generate()
    return t, g


@pytest.mark.parametrize("name", TREES)
def test_seeded_generate_is_bit_identical_to_reference(name):
    """Same numpy / random / torch seeds as the reference run -> identical tensors (structure AND values) and hash."""
    t, g = build_like_reference(name)
    t.assert_index_is_tree()
    assert t.hash == g["meta"]["hash"]
    for key, tensor in (("index", t.index_tensor), ("chance", t.chance_tensor), ("legal", t.legal_tensor),
                        ("value", t.value_tensor), ("expected_value", t.expected_value_tensor),
                        ("root_value", t.root_value_tensor), ("solution", t.solution_tensor)):
        assert_bits_equal(tensor.numpy(), g[key], key)


def test_saved_keys_and_save_load_roundtrip(tmp_path, monkeypatch):
    monkeypatch.setenv("RNAD_SAVE_DIR", str(tmp_path))
    t, _ = build_like_reference("c1")
    assert t.saved_keys == ["is_root", "device", "max_actions", "max_transitions", "row_actions", "col_actions", "depth_bound",
                            "transition_threshold", "terminal_values", "index_tensor", "value_tensor", "expected_value_tensor",
                            "legal_tensor", "chance_tensor", "root_value_tensor", "solution_tensor", "desc", "hash"]
    t.save("unit")
    u = Tree(max_actions=2)
    u.load("unit")
    assert u.hash == t.hash and torch.equal(u.index_tensor, t.index_tensor) and torch.equal(u.value_tensor, t.value_tensor)
    u.load("recent")
    assert u.depth_bound == 3


def test_assert_index_is_tree_rejects_broken_trees():
    t, _ = build_like_reference("small")
    t.index_tensor[5, 0, 0, 0] = 3  # duplicate / decreasing entry
    with pytest.raises(AssertionError):
        t.assert_index_is_tree()


def test_solver_matches_linear_programming():
    from scipy.optimize import linprog

    rng = np.random.default_rng(1)
    for trial in range(150):
        ra, ca = int(rng.integers(1, 6)), int(rng.integers(1, 6))
        M = (rng.choice([-1.0, 1.0], size=(ra, ca)) if trial % 2 else rng.standard_normal((ra, ca))).astype(np.float32)
        sol, val = rnad_hip.solve_matrix(torch.tensor(M), 5)
        x, y = sol[:ra].numpy().astype(np.float64), sol[5:5 + ca].numpy().astype(np.float64)
        assert abs(x.sum() - 1) < 1e-6 and abs(y.sum() - 1) < 1e-6 and x.min() >= 0 and y.min() >= 0
        c = np.zeros(ra + 1); c[-1] = -1  # noqa: E702
        r = linprog(c, A_ub=np.hstack([-M.T, np.ones((ca, 1))]), b_ub=np.zeros(ca), A_eq=np.array([[1.0] * ra + [0.0]]),
                    b_eq=[1], bounds=[(0, None)] * ra + [(None, None)])
        assert abs(-r.fun - val) < 1e-5
        assert (x @ M).min() >= val - 1e-5 and (M @ y).max() <= val + 1e-5  # (x, y) is an equilibrium


def test_equilibrium_choice_on_degenerate_matrices_is_the_documented_tie_break():
    """`Tree._solve` (tree.py:199-234) keeps `solutions[0]` after a STABLE sort of pygambit's `enummixed_solve` output by purity score.
    Which equilibrium that is depends on the order in which pygambit 16.0.2 (not vendored, not installable here) lists the extreme
    equilibria -- PARITY UNPINNED for `solution_tensor` whenever a matrix game has several.  What IS pinned here, on +-1 matrices with
    many equilibria: the native solver picks exactly the pair the reference's own selection rule picks from the list enumerated in the
    order INTEGRATION.md documents (supports by size, then rows, then columns lexicographically; pairs x-major), every candidate is
    an equilibrium of the same value, so NashConv(solution) and every value tensor are the same whichever one is kept."""
    import sys

    from _util import GOLDEN

    sys.path.insert(0, GOLDEN)
    from _pygambit_stub import extreme_strategies

    rng = np.random.default_rng(7)
    several = 0
    for trial in range(200):
        ra, ca = int(rng.integers(2, 5)), int(rng.integers(2, 5))
        M = rng.choice([-1.0, 1.0], size=(ra, ca)).astype(np.float32)
        xs, ys = extreme_strategies(M)
        assert xs and ys
        solutions = [list(x) + [0.0] * (4 - ra) + list(y) + [0.0] * (4 - ca) for x in xs for y in ys]  # tree.py:213-218 with max_actions = 4
        several += len(solutions) > 1
        value = None
        for sol in solutions:  # every listed pair is an equilibrium, all of one value
            x, y = np.array(sol[:ra]), np.array(sol[4:4 + ca])
            v = float(x @ M @ y)
            value = v if value is None else value
            assert abs(v - value) < 1e-9 and (x @ M).min() >= v - 1e-9 and (M @ y).max() <= v + 1e-9
        purity = lambda sol: -int(1 in sol[:4]) - int(1 in sol[4:])  # noqa: E731  (tree.py:227-229)
        want = sorted(solutions, key=purity)[0]  # list.sort is stable (tree.py:230)
        got, val = rnad_hip.solve_matrix(torch.tensor(M), 4)
        np.testing.assert_allclose(got.numpy(), np.array(want, np.float32), atol=1e-6)
        assert abs(val - value) < 1e-6
    assert several > 50, "the test wants matrices with several equilibria"


@pytest.mark.parametrize("A,C,depth,thr", [(2, 1, 3, 0.0), (3, 1, 4, 0.0), (3, 2, 3, 0.3), (5, 4, 2, 0.2)])
def test_native_generator_invariants(A, C, depth, thr):
    t = Tree(max_actions=A, max_transitions=C, depth_bound=depth, transition_threshold=thr)
    t.generate_native(seed=5)
    t.assert_index_is_tree()
    S = t.index_tensor.shape[0]
    if C == 1:
        assert S == 1 + sum((A * A) ** k for k in range(depth))  # regular tree: 1 + sum (A^2)^k
    ch = t.chance_tensor
    assert torch.allclose(ch[1:].sum(1), torch.ones(S - 1, A, A), atol=1e-6)
    assert (ch[1:][ch[1:] > 0] >= thr - 1e-7).all()
    assert ch[0, 0, 0, 0] == 1 and ch[0].sum() == 1 and t.legal_tensor[0].sum() == 1
    # expected_value = sum_t value * chance; child payoffs are the children's root values
    ev = (t.value_tensor * ch).sum(1, keepdim=True)
    assert torch.allclose(ev, t.expected_value_tensor, atol=1e-6)
    idx = t.index_tensor
    nz = idx != 0
    assert torch.allclose(t.value_tensor[nz], t.root_value_tensor[idx[nz], 0])
    term = (~nz) & (ch > 0)
    term[0] = False
    assert set(t.value_tensor[term].unique().tolist()) <= {-1.0, 1.0}
    # same seed -> same tree, other seed -> another tree
    u = Tree(max_actions=A, max_transitions=C, depth_bound=depth, transition_threshold=thr)
    u.generate_native(seed=5)
    assert torch.equal(u.value_tensor, t.value_tensor) and u.hash == t.hash
    u.generate_native(seed=6)
    assert not torch.equal(u.value_tensor, t.value_tensor)


def test_native_generator_solution_is_a_nash_equilibrium():
    """NashConv(solution) == 0 and row_best[root] == root_value (the invariant reference tests/test_nashconv.py aimed at)."""
    from oracle import oracle

    t = Tree(max_actions=3, max_transitions=2, depth_bound=3, transition_threshold=0.2)
    t.generate_native(seed=3, prune=(1, 3))
    t.assert_index_is_tree()
    sol = t.solution_tensor.numpy()
    rb, cb, reach, depth = oracle.nashconv(t.index_tensor.numpy(), t.value_tensor.numpy(), t.chance_tensor.numpy(),
                                           t.legal_tensor.numpy(), sol[1], sol)
    assert abs(rb[1] + cb[1]) < 1e-5
    assert abs(rb[1] - t.root_value_tensor[1, 0].item()) < 1e-5


def test_loads_a_tree_file_written_by_the_reference(tmp_path, monkeypatch):
    """tests/golden/ref_tree_c1.tar is `Tree.save()` of the reference itself (tree.py:385-415): same format, loads directly."""
    import os
    import shutil

    from _util import GOLDEN

    monkeypatch.setenv("RNAD_SAVE_DIR", str(tmp_path))
    os.makedirs(tmp_path / "saved_trees" / "from_reference")
    shutil.copy(os.path.join(GOLDEN, "ref_tree_c1.tar"), tmp_path / "saved_trees" / "from_reference" / "tree.tar")
    g = load_tree("c1")
    t = Tree(max_actions=3, depth_bound=1)  # deliberately a different shape: load() overwrites everything (tree.py:430-432)
    t.load("from_reference")
    assert (t.max_actions, t.max_transitions, t.depth_bound, t.hash) == (2, 1, 3, g["meta"]["hash"])
    for key, tensor in (("index", t.index_tensor), ("value", t.value_tensor), ("chance", t.chance_tensor), ("legal", t.legal_tensor),
                        ("expected_value", t.expected_value_tensor), ("root_value", t.root_value_tensor), ("solution", t.solution_tensor)):
        assert_bits_equal(tensor.numpy(), g[key], key)
    t.assert_index_is_tree()


def test_resumes_from_a_checkpoint_written_by_the_reference(tmp_path, monkeypatch):
    """saved_runs/<dir>/{params, <m>/<n>} as the reference writes them (rnad.py:208-209, :307-319): resuming picks the
    latest (m, n), restores hyper-parameters, the four nets and the optimizer (rnad.py:243-272)."""
    import os
    import shutil

    from _util import GOLDEN, load
    from learn.rnad import RNaD

    monkeypatch.setenv("RNAD_SAVE_DIR", str(tmp_path))
    run = tmp_path / "saved_runs" / "from_reference"
    os.makedirs(run / "1")
    shutil.copy(os.path.join(GOLDEN, "ref_run_params"), run / "params")
    shutil.copy(os.path.join(GOLDEN, "ref_run_ckpt_1_0"), run / "1" / "0")
    tree, tg = build_like_reference("c1")
    rn = RNaD(tree=tree, device=torch.device("cpu"), directory_name="from_reference", b1_adam=0.0)
    rn.initialize()
    g = load("run_c1")
    assert (rn.m, rn.n, rn.total_steps) == (1, 0, 3)
    assert (rn.eta, rn.lr, rn.gamma_averaging, rn.batch_size, rn.bounds, rn.delta_m) == (0.2, 1e-2, 0.1, 64, [2], [3])
    assert rn.net_params == {"type": "MLP", "max_actions": 2, "width": 16} and rn.tree_hash == tree.hash
    # the checkpoint at (m, n) = (1, 0) holds the nets after step 2 and the rotation at the end of m = 0 (rnad.py:528-531)
    for tag, net, key in (("net", rn.net, "s2_net_"), ("target", rn.net_target, "s2_net_target_"), ("reg", rn.net_reg, "s2_net_target_"),
                          ("reg_", rn.net_reg_, "s2_net_reg_")):
        for k, p in net.state_dict().items():
            assert_bits_equal(p.numpy(), g[key + k.replace(".", "_")], f"{tag} {k}")
    assert len(rn.optimizer.state_dict()["state"]) == 8 and rn.optimizer.state_dict()["param_groups"][0]["lr"] == 1e-2
    # a tree with another hash is refused (rnad.py:256-258)
    tree.hash += 1
    with pytest.raises(AssertionError):
        RNaD(tree=tree, device=torch.device("cpu"), directory_name="from_reference", b1_adam=0.0).initialize()
