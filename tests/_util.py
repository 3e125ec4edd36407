"""Shared helpers for the test-suite: fixture loading and bit-level comparisons."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.realpath(__file__)), "golden")
TREES = ("c1", "small", "ragged", "a5")


def load(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def load_tree(name):
    t = load("tree_" + name)
    t["meta"] = json.loads(str(t["meta"]))
    return t


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view({4: np.uint32, 8: np.uint64, 2: np.uint16, 1: np.uint8}[a.dtype.itemsize])


def assert_bits_equal(a, b, what=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    assert a.dtype == b.dtype, f"{what}: dtype {a.dtype} vs {b.dtype}"
    bad = bits(a) != bits(b)
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.size} elements differ bitwise; first at {np.argwhere(bad)[0]}"


def mlp_weights(d, prefix="w_"):
    keys = ("value_fc0_weight", "value_fc0_bias", "value_fc1_weight", "value_fc1_bias",
            "policy_fc0_weight", "policy_fc0_bias", "policy_fc1_weight", "policy_fc1_bias")
    return [d[prefix + k] for k in keys]
