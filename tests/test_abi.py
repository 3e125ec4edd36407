"""The C-ABI shared library builds, loads without a GPU and exports every symbol include/rnad_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
HEADER = os.path.join(ROOT, "include", "rnad_hip.h")
SO = os.path.join(ROOT, "r-nad_amd", "csrc", "librnad_hip.so")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(rnad_[a-z_0-9]+)\s*\(", src)
    return sorted(set(names))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("rnad_tree_create", "rnad_observe", "rnad_sample", "rnad_transition", "rnad_rollout_begin", "rnad_rollout_step",
                 "rnad_process_policy", "rnad_vtrace", "rnad_loss_v", "rnad_loss_nerd", "rnad_learn_fused", "rnad_nashconv",
                 "rnad_tree_generate", "rnad_solve_matrix", "rnad_last_error"):
        assert must in names


def test_library_loads_and_exports_every_declared_symbol():
    assert os.path.exists(SO), "build it first: make -C r-nad_amd/csrc (or __graft_entry__.build())"
    lib = ctypes.CDLL(SO)
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} is declared in include/rnad_hip.h but not exported"
    assert lib.rnad_version() >= 1


def test_binding_fails_loudly_without_a_gpu_tensor():
    import pytest
    import torch

    import rnad_hip

    with pytest.raises(rnad_hip.RnadHipError, match="GPU"):
        rnad_hip.process_policy(torch.zeros(4, 3), torch.ones(4, 3), 32, 0.03)
    with pytest.raises(rnad_hip.RnadHipError):
        rnad_hip.TreeHandle(torch.zeros(2, 1, 2, 2, dtype=torch.int64), torch.zeros(2, 1, 2, 2), torch.zeros(2, 1, 2, 2),
                            torch.zeros(2, 1, 2, 2), torch.zeros(2, 1, 2, 2), "cpu")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "r-nad_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower() or f == "tree_gen.cpp" and "oracle" not in text, (dirpath, f)
