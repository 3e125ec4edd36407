"""NashConv-vs-update curve on the GPU against the reference's own curves (tests/golden/curve_small.npz, produced by
tests/golden/make_curve.py from the imported reference: 3 seeds, 12 updates x 100 steps, batch 512, eta 0.2 on the golden
`small` tree, reference main.py:55-81 hyper-parameters).  Training is stochastic, so the comparison is a band."""
import numpy as np
import pytest
import torch

from _util import load

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tabular", (False, True))
def test_nashconv_curve_matches_reference_band(tmp_path, monkeypatch, tabular):
    """tabular=True: the same training with every net evaluated once per (player, state) (RNaD.tabular)."""
    from _gpu import DEV, golden_tree
    from learn.rnad import RNaD

    ref = load("curve_small")
    curves = ref["nashconv"]  # [seeds, M + 1]
    M, delta, B = int(ref["M"]), int(ref["delta_m"]), int(ref["batch"])
    lo, hi = curves.min(0), curves.max(0)
    tree, _ = golden_tree("small")
    monkeypatch.setenv("RNAD_SAVE_DIR", str(tmp_path))
    mine = []
    for seed in ((0,) if tabular else (0, 1)):
        torch.manual_seed(2000 + seed)
        rn = RNaD(tree=tree, device=DEV, directory_name=f"curve{seed}", eta=float(ref["eta"]), bounds=[M], delta_m=[delta],
                  lr=float(ref["lr"]), gamma_averaging=float(ref["gamma_averaging"]), batch_size=B, logit_clip=2, b1_adam=0.0,
                  net_params={"type": "MLP", "max_actions": 3, "width": 2**8})
        rn.initialize()
        rn.tabular = tabular
        nc0 = rn._evaluate_nashconv()
        rn._RNaD__resume(checkpoint_mod=10**9, expl_mod=1, log_mod=10**9)
        nc = [nc0] + [v for _, _, v in rn.nashconv_history] + [rn._evaluate_nashconv()]
        assert len(nc) == M + 1
        mine.append(nc)
    mine = np.array(mine)
    print("reference band lo", np.round(lo, 3), "\nreference band hi", np.round(hi, 3), "\nthis build        ", np.round(mine, 3))
    tol = 0.15
    assert (mine >= lo - tol).all() and (mine <= hi + tol).all(), (mine, lo, hi)
    assert (mine[:, -1] < 0.5).all() and (mine[:, 0] > 1.2).all()  # 1.5 -> ~0.35 like the reference (README plot: 1.2-1.4 -> 0.35-0.5)
