"""V-trace / NeuRD functions -- drop-in for reference learn/vtrace.py, each backed by one HIP kernel.

Same free functions, argument order and return values as the reference (which is itself an adaptation of
OpenSpiel's R-NaD): `process_policy`, `_player_others`, `v_trace`, `get_loss_v`, `get_loss_nerd`.  Inputs may be any
GPU tensors of the reference's shapes; they are made contiguous fp32/int32 for the kernels.  `RNaD.__learn` does not
go through these one by one -- it calls the fused kernel (rnad_hip.learn_fused) -- but they compute the same
expressions in the same order and are what tests compare against the reference's fixtures.
"""
from typing import Any, Sequence, Tuple

import torch

import rnad_hip


def _f32(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


def process_policy(policy: torch.Tensor, mask: torch.Tensor, n_disc, epsilon_threshold=0.03):
    """Threshold (eps) + discretise (n_disc) the learner policy.  reference learn/vtrace.py:24-55."""
    t_eff, batch_size, n_actions = policy.shape
    out = rnad_hip.process_policy(_f32(policy).view(-1, n_actions), _f32(mask).view(-1, n_actions), n_disc, epsilon_threshold)
    return out.view(t_eff, batch_size, n_actions)


def _player_others(player_ids: torch.Tensor, valid: torch.Tensor, player: int) -> torch.Tensor:
    """1 for the current player, -1 for others, 0 on invalid steps, shape [..., 1].  reference learn/vtrace.py:70-87."""
    res = 2 * (player_ids == player) - 1
    res = res * valid
    return torch.unsqueeze(res, dim=-1)


def v_trace(
    v: torch.Tensor,
    valid: torch.Tensor,
    player_id: torch.Tensor,
    acting_policy: torch.Tensor,
    merged_policy: torch.Tensor,
    merged_log_policy: torch.Tensor,
    player_others: torch.Tensor,
    actions_oh: torch.Tensor,
    reward: torch.Tensor,
    player: int,
    # Scalars below.
    eta: float,
    lambda_: float,
    c: float,
    rho: float,
    gamma=1.0,
) -> Tuple[Any, Any, Any]:
    """Two-player V-trace for `player`.  reference learn/vtrace.py:207-352.

    Returns (v_target [T,B,1], has_played [T,B] int64, learning_output [T,B,A]).  `player_others` must be
    `_player_others(player_id, valid, player)` as at the reference's only call site (learn/rnad.py:393); the kernel
    recomputes it.  `actions_oh` may also be an int32 `[T,B]` tensor of action ids.
    """
    T, B, A = acting_policy.shape
    acts = actions_oh.contiguous() if actions_oh.dtype == torch.int32 else _f32(actions_oh)
    vt, has_played, q = rnad_hip.vtrace(
        _f32(v).view(T, B), _f32(valid), player_id.to(torch.int32).contiguous(), _f32(acting_policy), _f32(merged_policy),
        _f32(merged_log_policy), acts, _f32(reward), player, eta, lambda_, c, rho, gamma)
    return vt.view(T, B, 1), has_played.to(torch.int64), q


class _LossV(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, v_targets, masks):
        vv = _f32(v)
        dv = torch.empty_like(vv)
        loss = torch.zeros((1,), dtype=torch.float64, device=v.device)
        for k, (vt, m) in enumerate(zip(v_targets, masks)):
            mf = _f32(m).view(-1)
            rnad_hip.loss_v(vv.view(-1), _f32(vt).view(-1), mf, rnad_hip.mask_sum(mf), 1.0, loss, dv.view(-1), k > 0)
        ctx.save_for_backward(dv)
        return loss[0].to(torch.float32)

    @staticmethod
    def backward(ctx, grad):
        (dv,) = ctx.saved_tensors
        return grad * dv, None, None


def get_loss_v(v_list: Sequence[torch.Tensor], v_target_list: Sequence[torch.Tensor], mask_list: Sequence[torch.Tensor]) -> torch.Tensor:
    """Critic loss sum_k sum(mask_k (v_k - target_k)^2) / max(sum(mask_k), 1).  reference learn/vtrace.py:377-393.
    Differentiable w.r.t. every entry of `v_list` (closed-form gradient).  At the reference's call site (learn/rnad.py:407) every entry
    is the same tensor: one pass of the kernel per player over it, one gradient buffer; distinct tensors (r06: the full signature) take
    one call per entry and the losses are added as the reference adds them (:393)."""
    if all(v is v_list[0] for v in v_list):
        return _LossV.apply(v_list[0], list(v_target_list), list(mask_list))
    return sum(_LossV.apply(v, [vt], [m]) for v, vt, m in zip(v_list, v_target_list, mask_list))


class _LossNerd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logit, pis, qs, masks, legal, clip, threshold):
        lg = _f32(logit)
        A = lg.shape[-1]
        dl = torch.empty_like(lg)
        loss = torch.zeros((1,), dtype=torch.float64, device=logit.device)
        legal_f = _f32(legal).view(-1, A)
        for k, (pi, q, m) in enumerate(zip(pis, qs, masks)):
            mf = _f32(m).view(-1)
            rnad_hip.loss_nerd(lg.view(-1, A), _f32(pi).view(-1, A), _f32(q).view(-1, A), mf, legal_f, rnad_hip.mask_sum(mf),
                               clip, threshold, 1.0, loss, dl.view(-1, A), k > 0)
        ctx.save_for_backward(dl)
        return loss[0].to(torch.float32)

    @staticmethod
    def backward(ctx, grad):
        (dl,) = ctx.saved_tensors
        return grad * dl, None, None, None, None, None, None


def get_loss_nerd(
    logit_list: Sequence[torch.Tensor],
    policy_list: Sequence[torch.Tensor],
    q_vr_list: Sequence[torch.Tensor],
    valid: torch.Tensor,
    player_ids: Sequence[torch.Tensor],
    legal_actions: torch.Tensor,
    importance_sampling_correction: Sequence[torch.Tensor],
    clip: float = 100,
    threshold: float = 2,
) -> torch.Tensor:
    """NeuRD loss.  reference learn/vtrace.py:396-431.  Differentiable w.r.t. `logit` (the force is detached there too).
    The importance weights multiply the advantage before clipping (:416); a weight per ROW (a scalar, or a last dimension of 1, as the
    reference passes: learn/rnad.py:409-410) commutes with the advantage -- adv is linear in q -- and is folded into q (`is_c * q`); a
    weight per ACTION does not commute and is refused.  At the reference's call site every entry of logit_list is the same tensor: one
    kernel pass per player over it; distinct tensors (r06: the full signature) take one call per entry, the losses added as at :431."""
    assert isinstance(importance_sampling_correction, list)
    for is_c in importance_sampling_correction:
        assert not torch.is_tensor(is_c) or is_c.dim() == 0 or is_c.shape[-1] == 1, \
            "get_loss_nerd: importance weights per action are not supported (one weight per row, shape [..., 1], or a scalar)"
    qs = [q * is_c for q, is_c in zip(q_vr_list, importance_sampling_correction)]
    masks = [valid * (player_ids == k) for k in range(len(logit_list))]
    if all(l is logit_list[0] for l in logit_list):
        return _LossNerd.apply(logit_list[0], list(policy_list), qs, masks, legal_actions, float(clip), float(threshold))
    return sum(_LossNerd.apply(lg, [pi], [q], [m], legal_actions, float(clip), float(threshold))
               for lg, pi, q, m in zip(logit_list, policy_list, qs, masks))
