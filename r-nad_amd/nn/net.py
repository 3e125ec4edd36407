"""MLP policy/value net -- drop-in for reference nn/net.py:18-85 (interface + parameter names kept).

The net is two parallel 2-layer perceptrons (10 756 parameters at A = 3, width 256).  north_star kept it in PyTorch-ROCm "unless
rocprof shows the GEMM is large enough to be a dense contraction"; round 1's profile did (the hidden activations of 12.6 M samples
made the torch MLP 99.5 % of a step), so `forward_logits` runs ONE fused fp32-MFMA HIP kernel that keeps the hidden layer in
registers (rnad_mlp_forward), with rnad_mlp_backward as its autograd backward.  The masked exp-normalise policy head (net.py:45-46,
:74-77) and the multinomial sampler (net.py:49) are HIP kernels too.  Shapes the fused kernels do not cover (width not a multiple of
32, non-fp32 weights, a weight image beyond the 160 KiB LDS) and CPU tensors fall back to four torch Linear calls.
`forward_batch` evaluates the net once over the flattened `[T*B, 2A^2]` trajectory -- or, on a tree that is small next to the
batch, over the tree's 2S distinct observations -- instead of a Python loop over t (net.py:67).

State-dict keys (`value_fc0.weight`, ...) are the reference's, so its checkpoints load unchanged.
"""
import torch
import torch.nn as nn

import rnad_hip


class MLP(nn.Module):
    def __init__(self, max_actions, width, device=torch.device("cpu:0"), dtype=torch.float):
        super().__init__()
        self.device = device
        self.value_fc0 = nn.Linear(2 * max_actions**2, width, device=device, dtype=dtype)
        self.value_fc1 = nn.Linear(width, 1, device=device, dtype=dtype)
        self.policy_fc0 = nn.Linear(2 * max_actions**2, width, device=device, dtype=dtype)
        self.policy_fc1 = nn.Linear(width, max_actions, device=device, dtype=dtype)
        self.max_actions = max_actions
        self.width = width
        self._seed = int(torch.randint(0, 2**62, (1,)).item())  # sampler stream of forward(); torch.manual_seed controls it
        self._calls = 0

    # ---------------------------------------------------------------- the two perceptrons
    def _weights(self):
        return [self.value_fc0.weight, self.value_fc0.bias, self.value_fc1.weight, self.value_fc1.bias,
                self.policy_fc0.weight, self.policy_fc0.bias, self.policy_fc1.weight, self.policy_fc1.bias]

    def pack(self):
        """The weight image the fused kernels read (rnad_mlp_pack).  Pack once per net and weight version and hand it to
        every `forward_logits(..., packed=...)` of a rollout; it is NOT cached here (in-place optimizers such as fused
        Adam do not bump tensor versions, so a cache could go stale silently)."""
        if not self._fusable():
            return None
        return rnad_hip.mlp_pack(self._weights(), self.max_actions)

    def _fusable(self):
        """The fused kernels keep ALL weights in the 160 KiB LDS of a CU and tile the hidden layer by 32."""
        w = self.value_fc0.weight
        A, W = self.max_actions, self.width
        image_floats = (2 * A * A + 1) * 2 * W + (1 + A) * W + 12
        return w.is_cuda and w.dtype == torch.float32 and W % 32 == 0 and image_floats * 4 <= 160 * 1024

    def forward_logits(self, input_batch, want_logits=True, want_value=True, packed=None, live=None):
        """obs [N, 2, A, A] (fp32 or fp16) -> logits [N, A], value [N, 1]   (net.py:40-43).

        ONE fused HIP kernel that keeps the hidden layer in registers (rnad_mlp_forward); under autograd it is an
        autograd node whose backward is rnad_mlp_backward (hidden layer recomputed on chip).  Shapes the kernels do not
        cover (width not a multiple of 32, non-fp32 weights, a weight image larger than the LDS) use four PyTorch-ROCm Linear calls.

        live: an rnad_hip.LiveRows over the N samples (ragged trajectories) -- the fused kernels then evaluate those rows only
        and return zeros elsewhere; the PyTorch fallback ignores it and evaluates everything."""
        A = self.max_actions
        if input_batch.is_cuda and self._fusable():
            if not torch.is_grad_enabled():
                return rnad_hip.mlp_forward(packed if packed is not None else self.pack(), self.width, input_batch.contiguous(), A, want_logits,
                                            want_value, live=live)
            if rnad_hip.mlp_backward_supported(A, self.width) and not input_batch.requires_grad:
                packed = packed if packed is not None else self.pack()
                if live is not None:
                    return rnad_hip.FusedMLPRows.apply(input_batch.contiguous(), A, packed, live, *self._weights())
                return rnad_hip.FusedMLP.apply(input_batch.contiguous(), A, packed, *self._weights())
        x = input_batch.reshape(-1, 2 * self.max_actions**2)
        if x.dtype != self.value_fc0.weight.dtype:
            x = x.to(self.value_fc0.weight.dtype)
        value = self.value_fc1(torch.relu(self.value_fc0(x))) if want_value else None
        logits = self.policy_fc1(torch.relu(self.policy_fc0(x))) if want_logits else None
        return logits, value

    @staticmethod
    def _mask(input_batch):
        return input_batch[:, 1, :, 0].to(torch.float).contiguous()  # filter_row (net.py:38)

    # ---------------------------------------------------------------- net.py:37-51
    def forward(self, input_batch):
        logits, value = self.forward_logits(input_batch)
        policy = rnad_hip.policy_head(logits.detach().contiguous(), mask=self._mask(input_batch))
        actions = rnad_hip.sample(policy, seed=self._seed, step=self._calls & 0xFFFFFF, stream_id=2).long()
        self._calls += 1
        return logits, policy, value, actions

    # ---------------------------------------------------------------- net.py:53-62
    def forward_policy(self, input_batch: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            logits, _ = self.forward_logits(input_batch, want_value=False)
        return rnad_hip.policy_head(logits.contiguous(), mask=self._mask(input_batch))

    # ---------------------------------------------------------------- net.py:64-85
    def forward_batch(self, episodes):
        """-> [logit, log_policy, policy, value], shapes [T,B,A] x3 and [T,B,1].  `logit` and `value` carry autograd;
        policy / log_policy come out of the HIP policy head and are constants (the reference's loss never
        differentiates through them: learn/vtrace.py:418, learn/rnad.py:377-382)."""
        T, B = episodes.t_eff + 1, episodes.batch_size
        A = self.max_actions
        logits = value = None
        tree = getattr(episodes, "tree", None)
        if (tree is not None and self._fusable() and rnad_hip.mlp_backward_supported(A, self.width)
                and episodes.indices.dtype == torch.int32 and episodes.indices.is_cuda and B <= 2**21):
            # the observations of a trajectory are rows of the tree's observation table (one per player and state): on a tree that
            # is small next to the batch the net is evaluated on those rows and each slot gathers its own.  Values are the per-slot
            # evaluation's bit for bit; under autograd the weight gradients are summed per row first (rnad_row_sums), i.e. they
            # equal the per-slot ones up to fp32 summation order.  An absorbed slot (index 0) gets the row of state 0, whose
            # observation differs from the zero padding of a collated batch -- those slots are masked by every consumer.
            handle = tree.handle()
            if 8 * handle.S <= T * B:
                table = handle.observations_table(getattr(episodes, "obs_half", False))
                idx = episodes.indices[:T].contiguous()
                if torch.is_grad_enabled():
                    logits, value = rnad_hip.TabularMLP.apply(table, idx, handle, A, self.pack(), *self._weights())
                else:
                    lt, vt = rnad_hip.mlp_forward(self.pack(), self.width, table, A)
                    rows = (idx.long() + (torch.arange(T, device=idx.device) & 1).view(T, 1) * handle.S).reshape(-1)
                    logits, value = lt.index_select(0, rows), vt.index_select(0, rows)
        if logits is None:
            logits, value = self.forward_logits(episodes.observations[:T])
        mask_bits = getattr(episodes, "mask_bits", None)
        if mask_bits is not None:
            policy, log_policy = rnad_hip.policy_head(logits.detach(), mask_bits=mask_bits[:T].reshape(-1), want_log=True)
        else:
            policy, log_policy = rnad_hip.policy_head(logits.detach(), mask=episodes.masks[:T].reshape(-1, A).contiguous(), want_log=True)
        return [logits.view(T, B, A), log_policy.view(T, B, A), policy.view(T, B, A), value.view(T, B, 1)]
