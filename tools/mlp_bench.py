#!/usr/bin/env python3
"""Micro-benchmark of the fused MLP kernels alone (N samples, A=3, width 256): ms and fp32 TFLOP/s on the matrix pipe."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "r-nad_amd"))
import torch  # noqa: E402

import rnad_hip  # noqa: E402

A = int(os.environ.get("MLP_BENCH_A", 3))
W = int(os.environ.get("MLP_BENCH_W", 256))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12 * 2**20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
torch.manual_seed(0)
K = 2 * A * A
shapes = [(W, K), (W,), (1, W), (1,), (W, K), (W,), (A, W), (A,)]
w = [torch.randn(s, device=dev) / s[-1] ** 0.5 for s in shapes]
x = torch.randn((N, 2, A, A), device=dev)
dl = torch.randn((N, A), device=dev)
dv = torch.randn((N, 1), device=dev)


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


fwd_flop = 2.0 * N * K * 2 * W  # executed MFMA flops of the first layer, both heads
_rem = (K + 1) % 16
_feat = 16 * ((K + 1) // 16 + (1 if _rem > 4 else 0)) + (4 if 0 < _rem <= 4 else 0)
bwd_flop = fwd_flop + 2.0 * N * _feat * 2 * W  # recompute + dW0 over the tile-padded augmented input
packed = rnad_hip.mlp_pack(w, A)
ms = timeit(lambda: rnad_hip.mlp_forward(packed, W, x, A))
print(f"forward  both heads: {ms:7.3f} ms  {fwd_flop / ms / 1e9:6.1f} TFLOP/s (MFMA-executed)")
ms = timeit(lambda: rnad_hip.mlp_forward(packed, W, x, A, want_value=False))
print(f"forward  policy only: {ms:7.3f} ms  {fwd_flop / 2 / ms / 1e9:6.1f} TFLOP/s")
ms = timeit(lambda: rnad_hip.mlp_backward(packed, w, x, A, dl, dv))
print(f"backward           : {ms:7.3f} ms  {bwd_flop / ms / 1e9:6.1f} TFLOP/s")

# the same kernels through a live-row list (ragged trajectories): every row listed, then a random 43 %
for frac in (1.0, 0.43):
    flags = (torch.rand((N,), device=dev) < frac).to(torch.int32)
    live = rnad_hip.compact_valid(flags)
    n = int(live.count.item())
    ms = timeit(lambda: rnad_hip.mlp_forward(packed, W, x, A, live=live))
    print(f"rows {frac:4.2f} forward both heads: {ms:7.3f} ms  {fwd_flop * n / N / ms / 1e9:6.1f} TFLOP/s on the listed rows")
    ms = timeit(lambda: rnad_hip.mlp_backward(packed, w, x, A, dl, dv, live=live))
    print(f"rows {frac:4.2f} backward          : {ms:7.3f} ms  {bwd_flop * n / N / ms / 1e9:6.1f} TFLOP/s on the listed rows")
ms = timeit(lambda: rnad_hip.compact_valid(flags))
print(f"compact_valid of {N} positions: {ms * 1e3:7.1f} us")
