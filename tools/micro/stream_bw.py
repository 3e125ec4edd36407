#!/usr/bin/env python3
"""What streaming stores / copies achieve on this GPU (the practical ceiling K1's 0.64 of the 8 TB/s spec peak is to be read against):
torch.fill_ and torch.copy_ over buffers beyond the 256 MiB Infinity Cache, timed with events.   python tools/micro/stream_bw.py"""
import torch

dev = torch.device("cuda:0")
for mb in (256, 906, 2048):
    n = mb * (1 << 20) // 4
    a = torch.empty((n,), dtype=torch.float32, device=dev)
    b = torch.empty((n,), dtype=torch.float32, device=dev)
    for name, fn, bytes_moved in (("fill (write only)", lambda: a.fill_(1.0), 4 * n), ("copy (read + write)", lambda: b.copy_(a), 8 * n)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"{mb:5d} MB {name:20s} {us:8.1f} us  {bytes_moved / us / 1e6:6.2f} TB/s  ({bytes_moved / us / 1e6 / 8.0:.2f} of 8 TB/s)")
