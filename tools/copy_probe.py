#!/usr/bin/env python3
"""device -> pinned host copy of a 101 MB map: DMA engine vs the copy kernel (bfg_copy_to_mapped_host) at several grid sizes"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from baryonforge_amd.engine import get_context
ctx = get_context()
n = 12 * 1024 * 1024
d = torch.rand(n, dtype=torch.float64, device=ctx.device)
h = torch.empty(n, dtype=torch.float64, pin_memory=True)
def t(fn):
    fn(); torch.cuda.synchronize()
    b = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t0)
    return b * 1e3
print(f"DMA copy: {t(lambda: h.copy_(d, non_blocking=True)):.2f} ms")
for plain in ("", "1"):
    for g in (32, 128, 512, 2048):
        os.environ["BFG_COPY_GRID"] = str(g)
        if plain: os.environ["BFG_COPY_PLAIN"] = "1"
        else: os.environ.pop("BFG_COPY_PLAIN", None)
        h.zero_()
        ms = t(lambda: ctx.copy_to_pinned(h, d))
        assert torch.equal(h, d.cpu())
        print(f"copy kernel, grid {g:5d}, {'plain' if plain else 'nontemporal'} stores: {ms:.2f} ms = {n * 8 / ms / 1e6:.1f} GB/s")
# both directions at once: H2D by DMA on one stream, D2H by DMA / by the copy kernel on another
os.environ.pop("BFG_COPY_PLAIN", None); os.environ["BFG_COPY_GRID"] = "32"
src = torch.empty(n, dtype=torch.float64, pin_memory=True); src.copy_(torch.rand(n, dtype=torch.float64))
d2 = torch.empty(n, dtype=torch.float64, device=ctx.device)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both(kernel):
    with torch.cuda.stream(s1):
        d2.copy_(src, non_blocking=True)
    with torch.cuda.stream(s2):
        if kernel: ctx.copy_to_pinned(h, d)
        else: h.copy_(d, non_blocking=True)
print(f"H2D (DMA) alone: {t(lambda: d2.copy_(src, non_blocking=True)):.2f} ms")
print(f"H2D (DMA) + D2H (DMA) on two streams: {t(lambda: both(False)):.2f} ms")
print(f"H2D (DMA) + D2H (copy kernel) on two streams: {t(lambda: both(True)):.2f} ms")
big = torch.rand(1 << 27, dtype=torch.float64, device=ctx.device)
def with_compute(kernel):
    both(kernel)
    for _ in range(4): big.mul_(1.0000001)           # a bandwidth-bound kernel on the default stream beside the copies
print(f"... + four 1 GiB element-wise kernels: DMA/DMA {t(lambda: with_compute(False)):.2f} ms, DMA/kernel {t(lambda: with_compute(True)):.2f} ms, compute alone {t(lambda: [big.mul_(1.0000001) for _ in range(4)]):.2f} ms")
