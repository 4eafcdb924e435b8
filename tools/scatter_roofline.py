#!/usr/bin/env python3
"""What the memory system gives a random 4-byte scatter: the un-permute of k_read_map_part is
packed[r] = value for 10 M reads r in partition order.  Measures (a) a plain permuted scatter of 10 M
int32 values (torch index_put_ with a random permutation: nothing but the scattered stores and two
coalesced loads), (b) the same with 8-byte values, (c) a coalesced copy, so that the kernel's time can
be read against the hardware's rate for this access pattern rather than against streaming HBM."""
import sys
import time

import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1)
perm = torch.randperm(n, generator=g).to(dev)
vals32 = torch.arange(n, dtype=torch.int32, device=dev)
vals64 = torch.arange(n, dtype=torch.int64, device=dev)
out32 = torch.zeros(n, dtype=torch.int32, device=dev)
out64 = torch.zeros(n, dtype=torch.int64, device=dev)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


flush = torch.zeros(1 << 28, dtype=torch.int32, device=dev)           # 1 GiB: larger than L2 + Infinity Cache


def timed_cold(fn, reps=10):
    """each repetition after the caches were swept by a 1 GiB fill (the pipeline's situation: the
    destination lines are not resident when the un-permute starts)"""
    tot = 0.0
    for _ in range(reps + 2):
        flush.add_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        if _ >= 2:
            tot += e0.elapsed_time(e1)
    return tot / reps


t_s32 = timed(lambda: out32.index_copy_(0, perm, vals32))
t_s32c = timed_cold(lambda: out32.index_copy_(0, perm, vals32))
t_g32c = timed_cold(lambda: torch.index_select(vals32, 0, perm, out=out32))
t_c32c = timed_cold(lambda: out32.copy_(vals32))
t_s64 = timed(lambda: out64.index_copy_(0, perm, vals64))
t_g32 = timed(lambda: torch.index_select(vals32, 0, perm, out=out32))
t_c32 = timed(lambda: out32.copy_(vals32))
print("random scatter, 4-byte values (index + value loads coalesced): %.3f ms = %.1f G stores/s" % (t_s32, n / t_s32 / 1e6))
print("  ... caches swept before each repetition:                      %.3f ms = %.1f G stores/s" % (t_s32c, n / t_s32c / 1e6))
print("random scatter, 8-byte values:                                 %.3f ms = %.1f G stores/s" % (t_s64, n / t_s64 / 1e6))
print("random gather,  4-byte values:                                 %.3f ms = %.1f G loads/s" % (t_g32, n / t_g32 / 1e6))
print("  ... gather, caches swept:                                     %.3f ms = %.1f G loads/s" % (t_g32c, n / t_g32c / 1e6))
print("  ... coalesced copy, caches swept:                             %.3f ms = %.0f GB/s" % (t_c32c, 8 * n / t_c32c / 1e6))
print("coalesced copy, 4-byte values:                                 %.3f ms = %.0f GB/s" % (t_c32, 8 * n / t_c32 / 1e6))
