#!/usr/bin/env python3
"""What this MI355X's HBM gives a plain kernel: write-only (fill), read + write (copy), read-only (sum) at 0.25-2 GB -- the ceilings the
HBM-bound rows are judged against (the guide's figure: ~6.3 TB/s achievable of 8 TB/s).  -> profiles/r05_hbm_ceiling.txt"""
import torch
dev = torch.device("cuda", 0)
for mb in (256, 688, 1216, 2048):
    n = mb * (1 << 20) // 8
    x = torch.empty(n, dtype=torch.float64, device=dev)
    y = torch.empty(n, dtype=torch.float64, device=dev)
    x.normal_()
    res = {}
    for name, fn, bytes_ in (("fill (write only)", lambda: y.zero_(), 8 * n), ("copy (read + write)", lambda: y.copy_(x), 16 * n),
                             ("sum (read only)", lambda: x.sum(), 8 * n)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        dt = e0.elapsed_time(e1) * 1e-3 / 20
        res[name] = bytes_ / dt / 1e12
    print(f"{mb:5d} MB: " + "   ".join(f"{k} {v:.2f} TB/s ({v / 8:.2f} of 8)" for k, v in res.items()))
