#!/usr/bin/env python3
"""What HBM write rate does this chip sustain?  (GPU box only.)  The decoder's fc1 + GELU GEMM writes 308 MB (h1 and gelu(h1)) and reads
39 MB in ~91 us = 3.8 TB/s; the question (round-4 review item 2) is whether its store pattern or the memory system sets that.
Pure fills and copies of the same byte counts through torch's own elementwise kernels (16 B per lane, grid-stride) give the ceilings."""
import torch
dev = torch.device("cuda:0")
def t(f, it=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
for mb in (77, 154, 308, 616, 1232):
    n = mb * 1000 * 1000 // 2
    a = torch.empty(n, dtype=torch.bfloat16, device=dev); b = torch.empty_like(a); c = torch.empty(n // 8, dtype=torch.bfloat16, device=dev)
    us_fill = t(lambda: a.fill_(1.0))
    us_copy = t(lambda: b.copy_(a))
    us_two = t(lambda: (a.fill_(1.0), b.fill_(2.0)))
    print(f"{mb:5d} MB: fill {us_fill:7.1f} us = {mb / us_fill:5.2f} TB/s written | copy {us_copy:7.1f} us = {2 * mb / us_copy:5.2f} TB/s (read + write) | two fills {us_two:7.1f} us = {2 * mb / us_two:5.2f} TB/s written")
