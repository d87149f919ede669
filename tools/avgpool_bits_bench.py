"""unit_avgpool_bwd_bits at the Res5 size (1024 RoIs x 49 bins x 2048 channels: 205 MB written): python tools/avgpool_bits_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unit_amd import ops as o

dev = torch.device("cuda:0")
R, C, K = 1024, 512, 2048
x = torch.randn(R, 7, 7, C, device=dev).bfloat16()
w = (torch.randn(K, 1, 1, C, device=dev) * 0.05).bfloat16()
y, bits, pooled = o.conv2d_ex(x, w, K, 1, 1, 0, relu=True, want_bits=True, pool_rows=49, want_y=True)
df = torch.randn(R, K, device=dev).bfloat16()
want = o.global_avgpool_bwd_relu(df, y)
got = o.avgpool_bwd_bits(df, bits, 7, 7)
print("equal", torch.equal(got, want))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    o.avgpool_bwd_bits(df, bits, 7, 7)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f"avgpool_bwd_bits {us:.1f} us  {R * 49 * K * 2 / us / 1e6:.2f} TB/s written")
