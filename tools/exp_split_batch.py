"""Experiment: res4 forward chain (23 bottleneck blocks) on 4 images in one stream vs 2+2 images on two streams."""
import sys, torch
sys.path.insert(0, ".")
from unit_amd import ops as o

dev = torch.device("cuda:0")
def make(n):
    x = torch.randn(n, 38, 63, 1024, device=dev).bfloat16()
    w1 = (torch.randn(256, 1, 1, 1024, device=dev) * 0.03).bfloat16()
    w2 = (torch.randn(256, 3, 3, 256, device=dev) * 0.03).bfloat16()
    w3 = (torch.randn(1024, 1, 1, 256, device=dev) * 0.03).bfloat16()
    return x, w1, w2, w3
def chain(x, w1, w2, w3, blocks=23):
    for _ in range(blocks):
        a = o.conv2d(x, w1, 256, 1, 1, 1, 0, relu=True)
        b = o.conv2d(a, w2, 256, 3, 3, 1, 1, relu=True)
        x = o.conv2d(b, w3, 1024, 1, 1, 1, 0, residual=x, relu=True)
    return x
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); 
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
x4 = make(4); xa = make(2); xb = make(2)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def one(): chain(*x4)
def two():
    main = torch.cuda.current_stream()
    s1.wait_stream(main); s2.wait_stream(main)
    with torch.cuda.stream(s1): chain(*xa)
    with torch.cuda.stream(s2): chain(*xb)
    main.wait_stream(s1); main.wait_stream(s2)
print("one stream, 4 images: %.3f ms" % timeit(one))
print("two streams, 2+2 images: %.3f ms" % timeit(two))
def half(): chain(*xa)
print("one stream, 2 images: %.3f ms" % timeit(half))
