"""Which HIP streams of this process share a hardware queue?  python tools/queue_probe.py [n_streams] [--nccl]
Two single-wave spin kernels (unit_debug_spin, ~300 us) on two streams take one spin time when the streams sit on different hardware queues
and two when they share one. Prints the matrix for the null stream and the first n torch.cuda.Stream() objects (PyTorch hands out streams of a
32-entry pool in creation order; the ROCm runtime maps all streams of the process onto GPU_MAX_HW_QUEUES = 4 queues)."""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from unit_amd._lib import check, lib

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8
if "--nccl" in sys.argv:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29633")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    t = torch.zeros(1024, device="cuda")
    dist.all_reduce(t)
dev = torch.device("cuda:0")
sink = torch.zeros(4, dtype=torch.int32, device=dev)
streams = [torch.cuda.default_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(n)]
names = ["null"] + [f"s{i}" for i in range(n)]
CYC = 600000          # ~300 us at 2 GHz


def spin(st):
    check(lib().unit_debug_spin(ctypes.c_longlong(CYC), ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(st.cuda_stream)), "spin")


def pair(a, b):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    spin(a)
    spin(b)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6


for st in streams:
    spin(st)
torch.cuda.synchronize()
one = min(pair(streams[1], streams[1]) for _ in range(3)) / 2
print("stream ids:", [(nm, st.stream_id) for nm, st in zip(names, streams)])
print(f"one spin: {one:.0f} us; entries = time of a pair / one spin (1 = different queues, 2 = same queue)")
print("      " + " ".join(f"{nm:>5s}" for nm in names))
for i, a in enumerate(streams):
    row = []
    for j, b in enumerate(streams):
        row.append("    -" if i == j else f"{min(pair(a, b) for _ in range(2)) / one:5.1f}")
    print(f"{names[i]:>5s} " + " ".join(row))
