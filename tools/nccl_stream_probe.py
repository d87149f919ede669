"""On which HIP stream does a torch.distributed collective of the nccl (= RCCL) backend run: its own internal stream, or the caller's current stream when
async_op=False?  Run under `rocprofv3 --kernel-trace` at world size 1 (an all_gather_into_tensor is then one device copy) and read the Stream_Id of the copies
next to the marker kernels:  rocprofv3 --kernel-trace -d out -o p --output-format csv -- python3 tools/nccl_stream_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from unit_amd import ops as o
from unit_amd._lib import lib, check
import ctypes


def spin(stream):
    check(lib().unit_debug_spin(20000, ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(stream.cuda_stream)), "debug_spin")

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.ones(1 << 22, device=dev); y = torch.empty(1 << 22, device=dev)
sink = torch.zeros(4, dtype=torch.int32, device=dev)
s = torch.cuda.Stream()
torch.cuda.synchronize()
with torch.cuda.stream(s):
    spin(s)                                                               # marker: the caller's stream
    dist.all_gather_into_tensor(y, x)                                     # sync form
    spin(s)
torch.cuda.synchronize()
z = torch.empty(1 << 21, device=dev)
with torch.cuda.stream(s):
    w = dist.all_gather_into_tensor(z, x[: 1 << 21], async_op=True)       # async form (half the size: tell the two copies apart)
    w.wait()
torch.cuda.synchronize()
dist.destroy_process_group()
