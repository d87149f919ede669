#!/bin/bash
# diagnostic: where do the cycles of the 512 -> 2048 (+ residual) Res5 launch go?  `tools/epi_stamp.sh build` here; `tools/epi_stamp.sh` on the GPU box
if [ "$1" = build ]; then
  python3 -c "import __graft_entry__ as g; g.build()"
  mkdir -p unit_amd/_build/epistamp
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -ffp-contract=off -std=c++17 -Wno-unused-value -DUNIT_EPI_STAMP=1 -c unit_amd/csrc/conv_igemm256p8.hip -o unit_amd/_build/epistamp/p8.o || exit 1
  objs=$(ls unit_amd/_build/*.o | grep -v conv_igemm256p8.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o unit_amd/_build/epistamp/libunit_hip.so $objs unit_amd/_build/epistamp/p8.o || exit 1
  exit 0
fi
export UNIT_HIP_LIB=$PWD/unit_amd/_build/epistamp/libunit_hip.so
python3 - <<'PY'
import ctypes, sys, torch
sys.path.insert(0, ".")
from unit_amd import ops as o, _lib
L = _lib.lib()
L.unit_debug_read_stamps.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * (32 * 16))()
def run(name, fn, reps=6):
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    L.unit_debug_read_stamps(buf)
    rows = [list(buf[i * 16:(i + 1) * 16]) for i in range(32)]
    rows = [r for r in rows if r[0] and r[13] > r[0]]
    print(name, "workgroups stamped:", len(rows))
    print("   wg    fill  mainloop  barrier  pre-blk " + " ".join(f"blk{b}" for b in range(8)) + "   epilogue   total  (cycles of s_memtime)")
    for r in sorted(rows, key=lambda r: r[14]):
        d = [r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3]] + [r[5 + b] - r[4 + b] for b in range(8)]
        print(f"{int(r[14]):5d} " + " ".join(f"{int(v):7d}" for v in d) + f"  {int(r[13] - r[3]):8d} {int(r[13] - r[0]):8d}")
dev = "cuda"
x = torch.randn(1024, 7, 7, 512, device=dev).bfloat16(); w = (torch.randn(2048, 1, 1, 512, device=dev) * 0.05).bfloat16()
res = torch.randn(1024, 7, 7, 2048, device=dev).bfloat16()
run("512->2048 +res +relu", lambda: o.conv2d(x, w, 2048, 1, 1, 1, 0, residual=res, relu=True, tile_cfg=16))
run("512->2048 plain", lambda: o.conv2d(x, w, 2048, 1, 1, 1, 0, relu=True, tile_cfg=16))
b = torch.randn(2048, device=dev)
run("512->2048 +bias +relu (straight-line passes, round 5)", lambda: o.conv2d(x, w, 2048, 1, 1, 1, 0, bias=b, relu=True, tile_cfg=16))
x3 = torch.randn(1024, 7, 7, 512, device=dev).bfloat16(); w3 = (torch.randn(512, 3, 3, 512, device=dev) * 0.05).bfloat16()
run("3x3 512->512 row-major tiles", lambda: o.conv2d(x3, w3, 512, 3, 3, 1, 1, relu=True, tile_cfg=22))
PY
