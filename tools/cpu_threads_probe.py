import os, sys, time, subprocess, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
code = """
import sys, time, torch, os
sys.path.insert(0, %r); sys.path.insert(0, %r + '/oracle')
n = int(sys.argv[1]); torch.set_num_threads(n)
import torch.nn.functional as F
x = torch.randn(4, 256, 38, 63); w = torch.randn(256, 256, 3, 3)
for _ in range(2): F.conv2d(x, w, padding=1)
t0 = time.time()
for _ in range(10): F.conv2d(x, w, padding=1)
a = (time.time() - t0) / 10
x = torch.randn(1024, 512, 7, 7); w = torch.randn(512, 512, 3, 3)
F.conv2d(x, w, padding=1)
t0 = time.time()
for _ in range(3): F.conv2d(x, w, padding=1)
b = (time.time() - t0) / 3
print(n, 'res4 3x3 ms', round(a * 1e3, 2), 'res5 3x3 ms', round(b * 1e3, 1), 'TF/s', round(2 * 1024 * 49 * 512 * 512 * 9 / b / 1e12, 2))
""" % (ROOT, ROOT)
for n in (16, 32, 64, 128, 256):
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(n))
    r = subprocess.run([sys.executable, "-c", code, str(n)], capture_output=True, text=True, env=env, timeout=300)
    print(r.stdout.strip(), r.stderr[-200:] if r.returncode else "")
