"""How often does the two-ranks-one-GPU `graph_whole` rehearsal die (SIGSEGV inside hipStreamEndCapture), with and
without deferred code-object loading?  python tools/dp_flake_probe.py <runs> [MODE=graph_whole|graph|replay|eager] [ENV=VALUE ...]"""
import os, socket, subprocess, sys, tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
W = os.path.join(os.path.dirname(HERE), "tests", "dp_rehearsal_worker.py")
runs = int(sys.argv[1])
env = dict(os.environ, PYTHONFAULTHANDLER="1")
mode = "graph_whole"
for kv in sys.argv[2:]:
    k, v = kv.split("=", 1)
    if k == "MODE":
        mode = v
    else:
        env[k] = v
bad = 0
for i in range(runs):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = str(s.getsockname()[1]); s.close()
    with tempfile.TemporaryDirectory() as d:
        ps = [subprocess.Popen([sys.executable, W, str(r), "2", port, os.path.join(d, f"r{r}.pt"), mode, "5"], stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, env=env) for r in range(2)]
        logs = [p.communicate(timeout=600)[0].decode(errors="replace") for p in ps]
    rc = [p.returncode for p in ps]
    if any(c != 0 for c in rc):
        bad += 1
        for r, lg in enumerate(logs):
            if ps[r].returncode is not None and ps[r].returncode < 0:
                print(f"run {i} rank {r} rc {ps[r].returncode}\n" + "\n".join(lg.splitlines()[-14:]), flush=True)
    print("run", i, rc, flush=True)
print("failed runs:", bad, "of", runs, sys.argv[2:])
