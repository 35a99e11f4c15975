"""Same-box A/B of an environment switch on the default bench config: tools/ab_env.py NAME VAL_A VAL_B [rounds].
Runs bench.py alternately (A, B, A, B, ...) in child processes and prints the sample-steps/s of every run (tools only)."""
import json
import os
import subprocess
import sys

name, va, vb = sys.argv[1:4]
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = {va: [], vb: []}
for _ in range(rounds):
    for v in (va, vb):
        env = dict(os.environ)
        env[name] = v
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "512", "--warmup", "128", "--no-cpu-baseline",
                              "--no-end-to-end"], env=env, capture_output=True, text=True).stdout
        line = [l for l in out.splitlines() if l.startswith("{")][-1]
        res[v].append(round(json.loads(line)["value"], 1))
for v in (va, vb):
    print(f"{name}={v}: {res[v]}  mean {sum(res[v]) / len(res[v]):.1f}")
