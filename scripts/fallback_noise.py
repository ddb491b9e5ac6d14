"""Prints how far one bf16 training step moves under each kernel-family switch (tests/test_fallbacks_gpu.py's children):
worst relative L2 difference of a gradient tensor and worst absolute difference relative to the largest gradient."""
import os, subprocess, sys, tempfile
import torch
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(HERE, "tests"))
from helpers import rel_l2
from test_fallbacks_gpu import SWITCHES

def run(switch, d):
    out = os.path.join(d, (switch or "default") + ".pt")
    env = {k: v for k, v in os.environ.items() if not k.startswith("MI_NO_")}
    if switch: env[switch] = "1"
    subprocess.run([sys.executable, os.path.join(HERE, "tests", "fallback_child.py"), out], env=env, check=True)
    return torch.load(out, weights_only=True)

with tempfile.TemporaryDirectory() as d:
    base = run(None, d)
    again = run(None, d)
    scale = max(float(g.abs().max()) for g in base["grads"].values())
    for name, got in [("(default again)", again)] + [(s, run(s, d)) for s in SWITCHES]:
        rows = [(rel_l2(got["grads"][k], g), float((got["grads"][k] - g).abs().max()) / scale, k) for k, g in base["grads"].items()]
        big = [r for r in rows if float(base["grads"][r[2]].abs().max()) > 1e-2 * scale]
        w = max(big)
        print(f"{name:22s} logits {rel_l2(got['logits'], base['logits']):.2e}  worst rel-l2 {w[0]:.3f} ({w[2]})  worst abs/scale {max(r[1] for r in rows):.2e}", flush=True)
