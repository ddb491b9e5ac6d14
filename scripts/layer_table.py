"""Markdown per-layer table from the per-launch lines bench.py prints with BENCH_LAYERS=1:
python scripts/layer_table.py bench.err > table.md   (columns: pass, layer, kernel, us, TFLOP/s, TB/s, bound)"""
import re
import sys

RIDGE = 2.5e15 / 8.0e12
rows = []
for line in open(sys.argv[1]):
    m = re.match(r"\s+(fwd|bwd) (.+?)\s+(\S+)\s+([\d.]+) us\s+([\d.]+) TF/s\s+([\d.]+) GB/s", line)
    if not m:
        continue
    ps, layer, kern, us, tf, gb = m.group(1), m.group(2).strip(), m.group(3), float(m.group(4)), float(m.group(5)), float(m.group(6))
    rows.append((ps, layer, kern, us, tf, gb))
print("| pass | layer | kernel | µs | TFLOP/s | TB/s (algorithmic bytes) | bound |")
print("|---|---|---|---|---|---|---|")
tot = {"fwd": 0.0, "bwd": 0.0}
for ps, layer, kern, us, tf, gb in rows:
    ai = (tf * 1e12) / (gb * 1e9) if gb > 0 else 0.0
    bound = "MFMA" if tf > 0 and ai > RIDGE else ("HBM" if gb > 0 else "launch")
    side = " (side stream)" if ps == "bwd" and ("wgrad" in layer or "reduce gradients" in layer) else (" (side stream)" if "pack weights" in layer else "")
    tot[ps] += us
    print(f"| {ps} | {layer}{side} | `{kern}` | {us:.1f} | {tf:.0f} | {gb / 1e3:.2f} | {bound} |" if tf > 0 else
          f"| {ps} | {layer}{side} | `{kern}` | {us:.1f} | – | {gb / 1e3:.2f} | {bound} |")
print(f"\nSum of the event-bracketed launch times: forward {tot['fwd'] / 1e3:.3f} ms, backward {tot['bwd'] / 1e3:.3f} ms (one stream; in production the side-stream rows overlap the rest).")
