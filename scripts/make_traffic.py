"""Builds profiles/<tag>_hbm_traffic.json from two rocprofv3 counter passes over bench.py
(--pmc FETCH_SIZE and --pmc WRITE_SIZE, csv output).  Units follow the MI355X notes: both counters are in
KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide streaming reads and is doubled here.
usage: python scripts/make_traffic.py <fetch csv> <write csv> <tag>"""
import collections
import csv
import json
import re
import sys


def label(name: str) -> str:
    """bench.py's step label of a kernel symbol."""
    m = re.search(r"(\w+)<(BF16|F32)((?:, [\w-]+)*)>", name)
    base = re.sub(r"^void |\(anonymous namespace\)::", "", name).split("(")[0]
    if not m:
        k16 = re.match(r"(conv2d_s?2?k16_kernel|wgrad_thick_s2_kernel|wgrad_thin_kernel)<(\d+)(?:, \d+)*>", base)
        if k16:
            return f"{k16.group(1)}<{k16.group(2)}>"
        base = re.sub(r"<.*$", "", base)
        if base in ("norm1_bwd_reduce_kernel", "norm1_bwd_apply_kernel"):
            return "norm1_bwd_kernels"
        return base
    kern, tr, rest = m.group(1), m.group(2), [a.strip() for a in m.group(3).split(",") if a.strip()]
    if kern == "c1_s2_fwd32_kernel":         # the strip form runs under the step label of the kernel it stands in for
        return f"c1_s2_fwd_kernel<{tr}>"
    if kern in ("norm_bwd_reduce_kernel", "norm_bwd_apply_kernel"):
        return f"norm_bwd_kernels<{tr}>"
    if kern in ("igemm_kernel", "igemm2d_kernel"):
        return f"igemm_kernel<{tr},{rest[0]},{rest[1]}>"
    if kern == "wgrad_kernel":
        return f"wgrad_kernel<{tr},{','.join(rest)}>"
    if kern == "wgrad_dma_kernel":           # same step label as the register-staged kernel it stands in for
        return f"wgrad_kernel<{tr},{','.join(rest[:5])}>"
    if kern == "conv2d_s2_kernel" and len(rest) > 1 and rest[1] == "false":    # tap tiles streamed with the patches
        return f"{kern}<{tr},{rest[0]},streamed>"
    if kern in ("conv2d_dma_kernel", "conv2d_phase_kernel", "conv2d_s2_kernel"):
        return f"{kern}<{tr},{rest[0]}>"
    return f"{kern}<{tr}>"


def load(path):
    d = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        k = label(r["Kernel_Name"])
        d[k][0] += float(r["Counter_Value"]); d[k][1] += 1
    return d


def by_kernel(path, counter, dst):
    """per-kernel-symbol sums of one counter pass, as committed under profiles/"""
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"]][0] += 1; d[r["Kernel_Name"]][1] += float(r["Counter_Value"])
    with open(dst, "w") as f:
        f.write(f"Kernel_Name,Dispatches,{counter}_sum_KiB\n")
        for k, (n, v) in sorted(d.items(), key=lambda kv: -kv[1][1]):
            f.write('"%s",%d,%.1f\n' % (k, n, v))


def main():
    fetch, write, tag = load(sys.argv[1]), load(sys.argv[2]), sys.argv[3]
    by_kernel(sys.argv[1], "FETCH_SIZE", f"profiles/{tag}_pmc_fetch_size_by_kernel.csv")
    by_kernel(sys.argv[2], "WRITE_SIZE", f"profiles/{tag}_pmc_write_size_by_kernel.csv")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, [0.0, 0]); w, nw = write.get(k, [0.0, 0])
        n = max(nf, nw)
        if n == 0 or k.startswith("at::") or "Cijk" in k:
            continue
        pair = 2 if k.startswith(("norm_bwd_kernels", "norm1_bwd_kernels")) else 1      # two kernels make one bench step
        launches = n / pair
        out[k] = {"bytes_per_launch": round((2.0 * f + w) * 1024.0 / launches), "read_bytes_per_launch": round(2.0 * f * 1024.0 / launches),
                  "write_bytes_per_launch": round(w * 1024.0 / launches), "launches_profiled": launches}
    json.dump({"note": "HBM-side bytes per launch from rocprofv3 FETCH_SIZE (x2, gfx950) and WRITE_SIZE; separate passes of "
                       "bench.py --steps 2 --warmup 1", "kernels": out}, open(f"profiles/{tag}_hbm_traffic.json", "w"), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["bytes_per_launch"] * kv[1]["launches_profiled"])[:12]:
        print(f"{k:40s} {v['bytes_per_launch'] / 1e6:9.1f} MB/launch x {v['launches_profiled']:.0f}")


if __name__ == "__main__":
    main()
