"""Joins the three counter passes of `scripts/profile_collect.sh level0` (SQ, TCC write side, TCP stalls) into one
table for the level-0 kernels: python scripts/pmc_level0_table.py [dir with pmc_l0_{sq,tcc,tcp}.csv]"""
import csv
import sys


def load(f):
    rows = list(csv.reader(open(f)))
    hdr = rows[0]
    return {r[0]: dict(zip(hdr[1:], map(float, r[1:]))) for r in rows[1:]}


base = (sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof").rstrip("/") + "/"
sq, tcc, tcp = load(base + "pmc_l0_sq.csv"), load(base + "pmc_l0_tcc.csv"), load(base + "pmc_l0_tcp.csv")
tags = ("c1_s2_fwd", "conv2d_phase_kernel<BF16, 1", "conv2d_k16_kernel", "conv2d_s2k16", "c1_s2_T", "norm_fwd_kernel<BF16, true, false",
        "norm_bwd_apply")
print("per launch: wave cycles (M), share of them parked at a wait / stalled at issue / issuing, VALU wave-instructions (M), store and load "
      "wave-instructions (k), L2->fabric write requests (M), stall cycles per write request, DRAM write-credit stalls (M), TCP pending-"
      "request / TCR / TA-data stall cycles (M), TCP write tag-conflict stall cycles (M)")
print(f"{'kernel':46s} {'calls':>5s} {'wavecyc':>8s} {'wait%':>6s} {'stall%':>6s} {'issue%':>6s} {'VALU':>6s} {'VMEM_WR':>8s} {'VMEM_RD':>8s} {'WRREQ':>6s} "
      f"{'stall/req':>9s} {'cred':>6s} {'pend':>6s} {'tcr':>6s} {'tadata':>6s} {'wrtag':>6s}")
for k, s in sq.items():
    if not any(t in k for t in tags):
        continue
    c, wc = s["calls"], s["SQ_WAVE_CYCLES"]
    t, p = tcc.get(k, {}), tcp.get(k, {})
    name = k.replace("void (anonymous namespace)::", "")[:46]
    print(f"{name:46s} {int(c):5d} {wc / c / 1e6:8.2f} {100 * s['SQ_WAIT_ANY'] / wc:6.0f} {100 * s['SQ_WAIT_INST_ANY'] / wc:6.0f} "
          f"{100 * s['SQ_ACTIVE_INST_ANY'] / wc:6.0f} {s['SQ_INSTS_VALU'] / c / 1e6:6.2f} {s['SQ_INSTS_VMEM_WR'] / c / 1e3:8.0f} "
          f"{s['SQ_INSTS_VMEM_RD'] / c / 1e3:8.0f} {t.get('TCC_EA0_WRREQ_sum', 0) / c / 1e6:6.2f} "
          f"{t.get('TCC_EA0_WRREQ_STALL_sum', 0) / max(1.0, t.get('TCC_EA0_WRREQ_sum', 1)):9.2f} "
          f"{t.get('TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum', 0) / c / 1e6:6.2f} {p.get('TCP_PENDING_STALL_CYCLES_sum', 0) / c / 1e6:6.2f} "
          f"{p.get('TCP_TCR_TCP_STALL_CYCLES_sum', 0) / c / 1e6:6.2f} {p.get('TCP_TCP_TA_DATA_STALL_CYCLES_sum', 0) / c / 1e6:6.2f} "
          f"{p.get('TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum', 0) / c / 1e6:6.2f}")
