"""What a dense bf16 GEMM reaches on this GPU (torch.matmul -> hipBLASLt / rocBLAS), on random and on all-zero operands:
the practical ceiling the wide convolution layers are compared with (the chip lowers its clock under dense MFMA load,
so this sits well below the 2.5 PFLOP/s data-sheet figure).  usage: python scripts/gemm_ceiling.py"""
import torch


def bench(m, n, k, zero):
    dev = torch.device("cuda", 0)
    a = (torch.zeros if zero else torch.randn)(m, k, device=dev, dtype=torch.bfloat16)
    b = (torch.zeros if zero else torch.randn)(k, n, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        torch.matmul(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 30
    e0.record()
    for _ in range(it):
        torch.matmul(a, b)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    return 2.0 * m * n * k / ms / 1e9


if __name__ == "__main__":
    for m, n, k in ((8192, 8192, 8192), (4096, 4096, 8192), (32768, 256, 2304), (32768, 512, 4608)):
        print(f"{m} x {n} x {k}: random {bench(m, n, k, False):7.0f} TFLOP/s, zeros {bench(m, n, k, True):7.0f} TFLOP/s")
