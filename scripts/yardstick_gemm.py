#!/usr/bin/env python3
"""Yardstick for the MFMA-bound convs: what a tuned library f16 GEMM (torch.matmul ->
hipBLASLt / rocBLAS) reaches on THIS box at the shapes of the path's matrix-bound convs, on random data, with the
clock the chip holds read beside it.  Not part of the product (the product links no BLAS): it prices the achievable
roof at the DVFS clock so that the kernels' targets are measured ones.

  python scripts/yardstick_gemm.py            -> one line per shape: TFLOP/s, share of 2.5 PFLOP/s, sclk seen
"""
import glob
import re
import threading
import time

import torch

SHAPES = [
    # (name, M, N, K)
    ("decoder conv 1056->1056 k3 (32 x 1024 frames)", 32768, 1056, 3168),
    ("decoder conv 1056->528 k3", 32768, 528, 3168),
    ("decoder conv 528->528 k3", 32768, 528, 1584),
    ("256-ch ResBlock conv k11 (rate 5)", 163840, 256, 2816),
    ("256-ch ResBlock conv k3", 163840, 256, 768),
    ("128-ch ResBlock conv k11 (rate 25)", 819200, 128, 1408),
    ("128-ch ResBlock conv k3", 819200, 128, 384),
    ("64-ch ResBlock conv k11 (rate 100)", 3276800, 64, 704),
    ("square 8192", 8192, 8192, 8192),
    ("square 4096", 4096, 4096, 4096),
]


def sclk_mhz():
    """current shader clock from sysfs (the starred line of pp_dpm_sclk), or None"""
    for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            for ln in open(f):
                if "*" in ln:
                    m = re.search(r"(\d+)\s*Mhz", ln, re.I)
                    if m:
                        return int(m.group(1))
        except OSError:
            pass
    return None


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    print("device:", torch.cuda.get_device_name(0))
    for name, M, N, K in SHAPES:
        a = (torch.rand(M, K, device=dev, dtype=torch.float32) * 2 - 1).to(torch.float16)
        b = (torch.rand(K, N, device=dev, dtype=torch.float32) * 2 - 1).to(torch.float16)
        for _ in range(3):
            c = a @ b
        torch.cuda.synchronize()
        flops = 2.0 * M * N * K
        # enough repetitions for >= 0.3 s of back-to-back launches (DVFS settles), clock sampled meanwhile
        t_one = None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        c = a @ b
        e1.record()
        torch.cuda.synchronize()
        t_one = e0.elapsed_time(e1) * 1e-3
        reps = max(10, int(0.5 / max(t_one, 1e-6)))
        clocks = []
        stop = threading.Event()

        def sample():
            while not stop.is_set():
                v = sclk_mhz()
                if v:
                    clocks.append(v)
                time.sleep(0.02)

        th = threading.Thread(target=sample)
        th.start()
        e0.record()
        for _ in range(reps):
            c = a @ b
        e1.record()
        torch.cuda.synchronize()
        stop.set()
        th.join()
        dt = e0.elapsed_time(e1) * 1e-3 / reps
        tf = flops / dt / 1e12
        clk = ("sclk %d..%d MHz" % (min(clocks), max(clocks))) if clocks else "sclk n/a"
        print("%-48s M=%-8d N=%-5d K=%-5d %8.1f us  %7.1f TFLOP/s = %.3f of 2.5 PF  (%s, %d reps)" % (
            name, M, N, K, dt * 1e6, tf, tf / 2500.0, clk, reps))
        del a, b, c


if __name__ == "__main__":
    main()
