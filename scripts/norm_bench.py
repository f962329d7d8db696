#!/usr/bin/env python
"""RMSNorm backward on the decoder's rows ([16 x 287, 3584] bf16, residual added, weight gradient wanted): time per call and, against a
float64 torch reference, the error of dx and dw.  The kernel variant is picked by the environment (read once per process):
    DXA_NORM_BWD_NO_SPLIT=1   round-4 kernel (one wave per row);  default: the split kernel;  DXA_NORM_BWD_ROWS=3: its 384-thread form"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dexbotic_amd import kernels as K  # noqa: E402


def main():
    dev = "cuda"
    torch.manual_seed(0)
    for rows, cols in ((4592, 3584), (8688, 3584), (13056, 2048), (543, 3584)):
        x = torch.randn(rows, cols, device=dev).bfloat16()
        dy = torch.randn(rows, cols, device=dev).bfloat16()
        res = torch.randn(rows, cols, device=dev).bfloat16()
        w = (1 + 0.1 * torch.randn(cols, device=dev)).bfloat16()
        y, rstd = K.rmsnorm_fwd(x, w, 1e-6)
        dw = torch.zeros(cols, device=dev, dtype=torch.float32)
        for _ in range(3):
            dx, _ = K.rmsnorm_bwd(dy, x, w, rstd, dw_out=dw, accumulate=False, want_dw=True, residual=res)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps):
            dx, _ = K.rmsnorm_bwd(dy, x, w, rstd, dw_out=dw, accumulate=False, want_dw=True, residual=res)
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / reps
        # reference (float64; x-hat rounded to bf16 before the weight-gradient product, as the kernel and HF's autocast do)
        xd, gd, wd, rs = x.double(), dy.double(), w.double(), rstd.double()[:, None]
        xh = xd * rs
        gw = gd * wd
        c2 = (gw * xh).sum(-1, keepdim=True) / cols
        dx_ref = rs * (gw - xh * c2) + res.double()
        dw_ref = (gd * xh.float().bfloat16().double()).sum(0)
        e_dx = float((dx.double() - dx_ref).abs().max() / dx_ref.abs().max())
        e_dw = float((dw.double() - dw_ref).abs().max() / dw_ref.abs().max())
        mb = rows * cols * 2 * 4 / 1e6
        print(f"rows {rows:6d} cols {cols:5d}: {us:8.1f} us per call (kernel + column sum)  {mb / us * 1e-3 * 1e3:6.2f} GB/s x1e3  dx err {e_dx:.2e}  dw err {e_dw:.2e}", flush=True)


if __name__ == "__main__":
    main()
