#!/usr/bin/env python3
"""tools/ubench/lib_gemm_rate.py — what the vendor GEMM library (hipBLASLt / rocBLAS behind torch.matmul) reaches on the encoder's four linear-layer
shapes at one 65,536-token batch, f16 in / fp32 accumulate: the yardstick for t5_gemm256x_kernel (profiles/r05/lib_gemm_rate.log)."""
import os, sys, json, time
import torch
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
shapes = {"qkv (N 12288, K 1024)": (12288, 1024), "attn out (N 1024, K 4096)": (1024, 4096), "ffn up (N 16384, K 1024)": (16384, 1024), "ffn down (N 1024, K 16384)": (1024, 16384)}
dev = "cuda"
res = {}
for name, (N, K) in shapes.items():
    a = torch.randn(M, K, device=dev, dtype=torch.float16) * 0.1
    w = torch.randn(N, K, device=dev, dtype=torch.float16) * 0.1
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    for _ in range(3): torch.matmul(a, w.t(), out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 10
    e0.record()
    for _ in range(it): torch.matmul(a, w.t(), out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    res[name] = {"ms": ms, "tflops": 2.0 * M * N * K / ms / 1e9}
    # with the ReLU behind it (what a fused epilogue would save)
    del a, w, out
tot_flop = sum(2.0 * M * N * K for N, K in shapes.values())
tot_ms = sum(r["ms"] for r in res.values())
res["all four layers of one block"] = {"ms": tot_ms, "tflops": tot_flop / tot_ms / 1e9}
res["env"] = {k: os.environ.get(k) for k in ("TORCH_BLAS_PREFER_HIPBLASLT", "PYTORCH_TUNABLEOP_ENABLED")}
print(json.dumps(res))
