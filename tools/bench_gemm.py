#!/usr/bin/env python3
"""The general contraction alone (ntx_gemm_f32; in a training step it takes the weight gradients, rows_kernel the rest), HIP-event timed.  GPU box only.
    python tools/bench_gemm.py [--m 262144]"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_tex_amd import _lib

ap = argparse.ArgumentParser(); ap.add_argument("--m", type=int, default=262144); ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda", 0)
M = a.m
def run(ak, Mg, N, K, lda, ldb):
    A = torch.randn((Mg if ak else K), lda, device=dev); B = torch.randn(K, ldb, device=dev); Cc = torch.empty(Mg, N, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    f = lambda: _lib.check(_lib.lib.ntx_gemm_f32(A.data_ptr(), lda, ak, B.data_ptr(), ldb, 0, Cc.data_ptr(), N, Mg, N, K, None, 0, st))
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    return ms, 2.0 * Mg * N * K / (ms * 1e-3) / 1e12
for name, args in [("forward / dX  [M,256] x [256,256]", (1, M, 256, 256, 256, 256)), ("forward  [M,340] x [337,256]", (1, M, 256, 337, 340, 256)),
                   ("dW  [M,256]^T x [M,256] (no split: one pass over M)", (0, 256, 256, M, 256, 256))]:
    ms, tf = run(*args)
    print(json.dumps({"what": name, "ms": round(ms, 4), "TFLOP/s": round(tf, 1), "frac_of_157.3": round(tf / 157.3, 3)}))
