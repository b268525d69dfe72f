"""what could ONE grouped launch for q|k and V^T buy?  The two launches read the same activation operand: q|k is
47 616 x 3584 (2604 tiles of 256^2 = 10.17 rounds on 256 CUs), V^T = W_v X^T is 1792 x 47 616 (1302 tiles = 5.09 rounds); grouped they would
be 3906 tiles = 15.26 rounds with ONE ragged last round instead of two.  A grouped launch cannot be faster than the plain GEMM with the same
tile count and the same K — X [47 616][1792] against W [5376][1792] (q | k | v side by side, untransposed V: the SAME 3906 tiles, one epilogue
kind, one tile walk) — so that GEMM's time is the bound, measured here next to the two launches the engine makes (interleaved, same box).

    python tools/grouped_qkv_bound.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kandinsky-5_amd"))
import torch
from kandinsky import _engine as E
N, D = 47616, 1792
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(N, D, device="cuda", generator=g).bfloat16()
wqk = (torch.randn(2 * D, D, device="cuda", generator=g) * 0.05).bfloat16()
wv = (torch.randn(D, D, device="cuda", generator=g) * 0.05).bfloat16()
wqkv = torch.cat([wqk, wv]).contiguous()
bqk, bv, bqkv = torch.randn(2 * D, device="cuda"), torch.randn(D, device="cuda"), torch.randn(3 * D, device="cuda")
oqk = torch.empty(N, 2 * D, dtype=torch.bfloat16, device="cuda")
ovt = torch.empty(D, N, dtype=torch.bfloat16, device="cuda")
oqkv = torch.empty(N, 3 * D, dtype=torch.bfloat16, device="cuda")


def two():
    E.gemm(x, wqk, out=oqk, bias=bqk, epilogue=E.EPI_BIAS)
    E.gemm(wv, x, out=ovt, bias=bv, epilogue=E.EPI_BIAS_M)      # V^T = W_v X^T, bias per row


def one():
    E.gemm(x, wqkv, out=oqkv, bias=bqkv, epilogue=E.EPI_BIAS)


def t(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20


b1 = b2 = 1e9
for rep in range(4):
    b2 = min(b2, t(two)); b1 = min(b1, t(one))
fl = 2.0 * N * 3 * D * D
print(f"q|k launch + V^T launch (what the engine runs): {b2 * 1e3:7.1f} us = {fl / b2 / 1e9:5.0f} TFLOP/s")
print(f"one 47616 x 5376 x 1792 launch (the bound of a grouped launch: same 3906 tiles): {b1 * 1e3:7.1f} us = {fl / b1 / 1e9:5.0f} TFLOP/s")
print(f"upper bound of the saving: {(b2 - b1) * 1e3:.1f} us per block = {32 * (b2 - b1):.2f} ms per step of ~520 ({100 * 32 * (b2 - b1) / 520:.2f} %)")
