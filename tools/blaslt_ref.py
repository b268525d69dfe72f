"""Reference point only (never on the product path): what the vendor library reaches on the engine's GEMM shapes on
this box, to judge the hand-written kernels against.  torch.matmul -> hipBLASLt."""
import torch, time
shapes = [("qk", 47616, 3584, 1792), ("out", 47616, 1792, 1792), ("ff1", 47616, 7168, 1792), ("ff2", 47616, 1792, 7168),
          ("4096^3", 4096, 4096, 4096), ("8192^3", 8192, 8192, 8192)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): torch.nn.functional.linear(a, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): torch.nn.functional.linear(a, w)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"hipBLASLt {name:8s} M={M} N={N} K={K}: {ms:8.3f} ms {2*M*N*K/ms/1e9:8.1f} TFLOP/s")
