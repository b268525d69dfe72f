import os, sys, subprocess
if len(sys.argv) > 1:
    sys.path.insert(0, "/root/repo/kandinsky-5_amd")
    import torch
    from kandinsky import _engine as E
    BF=torch.bfloat16
    torch.manual_seed(0)
    m,n,k=47616,1792,1792
    a, w = torch.randn(m, k, device="cuda").to(BF), (torch.randn(n, k, device="cuda") * 0.05).to(BF)
    bias = torch.randn(n, device="cuda").to(BF).float()
    ref = torch.empty(m, n, dtype=BF, device="cuda"); out = torch.empty(m, n, dtype=BF, device="cuda")
    E.gemm(a, w, bias, E.EPI_BIAS, out=ref, kernel=4)
    E.gemm(a, w, bias, E.EPI_BIAS, out=out, kernel=24)
    torch.cuda.synchronize()
    d=(out.float()-ref.float()).abs()
    print("dbg", os.environ.get("K5_GEMM_DBG"), "ok; differing outputs", (d>0).float().mean().item(), "max abs", d.max().item(), flush=True)
else:
    for dbg in ("176", "128", "48", "0"):
        r = subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, K5_GEMM_DBG=dbg), capture_output=True, text=True, timeout=120)
        print("K5_GEMM_DBG", dbg, "rc", r.returncode, (r.stdout + r.stderr).strip().splitlines()[-1][:200], flush=True)
