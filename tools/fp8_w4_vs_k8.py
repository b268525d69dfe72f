import os, sys, subprocess, json
ROOT = "/root/repo"
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "kandinsky-5_amd"))
    import torch
    from kandinsky import _engine as E
    L = E.lib()
    outs = {}
    for name, M, N, K, epi in [("bias_longK", 9472, 1792, 7168, E.EPI_BIAS), ("gate_longK", 9472, 1792, 7168, E.EPI_GATE), ("gate_shortK", 9472, 1792, 1792, E.EPI_GATE),
                               ("gate_289", 4100, 4352, 1792, E.EPI_GATE), ("gate_256", 4096, 4096, 512, E.EPI_GATE)]:
        g = torch.Generator(device="cuda").manual_seed(1)
        a8 = (torch.randn(M, K, device="cuda", generator=g)).to(torch.float8_e4m3fn).view(torch.uint8)
        w8 = (torch.randn(N, K, device="cuda", generator=g) * 0.5).to(torch.float8_e4m3fn).view(torch.uint8)
        ws = torch.rand(N, device="cuda", generator=g) * 0.01 + 0.001
        gate = torch.randn(N, device="cuda", generator=g)
        res = torch.randn(M, N, device="cuda", generator=g).bfloat16()
        rs = []
        for rep in range(3):
            out = res.clone()
            E.check(L.k5_gemm_fp8(a8.data_ptr(), w8.data_ptr(), ws.data_ptr(), out.data_ptr(), M, N, K, K, K, N, epi, out.data_ptr() if epi == E.EPI_GATE else None,
                                  N if epi == E.EPI_GATE else 0, gate.data_ptr() if epi == E.EPI_GATE else None, E.stream_ptr()))
            torch.cuda.synchronize()
            rs.append(out)
        print(name, "repeat-identical", all(torch.equal(rs[0], r) for r in rs))
        torch.save(rs[0].cpu(), f"/tmp/f8_{name}_{os.environ.get('K5_GEMM_FP8_V','d')}.pt")
else:
    import torch
    for v in ("8", "4"):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, K5_GEMM_FP8_V=v), check=True)
    for name in ("bias_longK", "gate_longK", "gate_shortK", "gate_289", "gate_256"):
        a, b = torch.load(f"/tmp/f8_{name}_8.pt").float(), torch.load(f"/tmp/f8_{name}_4.pt").float()
        d = (a - b).abs()
        bad = d > 0.05 + 0.03 * a.abs()
        idx = bad.nonzero()
        print(name, "max diff", d.max().item(), "bad", int(bad.sum()), "rows", sorted(set((idx[:, 0] // 16).tolist()))[:12], "cols", sorted(set((idx[:, 1] // 16).tolist()))[:12])
