"""Time the causal 3x3x3 convolution (k5_conv3d_bf16 / k5_conv3d_bf16_stats) on the layer shapes of one (5,64,96) decode tile.

    python tools/conv_bench.py [stats]

One line per distinct layer shape: source dims, channels, folded upsample, ms, TFLOP/s (2*27*Cin*Cout per output position), and
the share of the tile's conv time (count x ms).  Random operands.  K5_LIB selects a variant build (tools/build_variant.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kandinsky-5_amd"))
import torch
from kandinsky import _engine as E

# (count per tile, Cin, Cout, (up_t, up_s), source dims, residual epilogue)
LAYERS = [
    (5, 128, 128, (1, 1), (17, 512, 768), False),
    (1, 256, 128, (1, 1), (17, 512, 768), False),
    (1, 256, 256, (2, 2), (9, 256, 384), False),
    (5, 256, 256, (1, 1), (9, 256, 384), False),
    (1, 512, 256, (1, 1), (9, 256, 384), False),
    (1, 512, 512, (2, 2), (5, 128, 192), False),
    (6, 512, 512, (1, 1), (5, 128, 192), False),
    (1, 512, 512, (1, 2), (5, 64, 96), False),
    (10, 512, 512, (1, 1), (5, 64, 96), False),
]


def main():
    stats = "stats" in sys.argv[1:]
    L = E.lib()
    rows, total = [], 0.0
    for cnt, Cin, Cout, (ut, us), (Ts, Hs, Ws), res in LAYERS:
        To, Ho, Wo = (2 * Ts - 1 if ut == 2 else Ts), us * Hs, us * Ws
        M = To * Ho * Wo
        x = torch.randn(Ts * Hs * Ws, Cin, device="cuda").bfloat16()
        w = (torch.randn(Cout, 27 * Cin, device="cuda") * 0.02).bfloat16()
        b = torch.randn(Cout, device="cuda")
        out = torch.empty(M, Cout, dtype=torch.bfloat16, device="cuda")
        qs = torch.empty(max(1, L.k5_conv3d_stats_size(M, Cout) // 4), device="cuda") if stats else None

        def run():
            if stats:
                E.check(L.k5_conv3d_bf16_stats(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), Ts, Hs, Ws, Cin, Cout, ut, us, Cout,
                                               None, Cout, qs.data_ptr(), E.stream_ptr()))
            else:
                E.check(L.k5_conv3d_bf16(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), Ts, Hs, Ws, Cin, Cout, ut, us, Cout,
                                         None, Cout, E.stream_ptr()))
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 5
        a.record()
        for _ in range(it):
            run()
        e.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(e) / it
        fl = 2.0 * 27 * Cin * Cout * M
        rows.append((cnt, Cin, Cout, ut, us, Ts, Hs, Ws, ms, fl / ms / 1e9))
        total += cnt * ms
    for cnt, Cin, Cout, ut, us, Ts, Hs, Ws, ms, tf in rows:
        print(f"{cnt:2d} x  {Cin:3d} -> {Cout:3d}  up ({ut},{us})  src ({Ts},{Hs},{Ws}):  {ms:8.3f} ms  {tf:7.1f} TFLOP/s  {100 * cnt * ms / total:5.1f} %", flush=True)
    print(f"sum over the tile's 31 four-wave convs: {total:.2f} ms")


if __name__ == "__main__":
    main()
