import numpy as np, torch, math
# all finite bf16 inputs in [-12, 12]
bits = np.arange(65536, dtype=np.uint32) << 16
x = bits.view(np.float32)
x = x[np.isfinite(x) & (np.abs(x) <= 12)]
xd = x.astype(np.float64)
exact = 0.5 * xd * (1 + np.vectorize(math.erf)(xd / math.sqrt(2)))
f = np.float32
def gelu_26(x):
    z = x * f(0.70710678118654752440); ax = np.abs(z)
    t = f(1) / (f(0.3275911) * ax + f(1))
    p = f(1.061405429) * t + f(-1.453152027); p = p * t + f(1.421413741); p = p * t + f(-0.284496736); p = p * t + f(0.254829592)
    e = np.exp2(f(-1.44269504088896340736) * ax * ax).astype(f)
    r = -(p * t) * e + f(1)
    er = np.copysign(r, z)
    return (f(0.5) * x) * (f(1) + er)
def gelu_28(x):
    z = x * f(0.70710678118654752440); ax = np.abs(z)
    p = f(0.0000430638) * ax + f(0.0002765672); p = p * ax + f(0.0001520143); p = p * ax + f(0.0092705272); p = p * ax + f(0.0422820123); p = p * ax + f(0.0705230784); p = p * ax + f(1)
    with np.errstate(over='ignore'):
        p = p * p; p = p * p; p = p * p; p = p * p
        r = f(1) / p
    er = np.copysign(f(1) - r, z)
    return (f(0.5) * x) * (f(1) + er)
def bf16r(v): return torch.from_numpy(v.astype(np.float32)).bfloat16().float().numpy()
for name, fn in (("A&S 7.1.26 (ships)", gelu_26), ("A&S 7.1.28 (no exp)", gelu_28)):
    y = fn(x).astype(np.float64)
    err = np.abs(y - exact)
    b_ex, b_y = bf16r(exact), bf16r(y)
    diff = (b_ex != b_y)
    print(f"{name}: max abs err {err.max():.3e} at x={x[err.argmax()]:.4f}; bf16-rounded outputs that differ from bf16(exact): {diff.sum()} of {len(x)}; max rel err of those {np.max(np.abs(b_y[diff]-b_ex[diff])/np.maximum(np.abs(b_ex[diff]),1e-30)) if diff.any() else 0:.3e}")
