import torch, time
x = torch.randn(1, 32, 96, 96, 96, device="cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
v = x.reshape(-1).view(torch.int32)
print("int64 sum  %.1f us" % t(lambda: v.sum(dtype=torch.int64)))
print("int32 sum  %.1f us" % t(lambda: v.sum(dtype=torch.int32)))
print("fp32 sum   %.1f us" % t(lambda: x.sum()))
print("2d int32   %.1f us" % t(lambda: v.view(-1, 4096).sum(dim=1, dtype=torch.int32).sum(dtype=torch.int32)))
a, b = v.sum(dtype=torch.int32), v.sum(dtype=torch.int32)
print("deterministic", bool(a == b), int(a))
y = x.clone(); y.view(-1)[12345] += 1e-7 * y.view(-1)[12345].abs() + 1e-30
print("detects a 1-ulp-class change", bool(y.reshape(-1).view(torch.int32).sum(dtype=torch.int32) != a))
