"""torch.argmax on the device against numpy's argmax (first maximum wins) on tensors full of ties and NaNs.  python tools/exp/argmax_ties_probe.py"""
import numpy as np, torch
g = torch.Generator().manual_seed(0)
ok = True
for shape in ((1, 18, 256, 256), (3, 18, 64, 64), (2, 19, 33, 7)):
    for kind in ("ties", "all_equal", "nan"):
        m = torch.randint(0, 3, shape, generator=g).float()
        if kind == "all_equal": m[:] = 0.25
        if kind == "nan": m[torch.rand(shape, generator=g) < 0.05] = float("nan")
        a = np.argmax(m.numpy(), axis=1)
        b = torch.argmax(m.cuda(), dim=1).cpu().numpy()
        c = torch.argmax(m, dim=1).numpy()
        ok &= np.array_equal(a, b)
        print(shape, kind, "device == numpy:", np.array_equal(a, b), " torch-cpu == numpy:", np.array_equal(a, c))
print("all equal:", ok)
