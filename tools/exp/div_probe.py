import torch
dev = torch.device("cuda:0")
for N in (7, 12, 100, 129, 300):
    i = torch.arange(0, N ** 3, dtype=torch.long)
    h1 = (i.float() / N) % N
    h0 = ((i.float() / N) / N) % N
    d = i.to(dev)
    Nt = torch.tensor(float(N), device=dev)
    a1 = ((d.float() / N) % N).cpu(); a0 = (((d.float() / N) / N) % N).cpu()
    b1 = (torch.remainder(torch.div(d.float(), Nt), Nt)).cpu(); b0 = torch.remainder(torch.div(torch.div(d.float(), Nt), Nt), Nt).cpu()
    print(N, "scalar divisor equal:", torch.equal(a1, h1), torch.equal(a0, h0), " tensor divisor equal:", torch.equal(b1, h1), torch.equal(b0, h0))
