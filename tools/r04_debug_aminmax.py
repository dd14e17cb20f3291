import torch
dev = torch.device("cuda:0")
t = torch.randint(0, 14, (2, 1, 64, 224, 192), device=dev).float()
w = torch.randn(64, 64, 3, 3, 3, device=dev)
x = torch.randn(2, 64, 32, 56, 48, device=dev)
for name in ("aminmax", "minmax", "any"):
    cnt = torch.zeros((), dtype=torch.int64, device=dev)
    for i in range(300):
        y = torch.nn.functional.conv3d(x, w, padding=1)          # other work in flight
        if name == "aminmax":
            lo, hi = torch.aminmax(t)
            bad = (lo < 0) | (hi >= 14)
        elif name == "minmax":
            bad = (t.min() < 0) | (t.max() >= 14)
        else:
            bad = ((t < 0) | (t >= 14)).any()
        cnt += bad
        del y
    torch.cuda.synchronize()
    print(name, "eager flagged", int(cnt), "of 300", flush=True)
# inside a replayed graph
cnt = torch.zeros((), dtype=torch.int64, device=dev)
def body():
    y = torch.nn.functional.conv3d(x, w, padding=1)
    lo, hi = torch.aminmax(t)
    cnt.add_((lo < 0) | (hi >= 14))
    return y
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    body()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
cnt.zero_()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = body()
for i in range(300):
    g.replay()
torch.cuda.synchronize()
print("aminmax in graph flagged", int(cnt), "of 300", flush=True)
