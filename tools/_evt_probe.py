import torch, inspect
print(torch.__version__)
print(inspect.signature(torch.cuda.Event.__new__) if hasattr(torch.cuda.Event,'__new__') else '')
x = torch.randn(8192, 8192, device="cuda")
y = torch.empty_like(x)
s = torch.cuda.Stream()
evs = []
try:
    with torch.cuda.stream(s):
        for _ in range(3): y.copy_(x * 2)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(4):
                e0 = torch.cuda.Event(enable_timing=True, external=True); e1 = torch.cuda.Event(enable_timing=True, external=True)
                e0.record()
                torch.mul(x, 2.0, out=y)
                if i % 2: torch.mul(x, 3.0, out=y)
                e1.record()
                evs.append((e0, e1))
        for rep in range(3):
            g.replay()
            torch.cuda.synchronize()
            print([round(a.elapsed_time(b) * 1e3, 1) for a, b in evs])
except Exception as ex:
    print("FAILED:", type(ex).__name__, ex)
