import torch
dev = torch.device("cuda:0")
s = torch.cuda.Stream()
x = torch.zeros(1 << 24, device=dev)
evs = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=s):
        for i in range(4):
            evs[i].record(s)
            x.add_(1.0)
        evs[4].record(s)
    with torch.cuda.stream(s):
        g.replay(); g.replay()
    s.synchronize()
    print("ok", [evs[i].elapsed_time(evs[i + 1]) for i in range(4)])
except Exception as e:
    print("capture of timing events failed:", type(e).__name__, str(e)[:300])
    try:
        evs = [torch.cuda.Event(enable_timing=True, external=True) for _ in range(5)]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(4):
                evs[i].record(s)
                x.add_(1.0)
            evs[4].record(s)
        with torch.cuda.stream(s):
            g.replay(); g.replay()
        s.synchronize()
        print("external ok", [evs[i].elapsed_time(evs[i + 1]) for i in range(4)])
    except Exception as e2:
        print("external failed:", type(e2).__name__, str(e2)[:300])
