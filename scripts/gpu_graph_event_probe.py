"""Dev probe (GPU): can timing events be recorded inside a captured hipGraph (torch external events) and read after replay?"""
import torch
x = torch.randn(4096, 4096, device='cuda')
y = torch.empty_like(x)
torch.matmul(x, x, out=y)
torch.cuda.synchronize()
side = torch.cuda.Stream()
evs = []
g = torch.cuda.CUDAGraph()
side.wait_stream(torch.cuda.current_stream())
ok = True
try:
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for i in range(3):
                e0 = torch.cuda.Event(enable_timing=True, external=True)
                e1 = torch.cuda.Event(enable_timing=True, external=True)
                e0.record(side)
                torch.matmul(x, x, out=y)
                e1.record(side)
                evs.append((e0, e1))
except Exception as e:
    ok = False
    print('capture failed:', repr(e))
if ok:
    for r in range(3):
        g.replay()
        torch.cuda.synchronize()
        try:
            print('replay', r, [round(a.elapsed_time(b), 4) for a, b in evs])
        except Exception as e:
            print('elapsed_time failed:', repr(e))
            break
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); torch.matmul(x, x, out=y); t1.record(); torch.cuda.synchronize()
    print('eager matmul ms', t0.elapsed_time(t1))
