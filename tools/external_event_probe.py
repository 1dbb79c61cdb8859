"""Does an event recorded INSIDE a captured graph (torch.cuda.Event(external=True) -> hipEventRecordExternal node) release a waiter
on another stream when that point of the replay is reached (not at the end of the graph)?"""
import time, torch
x = torch.zeros(1 << 20, device="cuda"); y = torch.zeros(1 << 20, device="cuda"); z = torch.zeros(8, device="cuda")
def busy(t, n):
    for _ in range(n):
        t.mul_(1.0001).add_(0.5)
cap, s2 = torch.cuda.Stream(), torch.cuda.Stream()
try:
    ev = torch.cuda.Event(external=True)
except TypeError as e:
    print("no external events in this torch:", e); raise SystemExit
with torch.cuda.stream(cap):
    busy(x, 3); busy(y, 3)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=cap):
    busy(x, 20)            # first half (~short)
    ev.record()
    busy(y, 400)           # second half (~long)
torch.cuda.synchronize()
t_first = torch.cuda.Event(enable_timing=True); t_wait = torch.cuda.Event(enable_timing=True); t_end = torch.cuda.Event(enable_timing=True)
t0 = torch.cuda.Event(enable_timing=True)
for it in range(3):
    torch.cuda.synchronize()
    with torch.cuda.stream(cap):
        t0.record(); g.replay(); t_end.record()
    with torch.cuda.stream(s2):
        s2.wait_event(ev); z.add_(1.0); t_wait.record()
    torch.cuda.synchronize()
    print(f"replay {it}: waiter released after {t0.elapsed_time(t_wait):.3f} ms, graph finished after {t0.elapsed_time(t_end):.3f} ms")
