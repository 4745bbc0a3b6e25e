#!/usr/bin/env python3
"""Does replaying ONE captured graph back to back block the host until the previous replay has finished?  Host time of g.replay() for a
~10 ms graph of 2000 small kernels replayed five times in a row, against two captures of the same work replayed alternately."""
import time
import torch

dev = "cuda"
x = torch.randn(1024, 1024, device=dev)
ys = [torch.empty_like(x) for _ in range(2)]


def work(y):
    for _ in range(2000):
        torch.add(x, 1.0, out=y)


def capture(y):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        work(y)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        work(y)
    return g


ga, gb = capture(ys[0]), capture(ys[1])
for name, seq in (("one graph", [ga] * 6), ("two graphs alternating", [ga, gb] * 3)):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    host = []
    for g in seq:
        t = time.perf_counter()
        g.replay()
        host.append((time.perf_counter() - t) * 1e3)
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: host ms per replay call {[round(h, 2) for h in host]}, GPU total {e0.elapsed_time(e1):.2f} ms")

# ---- the same with three branches on three streams inside the captured graph (the trainer's micro-step has that shape)
side = [torch.cuda.Stream() for _ in range(2)]
zs = [torch.empty_like(x) for _ in range(3)]


def work3():
    cur = torch.cuda.current_stream()
    for st in side:
        st.wait_stream(cur)
    for k, st in enumerate(side):
        with torch.cuda.stream(st):
            for _ in range(700):
                torch.add(x, 1.0, out=zs[k + 1])
    for _ in range(700):
        torch.add(x, 1.0, out=zs[0])
    for st in side:
        cur.wait_stream(st)


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    work3()
torch.cuda.current_stream().wait_stream(s)
g3 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g3):
    work3()
torch.cuda.synchronize()
host = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(6):
    t = time.perf_counter()
    g3.replay()
    host.append((time.perf_counter() - t) * 1e3)
e1.record()
torch.cuda.synchronize()
print(f"three-branch graph: host ms per replay call {[round(h, 2) for h in host]}, GPU total {e0.elapsed_time(e1):.2f} ms")
