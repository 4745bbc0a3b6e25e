#!/usr/bin/env python3
"""How many branches of a captured hipGraph really run side by side?  n streams fork from the capture stream, each runs a chain of
K one-block spin kernels (~20 us each), join, replay: n lanes side by side take K x 20 us, serialised ones n x that.
Variants: every branch on a side stream, or the first branch kept on the capture stream (the step's layout)."""
import sys, time
import torch

K, CYC = 300, 42_000
dev = "cuda"
x = torch.zeros(1, device=dev)


def build(n, keep_first):
    side = [torch.cuda.Stream() for _ in range(n)]
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        with torch.cuda.graph(g, stream=cap):
            cur = torch.cuda.current_stream()
            used = []
            for i in reversed(range(n)):                     # side branches fork first; the capture stream's own chain is enqueued last
                st = None if (keep_first and i == 0) else side[i]
                if st is not None:
                    st.wait_stream(cur)
                    used.append(st)
                with (torch.cuda.stream(st) if st is not None else torch.cuda.stream(cur)):
                    for _ in range(K):
                        torch.cuda._sleep(CYC)
            for st in used:
                cur.wait_stream(st)
    return g


torch.cuda._sleep(CYC); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(K):
    torch.cuda._sleep(CYC)
e1.record(); torch.cuda.synchronize()
base = e0.elapsed_time(e1)
print(f"one chain of {K} spin kernels, eager: {base:.2f} ms ({base / K * 1e3:.1f} us per kernel)")
for n in (1, 2, 3, 4, 5, 6):
    for keep in (True, False):
        g = build(n, keep)
        g.replay(); torch.cuda.synchronize()
        e0.record()
        for _ in range(3):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 3
        print(f"{n} chains, first on the capture stream = {keep}: {t:7.2f} ms  = {t / base:4.2f} x one chain")
