#!/usr/bin/env python3
"""Timeline of the LAST training step in a rocprofv3 --kernel-trace csv: what each hardware queue (= chain of the captured
step) did, how much of the step the chip ran one / two / three kernels at once, and which kernels ran ALONE (nothing else
on the chip) -- the time a concurrent chain cannot hide.  Steps are delimited by the once-per-step lora_grad_reduce_kernel.

    python tools/step_timeline.py <kernel_trace.csv> [top] [last_step_out.csv]
"""
import collections
import csv
import sys


def short(n: str) -> str:
    if 'gemm_glds' in n:
        return 'glds ' + n[21:58]
    return n[:56]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?'),
           int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))) for r in rows]
    ev.sort()
    idx = [i for i, e in enumerate(ev) if 'lora_grad_reduce' in e[2]]
    step = ev[idx[-2] + 1:idx[-1] + 1]
    t0, t1 = step[0][0], max(e[1] for e in step)
    print(f"step span {(t1 - t0) / 1e6:.2f} ms, {len(step)} kernels, sum of durations {sum(e[1] - e[0] for e in step) / 1e6:.2f} ms")
    if len(sys.argv) > 3:
        with open(sys.argv[3], 'w') as f:
            f.write("start_us,dur_us,queue,blocks,kernel\n")
            for s, e, n, q, g in step:
                f.write(f"{(s - t0) / 1e3:.2f},{(e - s) / 1e3:.2f},{q},{g},{short(n)}\n")
    # per queue
    byq = collections.defaultdict(list)
    for e in step:
        byq[e[3]].append(e)
    print("\nper hardware queue:")
    for q, L in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        busy = sum(e[1] - e[0] for e in L)
        gaps = [L[i + 1][0] - L[i][1] for i in range(len(L) - 1)]
        pos = [g for g in gaps if g > 0]
        print(f"  queue {q}: {len(L):5d} kernels, first +{(L[0][0] - t0) / 1e6:6.2f} ms, last end +{(max(e[1] for e in L) - t0) / 1e6:6.2f} ms, "
              f"busy {busy / 1e6:6.2f} ms, gaps {sum(pos) / 1e6:6.2f} ms (median {sorted(pos)[len(pos) // 2] / 1e3 if pos else 0:.1f} us, "
              f"{sum(1 for g in pos if g > 20000)} over 20 us)")
    # concurrency sweep
    pts = []
    for i, e in enumerate(step):
        pts.append((e[0], 1, i))
        pts.append((e[1], -1, i))
    pts.sort()
    active = set()
    conc = collections.Counter()
    alone = collections.defaultdict(float)          # kernel -> ns it ran with nothing else on the chip
    alone_q = collections.defaultdict(float)
    prev = pts[0][0]
    for t, d, i in pts:
        if t > prev:
            conc[len(active)] += t - prev
            if len(active) == 1:
                j = next(iter(active))
                alone[short(step[j][2])] += t - prev
                alone_q[step[j][3]] += t - prev
        prev = t
        if d > 0:
            active.add(i)
        else:
            active.discard(i)
    print("\nkernels in flight (share of the step):")
    for k in sorted(conc):
        print(f"  {k}: {conc[k] / 1e6:6.2f} ms ({100 * conc[k] / (t1 - t0):4.1f} %)")
    print("\nran ALONE, by queue: " + ", ".join(f"{q}: {v / 1e6:.2f} ms" for q, v in sorted(alone_q.items(), key=lambda kv: -kv[1])))
    print("ran ALONE, by kernel:")
    for k, v in sorted(alone.items(), key=lambda kv: -kv[1])[:top]:
        print(f"  {k:58s} {v / 1e6:6.2f} ms")
    # per-queue kernel table
    for q, L in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        d = collections.defaultdict(list)
        for s, e, n, _, g in L:
            d[short(n)].append((e - s) / 1e3)
        print(f"\nqueue {q}: top kernels")
        for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:top]:
            print(f"  {k:58s} n={len(v):4d} mean={sum(v) / len(v):7.1f} us total={sum(v) / 1e3:6.2f} ms")


if __name__ == "__main__":
    main()
