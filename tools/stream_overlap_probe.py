"""Do kernels of a second stream run BESIDE a long sequence on the compute stream, or behind it?  (HIP maps streams onto a few hardware
queues; work that shares a queue runs in host submission order.)  Decides how GraphedDDPStep must issue its collectives.

    python tools/stream_overlap_probe.py
For each variant: stream A gets `n` big matmuls (eager or as ONE hipGraph launch), an event is recorded in front of them, stream B waits
for that event and runs one small kernel, B's completion event is timed against A's end.  "beside" = B finished while A was still busy."""
import torch

dev = torch.device("cuda", 0)
x = torch.randn(4096, 4096, device=dev)
small = torch.zeros(1024, device=dev)


def run(a_stream, b_stream, graphed, n=40, issue_b_first=False):
    torch.cuda.synchronize()
    start, e0, b_done, a_done = (torch.cuda.Event(enable_timing=True) for _ in range(4))
    g = None
    if graphed:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = x
            for _ in range(n):
                y = y @ x
        torch.cuda.synchronize()
    with torch.cuda.stream(a_stream):
        start.record()
        e0.record()

        def issue_a():
            if graphed:
                g.replay()
            else:
                y = x
                for _ in range(n):
                    y = y @ x
            a_done.record()

        def issue_b():
            with torch.cuda.stream(b_stream):
                b_stream.wait_event(e0)
                small.add_(1.0)
                b_done.record()

        if issue_b_first:
            issue_b(), issue_a()
        else:
            issue_a(), issue_b()
    torch.cuda.synchronize()
    return start.elapsed_time(b_done), start.elapsed_time(a_done)


default = torch.cuda.default_stream(dev)
side1, side2, hi = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev, priority=-1)
for name, a, b in (("A=default  B=side", default, side1), ("A=default  B=high-priority", default, hi), ("A=side     B=side2", side1, side2),
                   ("A=side     B=high-priority", side1, hi)):
    for graphed in (False, True):
        for first in (False, True):
            run(a, b, graphed, first)  # warm
            tb, ta = run(a, b, graphed, issue_b_first=first)
            print(f"{name:28s} {'graph' if graphed else 'eager'}  B issued {'before' if first else 'after '} A: B done at {tb:8.2f} ms, A done at {ta:8.2f} ms"
                  f"  -> {'BESIDE' if tb < 0.5 * ta else 'behind'}", flush=True)


def run2(graphed, variant, n=20):
    """A: segment 1, event e1, segment 2 (as GraphedDDPStep's B1 | B2); B waits for e1 and runs a small kernel.  B should finish at
    about HALF of A.  variant 0: both segments on the compute stream; 1: segment 2 on a second stream that waits for e1."""
    torch.cuda.synchronize()
    start, e1, b_done, a_done = (torch.cuda.Event(enable_timing=True) for _ in range(4))

    def seg():
        y = x
        for _ in range(n):
            y = y @ x

    g1 = g2 = None
    if graphed:
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1):
            seg()
        with torch.cuda.graph(g2, pool=g1.pool()):
            seg()
        torch.cuda.synchronize()
    s_a, s_a2, s_b = default, side2, side1
    start.record(s_a)
    with torch.cuda.stream(s_a):
        g1.replay() if graphed else seg()
        e1.record()
    if variant == 0:
        with torch.cuda.stream(s_a):
            g2.replay() if graphed else seg()
            a_done.record()
    else:
        with torch.cuda.stream(s_a2):
            s_a2.wait_event(e1)
            g2.replay() if graphed else seg()
            a_done.record()
    with torch.cuda.stream(s_b):
        s_b.wait_event(e1)
        small.add_(1.0)
        b_done.record()
    torch.cuda.synchronize()
    return start.elapsed_time(b_done), start.elapsed_time(a_done)


print("\nevent BETWEEN two segments of the compute stream, waited for by another stream (expected: B done at ~half of A):")
for graphed in (False, True):
    for variant in (0, 1):
        run2(graphed, variant)
        tb, ta = run2(graphed, variant)
        print(f"{'graph' if graphed else 'eager'} segments, segment 2 on {'the SAME stream' if variant == 0 else 'a second stream':16s}: B done at {tb:7.2f} ms, A done at {ta:7.2f} ms"
              f"  -> {'between the segments' if tb < 0.75 * ta else 'BEHIND both segments'}", flush=True)
