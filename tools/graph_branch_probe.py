#!/usr/bin/env python3
"""Do the parallel branches of a captured HIP graph run concurrently on this runtime?  Two chains of `torch.cuda._sleep` kernels (one
block each, ~0.5 ms) on two forked streams, eager and captured: serial execution takes 2x the time of one chain."""
import os, sys, time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PAYLOAD = os.environ.get('PROBE', 'sleep')
_T = {}

def chain(n, cyc):
    if PAYLOAD == 'sleep':
        for _ in range(n):
            torch.cuda._sleep(cyc)
        return
    from diffusion_tts_amd import ops
    key = torch.cuda.current_stream().cuda_stream
    if key not in _T:
        r, c = (16, 576) if PAYLOAD == 'conv16' else (64, 192)
        x = torch.randn(4, r, r, c, device='cuda').to(torch.bfloat16)
        w = (torch.randn(c, 3, 3, c, device='cuda') / (9 * c) ** 0.5).to(torch.bfloat16)
        _T[key] = (x, w, torch.randn(c, device='cuda'), torch.empty_like(x))
    x, w, b, out = _T[key]
    for _ in range(n):
        ops.conv2d(x, w, b, out=out, gn_stats=True)

def two(n, cyc, side):
    cur = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(cur)
    side.wait_event(ev)
    with torch.cuda.stream(side):
        chain(n, cyc)
        j = torch.cuda.Event(); j.record(side)
    chain(n, cyc)
    cur.wait_event(j)

def timeit(fn, reps=5):
    torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3

def main():
    n, cyc = 20, 200_000
    side = torch.cuda.Stream()
    chain(2, cyc)
    with torch.cuda.stream(side):
        chain(2, cyc)
    torch.cuda.synchronize()
    one = timeit(lambda: chain(n, cyc))
    eager2 = timeit(lambda: two(n, cyc, side))
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        chain(n, cyc)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        two(n, cyc, side)
    print(f'payload {PAYLOAD} env PACKET_CAPTURE={os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE")} QUEUES={os.environ.get("DEBUG_HIP_FORCE_GRAPH_QUEUES")}: '
          f'one chain eager {one:.2f} ms, two chains eager {eager2:.2f} ms | graph: one chain {timeit(g1.replay):.2f} ms, two branches {timeit(g2.replay):.2f} ms')

main()
