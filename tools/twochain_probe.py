#!/usr/bin/env python3
"""FEASIBILITY PROBE (round 6, last session): the per-GPU share of an 8-GPU search is 8 candidates, and a forward of 8 rows runs at the
latency floor of its ~640 dependent kernels (t(rows) ~ 4.4 ms + 0.5 ms x rows).  Would TWO chains of 4 rows, each on its own half of the chip
(hipExtStreamCreateWithCUMask) and each replayed from its own host thread, finish sooner than one chain of 8 rows on the whole chip?
Measures the ADM-64 denoiser forward (f16x3, HIP-graph replay, same kernels as the product path), ms per forward:
    8 rows whole chip | 4 rows whole chip | 4 rows on half the chip alone | 2 x 4 rows side by side (masked halves / two unmasked streams)
Timing only: the product path is not touched.  Each chain captures its graph with its own split-K workspace."""
import ctypes as C
import os
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('DTS_GRAPHS', '0')                 # the probe captures its own graphs around _device_forward
import torch
from diffusion_tts_amd import init as dinit, ops
from diffusion_tts_amd.config import adm_imagenet64
from diffusion_tts_amd.networks import EDMPrecond

hip = C.CDLL('libamdhip64.so')
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = C.c_int


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[sum(1 << b for b in range(32) if bits[32 * w + b]) for w in range(8)])
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    assert rc == 0, f'hipExtStreamCreateWithCUMask -> {rc}'
    return torch.cuda.ExternalStream(st.value)


class Chain:
    """one captured forward of `rows` rows with private inputs and a private split-K workspace"""

    def __init__(self, net, rows, seed):
        dev = net.device
        g = torch.Generator().manual_seed(seed)
        self.x = torch.randn(rows, 3, 64, 64, generator=g, dtype=torch.float64).to(dev)
        self.sigma = torch.tensor([2.0], dtype=torch.float64, device=dev)
        self.lab = torch.eye(1000)[torch.arange(rows) % 1000].to(dev).contiguous()
        self.ws = torch.empty(ops.CONV_WS_BYTES, dtype=torch.uint8, device=dev)
        key = (dev.type, dev.index)
        ops._CONV_WS[key] = self.ws
        for _ in range(2):                                # eager: one-time kernel attributes, allocator warm-up
            net._device_forward(self.x, self.sigma, self.lab)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = net._device_forward(self.x, self.sigma, self.lab)
        torch.cuda.synchronize()

    def run(self, stream, reps):
        with torch.cuda.stream(stream):
            for _ in range(reps):
                self.graph.replay()


def wall(jobs, reps=6, rounds=5):
    """jobs: list of (chain, stream); every job replays its graph `reps` times from its OWN host thread; wall ms per replay (median round)"""
    ts = []
    for _ in range(rounds):
        torch.cuda.synchronize()
        go = threading.Barrier(len(jobs) + 1)

        def work(ch, st):
            go.wait()
            ch.run(st, reps)
            st.synchronize()
        th = [threading.Thread(target=work, args=j) for j in jobs]
        for t in th:
            t.start()
        go.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        ts.append((time.perf_counter() - t0) / reps * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    dev = torch.device('cuda', 0)
    cfg = adm_imagenet64()
    sd, _ = dinit.refill_degenerate(dinit.edm_state_dict(cfg, 0), 0)
    net = EDMPrecond(cfg, sd, device=dev, dtype=ops.F16X3)
    c8 = Chain(net, 8, 1)
    c4a, c4b = Chain(net, 4, 2), Chain(net, 4, 3)
    c2 = [Chain(net, 2, 10 + i) for i in range(4)]
    full = torch.cuda.Stream()
    t8, t4 = wall([(c8, full)]), wall([(c4a, full)])
    t2 = wall([(c2[0], full)])
    print(f'whole chip, one chain: 8 rows {t8:.2f} ms   4 rows {t4:.2f} ms   2 rows {t2:.2f} ms per forward', flush=True)
    u1, u2 = torch.cuda.Stream(), torch.cuda.Stream()
    print(f'two unmasked streams, 2 x 4 rows side by side: {wall([(c4a, u1), (c4b, u2)]):.2f} ms per pair of forwards (one chain of 8 rows: {t8:.2f})', flush=True)
    for name, bits_a in (('by XCD (XCDs 0-3 | 4-7)', [(k % 8) < 4 for k in range(256)]), ('half of every XCD', [(k // 8) < 16 for k in range(256)])):
        sa, sb = masked_stream(bits_a), masked_stream([not v for v in bits_a])
        alone = wall([(c4a, sa)])
        pair = wall([(c4a, sa), (c4b, sb)])
        print(f'{name}: 4 rows on half the chip alone {alone:.2f} ms | 2 x 4 rows on complementary halves {pair:.2f} ms per pair (one chain of 8 rows: {t8:.2f})', flush=True)
    q = [masked_stream([(k % 8) // 2 == i for k in range(256)]) for i in range(4)]      # four quarters: two XCDs each
    print(f'four quarters (2 XCDs each), 4 x 2 rows side by side: {wall([(c2[i], q[i]) for i in range(4)]):.2f} ms per four forwards (one chain of 8 rows: {t8:.2f})', flush=True)
    # one chain on 1 / 2 / 4 / 8 XCDs: does a latency-bound forward care how much of the chip it has?
    for nx in (1, 2, 4, 8):
        st = masked_stream([(k % 8) < nx for k in range(256)])
        print(f'one chain on {nx} XCD(s): 2 rows {wall([(c2[0], st)]):.2f}   4 rows {wall([(c4a, st)]):.2f}   8 rows {wall([(c8, st)]):.2f} ms per forward', flush=True)
    # host side of one replay: time until graph.replay() returns (the GPU idle before it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(full):
        c8.graph.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f'host time of one replay() of the 8-row graph: {(t1 - t0) * 1e3:.2f} ms (GPU idle before the call)', flush=True)
    # same results whichever stream replays the graph
    ref = c4a.out.clone()
    c4a.run(masked_stream([(k % 8) < 4 for k in range(256)]), 1)
    torch.cuda.synchronize()
    print('4-row output identical on a masked stream:', bool(torch.equal(ref, c4a.out)))


if __name__ == '__main__':
    main()
