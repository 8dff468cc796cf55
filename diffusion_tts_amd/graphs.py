"""HIP-graph replay of the fixed-shape device forwards (the denoiser and the scorer's classifier).

One search iteration enqueues ~1000 kernels from Python (~30 us each on the host), and the host-side argmax at the end of
every iteration (edm/main.py:842) drains the queue, so each iteration restarts launch-bound: at N=64 the GPU idled ~3 ms of
every 44 ms, and at the small per-GPU batches of a sharded search the host, not the GPU, set the pace.  The forward of a given
(batch, dtype) is a fixed kernel sequence with fixed shapes, so it is captured once (stream capture of the very same C-ABI
launches) and replayed with one launch; inputs are copied into the captured input buffers, the output is copied out.
Shapes of at most SMALL_ROWS rows (the ragged rollout batches of an MCTS search: 1 .. 16 rows) have their own, larger budget: a small
forward runs at its launch floor (~620 dependent kernels), so there every row of padding and every eager forward is pure loss, while a
small graph's private pool is small.

Not a tracing compiler: nothing is rewritten or fused, the graph is the eager launch sequence.  The first calls of a shape run
eagerly (they also perform the one-time `hipFuncSetAttribute` calls), the third is captured; a module stops capturing new
shapes after 8 captures (MCTS rollouts change their batch size all the time).  `DTS_GRAPHS=0` disables it.
A refused capture (an out-of-memory private pool on the Nth ragged MCTS batch size, say) must not abort a long search -- nor, in a sharded
run, leave the other ranks waiting in the next collective: the SHAPE that was refused stays on eager launches (same kernels), a warning says
so once, `GraphCache.fallback` records why and `path_report()` shows it (main.py prints it).  A measurement must not report eager-launch
numbers under the graph-replay name: `DTS_GRAPHS_STRICT=1` (set by bench.py, the tools and the test-suite) turns a refusal into an error."""
import gc
import math
import os
import warnings
import weakref
from collections import OrderedDict

import torch

ENABLED = os.environ.get('DTS_GRAPHS', '1') != '0'
MAX_GRAPHS = 4                      # per module; each holds the activations of one forward in its private pool
MAX_CAPTURES = 8                    # per module: a caller with ever-changing batch sizes stays eager after that
SMALL_ROWS = 16                     # shapes with at most this many rows count against the SMALL budget instead
MAX_GRAPHS_SMALL = 16
MAX_CAPTURES_SMALL = 24
SIGHTINGS = 3                       # a shape is captured on its third call


class GraphCache:
    def __init__(self, fn, enabled=None):
        # a bound method is held weakly: the owner keeps this cache, and a strong reference back would make every network cyclic
        # garbage -- its captured graphs and their memory pools would then be released whenever the cycle collector happens to run,
        # possibly in the middle of ANOTHER graph's stream capture, where releasing a graph aborts the process
        self._fn = weakref.WeakMethod(fn) if hasattr(fn, '__self__') else (lambda: fn)
        self.enabled = ENABLED if enabled is None else enabled
        self.seen = {}
        self.graphs = OrderedDict()
        self.replays = 0
        self.captures = 0
        self.captures_by_class = [0, 0]  # [large, small]
        self.eager = 0                  # forwards launched kernel by kernel (first sightings of a shape, shapes beyond MAX_CAPTURES, graphs off)
        self.fallback = None            # why a capture was refused (the last refusal; without DTS_GRAPHS_STRICT=1)
        self.refused = set()            # shapes whose capture was refused: they stay on eager launches

    @property
    def fn(self):
        f = self._fn()
        if f is None:
            raise RuntimeError('GraphCache: the module this cache replays has been released')
        return f

    def __call__(self, *inputs):
        """inputs: device tensors (or None); returns a tensor the caller owns."""
        if not self.enabled:
            self.eager += 1
            return self.fn(*inputs)
        key = tuple(None if t is None else (tuple(t.shape), t.dtype, t.device.index) for t in inputs)
        ent = self.graphs.get(key)
        if ent is None:
            self.seen[key] = self.seen.get(key, 0) + 1
            small = self._small(key)
            if key in self.refused or self.seen[key] < SIGHTINGS or self.captures_by_class[small] >= (MAX_CAPTURES_SMALL if small else MAX_CAPTURES):
                self.eager += 1
                return self.fn(*inputs)
            ent = self._capture(key, inputs)
            if ent is None:
                self.eager += 1
                return self.fn(*inputs)
        graph, static_in, static_out = ent
        for s, t in zip(static_in, inputs):
            if s is not None:
                s.copy_(t)
        graph.replay()
        self.replays += 1
        return static_out.clone()

    @staticmethod
    def _small(key):
        # at most SMALL_ROWS rows of at most 3 x 128 x 128 elements each: a graph's private pool stays a few hundred MB even for the widest net
        shapes = [k[0] for k in key if k is not None and len(k[0]) > 0]
        rows = max((sh[0] for sh in shapes), default=0)
        per_row = max((math.prod(sh[1:]) for sh in shapes), default=0)
        return int(0 < rows <= SMALL_ROWS and per_row <= 3 * 128 * 128)

    def _capture(self, key, inputs):
        static_in = [None if t is None else t.clone() for t in inputs]
        fn = self.fn
        # no finaliser may run inside the capture (a collected graph / event / pool of some other module would issue HIP calls that are
        # illegal while a stream captures): collect what is collectable now, then keep the cycle collector off until the capture ends
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = fn(*static_in)
        except Exception as e:                       # capture refused: stay on the eager launch sequence (same kernels)
            if os.environ.get('DTS_GRAPHS_STRICT', '0') == '1':
                raise RuntimeError(f'HIP-graph capture failed ({type(e).__name__}: {e}); DTS_GRAPHS_STRICT=1: not continuing on eager launches '
                                   f'(same kernels, several times slower at small batches)') from e
            warnings.warn(f'HIP-graph capture of shape {key} failed ({type(e).__name__}: {e}): this shape continues on eager launches '
                          f'(DTS_GRAPHS_STRICT=1 makes this an error)')
            self.refused.add(key)
            self.fallback = f'{type(e).__name__}: {e}'
            return None
        finally:
            if gc_was_on:
                gc.enable()
        self.captures += 1
        small = self._small(key)
        self.captures_by_class[small] += 1
        mine = [k for k in self.graphs if self._small(k) == small]           # insertion order: oldest first
        while len(mine) >= (MAX_GRAPHS_SMALL if small else MAX_GRAPHS):
            del self.graphs[mine.pop(0)]
        self.graphs[key] = (graph, static_in, static_out)
        return self.graphs[key]

    def path_report(self):
        """which path this module's forwards took so far"""
        how = 'HIP-graph replay' if self.replays else ('eager launches' if self.eager else 'no forward yet')
        extra = f'; capture refused for {len(self.refused)} shape(s): {self.fallback}' if self.fallback else ('' if self.enabled else '; graphs disabled')
        return f'{how} ({self.replays} replays of {self.captures} captured graphs, {self.eager} eager forwards{extra})'
