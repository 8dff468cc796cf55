"""graphs.GraphCache on the CPU: it must not keep its owner alive (a strong reference back would make every network cyclic garbage whose
captured graphs are released at an arbitrary later time -- e.g. inside another module's stream capture, which aborts the process)."""
import gc
import weakref

import pytest
import torch

from diffusion_tts_amd.graphs import GraphCache


class _Owner:
    def __init__(self):
        self.cache = GraphCache(self.forward, enabled=False)

    def forward(self, x):
        return x + 1


def test_owner_is_released_by_reference_counting_alone():
    was = gc.isenabled()
    gc.disable()
    try:
        m = _Owner()
        assert torch.equal(m.cache(torch.zeros(2)), torch.ones(2))
        r = weakref.ref(m)
        cache = m.cache
        del m
        assert r() is None                              # no cycle: gone without the cycle collector
        with pytest.raises(RuntimeError, match='released'):
            cache(torch.zeros(1))
    finally:
        if was:
            gc.enable()


def test_plain_functions_are_held():
    c = GraphCache(lambda x: x * 2, enabled=False)
    assert torch.equal(c(torch.ones(1)), torch.full((1,), 2.0))
