"""CPU, world_size 2 over gloo: the candidate-sharding layer (parallel.CandidateShards) used by the N>1 path.
Checks the partition, the ONE reward all-gather per iteration (even and ragged N), that every rank takes the same
first-max argmax as the unsharded loop, and the owner broadcast used by rejection sampling."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q, cases=((8, 1), (7, 2), (64, 1), (3, 1))):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from diffusion_tts_amd.parallel import CandidateShards
        sh = CandidateShards()
        assert (sh.rank, sh.world) == (rank, world)
        out = {}
        for N, B in cases:
            spans = [sh.span(N, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == N and all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            g = torch.Generator().manual_seed(100 + N)                  # replicated "host RNG"
            full = torch.rand(N, B, generator=g)                        # rewards of all candidates (candidate-major)
            full[N // 2] = full.max() + 0.0                             # make an exact tie with the max -> first-max rule matters
            lo, hi = sh.span(N)
            before = sh.collectives
            got = sh.gather_rewards(full[lo:hi].reshape(-1).clone(), N, B)
            assert sh.collectives == before + 1                        # exactly one collective per search iteration
            assert torch.equal(got.reshape(N, B), full)
            best = got.reshape(N, B).argmax(dim=0)
            assert torch.equal(best, full.argmax(dim=0))
            out[(N, B)] = best.tolist()
            # rejection: the winner's image goes once from its owner to everyone
            win = int(best[0])
            img = torch.full((3, 4, 4), float(win)) if lo <= win < hi else torch.zeros(3, 4, 4)
            img = sh.broadcast_from_owner(img, win, N)
            assert torch.equal(img, torch.full((3, 4, 4), float(win)))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_candidate_sharding_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res = dict(res)
    assert res[0] == res[1]                                              # identical survivor on every rank


def test_candidate_sharding_world8_gloo():
    """The rank count of BASELINE config 3 (N = 64 candidates over 8 GPUs, 8 each), rehearsed over gloo on the CPU: contiguous spans that
    cover the candidates, ONE reward collective per decision, the same first-max survivor on all 8 ranks, the rejection winner broadcast
    from whichever rank owns it; also an uneven split (13 candidates) and a batch of two images."""
    world = 8
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, ((64, 1), (8, 1), (13, 1), (16, 2)))) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res = dict(res)
    assert len(res) == world and all(res[r] == res[0] for r in range(world))


def test_single_process_is_a_noop():
    from diffusion_tts_amd.parallel import CandidateShards
    sh = CandidateShards()
    assert (sh.rank, sh.world) == (0, 1) and sh.span(5) == (0, 5)
    x = torch.arange(5.0)
    assert sh.gather_rewards(x, 5, 1) is x and sh.collectives == 0


def _bulk_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from diffusion_tts_amd.bulk import rank_batches
        from diffusion_tts_amd.parallel import CandidateShards
        mine = [s for b in rank_batches(range(10, 31), 4, dist.get_rank(), dist.get_world_size()) for s in b.tolist()]
        off = CandidateShards(enabled=False)                              # whole searches per rank: no candidate sharding
        assert (off.rank, off.world, off.enabled) == (0, 1, False)
        loc = torch.arange(5, dtype=torch.float32)
        assert off.gather_rewards(loc, 5, 1) is loc and off.collectives == 0
        q.put((rank, mine))
    finally:
        dist.destroy_process_group()


def test_seed_sharding_world2_gloo():
    """bulk generation: the seeds are split like edm/generate.py:259-261 (tensor_split, rank::world); together the ranks cover
    every seed exactly once, with no collective on the data path."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bulk_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(got[0] + got[1]) == list(range(10, 31)) and not set(got[0]) & set(got[1])
    from diffusion_tts_amd.bulk import rank_batches, parse_int_list
    ref = torch.as_tensor(list(range(10, 31))).tensor_split(((21 - 1) // (4 * 2) + 1) * 2)
    assert got[0] == [s for b in ref[0::2] for s in b.tolist()] and got[1] == [s for b in ref[1::2] for s in b.tolist()]
    assert parse_int_list('1,2,5-8') == [1, 2, 5, 6, 7, 8]
    assert [b.tolist() for b in rank_batches([3, 4, 5], 64, 0, 1)] == [[3, 4, 5]]


def _scale_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import numpy as np
        from diffusion_tts_amd.hashing import builtin_scale
        from diffusion_tts_amd.parallel import CandidateShards
        sh = CandidateShards()
        own = [builtin_scale(i, k, n) for i in range(2) for k in range(2) for n in range(4)]     # salted per process
        fn = sh.replicate_scale_table(builtin_scale, 2, 2, 4)
        shared = [fn(i, k, n) for i in range(2) for k in range(2) for n in range(4)]
        np.random.seed(1000 + rank)                                       # ranks start with different numpy streams ...
        sh.sync_numpy_rng()                                               # ... and adopt rank 0's (edm/main.py:593)
        draws = np.random.randint(0, 1000, size=4).tolist()
        err = None
        try:
            sh.require_candidates(1, 'test')                              # N=1 < 2 ranks: same error on every rank, no hang
        except ValueError as e:
            err = str(e)
        sh.require_candidates(2, 'test')
        q.put((rank, own, shared, draws, err))
    finally:
        dist.destroy_process_group()


def test_hash_scale_table_and_numpy_rng_are_replicated_from_rank0():
    """the default `scale_fn` is Python's per-process salted hash(); ranks launched by torch.distributed.run
    disagree on it.  The sharded loop must use rank 0's table everywhere (== the single-process run)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = []
    old = os.environ.get('PYTHONHASHSEED')
    try:
        for r in range(2):
            os.environ['PYTHONHASHSEED'] = str(11 + r)                    # what independent launches give: different salts
            p = ctx.Process(target=_scale_worker, args=(r, 2, port, q))
            p.start()
            procs.append(p)
    finally:
        if old is None:
            os.environ.pop('PYTHONHASHSEED', None)
        else:
            os.environ['PYTHONHASHSEED'] = old
    got = {r[0]: r[1:] for r in (q.get(timeout=120) for _ in procs)}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][0] != got[1][0]                                         # the hazard is real: the salted tables differ
    assert got[0][1] == got[1][1] == got[0][0]                            # both ranks use rank 0's table
    assert got[0][2] == got[1][2]                                         # same numpy child picks
    assert got[0][3] == got[1][3] and 'N=1' in got[0][3]
