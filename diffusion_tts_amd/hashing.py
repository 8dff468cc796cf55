"""The epsilon-greedy step-size table.

The reference derives each candidate's step size from Python's builtin string hash
(`hash(f"{i}_{k}_{n}") % 1000 / 1000.0`, edm/main.py:776), so its results depend on PYTHONHASHSEED.
`builtin_scale` keeps that behaviour (drop-in).  `seed0_scale` reproduces, in any process, the table
the reference produces under PYTHONHASHSEED=0 (CPython then keys SipHash with an all-zero secret),
which is the setting the golden vectors under tests/golden were generated with.
"""
import sys

_M = (1 << 64) - 1


def _rotl(x, b):
    return ((x << b) | (x >> (64 - b))) & _M


def _siphash(data: bytes, k0: int, k1: int, c_rounds: int, d_rounds: int) -> int:
    v0 = k0 ^ 0x736F6D6570736575
    v1 = k1 ^ 0x646F72616E646F6D
    v2 = k0 ^ 0x6C7967656E657261
    v3 = k1 ^ 0x7465646279746573

    def rnd(v0, v1, v2, v3):
        v0 = (v0 + v1) & _M; v1 = _rotl(v1, 13); v1 ^= v0; v0 = _rotl(v0, 32)
        v2 = (v2 + v3) & _M; v3 = _rotl(v3, 16); v3 ^= v2
        v0 = (v0 + v3) & _M; v3 = _rotl(v3, 21); v3 ^= v0
        v2 = (v2 + v1) & _M; v1 = _rotl(v1, 17); v1 ^= v2; v2 = _rotl(v2, 32)
        return v0, v1, v2, v3

    n = len(data)
    b = (n & 0xFF) << 56
    full = n - (n % 8)
    for off in range(0, full, 8):
        m = int.from_bytes(data[off:off + 8], 'little')
        v3 ^= m
        for _ in range(c_rounds):
            v0, v1, v2, v3 = rnd(v0, v1, v2, v3)
        v0 ^= m
    b |= int.from_bytes(data[full:], 'little')
    v3 ^= b
    for _ in range(c_rounds):
        v0, v1, v2, v3 = rnd(v0, v1, v2, v3)
    v0 ^= b
    v2 ^= 0xFF
    for _ in range(d_rounds):
        v0, v1, v2, v3 = rnd(v0, v1, v2, v3)
    return (v0 ^ v1 ^ v2 ^ v3) & _M


def str_hash_seed0(s: str) -> int:
    """CPython `hash(s)` for an ASCII str when PYTHONHASHSEED=0 (siphash24 on 3.10, siphash13 on 3.11+)."""
    data = s.encode('ascii')
    if not data:
        return 0
    algo = sys.hash_info.algorithm
    if algo == 'siphash13':
        h = _siphash(data, 0, 0, 1, 3)
    elif algo == 'siphash24':
        h = _siphash(data, 0, 0, 2, 4)
    else:
        raise RuntimeError(f'unsupported str hash algorithm {algo!r}')
    if h >= 1 << 63:
        h -= 1 << 64
    return -2 if h == -1 else h


def builtin_scale(i: int, k: int, n: int) -> float:
    return hash(f"{i}_{k}_{n}") % 1000 / 1000.0


def seed0_scale(i: int, k: int, n: int) -> float:
    return str_hash_seed0(f"{i}_{k}_{n}") % 1000 / 1000.0
