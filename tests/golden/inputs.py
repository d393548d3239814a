"""Seeded generators of the golden inputs.  numpy 2.2 default_rng streams."""
import numpy as np


def rand1m():
    return np.random.default_rng(12345).integers(0, 255, 1 << 20, dtype=np.uint8)


def per3():
    return np.frombuffer((b"abc" * 22000)[:65536], dtype=np.uint8).copy()


def alla():
    return np.full(65536, ord("a"), np.uint8)


def fib():
    a, b = b"a", b"ab"
    while len(b) < 65536:
        a, b = b, b + a
    return np.frombuffer(b[:65536], dtype=np.uint8).copy()


def sig4_with_zero():
    return np.random.default_rng(3).integers(0, 4, 1 << 16, dtype=np.uint8)


def rand64m():
    return np.random.default_rng(7).integers(0, 255, 64 << 20, dtype=np.uint8)


# the fixtures every suite iterates over; LARGE ones are only run end to end on the GPU (tests/test_host.py)
LARGE_GENERATORS = {"rand64m": rand64m}
GENERATORS = {"rand1m": rand1m, "per3": per3, "alla": alla, "fib": fib, "sig4_with_zero": sig4_with_zero}
