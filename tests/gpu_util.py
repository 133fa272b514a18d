"""Helpers for the GPU parity tests: torch is only the device-memory allocator here."""
import numpy as np


class DeviceHelper:
    def __init__(self, torch):
        self.torch = torch
        self.device = torch.device("cuda:0")

    def to_device(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        return self.torch.from_numpy(a.view(np.int64).copy()).to(self.device)

    def empty(self, count):
        return self.torch.empty(int(count), dtype=self.torch.int64, device=self.device)

    def to_host(self, t):
        self.torch.cuda.synchronize()
        return t.cpu().numpy().view(np.uint64).copy()

    def sync(self):
        self.torch.cuda.synchronize()

    @property
    def stream(self):
        return self.torch.cuda.current_stream().cuda_stream


def rand_coeffs(rng, count, q, hi_mult=1):
    """uniform in [0, hi_mult*q) as uint64 (hi_mult up to 4: the lazy input range)"""
    hi = int(q) * hi_mult
    # numpy's integers() handles bounds up to 2^64 with dtype=uint64
    return rng.integers(0, hi, size=count, dtype=np.uint64)


def tables_for(orc, n, bits, count=1):
    """[(q, psi, tw, pre)] for the `count` largest primes below 2^bits"""
    out = []
    for k in range(count):
        q = orc.find_prime(bits, n, k)
        psi = orc.min_root(q, n)
        tw, pre = orc.make_tables(q, psi, n)
        out.append((q, psi, tw, pre))
    return out
