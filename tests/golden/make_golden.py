#!/usr/bin/env python3
"""Regenerates the golden fixtures in this directory.  Run from the repository root:

    python tests/golden/make_golden.py

The reference (joekurina/Agilex-NTT) ships no test vectors and cannot be built in this image
(it needs the oneAPI SYCL headers), so the vectors come from the CPU oracle
(oracle/ntt_oracle.c, a restatement of src/kernel/ntt.cpp's butterfly) and every vector is
cross-checked here against an independent O(n^2) evaluation of the closed-form contract
before it is written.  survey_anchors.json is NOT generated: it transcribes the values
SURVEY.md section 8c recorded from running the reference's own kernel code.

Input recipe: x[i] = splitmix64(state) % q drawn sequentially, state = seed;
primes = largest prime below 2^bits with q = 1 (mod 2n); psi = least primitive 2n-th root.
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import oracle as orc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def case(n, bits, frames, seed=42, full=False, naive_frames=1):
    q = orc.find_prime(bits, n)
    psi = orc.min_root(q, n)
    tw, pre = orc.make_tables(q, psi, n)
    x = orc.fill_splitmix(n * frames, seed, q)
    y = orc.forward(x, q, tw, pre, n)
    for f in range(min(frames, naive_frames)):  # second opinion
        ref = orc.naive_forward(x[f * n:(f + 1) * n], q, psi, n)
        assert np.array_equal(ref, y[f * n:(f + 1) * n]), (n, bits, f)
    d = {"n": n, "bits": bits, "q": str(q), "psi": str(psi), "frames": frames, "seed": seed,
         "fnv1a_words": "%016x" % orc.fnv1a_words(y),
         "first8": [str(int(v)) for v in y[:8]], "last8": [str(int(v)) for v in y[-8:]]}
    if full:
        d["input"] = [str(int(v)) for v in x]
        d["output"] = [str(int(v)) for v in y]
    return d


def main():
    cases = [
        case(32, 30, 2, full=True, naive_frames=2),
        case(1024, 30, 3, full=True, naive_frames=3),
        case(1024, 30, 1),
        case(4096, 60, 2, naive_frames=2),
        case(8192, 61, 1),
        case(16384, 60, 1),
        case(32768, 60, 1),
    ]
    with open(os.path.join(HERE, "forward_vectors.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py (oracle + naive cross-check)", "cases": cases}, f, indent=1)
    print("wrote forward_vectors.json with", len(cases), "cases")


if __name__ == "__main__":
    main()
