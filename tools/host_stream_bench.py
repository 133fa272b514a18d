"""tools/host_stream_bench.py -- SURVEY 8(f3): what the host<->device streaming pipeline (agx_ntt_forward_host_stream:
three pinned/device slots on three HIP streams, the GPU analogue of the reference's ntt_input_kernel /
ntt_output_kernel streaming, src/kernel/ntt.cpp:508-640) achieves on >= 1 GiB of host frames, against
  (a) the transfers alone  : H2D of the frames + D2H of the results from pinned memory (sequential and on two streams),
  (b) the serial schedule  : H2D everything -> one transform launch -> D2H everything (pinned, no overlap),
  (c) the host-side staging: the memcpy of pageable caller memory into / out of the pinned slots, which the
      pipeline also has to do (the ABI takes plain host pointers).
Usage: python tools/host_stream_bench.py [--n 4096 --frames 32768]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import agilex_ntt_amd as agx  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--frames", type=int, default=32768)
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
n, frames = args.n, args.frames
q = agx.find_primes(60, n, 1)[0]
plan = agx.Plan(n, [q])
nbytes = frames * n * 8
gib = nbytes / 2**30
rng = np.random.default_rng(1)
x = rng.integers(0, q, size=frames * n, dtype=np.uint64)


def best(fn, reps=args.reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts)


# the product path
out = plan.forward_host_stream(x, x, frames)        # warm-up (first touch of the pinned pools, clocks); also touches `out`
t_pipe = best(lambda: plan.forward_host_stream(x, x, frames, out=out))
t_pipe_fresh = best(lambda: plan.forward_host_stream(x, x, frames))   # output array allocated (and page-faulted) per call

# reference result for a spot check: device transform of the same frames
d = torch.from_numpy(x.view(np.int64)).cuda()
plan.forward(d.data_ptr(), d.data_ptr(), frames, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
assert np.array_equal(d.cpu().numpy().view(np.uint64), out), "pipeline result differs from the device-resident transform"

# (a) transfers alone, pinned
pin_in = torch.from_numpy(x.view(np.int64)).pin_memory()
pin_out = torch.empty_like(pin_in).pin_memory()
dev = torch.empty(frames * n, dtype=torch.int64, device="cuda")
dev2 = torch.empty_like(dev)


def seq_copy():
    dev.copy_(pin_in, non_blocking=True)
    pin_out.copy_(dev, non_blocking=True)
    torch.cuda.synchronize()


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def bidir_copy():
    with torch.cuda.stream(s1):
        dev.copy_(pin_in, non_blocking=True)
    with torch.cuda.stream(s2):
        pin_out.copy_(dev2, non_blocking=True)
    torch.cuda.synchronize()


seq_copy()
t_seq = best(seq_copy)
bidir_copy()
t_bidir = best(bidir_copy)


# (b) serial schedule with pinned memory: H2D all -> transform -> D2H all
def serial():
    dev.copy_(pin_in, non_blocking=True)
    plan.forward(dev.data_ptr(), dev.data_ptr(), frames, torch.cuda.current_stream().cuda_stream)
    pin_out.copy_(dev, non_blocking=True)
    torch.cuda.synchronize()


serial()
t_serial = best(serial)

# (c) host staging alone: pageable -> pinned and pinned -> pageable memcpy of the same bytes
pin_np = pin_in.numpy()
host_out = np.empty_like(x.view(np.int64))


def staging():
    np.copyto(pin_np, x.view(np.int64))
    np.copyto(host_out, pin_np)


staging()
t_stage = best(staging)

moved = 2 * gib     # frames in + results out
print(f"n={n}, {frames} frames = {gib:.2f} GiB in + {gib:.2f} GiB out, one 60-bit modulus")
print(f"  agx_ntt_forward_host_stream (pageable host pointers, 3 slots x 3 streams): {t_pipe*1e3:8.1f} ms  {moved/t_pipe:6.2f} GiB/s moved  {frames/t_pipe/1e6:6.3f} M NTT/s")
print(f"  the same into a freshly allocated (untouched) output array:                {t_pipe_fresh*1e3:8.1f} ms  {moved/t_pipe_fresh:6.2f} GiB/s moved")
print(f"  (a) transfers only, pinned, H2D then D2H on one stream:                  {t_seq*1e3:8.1f} ms  {moved/t_seq:6.2f} GiB/s")
print(f"  (a') transfers only, pinned, H2D and D2H on two streams:                  {t_bidir*1e3:8.1f} ms  {moved/t_bidir:6.2f} GiB/s")
print(f"  (b) serial H2D -> transform -> D2H, pinned, one stream:                   {t_serial*1e3:8.1f} ms  {moved/t_serial:6.2f} GiB/s")
print(f"  (c) host staging alone (pageable -> pinned + pinned -> pageable memcpy):  {t_stage*1e3:8.1f} ms  {moved/t_stage:6.2f} GiB/s")
print(f"  pipeline / bidirectional-transfer ceiling = {t_bidir/t_pipe:.2f}; pipeline vs serial schedule + staging = {(t_serial + t_stage)/t_pipe:.2f}x")
plan.close()
