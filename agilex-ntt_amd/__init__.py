"""agilex-ntt_amd: MI355X-native batched negacyclic NTT engine -- thin Python host layer.

The product is the C-ABI shared library ``lib/libagxntt.so`` (``include/agx_ntt.h``), built
from ``csrc/`` by hipcc for gfx950.  This module only binds that ABI with ctypes so that
bench.py and the tests can drive it from Python with torch providing device memory and
streams.  There is no CPU fallback: if the library is missing, loading fails loudly.

The directory name has a hyphen (it mirrors the reference repository's name), so import it
through the ``agilex_ntt_amd`` shim at the repository root.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AGX_NTT_LIB") or os.path.join(_HERE, "lib", "libagxntt.so")   # override: A/B builds in tools/
INCLUDE_DIR = os.path.join(os.path.dirname(_HERE), "include")

AGX_OK = 0
VARIANT_AUTO, VARIANT_LDS_RADIX2, VARIANT_REGBLOCK = 0, 1, 2
VARIANT_REGBLOCK_BASE = 256  # + registry index: A/B measurements only

_u64 = ctypes.c_uint64
_u32 = ctypes.c_uint32
_i64 = ctypes.c_int64
_int = ctypes.c_int
_vp = ctypes.c_void_p
_p64 = ctypes.POINTER(ctypes.c_uint64)

# name -> (restype, argtypes): every symbol include/agx_ntt.h declares
ABI = {
    "agx_ntt_strerror": (ctypes.c_char_p, [_int]),
    "agx_ntt_last_hip_error": (_int, []),
    "agx_ntt_device_count": (_int, [ctypes.POINTER(_int)]),
    "agx_ntt_forward_host": (_int, [_p64, _p64, _p64, _p64, _p64, _p64, _u32, _u32]),
    "agx_ntt_forward_host_stream": (_int, [_vp, _p64, _p64, _p64, _u64]),
    "agx_ntt_inverse_host_stream": (_int, [_vp, _p64, _p64, _u64]),
    "agx_ntt_release_caches": (_int, []),
    "agx_ntt_plan_create": (_int, [ctypes.POINTER(_vp), _u32, _u32, _p64, _p64, _p64, _p64, _p64]),
    "agx_ntt_plan_create_auto": (_int, [ctypes.POINTER(_vp), _u32, _u32, _p64, _p64]),
    "agx_ntt_plan_destroy": (_int, [_vp]),
    "agx_ntt_plan_set_variant": (_int, [_vp, _int]),
    "agx_ntt_plan_info": (_int, [_vp, ctypes.POINTER(_u32), ctypes.POINTER(_u32), ctypes.POINTER(_int), ctypes.POINTER(_int)]),
    "agx_ntt_plan_get_modulus": (_int, [_vp, _u32, _p64, _p64]),
    "agx_ntt_forward": (_int, [_vp, _vp, _vp, _u64, _vp]),
    "agx_ntt_forward_lazy": (_int, [_vp, _vp, _vp, _u64, _vp]),
    "agx_ntt_inverse": (_int, [_vp, _vp, _vp, _u64, _vp]),
    "agx_ntt_forward_strided": (_int, [_vp, _vp, _vp, _u64, _i64, _i64, _vp]),
    "agx_ntt_inverse_strided": (_int, [_vp, _vp, _vp, _u64, _i64, _i64, _vp]),
    "agx_ntt_pointwise": (_int, [_vp, _vp, _vp, _vp, _u64, _vp]),
    "agx_ntt_polymul": (_int, [_vp, _vp, _vp, _vp, _vp, _u64, _vp]),
    "agx_ntt_fill_synthetic": (_int, [_vp, _vp, _u64, _u64, _u64, _vp]),
    "agx_ntt_find_primes": (_int, [_u32, _u32, _u32, _p64]),
    "agx_ntt_min_root": (_int, [_u64, _u32, _p64]),
    "agx_ntt_make_tables": (_int, [_u64, _u64, _u32, _p64, _p64]),
    "agx_ntt_make_inverse_tables": (_int, [_u64, _u64, _u32, _p64, _p64]),
    # groups: the same calls over several GPUs (one shard, one host thread, one stream per listed device)
    "agx_ntt_shard_range": (_int, [_u64, _u32, _u32, _p64, _p64]),
    "agx_ntt_group_create": (_int, [ctypes.POINTER(_vp), ctypes.POINTER(_int), _u32, _u32, _u32, _p64, _p64, _p64, _p64, _p64]),
    "agx_ntt_group_create_auto": (_int, [ctypes.POINTER(_vp), ctypes.POINTER(_int), _u32, _u32, _u32, _p64, _p64]),
    "agx_ntt_group_destroy": (_int, [_vp]),
    "agx_ntt_group_info": (_int, [_vp, ctypes.POINTER(_u32), ctypes.POINTER(_u32), ctypes.POINTER(_u32)]),
    "agx_ntt_group_shard": (_int, [_vp, _u32, ctypes.POINTER(_int), ctypes.POINTER(_vp), ctypes.POINTER(_vp)]),
    "agx_ntt_group_forward_host": (_int, [_vp, _p64, _p64, _p64, _u64]),
    "agx_ntt_group_inverse_host": (_int, [_vp, _p64, _p64, _u64]),
    "agx_ntt_group_forward": (_int, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _p64]),
    "agx_ntt_group_inverse": (_int, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), _p64]),
    "agx_ntt_group_polymul": (_int, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), _p64]),
    "agx_ntt_group_synchronize": (_int, [_vp]),
}


class AgxError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        msg = lib().agx_ntt_strerror(status).decode()
        super().__init__(f"{where}: agx status {status} ({msg}), hip error {lib().agx_ntt_last_hip_error()}")


def build(verbose=False, with_diag=False):
    """Compile the gfx950 library in-tree (hipcc cross-compiles without a GPU).  with_diag: lib/libagxntt_diag.so in the same make
    invocation, so the objects of both libraries compile side by side."""
    cmd = ["make", "-C", _HERE, f"-j{min(8, os.cpu_count() or 1)}", "build"] + (["diag"] if with_diag else [])
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7) and ask for
    it by file name, so a process that loaded /opt/rocm's copy first ends up with two HIP
    runtimes: torch's device pointers and streams would then mean nothing to this library.
    Loading torch's copy by path before libagxntt.so makes both resolve to the same runtime,
    whatever the import order.  Without torch installed the system runtime is used."""
    import importlib.util

    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)


def lib():
    """Load libagxntt.so.  Raises if it has not been built: there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `make -C {_HERE} build` (or __graft_entry__.build())")
        _share_hip_runtime_with_torch()
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in ABI.items():
            fn = getattr(L, name)  # AttributeError if the ABI and the header drift apart
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(status, where):
    if status != AGX_OK:
        raise AgxError(status, where)


def _np_ptr(a):
    import numpy as np

    assert isinstance(a, np.ndarray) and a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_p64)


def kernel_source_sha16():
    """sha256[:16] over the DEVICE sources the kernels are built from (csrc/*.hip and the headers they include, sorted by
    name; not the host-side agx_ntt.cpp / host_math.*): ties a committed PMC figure (profiles/hbm_traffic.json) to the
    kernels it was measured on"""
    import glob
    import hashlib

    h = hashlib.sha256()
    files = glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + [os.path.join(_HERE, "csrc", f) for f in ("rb_frame.hpp", "rb_kernels.hpp", "rb32_kernels.hpp", "wp_kernels.hpp", "rb_stream_opts.hpp", "rb_registry.hpp", "modarith.hpp", "ntt_kernels.hpp")]
    for f in sorted(files):
        h.update(os.path.relpath(f, _HERE).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


DIAG_LIB_PATH = os.path.join(_HERE, "lib", "libagxntt_diag.so")   # `make diag`: product + diagnostics kernels (tools/agx_ntt_diag.h)


def build_diag():
    """Compile lib/libagxntt_diag.so (trace twin, streaming A/B kernels); load it with AGX_NTT_LIB=DIAG_LIB_PATH."""
    subprocess.check_call(["make", "-s", "-C", _HERE, f"-j{min(8, os.cpu_count() or 1)}", "diag"])
    return DIAG_LIB_PATH


def debug_set_trace_buffer(d_buf, nbytes):
    """Diagnostics (tools/timeline.py), diag library only: where the registry's trace kernel writes its phase
    stamps; (0, 0) = off.  The product library does not export the hook."""
    try:
        fn = lib().agx_ntt_debug_set_trace_buffer
    except AttributeError:
        raise RuntimeError("agx_ntt_debug_set_trace_buffer is only in lib/libagxntt_diag.so: `make -C agilex-ntt_amd diag` "
                           "and run with AGX_NTT_LIB=<that file>") from None
    fn.restype, fn.argtypes = _int, [_vp, _u64]
    _check(fn(d_buf, nbytes), "debug_set_trace_buffer")


def device_count():
    c = _int(0)
    _check(lib().agx_ntt_device_count(ctypes.byref(c)), "device_count")
    return c.value


# ---------------------------------------------------------------------------------------
# host math
# ---------------------------------------------------------------------------------------
def find_primes(bits, n, count=1):
    import numpy as np

    out = np.zeros(count, dtype=np.uint64)
    _check(lib().agx_ntt_find_primes(bits, n, count, _np_ptr(out)), "find_primes")
    return [int(v) for v in out]


def min_root(q, n):
    r = _u64(0)
    _check(lib().agx_ntt_min_root(q, n, ctypes.byref(r)), "min_root")
    return r.value


def make_tables(q, psi, n, inverse=False):
    import numpy as np

    tw = np.zeros(n, dtype=np.uint64)
    pre = np.zeros(n, dtype=np.uint64)
    fn = lib().agx_ntt_make_inverse_tables if inverse else lib().agx_ntt_make_tables
    _check(fn(q, psi, n, _np_ptr(tw), _np_ptr(pre)), "make_tables")
    return tw, pre


# ---------------------------------------------------------------------------------------
# one-shot host path: the reference's ntt_input_kernel + fwd_ntt_kernel + ntt_output_kernel
# ---------------------------------------------------------------------------------------
def forward_host(in1, in2, modulus, twiddles, precons, n, num_frames):
    import numpy as np

    out = np.zeros(num_frames * n, dtype=np.uint64)
    mod = np.array([modulus], dtype=np.uint64)
    _check(lib().agx_ntt_forward_host(_np_ptr(in1), _np_ptr(in2), _np_ptr(mod), _np_ptr(twiddles), _np_ptr(precons),
                                      _np_ptr(out), n, num_frames), "forward_host")
    return out


# ---------------------------------------------------------------------------------------
# plans and device-pointer calls (pointers are plain integers, e.g. torch.Tensor.data_ptr())
# ---------------------------------------------------------------------------------------
class Plan:
    """Device-resident tables for `moduli` at size n.  Tables are generated by the library
    unless `tables=(tw, pre[, itw, ipre])` (each [num_primes, n] uint64) is given."""

    def __init__(self, n, moduli, psi=None, tables=None):
        import numpy as np

        self.n = int(n)
        self.moduli = [int(q) for q in moduli]
        mods = np.array(self.moduli, dtype=np.uint64)
        self._h = _vp(None)
        if tables is None:
            psi_arr = None if psi is None else np.array([int(p) for p in psi], dtype=np.uint64)
            _check(lib().agx_ntt_plan_create_auto(ctypes.byref(self._h), self.n, len(self.moduli), _np_ptr(mods),
                                                  None if psi_arr is None else _np_ptr(psi_arr)), "plan_create_auto")
        else:
            tw, pre = (np.ascontiguousarray(t, dtype=np.uint64) for t in tables[:2])
            itw = ipre = None
            if len(tables) == 4:
                itw, ipre = (np.ascontiguousarray(t, dtype=np.uint64) for t in tables[2:])
            _check(lib().agx_ntt_plan_create(ctypes.byref(self._h), self.n, len(self.moduli), _np_ptr(mods), _np_ptr(tw),
                                             _np_ptr(pre), None if itw is None else _np_ptr(itw),
                                             None if ipre is None else _np_ptr(ipre)), "plan_create")

    @property
    def num_primes(self):
        return len(self.moduli)

    def psi(self, k):
        q, r = _u64(0), _u64(0)
        _check(lib().agx_ntt_plan_get_modulus(self._h, k, ctypes.byref(q), ctypes.byref(r)), "plan_get_modulus")
        return r.value

    def set_variant(self, variant):
        _check(lib().agx_ntt_plan_set_variant(self._h, variant), "plan_set_variant")

    def forward(self, d_in, d_out, batch, stream=0):
        _check(lib().agx_ntt_forward(self._h, d_in, d_out, batch, stream), "forward")

    def forward_lazy(self, d_in, d_out, batch, stream=0):
        _check(lib().agx_ntt_forward_lazy(self._h, d_in, d_out, batch, stream), "forward_lazy")

    def inverse(self, d_in, d_out, batch, stream=0):
        _check(lib().agx_ntt_inverse(self._h, d_in, d_out, batch, stream), "inverse")

    def forward_strided(self, d_in, d_out, batch, prime_stride, poly_stride, stream=0):
        _check(lib().agx_ntt_forward_strided(self._h, d_in, d_out, batch, prime_stride, poly_stride, stream), "forward_strided")

    def inverse_strided(self, d_in, d_out, batch, prime_stride, poly_stride, stream=0):
        _check(lib().agx_ntt_inverse_strided(self._h, d_in, d_out, batch, prime_stride, poly_stride, stream), "inverse_strided")

    def pointwise(self, d_a, d_b, d_c, batch, stream=0):
        _check(lib().agx_ntt_pointwise(self._h, d_a, d_b, d_c, batch, stream), "pointwise")

    def polymul(self, d_a, d_b, d_c, d_scratch, batch, stream=0):
        _check(lib().agx_ntt_polymul(self._h, d_a, d_b, d_c, d_scratch, batch, stream), "polymul")

    def fill_synthetic(self, d_out, batch, first_poly=0, seed=42, stream=0):
        _check(lib().agx_ntt_fill_synthetic(self._h, d_out, batch, first_poly, seed, stream), "fill_synthetic")

    def forward_host_stream(self, in1, in2, num_frames, out=None):
        """host frames through this (single-modulus) plan with overlapped transfers; `out` (optional) is a caller-owned
        uint64 array of num_frames*n elements (a fresh np.zeros array costs a page fault per 4 KiB when it is first written)"""
        import numpy as np

        if out is None:
            out = np.zeros(num_frames * self.n, dtype=np.uint64)
        _check(lib().agx_ntt_forward_host_stream(self._h, _np_ptr(in1), _np_ptr(in2), _np_ptr(out), num_frames), "forward_host_stream")
        return out

    def inverse_host_stream(self, in1, num_frames, out=None):
        import numpy as np

        if out is None:
            out = np.zeros(num_frames * self.n, dtype=np.uint64)
        _check(lib().agx_ntt_inverse_host_stream(self._h, _np_ptr(in1), _np_ptr(out), num_frames), "inverse_host_stream")
        return out

    def close(self):
        if self._h:
            lib().agx_ntt_plan_destroy(self._h)
            self._h = _vp(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_block(num_frames, num_shards, index):
    """(first, count) of shard `index` when num_frames frames are dealt to num_shards shards (agx_ntt_shard_range: the reference's
    minibatch sizes, src/kernel/ntt.cpp:526-536, as contiguous blocks); pure arithmetic, no device needed"""
    first, count = _u64(0), _u64(0)
    _check(lib().agx_ntt_shard_range(num_frames, num_shards, index, ctypes.byref(first), ctypes.byref(count)), "shard_range")
    return first.value, count.value


class DeviceGroup:
    """One shard (plan + stream + staging + host thread) per entry of `devices` (agx_ntt_group_*): host frames are dealt in contiguous
    blocks, device-pointer calls take one pointer per shard.  No collective anywhere."""

    def __init__(self, devices, n, moduli, psi=None, tables=None):
        import numpy as np

        self.n = int(n)
        self.devices = [int(d) for d in devices]
        self.moduli = [int(q) for q in moduli]
        mods = np.array(self.moduli, dtype=np.uint64)
        devs = (_int * len(self.devices))(*self.devices)
        self._h = _vp(None)
        if tables is None:
            psi_arr = None if psi is None else np.array([int(p) for p in psi], dtype=np.uint64)
            _check(lib().agx_ntt_group_create_auto(ctypes.byref(self._h), devs, len(self.devices), self.n, len(self.moduli), _np_ptr(mods),
                                                   None if psi_arr is None else _np_ptr(psi_arr)), "group_create_auto")
        else:
            tw, pre = (np.ascontiguousarray(t, dtype=np.uint64) for t in tables[:2])
            itw = ipre = None
            if len(tables) == 4:
                itw, ipre = (np.ascontiguousarray(t, dtype=np.uint64) for t in tables[2:])
            _check(lib().agx_ntt_group_create(ctypes.byref(self._h), devs, len(self.devices), self.n, len(self.moduli), _np_ptr(mods), _np_ptr(tw),
                                              _np_ptr(pre), None if itw is None else _np_ptr(itw), None if ipre is None else _np_ptr(ipre)), "group_create")

    @property
    def num_shards(self):
        return len(self.devices)

    def shard(self, index):
        """(device, plan handle, stream handle) of shard `index`; the handles belong to the group"""
        dev, plan, stream = _int(0), _vp(None), _vp(None)
        _check(lib().agx_ntt_group_shard(self._h, index, ctypes.byref(dev), ctypes.byref(plan), ctypes.byref(stream)), "group_shard")
        return dev.value, plan.value, stream.value

    def forward_host(self, in1, in2, num_frames, out=None):
        import numpy as np

        if out is None:
            out = np.zeros(num_frames * self.n, dtype=np.uint64)
        _check(lib().agx_ntt_group_forward_host(self._h, _np_ptr(in1), _np_ptr(in2), _np_ptr(out), num_frames), "group_forward_host")
        return out

    def inverse_host(self, in1, num_frames, out=None):
        import numpy as np

        if out is None:
            out = np.zeros(num_frames * self.n, dtype=np.uint64)
        _check(lib().agx_ntt_group_inverse_host(self._h, _np_ptr(in1), _np_ptr(out), num_frames), "group_inverse_host")
        return out

    def _ptrs(self, ptrs):
        return None if ptrs is None else (_vp * self.num_shards)(*[int(p) if p else None for p in ptrs])

    def _batches(self, batch):
        return (_u64 * self.num_shards)(*[int(b) for b in batch])

    def forward(self, d_in, d_out, batch):
        _check(lib().agx_ntt_group_forward(self._h, self._ptrs(d_in), self._ptrs(d_out), self._batches(batch)), "group_forward")

    def inverse(self, d_in, d_out, batch):
        _check(lib().agx_ntt_group_inverse(self._h, self._ptrs(d_in), self._ptrs(d_out), self._batches(batch)), "group_inverse")

    def polymul(self, d_a, d_b, d_c, batch, d_scratch=None):
        _check(lib().agx_ntt_group_polymul(self._h, self._ptrs(d_a), self._ptrs(d_b), self._ptrs(d_c), self._ptrs(d_scratch), self._batches(batch)), "group_polymul")

    def synchronize(self):
        _check(lib().agx_ntt_group_synchronize(self._h), "group_synchronize")

    def close(self):
        if self._h:
            lib().agx_ntt_group_destroy(self._h)
            self._h = _vp(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# sharding of independent frames over ranks lives in distributed.py (no collective on the data path)
from .distributed import Group, aggregate_throughput, shard_range  # noqa: E402,F401
