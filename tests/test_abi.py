"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/agx_ntt.h declares, validates arguments, and its host math agrees with the oracle.
No compute call can succeed here (no GPU): the product path must fail loudly, not fall back."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "agx_ntt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(agx_ntt_\w+)\s*\(", text)))


def test_header_and_binding_agree(agx):
    declared = _declared_symbols()
    assert len(declared) >= 20
    assert sorted(agx.ABI) == declared


def test_library_exports_every_declared_symbol(agx):
    raw = ctypes.CDLL(agx.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(raw, name), name


def test_library_exports_nothing_but_the_c_abi(agx):
    """built with -fvisibility=hidden and csrc/exports.map: the dynamic symbol table holds the agx_ntt_* entry points only (no
    C++ internals, no kernel host stubs), and exactly the ones the header declares"""
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", agx.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert names == _declared_symbols(), [n for n in names if not n.startswith("agx_ntt_")][:10]


def test_no_torch_or_oracle_in_product_library(agx):
    """the shipped library links neither torch nor the oracle"""
    import subprocess

    needed = subprocess.run(["readelf", "-d", agx.LIB_PATH], capture_output=True, text=True).stdout
    assert "torch" not in needed and "oracle" not in needed
    assert "amdhip64" in needed


def test_release_caches_without_a_device(agx):
    """agx_ntt_release_caches has nothing to free on a box without a GPU and must say so quietly (no HIP call may be required for it)"""
    assert agx.lib().agx_ntt_release_caches() == 0
    assert agx.lib().agx_ntt_release_caches() == 0


def test_strerror(agx):
    L = agx.lib()
    assert L.agx_ntt_strerror(0) == b"success"
    for code in range(1, 10):
        assert L.agx_ntt_strerror(code) not in (b"", b"unknown status")
    assert L.agx_ntt_strerror(1234) == b"unknown status"


@pytest.mark.parametrize("n,bits", [(32, 30), (1024, 30), (4096, 60), (8192, 61), (16384, 60), (32768, 60)])
def test_host_math_matches_oracle(agx, orc, n, bits):
    qs = agx.find_primes(bits, n, 3)
    assert qs == [orc.find_prime(bits, n, k) for k in range(3)]
    q = qs[0]
    psi = agx.min_root(q, n)
    assert psi == orc.min_root(q, n)
    tw, pre = agx.make_tables(q, psi, n)
    otw, opre = orc.make_tables(q, psi, n)
    assert np.array_equal(tw, otw) and np.array_equal(pre, opre)
    itw, ipre = agx.make_tables(q, psi, n, inverse=True)
    oitw, oipre = orc.make_inv_tables(q, psi, n)
    assert np.array_equal(itw, oitw) and np.array_equal(ipre, oipre)


def test_argument_validation(agx):
    L = agx.lib()
    out = np.zeros(4, dtype=np.uint64)
    p64 = ctypes.POINTER(ctypes.c_uint64)
    ptr = out.ctypes.data_as(p64)
    assert L.agx_ntt_find_primes(60, 4096, 1, None) == 1          # NULL
    assert L.agx_ntt_find_primes(60, 1000, 1, ptr) == 2           # not a power of two
    assert L.agx_ntt_find_primes(60, 65536, 1, ptr) == 2          # beyond AGX_NTT_MAX_N
    assert L.agx_ntt_find_primes(63, 4096, 1, ptr) == 5           # q must stay below 2^62
    r = ctypes.c_uint64(0)
    assert L.agx_ntt_min_root(65537, 16384, ctypes.byref(r)) == 0  # the reference's example modulus (main.cpp:55)
    assert L.agx_ntt_min_root(65536, 16, ctypes.byref(r)) == 3     # even
    assert L.agx_ntt_min_root(97, 64, ctypes.byref(r)) == 3        # 96 not divisible by 128
    assert L.agx_ntt_min_root((1 << 62) + 1, 2, ctypes.byref(r)) == 3  # too large
    assert L.agx_ntt_make_tables(97, 5, 16, ptr, ptr) == 4         # 5 is not a primitive 32nd root mod 97
    h = ctypes.c_void_p(None)
    mods = np.array([97], dtype=np.uint64)
    mp = mods.ctypes.data_as(p64)
    assert L.agx_ntt_plan_create_auto(ctypes.byref(h), 24, 1, mp, None) == 2
    assert L.agx_ntt_plan_create_auto(ctypes.byref(h), 16, 0, mp, None) == 5
    assert L.agx_ntt_plan_create_auto(None, 16, 1, mp, None) == 1
    mods[0] = 33  # 1 mod 32 but composite
    assert L.agx_ntt_plan_create_auto(ctypes.byref(h), 16, 1, mp, None) == 3
    assert L.agx_ntt_forward(None, None, None, 1, None) == 1
    assert L.agx_ntt_plan_destroy(None) == 0


def test_product_fails_loudly_without_gpu(agx):
    """there is no CPU fallback behind the boundary: without a device every compute entry
    point reports AGX_ERR_NO_DEVICE"""
    if agx.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(agx.AgxError) as ei:
        agx.Plan(16, [97])
    assert ei.value.status == 6
    q = agx.find_primes(30, 32)[0]
    tw, pre = agx.make_tables(q, agx.min_root(q, 32), 32)
    x = np.zeros(32, dtype=np.uint64)
    with pytest.raises(agx.AgxError) as ei:
        agx.forward_host(x, x, q, tw, pre, 32, 1)
    assert ei.value.status == 6


def test_product_sources_never_reference_the_oracle():
    pkg = os.path.join(ROOT, "agilex-ntt_amd")
    for dirpath, _, files in os.walk(pkg):
        if any(part in ("build", "lib", "bin", "__pycache__") for part in dirpath.split(os.sep)):
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower(), os.path.join(dirpath, f)


def test_bench_cpu_baseline_leg_runs(orc):
    """bench.py's cpu_baseline leg (the oracle timed on host cores) on a tiny budget"""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    os.environ["AGX_BENCH_CPU_THREADS"] = "2"
    try:
        r = bench.cpu_baseline(target_seconds=0.3)
    finally:
        del os.environ["AGX_BENCH_CPU_THREADS"]
    assert r["kind"] == "port" and r["unit"] == "NTT/s" and r["cores"] == 2
    assert r["value"] > 1000 and r["single_core_value"] > 500
    assert bench.host_cores() >= 1


def test_diag_header_declares_only_the_debug_hook():
    """tools/agx_ntt_diag.h (lib/libagxntt_diag.so) adds exactly one symbol to the boundary, and the public header none of it"""
    text = open(os.path.join(ROOT, "tools", "agx_ntt_diag.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    assert sorted(set(re.findall(r"\b(agx_ntt_\w+)\s*\(", text))) == ["agx_ntt_debug_set_trace_buffer"]
    assert "agx_ntt_debug_set_trace_buffer" not in _declared_symbols()


def test_kernel_source_hash_tracks_the_sources(agx, tmp_path):
    """profiles/hbm_traffic.json is tied to the kernel sources by agx.kernel_source_sha16(): stable, 16 hex digits"""
    h = agx.kernel_source_sha16()
    assert re.fullmatch(r"[0-9a-f]{16}", h) and h == agx.kernel_source_sha16()


def test_bench_power_sampler_reads_amdgpu_sysfs_layout(tmp_path):
    """bench.py's PowerSampler on a fake amdgpu sysfs tree: microwatts -> watts, the starred pp_dpm_sclk level, window medians"""
    import importlib.util
    import time

    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    hw = tmp_path / "hwmon" / "hwmon3"
    hw.mkdir(parents=True)
    (hw / "power1_average").write_text("1370000000\n")
    (hw / "power1_cap").write_text("1400000000\n")
    (tmp_path / "pp_dpm_sclk").write_text("0: 500Mhz\n1: 1744Mhz *\n2: 2400Mhz\n")
    s = bench.PowerSampler.__new__(bench.PowerSampler)
    s.samples, s.power_file, s.cap_file, s.sclk_file = [], str(hw / "power1_average"), str(hw / "power1_cap"), str(tmp_path / "pp_dpm_sclk")
    t0 = time.perf_counter()
    for _ in range(3):
        s.samples.append((time.perf_counter(),) + s._read())
    w = s.window(t0, time.perf_counter())
    assert w["socket_power_w_median"] == 1370.0 and w["power_cap_w"] == 1400.0 and w["sclk_mhz_median"] == 1744 and w["samples"] == 3
    assert s.window(t0 - 10, t0 - 5) is None
