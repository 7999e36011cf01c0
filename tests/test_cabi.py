"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol include/pgibbs.h
declares, host-only entry points work, and compute entry points fail loudly without a GPU."""
import ctypes
import os
import random
import re

import numpy as np
import pytest

from protein_gibbs_sampler_amd import _lib, pyrandom

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "pgibbs.h")).read()
    declared = set(re.findall(r"\b(pg_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), "library does not export " + name
    assert declared == {n for n, _, _ in _lib.SIGNATURES}       # the ctypes table covers the whole header


def test_version_and_error_string():
    L = _lib.lib()
    assert b"gfx950" in L.pg_version()
    assert L.pg_pyrandom_seed(None, None, 0) == _lib.PG_ERR_INVALID
    assert b"key word" in L.pg_last_error()


@pytest.mark.parametrize("seed", [0, 1, 12345, 2**40 + 17])
def test_native_pyrandom_matches_interpreter(seed):
    r = pyrandom.NativePyRandom()
    r.seed(seed)
    random.seed(seed)
    for n, k in [(25, 2), (256, 25), (257, 25), (512, 51), (10, 10), (21, 6), (22, 6), (85, 6), (86, 6), (5, 0)]:
        assert r.sample(list(range(1, n + 1)), k) == random.sample(range(1, n + 1), k)
        a, b = list(range(n)), list(range(n))
        r.shuffle(a)
        random.shuffle(b)
        assert a == b
        assert r.choices(7, 5) == random.choices(range(7), k=5)
        assert r.random() == random.random()
        assert r.getrandbits(17) == random.getrandbits(17)
    assert r.getstate() == random.getstate()


def test_global_table_leaves_interpreter_rng_where_reference_would():
    random.seed(42)
    want = [[random.sample(range(1, 257), 25) for _ in range(8)] for _ in range(3)]
    after = random.getrandbits(32)
    random.seed(42)
    got = pyrandom.global_sample_table(list(range(1, 257)), 25, 24).reshape(3, 8, 25).tolist()
    assert got == want
    assert random.getrandbits(32) == after


def test_sample_larger_than_population_raises():
    with pytest.raises(ValueError):
        pyrandom.NativePyRandom().sample_table([1, 2, 3], 4, 1)


def test_compute_entry_points_fail_loudly_without_gpu():
    L = _lib.lib()
    if L.pg_device_count() > 0:
        pytest.skip("GPU present")
    x = np.zeros((4, 64), dtype=np.float32)
    rc = L.pg_dbg_layernorm(0, _lib.ptr(x), _lib.ptr(x), _lib.ptr(x), _lib.ptr(x), 4, 64, 1e-5)
    assert rc == _lib.PG_ERR_NO_DEVICE
    cfg = _lib.ModelConfig(arch=1, vocab=33, d_model=128, n_layers=1, n_heads=2, d_ffn=256, max_positions=32, pad_idx=1,
                           mask_idx=32, cls_idx=0, eos_idx=2, token_dropout=1, max_msa_rows=0, layer_norm_eps=1e-5)
    h = ctypes.c_void_p()
    assert L.pg_engine_create(ctypes.byref(cfg), None, 0, 0, 0, ctypes.byref(h)) == _lib.PG_ERR_NO_DEVICE
    assert b"no CPU fallback" in L.pg_last_error()


def test_comm_entry_points_reject_bad_arguments_and_need_a_gpu():
    """pg_comm_* (the C-ABI form of the one RCCL gather): argument checks run on the host; joining a communicator
    without a GPU is PG_ERR_NO_DEVICE, never a silent single-process fallback."""
    L = _lib.lib()
    h = ctypes.c_void_p()
    ident = ctypes.create_string_buffer(_lib.PG_COMM_ID_BYTES)
    assert L.pg_comm_unique_id(None) == _lib.PG_ERR_INVALID
    assert L.pg_comm_create(0, 1, None, 0, ctypes.byref(h)) == _lib.PG_ERR_INVALID
    assert L.pg_comm_create(2, 2, ident, 0, ctypes.byref(h)) == _lib.PG_ERR_INVALID and b"rank" in L.pg_last_error()
    if L.pg_device_count() == 0:
        assert L.pg_comm_create(0, 1, ident, 0, ctypes.byref(h)) == _lib.PG_ERR_NO_DEVICE
        assert not h.value
    assert L.pg_gather_tokens(None, None, None, 0, 1, None, None) == _lib.PG_ERR_INVALID
    assert L.pg_comm_rank(None) == -1 and L.pg_comm_world(None) == 0
    L.pg_comm_destroy(None)


def test_header_is_plain_c_and_the_c_client_fails_loudly_without_a_gpu(tmp_path):
    """include/pgibbs.h must bind from C, not only from C++ / ctypes: examples/pgibbs_client.c (C99, -Wall -Werror) compiles and
    links against the library with gcc; without a GPU it stops at pg_device_count() with a message and a non-zero status -- it
    does not compute anything on the host."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/lib/libamdhip64.so"):
        pytest.skip("needs gcc and the ROCm runtime library")
    lib_dir = os.path.join(ROOT, "protein_gibbs_sampler_amd", "lib")
    exe = tmp_path / "client"
    p = subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "pgibbs_client.c"), "-L", lib_dir, "-lpgibbs", "-L", "/opt/rocm/lib", "-lamdhip64",
                        "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", str(exe)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    if _lib.lib().pg_device_count() == 0:
        r = subprocess.run([str(exe), str(tmp_path / "w.bin"), str(tmp_path / "o.bin")], capture_output=True, text=True, timeout=120)
        assert r.returncode == 3 and "no MI355X" in r.stderr and not (tmp_path / "o.bin").exists()
