"""The boundary from a language that is not Python: examples/pgibbs_client.c (plain C99, gcc, no torch, no HIP headers) builds an
ESM-1b-shaped model from named tensors through include/pgibbs.h, draws its positions with the CPython-exact generator, runs the whole
Gibbs loop in one pg_esm_gibbs_run call and sends the tokens through the C-ABI collective (pg_comm_* / pg_gather_tokens, world of
one).  Checked here against the interpreter's own `random` and, bit for bit, against the Python path on the same weights."""
import os
import random
import shutil
import struct
import subprocess
import warnings

import numpy as np
import pytest

from protein_gibbs_sampler_amd import _lib, models, weights

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read_weights(path):
    sd = {}
    with open(path, "rb") as f:
        (n,) = struct.unpack("<i", f.read(4))
        for _ in range(n):
            (ln,) = struct.unpack("<i", f.read(4))
            name = f.read(ln).decode()
            (numel,) = struct.unpack("<q", f.read(8))
            sd[name] = np.frombuffer(f.read(4 * numel), dtype=np.float32).copy()
    return sd


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_plain_c_client_matches_the_python_path(tmp_path):
    rocm_lib = "/opt/rocm/lib"
    exe = tmp_path / "pgibbs_client"
    lib_dir = os.path.join(ROOT, "protein_gibbs_sampler_amd", "lib")
    cc = ["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "pgibbs_client.c"),
          "-L", lib_dir, "-lpgibbs", "-L", rocm_lib, "-lamdhip64", "-Wl,-rpath," + lib_dir, "-Wl,-rpath," + rocm_lib, "-lm", "-o", str(exe)]
    p = subprocess.run(cc, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    wfile, ofile = tmp_path / "w.bin", tmp_path / "o.bin"
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([str(exe), str(wfile), str(ofile)], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert "gfx950" in p.stdout and "24 draws" in p.stdout

    B, T, ITERS, P, V = 3, 20, 2, 4, 33
    raw = np.fromfile(ofile, dtype=np.int32)
    start, idx = raw[:B * T].reshape(B, T), raw[B * T:B * T + ITERS * B * P].reshape(ITERS, B, P)
    o = B * T + ITERS * B * P
    tokens, sampled = raw[o:o + B * T].reshape(B, T), raw[o + B * T:o + B * T + ITERS * B * P].reshape(ITERS, B, P)
    logits = raw[o + B * T + ITERS * B * P:].view(np.float32).reshape(ITERS, B, P, V)
    # positions: random.seed(12345); random.sample(range(1, 19), 4) per (iteration, chain) -- /root/reference/src/pgen/esm_sampler.py:245
    random.seed(12345)
    want_idx = [[random.sample(range(1, T - 1), P) for _ in range(B)] for _ in range(ITERS)]
    assert idx.tolist() == want_idx
    # the Python path on the same named tensors
    sd = _read_weights(wfile)
    cfg = weights.make_config(weights.ESM1B_CONFIG, d_model=128, n_layers=2, n_heads=2, d_ffn=256, max_positions=64)
    shapes = weights.tensor_shapes(cfg)
    assert set(sd) == set(shapes)
    sd = {k: v.reshape(shapes[k]) for k, v in sd.items()}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        lm = models.ESM1b(state_dict=sd, config=cfg, precision="bf16").model.to("cuda:0")
    tok = start.copy()
    params = _lib.make_sample_params(True, 32, 0, float("inf"), 1.0, list(range(4, 24)), rng_seed=99)
    lg, st = lm.gibbs_run(tok, idx.copy(), params, want_logits=True, want_tokens=True)
    assert np.array_equal(tok, tokens) and np.array_equal(st, sampled)
    assert np.isfinite(logits).all() and np.array_equal(lg.view(np.uint32), logits.view(np.uint32))
    assert (tokens != start).any() and ((tokens[:, 1:-1] >= 4) & (tokens[:, 1:-1] <= 23)).all()
