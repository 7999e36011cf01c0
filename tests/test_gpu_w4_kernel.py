"""The experimental 4-wave and 16-wave 256x256 GEMM tile kernels (csrc/gemm_w4.hip, gemm_w16.hip; not in the default dispatch, see
DESIGN.md section 4) stay
correct: every epilogue, both MFMA shapes, K from 64 to 1280, multi-tile grids -- against numpy, through the C ABI's pg_dbg_gemm
in a child process (the kernel choice is an environment switch read once per process)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("variant", ["40", "51", "80"])    # 40 / 51: 4 waves with 16x16x32 / 32x32x16 MFMAs, 80: 16 waves
def test_w4_kernel_against_numpy(variant):
    env = dict(os.environ, PGIBBS_GEMM=variant)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "w4_debug.py")], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert p.stdout.count("bad 0 of") >= 9


@pytest.mark.parametrize("big", ["w4", "w16"])
def test_w4_in_the_engine_dispatch_keeps_shards_bit_identical(big):
    """PGIBBS_GEMM_BIG=w4 / w16 swaps the kernel into the engine's big-tile slot: it accumulates in the same k order with the
    same MFMA instruction as the other tile kernels, so the shard-invariance test of the full-size engine still holds."""
    env = dict(os.environ, PGIBBS_GEMM_BIG=big)
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_fullsize.py"), "-q", "-x", "-k",
                        "rows_are_independent"], capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:]
