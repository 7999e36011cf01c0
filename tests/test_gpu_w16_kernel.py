"""The 16-wave 256x256 GEMM tile kernel (csrc/gemm_w16.hip) forced into every big-tile slot: every epilogue, K from 64 to 1280,
multi-tile grids -- against numpy, through the C ABI's pg_dbg_gemm in a child process (the kernel choice is an environment switch
read once per process)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("variant", ["80"])    # 80: the 16-wave kernel for every 256-multiple shape
def test_w16_kernel_against_numpy(variant):
    env = dict(os.environ, PGIBBS_GEMM=variant)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "tile_gemm_debug.py")], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert p.stdout.count("bad 0 of") >= 9


@pytest.mark.parametrize("big", ["w16", "pp"])
def test_forced_big_tile_kernel_keeps_shards_bit_identical(big):
    """PGIBBS_GEMM_BIG=w16 / pp swaps the kernel into the engine's big-tile slot: it accumulates in the same k order with the
    same MFMA instruction as the other tile kernels, so the shard-invariance test of the full-size engine still holds."""
    env = dict(os.environ, PGIBBS_GEMM_BIG=big)
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_fullsize.py"), "-q", "-x", "-k",
                        "rows_are_independent"], capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:]
