"""Child process of tests/test_gpu_rccl_single.py: a world_size-1 `nccl` (= RCCL) process group on the one GPU of the box, and the
product's collective code paths executed through it on device tensors.  Prints OK lines; any failure raises."""
import os
import random
import socket
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from protein_gibbs_sampler_amd import _lib, esm_sampler, models, sharding, weights  # noqa: E402

s = socket.socket()
s.bind(("127.0.0.1", 0))
port = s.getsockname()[1]
s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
assert dist.get_backend() == "nccl"
ones = torch.ones(1, dtype=torch.int64, device=dev)
dist.all_reduce(ones)
assert int(ones.item()) == 1
print("OK all_reduce on RCCL, ranks seen", int(ones.item()))

# the one collective of the product, both forms, on device tensors
t = torch.arange(5 * 7, dtype=torch.int32, device=dev).reshape(5, 7)
eq = sharding.gather_tokens(dist, t, [5])
assert eq.device.type == "cuda" and (eq == t).all()
rag = sharding.gather_tokens(dist, t, [5], force_padded=True)
assert rag.shape == t.shape and (rag == t).all()
print("OK gather_tokens: all_gather_into_tensor and padded all_gather")

# the same collective through the C ABI (pg_comm_* / pg_gather_tokens: the library opens RCCL itself), equal and padded forms
comm = sharding.NativeComm(0, 1, 0)
assert _lib.lib().pg_comm_rank(comm.handle) == 0 and _lib.lib().pg_comm_world(comm.handle) == 1
nat = comm.gather_tokens(t, [5])
torch.cuda.synchronize()
assert nat.device.type == "cuda" and (nat == t).all()
os.environ["PGIBBS_GATHER_FORCE_PADDED"] = "1"
nat2 = comm.gather_tokens(t.reshape(5, 7, 1), [5])
torch.cuda.synchronize()
del os.environ["PGIBBS_GATHER_FORCE_PADDED"]
assert nat2.shape == (5, 7, 1) and (nat2.reshape(5, 7) == t).all()
import ctypes  # noqa: E402
bad = (ctypes.c_int64 * 1)(4)
assert _lib.lib().pg_gather_tokens(comm.handle, None, ctypes.c_void_p(t.data_ptr()), 5, 7, bad, ctypes.c_void_p(nat.data_ptr())) == _lib.PG_ERR_INVALID
comm.close()
print("OK pg_gather_tokens through the C ABI (RCCL opened by libpgibbs.so)")

# run_sharded through the real engine under that group == the direct call
cfg = weights.make_config(weights.ESM1B_CONFIG, d_model=256, n_layers=2, d_ffn=512, max_positions=64)
sd = weights.synthetic_state_dict(cfg, seed=3, std=0.05, embed_std=0.3, ln_jitter=0.1)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    model = models.ESM1b(state_dict=sd, config=cfg)
smp = esm_sampler.ESM_sampler(model, device="cuda:0")
lm = model.model
rng = np.random.default_rng(0)
tok0 = np.concatenate([np.zeros((6, 1)), rng.integers(4, 24, (6, 30)), np.full((6, 1), 2)], axis=1).astype(np.int32)
table = np.stack([np.stack([rng.choice(np.arange(1, 31), 4, replace=False) for _ in range(6)]) for _ in range(3)]).astype(np.int32)
params = _lib.make_sample_params(True, 32, 0, float("inf"), 1.0, smp.valid_aa_idx, 99)
direct = tok0.copy()
lm.gibbs_run(direct, table, params)


def run_block(ltok, ltable, base):
    params.row_id_base = base & 0xFFFFFFFF
    lm.gibbs_run(ltok, ltable, params)


ctx = sharding.DistContext(dist)
got = sharding.run_sharded(ctx, tok0.copy(), table, 0, 1, run_block, "cuda:0")
assert (got == direct).all()
print("OK run_sharded (engine + RCCL all-gather) == direct call")

# the sampler's own sharded entry (world_size 1 is below dist_context()'s threshold, so hand it the context's pieces)
random.seed(1)
smp.draw_seed = 5
a = smp.generate(4, "MEPAATGQEAEECAHSGRGEAWEEV", batch_size=4, num_iters=2, num_positions=3, show_progress_bar=False)
assert len(a) == 4
sharding.sync_host_rng(ctx)
assert sharding.broadcast_object(ctx, 123) == 123
sharding.check_same_job(ctx, sharding.job_digest("x", 1), "test")
print("OK broadcast_object / sync_host_rng / check_same_job on RCCL")
dist.barrier()
dist.destroy_process_group()
print("DONE")
