"""Time split of generate_single at the default pgen_msa_revised shape (alignment_size 32, templates of ~300 residues), 1 and 4
templates per native call.  Run on the GPU box: python tools/probes/pgen_msa_default_shape_probe.py"""
import random, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench_msa
from protein_gibbs_sampler_amd import esm_msa_sampler

wrapper, lm, cfg = bench_msa.build("bf16")
valid_idx = bench_msa._valid_idx(wrapper)
rng = np.random.default_rng(1)
R, L, steps, passes = 32, 300, 10, 3
s = esm_msa_sampler.ESM_MSA_sampler(wrapper, device="cuda:0")
s.draw_seed = 0
inv = {wrapper.alphabet.get_idx(t): t for t in "-ACDEFGHIKLMNPQRSTVWY"}
toks = bench_msa.random_msa_tokens(rng, valid_idx, 4, R, L)
msas = [["".join(inv[int(t)] for t in row[1:]) for row in toks[b]] for b in range(4)]
for mb in (1, 4, 1, 4):
    random.seed(0)
    s.generate_single_batch(msas[:mb], steps=steps, passes=1, burn_in=1, target_index=0, k=1, max_batch=mb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.generate_single_batch(msas, steps=steps, passes=passes, burn_in=2, target_index=0, k=1, max_batch=mb)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    lm.prof_enable(True); lm.prof_reset()
    s.generate_single_batch(msas[:mb], steps=steps, passes=1, burn_in=1, target_index=0, k=1, max_batch=mb)
    split = {c: round(v / steps, 3) for c, v in bench_msa._split(lm).items()}
    lm.prof_enable(False)
    fl = bench_msa.msa_flops_per_forward(cfg, 1, R, L + 1) * steps * passes * 4
    print("templates per call %d: %.2f ms per template-forward, %.0f TFLOP/s model, split per forward %s" % (mb, 1e3 * el / (steps * passes * 4), fl / el / 1e12, split))
