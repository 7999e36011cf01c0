"""(f3) fair-esm checkpoint file -> loader -> HIP engine -> logits against the oracle, for both on-disk layouts
(/root/reference/src/pgen/models.py:61,86 obtain weights through `esm.pretrained.*`, i.e. through this key mapping);
and the out-of-range target positions the reference answers with IndexError (esm_sampler.py:234,262)."""
import argparse
import random
import warnings

import numpy as np
import pytest
import torch

from oracle.esm_forward import EsmConfig, esm1b_forward
from oracle.msa_forward import MsaConfig, msa_forward
from protein_gibbs_sampler_amd import _lib, esm_msa_sampler, esm_sampler, models, weights

pytestmark = pytest.mark.gpu
BF16_REL = 0.03      # bf16 throughput mode: max logit error held to 3 % of the logit spread (std); strict mode: 1e-3 absolute


def _write_pt(path, sd, cfg, arch):
    disk = {k: torch.from_numpy(v) for k, v in weights.to_fair_esm_checkpoint_layout(sd, cfg).items()}
    disk["encoder.lm_head.weight"] = disk.pop("encoder.sentence_encoder.embed_tokens.weight")      # tied copy only
    disk["encoder.sentence_encoder.contact_head.regression.weight"] = torch.zeros(1, 2 * cfg["n_layers"] * cfg["n_heads"])
    torch.save({"model": disk, "args": argparse.Namespace(arch=arch)}, path)


@pytest.mark.parametrize("precision,tol", [("bf16", None), ("fp32", 1e-3)])
def test_esm1b_checkpoint_file_drives_the_engine(tmp_path, precision, tol):
    cfg = weights.make_config(weights.ESM1B_CONFIG, d_model=128, n_layers=3, d_ffn=256, max_positions=64)
    sd = weights.synthetic_state_dict(cfg, seed=8, std=0.08, embed_std=0.5, ln_jitter=0.1)
    path = tmp_path / "esm1b_t33_650M_UR50S.pt"
    _write_pt(path, sd, cfg, "roberta_large")
    model = models.ESM1b(checkpoint=str(path), config=cfg, precision=precision)     # no synthetic opt-in needed: a file is given
    s = esm_sampler.ESM_sampler(model, device="cuda:0")
    tok = s.get_init_seq("MEPAATGQEAEECAHSGRGEAWEEV", 30, 2).numpy()
    tok[1, 4] = 32
    got = model.model.forward_logits(tok)
    ref_sd = dict(sd)
    ref_sd["embed_tokens.weight"] = sd["embed_tokens.weight"].copy()
    ref_sd["embed_tokens.weight"][32] = 0            # fair-esm zeroes the <mask> embedding row of ESM-1b checkpoints
    want = esm1b_forward(ref_sd, EsmConfig(d_model=128, n_layers=3, n_heads=2, d_ffn=256, max_pos=64), tok)
    err = np.abs(got - want).max()
    print("\ncheckpoint -> engine (ESM-1b layout, %s): max|engine - oracle| = %.3e (logit std %.2f)" % (precision, err, want.std()))
    assert err < (tol if tol else BF16_REL * want.std())
    assert np.abs(got[..., 32] - sd["lm_head.bias"][32]).max() < 1e-6       # logit[<mask>] = bias only (zeroed tied row)


@pytest.mark.parametrize("precision,tol", [("bf16", None), ("fp32", 1e-3)])
def test_msa1b_checkpoint_file_drives_the_engine(tmp_path, precision, tol):
    """The MSA layout has row/column exchanged on disk: loading it unswapped gives different logits (checked too)."""
    cfg = weights.make_config(weights.MSA1B_CONFIG, d_model=128, n_layers=2, d_ffn=256, max_positions=64, max_msa_rows=8)
    sd = weights.synthetic_state_dict(cfg, seed=9, std=0.08, embed_std=0.5, ln_jitter=0.1)
    path = tmp_path / "esm_msa1b_t12_100M_UR50S.pt"
    _write_pt(path, sd, cfg, "msa_transformer")
    model = models.ESM_MSA1(checkpoint=str(path), config=cfg, precision=precision)
    s = esm_msa_sampler.ESM_MSA_sampler(model, device="cuda:0")
    tok = s.get_init_msa(["ACDEFGHIKLMN", "AC-EFGHIKLMN", "ACDEFGH-KLMV", "MCDEFGHIKLMN"], 14, 2).numpy()
    tok[0, 1, 3] = 32
    got = model.model.forward_logits(tok)
    ocfg = MsaConfig(d_model=128, n_layers=2, n_heads=2, d_ffn=256, max_pos=64, max_rows=8)
    want = msa_forward(sd, ocfg, tok)
    err = np.abs(got - want).max()
    unswapped = {weights._swap_row_column(k): v for k, v in sd.items()}
    swap_effect = np.abs(msa_forward(unswapped, ocfg, tok) - want).max()
    print("\ncheckpoint -> engine (MSA layout, %s): max|engine - oracle| = %.3e (logit std %.2f; a missed row/column swap would "
          "move logits by %.2f)" % (precision, err, want.std(), swap_effect))
    assert err < (tol if tol else BF16_REL * want.std())
    assert swap_effect > 0.5 and err < swap_effect / 3           # the swap is not a no-op on these weights, and it was applied


def test_models_refuse_to_run_without_checkpoint():
    with pytest.raises(FileNotFoundError, match="synthetic"):
        models.ESM1b(config=weights.make_config(weights.ESM1B_CONFIG, d_model=128, n_layers=1, d_ffn=256))


# ---- target positions outside the token row -----------------------------------------------------------------------------
@pytest.fixture(scope="module")
def small_sampler():
    cfg = weights.make_config(weights.ESM1B_CONFIG, d_model=128, n_layers=2, d_ffn=256, max_positions=64)
    model = models.ESM1b(state_dict=weights.synthetic_state_dict(cfg, seed=2, std=0.08, embed_std=0.5), config=cfg)
    return esm_sampler.ESM_sampler(model, device="cuda:0")


def test_generate_rejects_out_of_range_indexes_like_the_reference(small_sampler):
    s = small_sampler
    seed = "MEPAATGQEAEECAHSGRGEAWEEV"                        # T = 27 tokens
    with pytest.raises(IndexError, match="out of bounds"):
        s.generate(1, seed, num_iters=1, indexes=[3, 27], show_progress_bar=False)
    with pytest.raises(IndexError, match="out of bounds"):
        s.generate(1, seed, num_iters=1, indexes=[-28], show_progress_bar=False)
    # negative positions count from the end of the token row, as `batch[b][kk]` does: -2 is the last residue (26 is <eos>)
    s.draw_seed = 4
    random.seed(0)
    a = s.generate(2, seed, batch_size=2, num_iters=2, indexes=[-2, 5], top_k=1, burnin=0, show_progress_bar=False)
    random.seed(0)
    b = s.generate(2, seed, batch_size=2, num_iters=2, indexes=[25, 5], top_k=1, burnin=0, show_progress_bar=False)
    assert a == b and all(len(x) == 25 for x in a)


def test_c_abi_rejects_out_of_range_positions(small_sampler):
    lm = small_sampler.model.model
    tok = np.zeros((2, 27), dtype=np.int32) + 5
    params = _lib.make_sample_params(True, 32, 0, float("inf"), None, small_sampler.valid_aa_idx, 1)
    before = tok.copy()
    with pytest.raises(_lib.PgError, match="out of range"):
        lm.gibbs_run(tok, np.asarray([[[3, 27], [1, 2]]], dtype=np.int32), params)
    assert (tok == before).all()
    # the *_device kernels skip such entries instead of touching memory outside the row
    d_tok = torch.from_numpy(tok).cuda()
    d_idx = torch.tensor([[3, 1 << 20], [2, 27]], dtype=torch.int32).cuda()
    import ctypes
    _lib.check(_lib.lib().pg_mask_scatter_device(None, ctypes.c_void_p(d_tok.data_ptr()), 2, 27, ctypes.c_void_p(d_idx.data_ptr()),
                                                 None, 2, 2, 32))
    torch.cuda.synchronize()
    want = before.copy()
    want[0, 3] = want[1, 2] = 32
    assert (d_tok.cpu().numpy() == want).all()


def test_ragged_list_seeds_mask_pad_keys_in_the_gibbs_run(small_sampler):
    """List seeds longer than max_len leave no <mask> padding and the batch converter pads the shorter rows with <pad>
    (esm_sampler.py:113-115): the Gibbs run must then mask <pad> keys in attention as the scoring entry points do."""
    s = small_sampler
    lm = s.model.model
    tok = np.full((2, 12), 1, dtype=np.int32)
    tok[0] = [0, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 2]
    tok[1, :7] = [0, 5, 6, 7, 8, 9, 2]
    params = _lib.make_sample_params(True, 32, 0, float("inf"), None, s.valid_aa_idx, 3)
    idx = np.asarray([[[2, 4], [2, 4]]], dtype=np.int32)
    t = tok.copy()
    lg, st = lm.gibbs_run(t, idx, params, want_logits=True, want_tokens=True)
    masked = tok.copy()
    masked[0, [2, 4]] = masked[1, [2, 4]] = 32
    full = lm.forward_logits(masked)                           # this entry point scans for <pad> and masks the keys
    ref = np.stack([full[b, idx[0, b]] for b in range(2)])
    assert np.abs(lg[0] - ref).max() < 2e-2
    alone = lm.forward_logits(masked[1:2, :7])                 # the short chain on its own, no padding at all
    assert np.abs(lg[0, 1] - alone[0, [2, 4]]).max() < 5e-2
