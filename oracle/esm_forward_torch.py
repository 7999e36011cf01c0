"""ORACLE (test infrastructure only -- never imported by the product path).

The CPU BASELINE leg of bench.py: the reference's per-iteration work
(/root/reference/src/pgen/esm_sampler.py:209-234) stated in torch CPU ops, the way
the reference itself runs on a CPU (fair-esm fp32 modules under PyTorch with all
host cores).  fair-esm is a third-party package that is not
installed here, so the forward below is the same restatement as
oracle/esm_forward.py written with the torch operators
fair-esm's modules call -- F.embedding, F.layer_norm, F.linear, bmm + softmax,
F.gelu (exact erf) -- on weights keyed by fair-esm's state-dict names.

  torch_state(w)             numpy state dict -> torch fp32 tensors (shared memory)
  esm1b_forward(w, cfg, tok) tokens int64 [B,T] -> logits fp32 [B,T,V]   (every row: the reference
                             evaluates `model(batch)["logits"]` in full, esm_sampler.py:223)
  generate_step(...)         esm_sampler.py:8-45 with torch.topk / Categorical, one call per position
  gibbs_iterations(...)      mask -> forward -> per-(b, kk) generate_step -> write-back, esm_sampler.py:209-234

It is checked against the numpy oracle in tests/test_oracle_forward.py (same weights,
same tokens, 2e-4); the numpy oracle stays the checker of the engine's logits.
"""
import time

import numpy as np
import torch
import torch.nn.functional as F


def torch_state(w):
    return {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in w.items()}


def _embed(w, cfg, tokens):
    pad = tokens.eq(cfg.pad_idx)
    x = F.embedding(tokens, w["embed_tokens.weight"])
    if cfg.token_dropout:
        is_mask = tokens.eq(cfg.mask_idx)
        x = x.masked_fill(is_mask.unsqueeze(-1), 0.0)
        src_len = (~pad).sum(-1).to(x.dtype)
        ratio = is_mask.sum(-1).to(x.dtype) / src_len
        x = x * ((1 - 0.15 * 0.8) / (1 - ratio))[:, None, None]
    nonpad = (~pad).long()
    pos = torch.cumsum(nonpad, dim=1) * nonpad + cfg.pad_idx
    x = x + F.embedding(pos, w["embed_positions.weight"])
    d = x.shape[-1]
    x = F.layer_norm(x, (d,), w["emb_layer_norm_before.weight"], w["emb_layer_norm_before.bias"], 1e-5)
    return x * (~pad).unsqueeze(-1).to(x.dtype), pad


def _mha(w, p, cfg, h, pad):
    B, T, d = h.shape
    H = cfg.n_heads
    dh = d // H
    q = F.linear(h, w[p + "q_proj.weight"], w[p + "q_proj.bias"]) * dh ** -0.5
    k = F.linear(h, w[p + "k_proj.weight"], w[p + "k_proj.bias"])
    v = F.linear(h, w[p + "v_proj.weight"], w[p + "v_proj.bias"])
    q = q.view(B, T, H, dh).transpose(1, 2).reshape(B * H, T, dh)
    k = k.view(B, T, H, dh).transpose(1, 2).reshape(B * H, T, dh)
    v = v.view(B, T, H, dh).transpose(1, 2).reshape(B * H, T, dh)
    a = torch.bmm(q, k.transpose(1, 2))
    if bool(pad.any()):
        a = a.view(B, H, T, T).masked_fill(pad[:, None, None, :], float("-inf")).view(B * H, T, T)
    ctx = torch.bmm(F.softmax(a, dim=-1), v).view(B, H, T, dh).transpose(1, 2).reshape(B, T, d)
    return F.linear(ctx, w[p + "out_proj.weight"], w[p + "out_proj.bias"])


@torch.no_grad()
def esm1b_forward(w, cfg, tokens):
    tokens = torch.as_tensor(np.asarray(tokens), dtype=torch.int64)
    x, pad = _embed(w, cfg, tokens)
    d = x.shape[-1]
    for i in range(cfg.n_layers):
        p = "layers.%d." % i
        h = F.layer_norm(x, (d,), w[p + "self_attn_layer_norm.weight"], w[p + "self_attn_layer_norm.bias"], 1e-5)
        x = x + _mha(w, p + "self_attn.", cfg, h, pad)
        h = F.layer_norm(x, (d,), w[p + "final_layer_norm.weight"], w[p + "final_layer_norm.bias"], 1e-5)
        h = F.gelu(F.linear(h, w[p + "fc1.weight"], w[p + "fc1.bias"]))
        x = x + F.linear(h, w[p + "fc2.weight"], w[p + "fc2.bias"])
    x = F.layer_norm(x, (d,), w["emb_layer_norm_after.weight"], w["emb_layer_norm_after.bias"], 1e-5)
    h = F.gelu(F.linear(x, w["lm_head.dense.weight"], w["lm_head.dense.bias"]))
    h = F.layer_norm(h, (d,), w["lm_head.layer_norm.weight"], w["lm_head.layer_norm.bias"], 1e-5)
    return F.linear(h, w["embed_tokens.weight"]) + w["lm_head.bias"]


def generate_step(out, gen_idx, top_k=0, temperature=None, sample=False, valid_idx=None):
    """esm_sampler.py:8-45: one position of one chain."""
    logits = out[gen_idx]
    if temperature is not None:
        logits = logits / temperature
    sub = logits[valid_idx]
    n = sub.shape[0]
    k = n if (sample or top_k <= 0 or top_k > n) else top_k
    kth_vals, kth_idx = sub.topk(k)
    dist = torch.distributions.categorical.Categorical(logits=kth_vals)
    return valid_idx[kth_idx[dist.sample()]]


@torch.no_grad()
def gibbs_iterations(w, cfg, tokens, targets, valid_idx, top_k=0, temperature=1.0, sample=True):
    """`targets[it][b]` = the positions of chain b resampled in iteration it.  Returns (tokens, seconds in the forward,
    seconds in the per-position loop)."""
    tok = torch.as_tensor(np.asarray(tokens), dtype=torch.int64).clone()
    vi = torch.as_tensor(valid_idx, dtype=torch.int64)
    t_fwd = t_loop = 0.0
    for tgt in targets:
        t0 = time.perf_counter()
        for b, kks in enumerate(tgt):
            tok[b, torch.as_tensor(kks)] = cfg.mask_idx
        out = esm1b_forward(w, cfg, tok)
        t1 = time.perf_counter()
        for b, kks in enumerate(tgt):
            for kk in kks:
                tok[b, kk] = generate_step(out[b], int(kk), top_k=top_k, temperature=temperature, sample=sample, valid_idx=vi)
        t_loop += time.perf_counter() - t1
        t_fwd += t1 - t0
    return tok.numpy(), t_fwd, t_loop
