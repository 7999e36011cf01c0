"""Invariants pinning oracle/msa_forward.py (no independent MSA-Transformer implementation exists offline,
see the module header: "parity unpinned" against the reference's own KATs)."""
import numpy as np

from oracle import esm_forward as E
from oracle.msa_forward import MsaConfig, column_attention, msa_embed, msa_forward, row_attention, synthetic_msa_weights

CFG = MsaConfig(d_model=128, n_layers=2, n_heads=2, d_ffn=256, max_pos=40, max_rows=16)
W = synthetic_msa_weights(CFG, seed=2, std=0.08, embed_std=0.5, ln_jitter=0.1)
RNG = np.random.default_rng(0)


def _tokens(B, R, C):
    t = RNG.integers(4, 24, (B, R, C))
    t[..., 0] = 0
    t[RNG.random((B, R, C)) < 0.1] = 32
    t[..., 0] = 0
    return t


def test_shapes_and_batch_independence():
    t = _tokens(3, 4, 9)
    a = msa_forward(W, CFG, t)
    assert a.shape == (3, 4, 9, 33) and np.isfinite(a).all() and a.std() > 0.5
    b = msa_forward(W, CFG, t[1:2])
    assert np.abs(a[1:2] - b).max() < 2e-5                 # MSAs in a batch never interact


def test_column_attention_single_row_shortcut():
    h = RNG.standard_normal((2, 1, 7, 128)).astype(np.float32)
    p = "layers.0.column_self_attention.layer."
    short = column_attention(W, p, CFG, h)
    # general path with R == 1: softmax over one key is exactly 1
    H, dh = 2, 64
    v = E.linear(h, W[p + "v_proj.weight"], W[p + "v_proj.bias"])
    full = E.linear(v, W[p + "out_proj.weight"], W[p + "out_proj.bias"])
    assert np.abs(short - full).max() < 1e-6


def test_row_permutation_equivariance_of_attention_blocks():
    """Tied row attention sums over rows; column attention attends over rows: permuting the rows of the
    input permutes the rows of each block's output."""
    h = RNG.standard_normal((1, 5, 6, 128)).astype(np.float32)
    perm = np.array([3, 0, 4, 1, 2])
    for fn, p in ((row_attention, "layers.0.row_self_attention.layer."), (column_attention, "layers.1.column_self_attention.layer.")):
        a = fn(W, p, CFG, h)
        b = fn(W, p, CFG, h[:, perm])
        assert np.abs(a[:, perm] - b).max() < 2e-5


def test_row_attention_is_tied_and_scaled():
    """One attention map per head shared by all rows, logits scaled by dh^-0.5 / sqrt(R)."""
    B, R, C, d, H, dh = 1, 3, 5, 128, 2, 64
    h = RNG.standard_normal((B, R, C, d)).astype(np.float32)
    p = "layers.0.row_self_attention.layer."
    got = row_attention(W, p, CFG, h)
    q = (h @ W[p + "q_proj.weight"].T + W[p + "q_proj.bias"]).reshape(R, C, H, dh)
    k = (h @ W[p + "k_proj.weight"].T + W[p + "k_proj.bias"]).reshape(R, C, H, dh)
    v = (h @ W[p + "v_proj.weight"].T + W[p + "v_proj.bias"]).reshape(R, C, H, dh)
    ctx = np.zeros((R, C, H, dh))
    for hh in range(H):
        a = np.zeros((C, C))
        for r in range(R):
            a += q[r, :, hh].astype(np.float64) @ k[r, :, hh].astype(np.float64).T
        a *= dh ** -0.5 / np.sqrt(R)
        pr = np.exp(a - a.max(-1, keepdims=True))
        pr /= pr.sum(-1, keepdims=True)
        for r in range(R):
            ctx[r, :, hh] = pr @ v[r, :, hh]
    want = ctx.reshape(1, R, C, d) @ W[p + "out_proj.weight"].T + W[p + "out_proj.bias"]
    assert np.abs(got - want).max() < 1e-4


def test_embedding_matches_esm_oracle_pieces():
    """Token + learned positions + LN reuse the HF-corroborated ESM-1b oracle code paths; the MSA row
    embedding is a plain broadcast add before the LayerNorm."""
    t = _tokens(1, 3, 8)
    x = msa_embed(W, CFG, t)
    raw = W["embed_tokens.weight"][t] + W["embed_positions.weight"][np.arange(8) + 2][None, None] \
        + W["msa_position_embedding"].reshape(-1, 128)[:3][None, :, None, :]
    want = E.layer_norm(raw, W["emb_layer_norm_before.weight"], W["emb_layer_norm_before.bias"])
    assert np.abs(x - want).max() < 1e-5


# ---- second pin: fair-esm's own tensor layout and its chunked (max_tokens_per_msa) paths -----------------------------------
def test_oracle_equals_fair_esm_layout_restatement_and_chunked_paths():
    """"chunked == unchunked".  tests/_msa_alt.py restates the forward in fair-esm's R x C x B x D layout
    with its einsum strings (torch); it must agree with oracle/msa_forward.py (B x R x C x D, numpy), and its memory-bounded
    row-chunk / column-chunk paths must agree with the one-shot path."""
    from _msa_alt import msa_forward_alt
    ck = dict(d_model=128, n_layers=2, n_heads=2, d_ffn=256, max_pos=80, max_rows=16)
    ocfg = MsaConfig(**ck)
    sd = synthetic_msa_weights(ocfg, seed=21, std=0.08, embed_std=0.5, ln_jitter=0.1)
    rng = np.random.default_rng(4)
    for (B, R, C) in [(2, 5, 23), (1, 1, 9), (1, 12, 40)]:
        tok = rng.integers(4, 24, (B, R, C))
        tok[rng.random((B, R, C)) < 0.1] = 30
        tok[rng.random((B, R, C)) < 0.1] = 32
        tok[..., 0] = 0
        want = msa_forward(sd, ocfg, tok)
        alt = msa_forward_alt(sd, 2, 2, tok)
        assert np.abs(alt - want).max() < 5e-5 * max(1.0, np.abs(want).max())
        for max_tokens in (C * 2, C + 3, 7):                    # 2 rows per chunk / 1 row + ragged columns / tiny
            chunked = msa_forward_alt(sd, 2, 2, tok, max_tokens_per_msa=max_tokens)
            assert np.abs(chunked - alt).max() < 2e-5 * max(1.0, np.abs(alt).max())


def test_padded_ragged_msa_batch_oracle_vs_fair_esm_layout_restatement():
    """Ragged MSA lists (the reference's unmasked log_likelihood_batch pads them to one [n, R_max, C_max] tensor,
    /root/reference/src/pgen/esm_msa_sampler.py:341,416-431): <pad> rows and <pad> columns under fair-esm's padding semantics -- q
    zeroed at padded positions of the tied row attention, key columns that are <pad> in row 0 and key rows that are <pad> at a
    column filled with -10000, the 1/sqrt(R) from the padded row count.  The two restatements must agree, the padded MSA's real
    positions must differ from the same MSA scored alone (the artefact the reference has: R_max enters the scaling), and a batch
    without <pad> must not change."""
    from _msa_alt import msa_forward_alt
    ck = dict(d_model=128, n_layers=2, n_heads=2, d_ffn=256, max_pos=80, max_rows=16)
    ocfg = MsaConfig(**ck)
    sd = synthetic_msa_weights(ocfg, seed=23, std=0.08, embed_std=0.5, ln_jitter=0.1)
    rng = np.random.default_rng(6)
    shapes = [(6, 21), (4, 17), (6, 12)]                        # (rows, columns incl. <cls>) of three MSAs
    R, C = 6, 21
    tok = np.full((3, R, C), 1, dtype=np.int64)
    for b, (r, c) in enumerate(shapes):
        tok[b, :r, :c] = rng.integers(4, 24, (r, c))
        tok[b, :r, 0] = 0
    want = msa_forward(sd, ocfg, tok)
    alt = msa_forward_alt(sd, 2, 2, tok)
    assert np.isfinite(want).all()
    for b, (r, c) in enumerate(shapes):
        assert np.abs(alt[b, :r, :c] - want[b, :r, :c]).max() < 5e-5 * max(1.0, np.abs(want).max())
    # MSA 0 has no padding of its own but shares the batch: same as alone (fair-esm masks per MSA; R_max == its own R)
    alone0 = msa_forward(sd, ocfg, tok[:1])
    assert np.abs(alone0[0] - want[0]).max() < 2e-5
    # MSA 1 padded from 4 to 6 rows: the row-attention scaling uses 6 -> differs from scoring it alone
    alone1 = msa_forward(sd, ocfg, tok[1:2, :4, :17])
    assert np.abs(alone1[0] - want[1, :4, :17]).max() > 1e-3
    # MSA 2 padded in columns only (same rows): equals scoring it alone up to rounding -- pad columns are masked keys, masked column-attention rows
    alone2 = msa_forward(sd, ocfg, tok[2:3, :, :12])
    assert np.abs(alone2[0] - want[2, :, :12]).max() < 5e-5 * max(1.0, np.abs(want).max())


# ---- the two attention forms that only the MSA Transformer has, against torch's own attention code ------------------------------
# No MSA Transformer exists offline, but both of its attention blocks are ordinary attention in disguise, and torch's
# implementations of ordinary attention share no code with the oracle:
#   * column attention = multi-head attention along the R rows of every column -> torch.nn.MultiheadAttention (with a key-padding
#     mask for the <pad> case; fair-esm's finite -10000 fill and torch's -inf agree wherever a column has a real key);
#   * TIED row attention: scores summed over the rows with q scaled by dh^-0.5 / sqrt(R) = scaled-dot-product attention over the
#     C columns whose per-head feature vector is the concatenation of the R rows' features (dimension R * dh, default scale
#     1 / sqrt(R * dh)) -> torch.nn.functional.scaled_dot_product_attention.
def _proj_weights(p):
    import torch
    return {k: torch.from_numpy(W[p + k]) for k in ("q_proj.weight", "q_proj.bias", "k_proj.weight", "k_proj.bias", "v_proj.weight",
                                                    "v_proj.bias", "out_proj.weight", "out_proj.bias")}


def test_column_attention_against_torch_multihead_attention():
    import torch
    p = "layers.1.column_self_attention.layer."
    B, R, C, d, H = 2, 6, 5, CFG.d_model, CFG.n_heads
    h = RNG.standard_normal((B, R, C, d)).astype(np.float32)
    pw = _proj_weights(p)
    m = torch.nn.MultiheadAttention(d, H, bias=True, batch_first=True)
    with torch.no_grad():
        m.in_proj_weight.copy_(torch.cat([pw["q_proj.weight"], pw["k_proj.weight"], pw["v_proj.weight"]]))
        m.in_proj_bias.copy_(torch.cat([pw["q_proj.bias"], pw["k_proj.bias"], pw["v_proj.bias"]]))
        m.out_proj.weight.copy_(pw["out_proj.weight"])
        m.out_proj.bias.copy_(pw["out_proj.bias"])
        x = torch.from_numpy(h).permute(0, 2, 1, 3).reshape(B * C, R, d)          # one sequence of R rows per (msa, column)
        want = m(x, x, x, need_weights=False)[0].reshape(B, C, R, d).permute(0, 2, 1, 3).numpy()
        got = column_attention(W, p, CFG, h)
        assert np.abs(got - want).max() < 3e-5 * max(1.0, np.abs(want).max())
        # <pad> keys (ragged MSA lists): rows 4, 5 of MSA 1 are padding in every column
        pad = np.zeros((B, R, C), bool)
        pad[1, 4:] = True
        kpm = torch.from_numpy(pad).permute(0, 2, 1).reshape(B * C, R)
        want = m(x, x, x, key_padding_mask=kpm, need_weights=False)[0].reshape(B, C, R, d).permute(0, 2, 1, 3).numpy()
        got = column_attention(W, p, CFG, h, pad)
        real = ~pad                                                               # outputs AT padded positions are never used
        assert np.abs(got - want)[real].max() < 3e-5 * max(1.0, np.abs(want).max())


def test_tied_row_attention_against_torch_scaled_dot_product_attention():
    import torch
    import torch.nn.functional as F
    p = "layers.0.row_self_attention.layer."
    B, R, C, d, H = 2, 5, 9, CFG.d_model, CFG.n_heads
    dh = d // H
    h = RNG.standard_normal((B, R, C, d)).astype(np.float32)
    pw = _proj_weights(p)
    with torch.no_grad():
        x = torch.from_numpy(h)
        q, k, v = (F.linear(x, pw[n + "_proj.weight"], pw[n + "_proj.bias"]).reshape(B, R, C, H, dh) for n in "qkv")
        # per head: C "tokens" with R * dh features each (the rows concatenated); default scale = (R * dh)^-0.5
        cat = lambda t: t.permute(0, 3, 2, 1, 4).reshape(B, H, C, R * dh)          # noqa: E731
        ctx = F.scaled_dot_product_attention(cat(q), cat(k), cat(v))              # [B, H, C, R * dh]
        ctx = ctx.reshape(B, H, C, R, dh).permute(0, 3, 2, 1, 4).reshape(B, R, C, d)
        want = F.linear(ctx, pw["out_proj.weight"], pw["out_proj.bias"]).numpy()
    got = row_attention(W, p, CFG, h)
    assert np.abs(got - want).max() < 3e-5 * max(1.0, np.abs(want).max())
    # one row: plain attention along the columns, i.e. torch.nn.MultiheadAttention once more
    m = torch.nn.MultiheadAttention(d, H, bias=True, batch_first=True)
    with torch.no_grad():
        m.in_proj_weight.copy_(torch.cat([pw["q_proj.weight"], pw["k_proj.weight"], pw["v_proj.weight"]]))
        m.in_proj_bias.copy_(torch.cat([pw["q_proj.bias"], pw["k_proj.bias"], pw["v_proj.bias"]]))
        m.out_proj.weight.copy_(pw["out_proj.weight"])
        m.out_proj.bias.copy_(pw["out_proj.bias"])
        x1 = torch.from_numpy(h[:, :1]).reshape(B, C, d)
        want1 = m(x1, x1, x1, need_weights=False)[0].reshape(B, 1, C, d).numpy()
    assert np.abs(row_attention(W, p, CFG, h[:, :1]) - want1).max() < 3e-5 * max(1.0, np.abs(want1).max())
