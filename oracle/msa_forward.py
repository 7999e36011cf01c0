"""ORACLE (test infrastructure only -- never imported by the product path).

fp32 CPU restatement of the ESM-MSA-1b forward pass the reference calls as
`self.model.model(batch)["logits"]` (/root/reference/src/pgen/esm_msa_sampler.py:136,236).

The arithmetic lives in the third-party package `fair-esm` (`esm.model.msa_transformer.MSATransformer`,
`esm.axial_attention.RowSelfAttention / ColumnSelfAttention`; unpinned git HEAD, conda_env.yml:21), which
is not installed here and has no second implementation in this container.  Restated from its published
algorithm:

  x = embed_tokens[tok] + embed_positions[pos per row] + msa_position_embedding[:, :R]
  x = LN_before(x)
  12 x { x += out_proj(RowAttn(LN(x)));  x += out_proj(ColAttn(LN(x)));  x += fc2(gelu(fc1(LN(x)))) }
  x = LN_after(x);  logits = RobertaLMHead(x)  (tied decoder)
  RowAttn (tied):  q *= dh^-0.5 / sqrt(R);  A[h,i,j] = sum_r sum_d q[r,i,h,d] k[r,j,h,d];  P = softmax_j A;
                   ctx[r,i] = sum_j P[h,i,j] v[r,j,h,:]
  ColAttn:         per column c, attention along the R rows, q *= dh^-0.5

PARITY UNPINNED: the reference's numeric KATs for this boundary
(/root/reference/test/test_esm_msa_sampler.py:248-397, 561-565) need the pretrained
esm_msa1b_t12_100M_UR50S checkpoint, unavailable offline, and no third-party implementation exists
here.  The restatement is checked (tests/test_oracle_msa.py) by the invariants fair-esm's own code
relies on -- R == 1 column-attention shortcut, row-permutation equivariance, independence of MSAs in a
batch, agreement of the shared sub-blocks (embedding, LayerNorm, FFN, LM head) with the
HF-corroborated ESM-1b oracle -- and against tests/_msa_alt.py, a second restatement written in
fair-esm's own R x C x B x D layout with its einsum strings ("rinhd,rjnhd->hnij", "icnhd,jcnhd->hcnij")
in torch, including fair-esm's chunked `max_tokens_per_msa` paths (chunked == unchunked).
Since round 4 the two attention forms that only this model has are also checked against torch's own attention code, which shares
nothing with this file: column attention against torch.nn.MultiheadAttention along the rows of every column (with and without a
key-padding mask), tied row attention against torch.nn.functional.scaled_dot_product_attention over per-head feature vectors that
concatenate the R rows (the tied map with q * dh^-0.5 / sqrt(R) IS scaled-dot-product attention of dimension R * dh).  What no
independent code confirms is only the ASSEMBLY (the embedding sum with msa_position_embedding, the order row -> column -> FFN).
"""
import numpy as np

from .esm_forward import F32, gelu, layer_norm, linear, lm_head, softmax_lastdim


class MsaConfig:
    def __init__(self, vocab=33, d_model=768, n_layers=12, n_heads=12, d_ffn=3072, max_pos=1024, max_rows=1024,
                 pad_idx=1, mask_idx=32, cls_idx=0, eos_idx=2):
        self.vocab, self.d_model, self.n_layers, self.n_heads, self.d_ffn = vocab, d_model, n_layers, n_heads, d_ffn
        self.max_pos, self.max_rows = max_pos, max_rows
        self.pad_idx, self.mask_idx, self.cls_idx, self.eos_idx = pad_idx, mask_idx, cls_idx, eos_idx


def msa_embed(w, cfg, tokens):
    tokens = np.asarray(tokens)
    B, R, C = tokens.shape
    if R > cfg.max_rows:
        raise RuntimeError("MSA has more rows than msa_position_embedding")
    pad = tokens == cfg.pad_idx
    x = w["embed_tokens.weight"][tokens].astype(F32)
    nonpad = (~pad).astype(np.int64)
    pos = np.cumsum(nonpad, axis=2) * nonpad + cfg.pad_idx
    x = x + w["embed_positions.weight"][pos]
    x = x + w["msa_position_embedding"].reshape(-1, cfg.d_model)[:R][None, :, None, :]
    x = layer_norm(x, w["emb_layer_norm_before.weight"], w["emb_layer_norm_before.bias"])
    return np.where(pad[..., None], F32(0), x).astype(F32)


def row_attention(w, p, cfg, h, pad=None):
    """Tied row attention: one C x C map per head shared by all rows of an MSA.
    pad (bool [B,R,C], only when the batch holds <pad>: ragged MSA lists, esm_msa_sampler.py:341) -- fair-esm's RowSelfAttention
    zeroes q at padded positions ("we take a sum across the alignment axis") and fills the scores of the key columns that are
    <pad> in ROW 0 with -10000 (a finite fill, not -inf); the 1/sqrt(R) uses the padded row count."""
    B, R, C, d = h.shape
    H = cfg.n_heads
    dh = d // H
    scale = F32(dh ** -0.5) / F32(np.sqrt(R))
    q = linear(h, w[p + "q_proj.weight"], w[p + "q_proj.bias"]).reshape(B, R, C, H, dh) * scale
    k = linear(h, w[p + "k_proj.weight"], w[p + "k_proj.bias"]).reshape(B, R, C, H, dh)
    v = linear(h, w[p + "v_proj.weight"], w[p + "v_proj.bias"]).reshape(B, R, C, H, dh)
    if pad is not None:
        q = np.where(pad[..., None, None], F32(0), q)
    # A[b,h,i,j] = sum_{r,d} q[b,r,i,h,d] k[b,r,j,h,d] as one matrix product per (b, h) over the R * dh concatenated features, and
    # ctx[b,r,i,h,:] = sum_j P[b,h,i,j] v[b,r,j,h,:] likewise ("brihd,brjhd->bhij" / "bhij,brjhd->brihd": numpy's einsum walks these
    # in single-threaded C loops, 40 % of a 128 x 513 alignment's 400 s; the BLAS products are the same sums in another order)
    qc = np.ascontiguousarray(q.transpose(0, 3, 2, 1, 4)).reshape(B, H, C, R * dh)
    kc = np.ascontiguousarray(k.transpose(0, 3, 2, 1, 4)).reshape(B, H, C, R * dh)
    a = np.matmul(qc, kc.transpose(0, 1, 3, 2)).astype(F32)                        # [B, H, C(i), C(j)]
    if pad is not None:
        a = np.where(pad[:, 0][:, None, None, :], F32(-10000.0), a)
    pr = softmax_lastdim(a)
    vc = np.ascontiguousarray(v.transpose(0, 3, 2, 1, 4)).reshape(B, H, C, R * dh)  # [B, H, C(j), R * dh]
    ctx = np.matmul(pr, vc).astype(F32).reshape(B, H, C, R, dh).transpose(0, 3, 2, 1, 4).reshape(B, R, C, d)
    return linear(ctx, w[p + "out_proj.weight"], w[p + "out_proj.bias"])


def column_attention(w, p, cfg, h, pad=None):
    """pad: fair-esm's ColumnSelfAttention fills the scores of key rows that are <pad> at that column with -10000."""
    B, R, C, d = h.shape
    H = cfg.n_heads
    dh = d // H
    if R == 1:   # fair-esm shortcut: softmax over one key is 1
        return linear(linear(h, w[p + "v_proj.weight"], w[p + "v_proj.bias"]), w[p + "out_proj.weight"], w[p + "out_proj.bias"])
    q = linear(h, w[p + "q_proj.weight"], w[p + "q_proj.bias"]).reshape(B, R, C, H, dh) * F32(dh ** -0.5)
    k = linear(h, w[p + "k_proj.weight"], w[p + "k_proj.bias"]).reshape(B, R, C, H, dh)
    v = linear(h, w[p + "v_proj.weight"], w[p + "v_proj.bias"]).reshape(B, R, C, H, dh)
    # per (b, h, column c): attention along the rows ("bichd,bjchd->bhcij" / "bhcij,bjchd->bichd" as batched matrix products)
    qc = q.transpose(0, 3, 2, 1, 4)                                                  # [B, H, C, R(i), dh]
    kc = k.transpose(0, 3, 2, 4, 1)                                                  # [B, H, C, dh, R(j)]
    a = np.matmul(qc, kc).astype(F32)                                                # [B, H, C, R(i), R(j)]
    if pad is not None:
        a = np.where(pad.transpose(0, 2, 1)[:, None, :, None, :], F32(-10000.0), a)      # [B,1,C,1,R(j)]
    pr = softmax_lastdim(a)
    ctx = np.matmul(pr, v.transpose(0, 3, 2, 1, 4)).astype(F32)                      # [B, H, C, R(i), dh]
    ctx = ctx.transpose(0, 3, 2, 1, 4).reshape(B, R, C, d)
    return linear(ctx, w[p + "out_proj.weight"], w[p + "out_proj.bias"])


def msa_trunk(w, cfg, tokens):
    x = msa_embed(w, cfg, tokens)
    pad = np.asarray(tokens) == cfg.pad_idx
    pad = pad if pad.any() else None              # fair-esm: `if not padding_mask.any(): padding_mask = None`
    for i in range(cfg.n_layers):
        p = "layers.%d." % i
        h = layer_norm(x, w[p + "row_self_attention.layer_norm.weight"], w[p + "row_self_attention.layer_norm.bias"])
        x = (x + row_attention(w, p + "row_self_attention.layer.", cfg, h, pad)).astype(F32)
        h = layer_norm(x, w[p + "column_self_attention.layer_norm.weight"], w[p + "column_self_attention.layer_norm.bias"])
        x = (x + column_attention(w, p + "column_self_attention.layer.", cfg, h, pad)).astype(F32)
        h = layer_norm(x, w[p + "feed_forward_layer.layer_norm.weight"], w[p + "feed_forward_layer.layer_norm.bias"])
        h = gelu(linear(h, w[p + "feed_forward_layer.layer.fc1.weight"], w[p + "feed_forward_layer.layer.fc1.bias"]))
        x = (x + linear(h, w[p + "feed_forward_layer.layer.fc2.weight"], w[p + "feed_forward_layer.layer.fc2.bias"])).astype(F32)
    return layer_norm(x, w["emb_layer_norm_after.weight"], w["emb_layer_norm_after.bias"])


def msa_forward(w, cfg, tokens):
    """tokens int [B,R,C] -> logits fp32 [B,R,C,V]."""
    return lm_head(w, msa_trunk(w, cfg, tokens))


def synthetic_msa_weights(cfg, seed=0, std=0.02, embed_std=None, ln_jitter=0.0):
    rng = np.random.default_rng(seed)
    d, f, V = cfg.d_model, cfg.d_ffn, cfg.vocab
    es = std if embed_std is None else embed_std

    def n(*shape, s=std):
        return (rng.standard_normal(shape, dtype=np.float32) * np.float32(s)).astype(F32)

    def ln(prefix, w):
        w[prefix + ".weight"] = (1.0 + ln_jitter * rng.standard_normal(d, dtype=np.float32)).astype(F32)
        w[prefix + ".bias"] = (ln_jitter * rng.standard_normal(d, dtype=np.float32)).astype(F32)

    w = {"embed_tokens.weight": n(V, d, s=es), "embed_positions.weight": n(cfg.max_pos + cfg.pad_idx + 1, d, s=es),
         "msa_position_embedding": n(1, cfg.max_rows, 1, d, s=es)}
    ln("emb_layer_norm_before", w)
    for i in range(cfg.n_layers):
        p = "layers.%d." % i
        for blk in ("row_self_attention", "column_self_attention"):
            for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
                w[p + blk + ".layer." + nm + ".weight"] = n(d, d)
                w[p + blk + ".layer." + nm + ".bias"] = n(d)
            ln(p + blk + ".layer_norm", w)
        w[p + "feed_forward_layer.layer.fc1.weight"] = n(f, d)
        w[p + "feed_forward_layer.layer.fc1.bias"] = n(f)
        w[p + "feed_forward_layer.layer.fc2.weight"] = n(d, f)
        w[p + "feed_forward_layer.layer.fc2.bias"] = n(d)
        ln(p + "feed_forward_layer.layer_norm", w)
    ln("emb_layer_norm_after", w)
    w["lm_head.dense.weight"] = n(d, d)
    w["lm_head.dense.bias"] = n(d)
    ln("lm_head.layer_norm", w)
    w["lm_head.bias"] = n(V)
    return w
