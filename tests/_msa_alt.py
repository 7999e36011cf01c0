"""A second, independently laid-out restatement of the ESM-MSA-1b forward, used only to cross-check oracle/msa_forward.py.

oracle/msa_forward.py works on [B, R, C, D] numpy arrays with `brihd` einsums.  This file follows fair-esm's OWN tensor
layout and contraction strings (esm/axial_attention.py, [recalled]: the package is absent here): activations are permuted
to [R, C, B, D] before the layer stack (`x.permute(1, 2, 0, 3)`), tied row attention contracts
"rinhd,rjnhd->hnij" / "hnij,rjnhd->rinhd", column attention "icnhd,jcnhd->hcnij" / "hcnij,jcnhd->icnhd", and both have the
memory-bounded paths fair-esm takes when R*C > max_tokens_per_msa without grad: row attention accumulates the score map over
row chunks (scaling still from the FULL row count) and then applies the shared probabilities chunk by chunk; column
attention processes column chunks independently.  Written in torch so that not even the BLAS calls are shared.
"""
import math

import torch
import torch.nn.functional as F


def _lin(x, w, p):
    return F.linear(x, torch.from_numpy(w[p + ".weight"]), torch.from_numpy(w[p + ".bias"]))


def _ln(x, w, p):
    return F.layer_norm(x, (x.shape[-1],), torch.from_numpy(w[p + ".weight"]), torch.from_numpy(w[p + ".bias"]), 1e-5)


def _row_attention(w, p, x, H, max_tokens, padding_mask=None):
    R, C, B, D = x.shape
    dh = D // H
    scaling = (dh ** -0.5) / math.sqrt(R)                       # align_scaling: from the full number of rows

    def weights(xc, pm):                                        # pm: padding mask [B, rows of this chunk, C] or None
        q = _lin(xc, w, p + "q_proj").view(xc.shape[0], C, B, H, dh) * scaling
        k = _lin(xc, w, p + "k_proj").view(xc.shape[0], C, B, H, dh)
        if pm is not None:
            # "Zero out any padded aligned positions - this is important since we take a sum across the alignment axis."
            q = q * (1 - pm.permute(1, 2, 0).unsqueeze(3).unsqueeze(4).to(q))
        a = torch.einsum("rinhd,rjnhd->hnij", q, k)
        if pm is not None:
            a = a.masked_fill(pm[:, 0].unsqueeze(0).unsqueeze(2), -10000)      # row 0 OF THE CHUNK, as fair-esm's batched path does
        return a

    def update(xc, probs):
        v = _lin(xc, w, p + "v_proj").view(xc.shape[0], C, B, H, dh)
        ctx = torch.einsum("hnij,rjnhd->rinhd", probs, v).contiguous().view(xc.shape[0], C, B, D)
        return _lin(ctx, w, p + "out_proj")

    if R * C <= max_tokens:
        return update(x, weights(x, padding_mask).softmax(-1))
    max_rows = max(1, max_tokens // C)
    attns = 0
    for s in range(0, R, max_rows):
        attns = attns + weights(x[s:s + max_rows], None if padding_mask is None else padding_mask[:, s:s + max_rows])
    probs = attns.softmax(-1)
    return torch.cat([update(x[s:s + max_rows], probs) for s in range(0, R, max_rows)], 0)


def _column_attention(w, p, x, H, max_tokens, padding_mask=None):
    R, C, B, D = x.shape
    dh = D // H

    def block(xc, pm=None):
        if R == 1:
            return _lin(_lin(xc, w, p + "v_proj"), w, p + "out_proj")
        c = xc.shape[1]
        q = _lin(xc, w, p + "q_proj").view(R, c, B, H, dh) * dh ** -0.5
        k = _lin(xc, w, p + "k_proj").view(R, c, B, H, dh)
        v = _lin(xc, w, p + "v_proj").view(R, c, B, H, dh)
        a = torch.einsum("icnhd,jcnhd->hcnij", q, k)
        if pm is not None:
            a = a.masked_fill(pm.permute(2, 0, 1).unsqueeze(0).unsqueeze(3), -10000)
        probs = a.softmax(-1)
        ctx = torch.einsum("hcnij,jcnhd->icnhd", probs, v).contiguous().view(R, c, B, D)
        return _lin(ctx, w, p + "out_proj")

    if R * C <= max_tokens:
        return block(x, padding_mask)
    max_cols = max(1, max_tokens // R)
    return torch.cat([block(x[:, s:s + max_cols], None if padding_mask is None else padding_mask[:, :, s:s + max_cols])
                      for s in range(0, C, max_cols)], 1)


def msa_forward_alt(w, n_layers, n_heads, tokens, pad_idx=1, max_tokens_per_msa=2 ** 14):
    """tokens int [B, R, C] -> logits float32 [B, R, C, V], fair-esm layout, optional chunked attention."""
    tok = torch.as_tensor(tokens, dtype=torch.long)
    B, R, C = tok.shape
    E = torch.from_numpy(w["embed_tokens.weight"])
    pad = tok.eq(pad_idx)
    x = E[tok]
    nonpad = (~pad).long()
    pos = torch.cumsum(nonpad, dim=2) * nonpad + pad_idx
    x = x + torch.from_numpy(w["embed_positions.weight"])[pos.view(B * R, C)].view(B, R, C, -1)
    x = x + torch.from_numpy(w["msa_position_embedding"])[:, :R]
    x = _ln(x, w, "emb_layer_norm_before")
    x = x * (1 - pad.unsqueeze(-1).type_as(x))
    x = x.permute(1, 2, 0, 3)                                   # B x R x C x D -> R x C x B x D
    pm = pad if bool(pad.any()) else None                       # `if not padding_mask.any(): padding_mask = None`
    for i in range(n_layers):
        p = "layers.%d." % i
        x = x + _row_attention(w, p + "row_self_attention.layer.", _ln(x, w, p + "row_self_attention.layer_norm"), n_heads,
                               max_tokens_per_msa, pm)
        x = x + _column_attention(w, p + "column_self_attention.layer.", _ln(x, w, p + "column_self_attention.layer_norm"),
                                  n_heads, max_tokens_per_msa, pm)
        h = _ln(x, w, p + "feed_forward_layer.layer_norm")
        h = F.gelu(_lin(h, w, p + "feed_forward_layer.layer.fc1"))
        x = x + _lin(h, w, p + "feed_forward_layer.layer.fc2")
    x = _ln(x, w, "emb_layer_norm_after").permute(2, 0, 1, 3)   # R x C x B x D -> B x R x C x D
    h = _ln(F.gelu(_lin(x, w, "lm_head.dense")), w, "lm_head.layer_norm")
    return (F.linear(h, E) + torch.from_numpy(w["lm_head.bias"])).numpy()
