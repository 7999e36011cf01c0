"""ESM-1 oracle (oracle/esm1_forward.py): the pieces that differ from ESM-1b, against independent implementations available
offline -- torch.nn.MultiheadAttention(add_bias_kv=True) for the attention block with the extra bias_k / bias_v key (the same
fairseq-derived semantics fair-esm's MultiheadAttention has), torch LayerNorm at eps 1e-12, and closed-form properties of the
sinusoidal table.  The whole model stays "parity unpinned" against the reference (its KATs need the 43 M checkpoint)."""
import numpy as np
import torch

from oracle import esm1_forward as E


def test_sinusoidal_table_properties():
    t = E.sinusoidal_table(40, 768, 1)
    assert t.shape == (40, 768) and (t[1] == 0).all()                       # padding row
    assert np.allclose(t[0, :384], 0) and np.allclose(t[0, 384:], 1)        # position 0: sin 0, cos 0
    inv = np.exp(np.arange(384) * -(np.log(10000.0) / 383))
    assert np.allclose(t[7, :384], np.sin(7 * inv), atol=2e-6) and np.allclose(t[7, 384:], np.cos(7 * inv), atol=2e-6)
    assert abs(inv[-1] - 1e-4) < 1e-9                                        # last frequency = 1 / 10000


def test_attention_with_bias_kv_matches_torch_multihead_attention():
    cfg = E.Esm1Config(d_model=128, n_layers=1, n_heads=2, d_ffn=256)
    w = E.synthetic_esm1_weights(cfg, seed=3, std=0.08)
    rng = np.random.default_rng(0)
    B, T, d = 3, 11, 128
    h = rng.standard_normal((B, T, d)).astype(np.float32)
    pad = np.zeros((B, T), bool)
    pad[1, 8:] = True                                                        # a right-padded row: key_padding_mask
    got = E.mha_bias_kv(w, "layers.0.self_attn.", cfg, h, pad)
    m = torch.nn.MultiheadAttention(d, 2, bias=True, add_bias_kv=True, batch_first=True)
    p = "layers.0.self_attn."
    with torch.no_grad():
        m.in_proj_weight.copy_(torch.from_numpy(np.concatenate([w[p + "q_proj.weight"], w[p + "k_proj.weight"], w[p + "v_proj.weight"]])))
        m.in_proj_bias.copy_(torch.from_numpy(np.concatenate([w[p + "q_proj.bias"], w[p + "k_proj.bias"], w[p + "v_proj.bias"]])))
        m.out_proj.weight.copy_(torch.from_numpy(w[p + "out_proj.weight"]))
        m.out_proj.bias.copy_(torch.from_numpy(w[p + "out_proj.bias"]))
        m.bias_k.copy_(torch.from_numpy(w[p + "bias_k"]).view(1, 1, d))
        m.bias_v.copy_(torch.from_numpy(w[p + "bias_v"]).view(1, 1, d))
        x = torch.from_numpy(h)
        want = m(x, x, x, key_padding_mask=torch.from_numpy(pad), need_weights=False)[0].numpy()
    ok = ~pad                                                                # queries at <pad> rows are never read
    assert np.abs(got[ok] - want[ok]).max() < 2e-5


def test_whole_forward_shapes_and_pad_independence():
    cfg = E.Esm1Config(d_model=128, n_layers=2, n_heads=2, d_ffn=256)
    w = E.synthetic_esm1_weights(cfg, seed=4, std=0.05, embed_std=0.3, ln_jitter=0.1)
    rng = np.random.default_rng(1)
    tok = np.concatenate([np.full((2, 1), 32), rng.integers(4, 24, (2, 9))], axis=1)
    tok[0, 4] = 33
    out = E.esm1_forward(w, cfg, tok)
    assert out.shape == (2, 10, 35) and np.isfinite(out).all()
    # right padding does not change the real positions (key-padding mask + positions from cumsum(non-pad))
    padded = np.concatenate([tok, np.full((2, 3), 1)], axis=1)
    out2 = E.esm1_forward(w, cfg, padded)
    assert np.abs(out2[:, :10] - out).max() < 2e-5
    # LayerNorm at eps 1e-12 equals torch's
    x = rng.standard_normal((5, 128)).astype(np.float32)
    g, b = w["layers.0.self_attn_layer_norm.weight"], w["layers.0.self_attn_layer_norm.bias"]
    want = torch.nn.functional.layer_norm(torch.from_numpy(x), (128,), torch.from_numpy(g), torch.from_numpy(b), 1e-12).numpy()
    assert np.abs(E.layer_norm(x, g, b, 1e-12) - want).max() < 2e-6
