#!/usr/bin/env python3
"""Which 16-bit rounding contributes what to the throughput mode's logit error?

CPU study on the fp32 oracle (checker infrastructure; nothing here is product code): the full-size ESM-1b forward
(33 layers, d = 1280) on the bench's realistic-scale synthetic weights, with the engine's rounding points emulated one at a
time -- the stored 16-bit operands of the bf16 / fp16 throughput modes:

  W    every projection weight (W_q carries the folded 64^-0.5: exact)       csrc/engine.hip Uploader::dense
  ln   LayerNorm outputs feeding QKV / fc1 / the LM-head dense                elementwise.hip layernorm_bf16_kernel
  qkv  q (pre-scaled), k, v as stored by the QKV epilogue                     gemm epilogue EPI_BF16
  p    unnormalised softmax numerators exp(s - max) fed to the P.V MFMA       attention.hip (sum and 1/sum stay fp32)
  ctx  attention context rows                                                 attention.hip store
  ffn  GELU(fc1) rows                                                         gemm epilogue EPI_BF16_GELU

Accumulation is fp32 everywhere (as the MFMA accumulates), the residual stream, LayerNorm statistics, softmax and the LM-head
tail are fp32.  Output: max / mean |logit error| against the unrounded fp32 forward, per site and dtype, as a table.

  python tests/rounding_ablation.py [--chains 3] [--layers 33] [--out profiles/r04_rounding_ablation.txt]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import esm_forward as O  # noqa: E402

F32 = np.float32
SITES = ("W", "ln", "qkv", "p", "ctx", "ffn")


def rnd_bf16(a):
    u = np.ascontiguousarray(a, dtype=F32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(F32).reshape(np.shape(a))


def rnd_f16(a):
    return np.asarray(a, dtype=F32).astype(np.float16).astype(F32)


def forward(w, cfg, tokens, sites, rnd):
    """oracle.esm_forward.esm1b_forward with `rnd` applied at the sites named in `sites`."""
    r = {s: (rnd if s in sites else (lambda a: a)) for s in SITES}
    x, pad = O.esm1b_embed(w, cfg, tokens)
    B, T, d = x.shape
    H, dh = cfg.n_heads, d // cfg.n_heads
    for i in range(cfg.n_layers):
        p = "layers.%d." % i
        h = r["ln"](O.layer_norm(x, w[p + "self_attn_layer_norm.weight"], w[p + "self_attn_layer_norm.bias"]))
        a = p + "self_attn."
        q = r["qkv"](O.linear(h, r["W"](w[a + "q_proj.weight"]) * F32(dh ** -0.5), w[a + "q_proj.bias"] * F32(dh ** -0.5)))
        k = r["qkv"](O.linear(h, r["W"](w[a + "k_proj.weight"]), w[a + "k_proj.bias"]))
        v = r["qkv"](O.linear(h, r["W"](w[a + "v_proj.weight"]), w[a + "v_proj.bias"]))
        q = q.reshape(B, T, H, dh).transpose(0, 2, 1, 3)
        k = k.reshape(B, T, H, dh).transpose(0, 2, 1, 3)
        v = v.reshape(B, T, H, dh).transpose(0, 2, 1, 3)
        s = (q @ k.transpose(0, 1, 3, 2)).astype(F32)
        e = np.exp(s - s.max(axis=-1, keepdims=True), dtype=F32)
        ctx = ((r["p"](e) @ v) / e.sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)      # fp32 sum of the UNROUNDED numerators
        ctx = r["ctx"](ctx.transpose(0, 2, 1, 3).reshape(B, T, d))
        x = x + O.linear(ctx, r["W"](w[a + "out_proj.weight"]), w[a + "out_proj.bias"])
        h = r["ln"](O.layer_norm(x, w[p + "final_layer_norm.weight"], w[p + "final_layer_norm.bias"]))
        h = r["ffn"](O.gelu(O.linear(h, r["W"](w[p + "fc1.weight"]), w[p + "fc1.bias"])))
        x = (x + O.linear(h, r["W"](w[p + "fc2.weight"]), w[p + "fc2.bias"])).astype(F32)
    x = r["ln"](O.layer_norm(x, w["emb_layer_norm_after.weight"], w["emb_layer_norm_after.bias"]))
    h = O.gelu(O.linear(x, r["W"](w["lm_head.dense.weight"]), w["lm_head.dense.bias"]))
    h = O.layer_norm(h, w["lm_head.layer_norm.weight"], w["lm_head.layer_norm.bias"])
    return (h @ w["embed_tokens.weight"].T + w["lm_head.bias"]).astype(F32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, default=3)
    ap.add_argument("--layers", type=int, default=33)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    cfg = O.EsmConfig(n_layers=args.layers)
    w = O.synthetic_esm_weights(cfg, seed=0, std=0.025, embed_std=0.3, ln_jitter=0.1)
    rng = np.random.default_rng(7)
    tok = np.concatenate([np.zeros((args.chains, 1), np.int64), rng.integers(4, 24, (args.chains, 256)), np.full((args.chains, 1), 2)], axis=1)
    tok[:, 3:250:10] = 32                                        # 25 masked positions per chain, as a Gibbs iteration has
    t0 = time.time()
    ref = forward(w, cfg, tok, (), None)
    lines = ["# rounding ablation on the fp32 oracle: ESM-1b %d layers x d=1280, %d chains x 258 tokens, logit std %.2f, max |logit| %.1f"
             % (cfg.n_layers, args.chains, ref.std(), np.abs(ref).max()),
             "# (one unrounded forward: %.0f s on this host)" % (time.time() - t0),
             "%-28s %12s %12s %12s %12s" % ("rounded site(s)", "bf16 max", "bf16 mean", "fp16 max", "fp16 mean")]
    print("\n".join(lines), flush=True)
    for sites in [(s,) for s in SITES] + [("ln", "qkv", "p", "ctx", "ffn"), SITES]:
        row = []
        for rnd in (rnd_bf16, rnd_f16):
            d = np.abs(forward(w, cfg, tok, sites, rnd) - ref)
            row += [d.max(), d.mean()]
        name = "+".join(sites) if len(sites) < 5 else ("all activations" if len(sites) == 5 else "all (the engine's mode)")
        lines.append("%-28s %12.3e %12.3e %12.3e %12.3e" % (name, *row))
        print(lines[-1], flush=True)
    if args.out:
        open(args.out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
