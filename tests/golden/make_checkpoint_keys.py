#!/usr/bin/env python3
"""Writes tests/golden/fair_esm_checkpoint_keys.json: the LITERAL on-disk key names and shapes of the two checkpoints the
reference loads through fair-esm (/root/reference/src/pgen/models.py:61,86):

  esm1b_t33_650M_UR50S.pt          args.arch = "roberta_large"      (esm.pretrained.esm1b_t33_650M_UR50S)
  esm_msa1b_t12_100M_UR50S.pt      args.arch = "msa_transformer"    (esm.pretrained.esm_msa1b_t12_100M_UR50S)

The names are spelled out here from the module tree of fair-esm (ProteinBertModel / MSATransformer, RobertaLMHead,
TransformerLayer, AxialTransformerLayer + NormalizedResidualBlock, ContactPredictionHead) under the fairseq prefixes of the v1
checkpoints -- NOT produced by protein_gibbs_sampler_amd.weights.to_fair_esm_checkpoint_layout (whose inverse the loader is), so
the loader is checked against an independent statement of the layout.  [recalled]: fair-esm and the .pt files are not available
offline; the day a checkpoint is supplied, `python tests/golden/make_checkpoint_keys.py --verify file.pt` compares this list
with the file's own keys.  In the MSA checkpoint "row" and "column" are exchanged on disk (fair-esm's loader swaps them back).
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def esm1b():
    d, f, V, L = 1280, 5120, 33, 33
    k = {"encoder.sentence_encoder.embed_tokens.weight": [V, d],
         "encoder.sentence_encoder.embed_positions.weight": [1026, d],
         "encoder.sentence_encoder.emb_layer_norm_before.weight": [d], "encoder.sentence_encoder.emb_layer_norm_before.bias": [d],
         "encoder.sentence_encoder.emb_layer_norm_after.weight": [d], "encoder.sentence_encoder.emb_layer_norm_after.bias": [d],
         "encoder.lm_head.weight": [V, d], "encoder.lm_head.bias": [V],
         "encoder.lm_head.dense.weight": [d, d], "encoder.lm_head.dense.bias": [d],
         "encoder.lm_head.layer_norm.weight": [d], "encoder.lm_head.layer_norm.bias": [d]}
    for i in range(L):
        p = "encoder.sentence_encoder.layers.%d." % i
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            k[p + "self_attn." + n + ".weight"] = [d, d]
            k[p + "self_attn." + n + ".bias"] = [d]
        k[p + "self_attn_layer_norm.weight"] = [d]
        k[p + "self_attn_layer_norm.bias"] = [d]
        k[p + "fc1.weight"] = [f, d]
        k[p + "fc1.bias"] = [f]
        k[p + "fc2.weight"] = [d, f]
        k[p + "fc2.bias"] = [d]
        k[p + "final_layer_norm.weight"] = [d]
        k[p + "final_layer_norm.bias"] = [d]
    # merged in by fair-esm from the separate *-contact-regression.pt; the samplers never read it
    extra = {"contact_head.regression.weight": [1, 660], "contact_head.regression.bias": [1]}
    return {"arch": "roberta_large", "file": "esm1b_t33_650M_UR50S.pt", "keys": k, "ignored_keys": extra}


def msa1b():
    d, f, V, L = 768, 3072, 33, 12
    k = {"encoder.sentence_encoder.embed_tokens.weight": [V, d],
         "encoder.sentence_encoder.embed_positions.weight": [1026, d],
         "encoder.sentence_encoder.msa_position_embedding": [1, 1024, 1, d],
         "encoder.sentence_encoder.emb_layer_norm_before.weight": [d], "encoder.sentence_encoder.emb_layer_norm_before.bias": [d],
         "encoder.sentence_encoder.emb_layer_norm_after.weight": [d], "encoder.sentence_encoder.emb_layer_norm_after.bias": [d],
         "encoder.lm_head.weight": [V, d], "encoder.lm_head.bias": [V],
         "encoder.lm_head.dense.weight": [d, d], "encoder.lm_head.dense.bias": [d],
         "encoder.lm_head.layer_norm.weight": [d], "encoder.lm_head.layer_norm.bias": [d]}
    for i in range(L):
        p = "encoder.sentence_encoder.layers.%d." % i
        for blk in ("row_self_attention", "column_self_attention"):
            for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
                k[p + blk + ".layer." + n + ".weight"] = [d, d]
                k[p + blk + ".layer." + n + ".bias"] = [d]
            k[p + blk + ".layer_norm.weight"] = [d]
            k[p + blk + ".layer_norm.bias"] = [d]
        k[p + "feed_forward_layer.layer.fc1.weight"] = [f, d]
        k[p + "feed_forward_layer.layer.fc1.bias"] = [f]
        k[p + "feed_forward_layer.layer.fc2.weight"] = [d, f]
        k[p + "feed_forward_layer.layer.fc2.bias"] = [d]
        k[p + "feed_forward_layer.layer_norm.weight"] = [d]
        k[p + "feed_forward_layer.layer_norm.bias"] = [d]
    extra = {"contact_head.regression.weight": [1, 144], "contact_head.regression.bias": [1]}
    return {"arch": "msa_transformer", "file": "esm_msa1b_t12_100M_UR50S.pt", "keys": k, "ignored_keys": extra,
            "note": "on disk the tensors named row_self_attention belong to the module's column_self_attention and vice versa"}


def main():
    out = {"esm1b": esm1b(), "msa1b": msa1b()}
    if len(sys.argv) > 2 and sys.argv[1] == "--verify":
        import torch
        blob = torch.load(sys.argv[2], map_location="cpu", weights_only=False)
        arch = getattr(blob.get("args"), "arch", None) or (blob.get("args") or {}).get("arch")
        want = next(v for v in out.values() if v["arch"] == arch)
        have = {k: list(v.shape) for k, v in blob["model"].items()}
        missing = sorted(set(want["keys"]) - set(have))
        extra = sorted(set(have) - set(want["keys"]) - set(want["ignored_keys"]))
        bad = sorted(k for k in want["keys"] if k in have and have[k] != want["keys"][k])
        print("arch %s: %d keys in the file, %d expected; missing %s; unexpected %s; shape mismatches %s"
              % (arch, len(have), len(want["keys"]), missing[:5], extra[:5], bad[:5]))
        sys.exit(1 if (missing or bad) else 0)
    with open(os.path.join(HERE, "fair_esm_checkpoint_keys.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print({k: len(v["keys"]) for k, v in out.items()})


if __name__ == "__main__":
    main()
