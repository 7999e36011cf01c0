#!/bin/bash
# Tied row attention: K_r / V_r tiles by LDS-DMA into a three-buffer ring (PGIBBS_MSA_ROW_DMA=1, round 5) against the register-staged
# tiles (=0).  Needs tools/probes/msa_row_attention_lds_dma.patch applied to csrc/msa_attention.hip (the variant lost and was taken
# out of the library: profiles/r05_msa_row_attention_lds_dma_ab.txt).  (1) logits of an MSA forward bit for bit, several shapes incl. the split-R form; (2) configs 4 and 5, interleaved.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
cat > /tmp/rowdma_logits.py <<PY
import sys, numpy as np, warnings
sys.path.insert(0, "$ROOT")
from protein_gibbs_sampler_amd import models, weights
cfg = weights.make_config(weights.MSA1B_CONFIG, n_layers=2)
sd = weights.synthetic_state_dict(cfg, seed=4, std=0.04, embed_std=0.3, ln_jitter=0.1)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    lm = models.ESM_MSA1(state_dict=sd, config=cfg, precision=sys.argv[2]).model.to("cuda:0")
rng = np.random.default_rng(0)
out = {}
for B, R, C in ((3, 8, 97), (2, 16, 257), (9, 32, 257), (1, 32, 301), (1, 24, 161), (5, 4, 113), (2, 7, 384)):
    tok = np.concatenate([np.zeros((B, R, 1), np.int64), rng.integers(4, 24, (B, R, C - 1))], axis=2)
    out["%d_%d_%d" % (B, R, C)] = lm.forward_logits(tok)
np.savez(sys.argv[1], **out)
PY
for prec in bf16 fp16; do
  PGIBBS_MSA_ROW_DMA=0 python /tmp/rowdma_logits.py /tmp/rowdma_0.npz $prec 2>/dev/null
  PGIBBS_MSA_ROW_DMA=1 python /tmp/rowdma_logits.py /tmp/rowdma_1.npz $prec 2>/dev/null
  python3 - $prec <<'PY'
import numpy as np, sys
a, b = np.load("/tmp/rowdma_0.npz"), np.load("/tmp/rowdma_1.npz")
for k in a.files:
    same = np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32))
    print("%s logits %-12s finite %s  DMA == register-staged bit for bit: %s  (max |diff| %g)" % (sys.argv[1], k, np.isfinite(b[k]).all(), same, np.abs(a[k] - b[k]).max()))
PY
done
for rep in 1 2; do
  CFG=4 bash tools/r03_exp_msa.sh "regs:PGIBBS_MSA_ROW_DMA=0" "dma:PGIBBS_MSA_ROW_DMA=1"
done
CFG=5 bash tools/r03_exp_msa.sh "regs:PGIBBS_MSA_ROW_DMA=0" "dma:PGIBBS_MSA_ROW_DMA=1"
