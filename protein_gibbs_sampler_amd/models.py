"""Model wrappers with the three attributes the samplers expect: `.model`, `.alphabet`,
`.batch_converter` -- the plug-in contract of /root/reference/src/pgen/models.py:59-88.

`.model` is a NativeMaskedLM (HIP engine handle) instead of a fair-esm nn.Module.  Weights: a real
fair-esm checkpoint when one is given or found in torch's hub cache.  Without one the constructor RAISES
(as `esm.pretrained.*` fails in the reference when it cannot load) unless the caller opts in to seeded
synthetic weights of the same architecture with `synthetic=True` (benchmarks, tests): samples from those
are not biologically meaningful.
"""

from . import weights as _w
from .alphabet import Alphabet
from .engine import NativeMaskedLM


def _resolve_weights(cfg, state_dict, checkpoint, filename, seed, synthetic, explicit_config=False):
    """-> (state dict, config).  A checkpoint file brings its own hyper-parameters (weights.config_from_checkpoint): `cfg` is the
    wrapper's architecture and the default for files that carry none."""
    if state_dict is not None:
        return state_dict, cfg
    path = checkpoint or _w.find_cached_checkpoint(filename)
    if path:
        return _w.load_fair_esm_checkpoint(path, cfg, return_config=True, explicit_config=explicit_config)
    if not synthetic:
        raise FileNotFoundError(
            "no %s checkpoint: pass checkpoint=<path to the fair-esm .pt file> (CLI: --checkpoint) or place it in "
            "~/.cache/torch/hub/checkpoints/.  Seeded synthetic weights of the same architecture are available only as an "
            "explicit opt-in (synthetic=True / --synthetic-weights): output sampled from them is not biologically "
            "meaningful." % filename)
    return _w.synthetic_state_dict(cfg, seed=seed), cfg


class _Wrapper:
    def __init__(self, cfg, alphabet, msa, state_dict, checkpoint, filename, seed, precision, synthetic, explicit_config=False):
        sd, cfg = _resolve_weights(cfg, state_dict, checkpoint, filename, seed, synthetic, explicit_config)
        self.cfg = cfg
        self.alphabet = alphabet
        self.batch_converter = alphabet.get_batch_converter(msa=msa)
        self.model = NativeMaskedLM(cfg, sd, precision)


class ESM1b(_Wrapper):
    """esm1b_t33_650M_UR50S (models.py:59-62)."""

    def __init__(self, state_dict=None, checkpoint=None, seed=0, precision="auto", config=None, synthetic=False):
        super().__init__(config or dict(_w.ESM1B_CONFIG), Alphabet(True, True), False, state_dict, checkpoint,
                         "esm1b_t33_650M_UR50S.pt", seed, precision, synthetic, config is not None)


class ESM1v(_Wrapper):
    """esm1v_t33_650M_UR90S (models.py:64-67): same architecture as ESM-1b, different weights."""

    def __init__(self, state_dict=None, checkpoint=None, seed=0, precision="auto", config=None, synthetic=False):
        super().__init__(config or dict(_w.ESM1B_CONFIG), Alphabet(True, True), False, state_dict, checkpoint,
                         "esm1v_t33_650M_UR90S_1.pt", seed, precision, synthetic, config is not None)


class ESM_MSA1(_Wrapper):
    """esm_msa1b_t12_100M_UR50S (models.py:84-88) with the reference's patched MSA batch converter."""

    def __init__(self, state_dict=None, checkpoint=None, seed=0, precision="auto", config=None, synthetic=False):
        super().__init__(config or dict(_w.MSA1B_CONFIG), Alphabet(True, False), True, state_dict, checkpoint,
                         "esm_msa1b_t12_100M_UR50S.pt", seed, precision, synthetic, config is not None)


class _ESM1(_Wrapper):
    """ESM-1 family (models.py:69-82): the 35-token "ESM-1" alphabet, <cls> prepended, no <eos>."""
    _cfg, _file = None, None

    def __init__(self, state_dict=None, checkpoint=None, seed=0, precision="auto", config=None, synthetic=False):
        super().__init__(config or dict(self._cfg), Alphabet(True, False, arch="ESM-1"), False, state_dict, checkpoint,
                         self._file, seed, precision, synthetic, config is not None)


class ESM6(_ESM1):
    """esm1_t6_43M_UR50S (models.py:69-72): the model the reference's own unit tests load (test/test_esm_sampler.py:10-18)."""
    _cfg, _file = _w.ESM1_T6_CONFIG, "esm1_t6_43M_UR50S.pt"


class ESM12(_ESM1):
    """esm1_t12_85M_UR50S (models.py:74-77)."""
    _cfg, _file = _w.ESM1_T12_CONFIG, "esm1_t12_85M_UR50S.pt"


class ESM34(_ESM1):
    """esm1_t34_670M_UR50S (models.py:79-82)."""
    _cfg, _file = _w.ESM1_T34_CONFIG, "esm1_t34_670M_UR50S.pt"
