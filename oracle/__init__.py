"""CPU oracle for the AV-former transformer hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: it is
imported by ``tests/``, by ``__graft_entry__.smoke()`` and by ``bench.py``'s
``cpu_baseline`` leg, and only as the checker / the timed CPU baseline.  The product
path (the ``*_amd`` package) never imports it and fails loudly if its HIP library is
missing.

Pinning: the reference repository has no tests and no golden vectors of its own
(SURVEY.md section 4), so this restatement is pinned by fixtures generated from the
reference's own ``models/heads.py``, ``models/loss.py``, ``models/tformer.py`` and
``models/vformer.py`` imported unmodified in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``; checked by
``tests/test_oracle_golden.py``).
"""
from .reference_math import (  # noqa: F401
    AU_POS_WEIGHT,
    au_former_forward,
    va_former_forward,
    au_head_forward,
    au_loss,
    attention_forward,
    feedforward_forward,
    gelu_tanh,
    layer_forward,
    layernorm,
    resformer_tokens_forward,
    tformer_forward,
    transformer_forward,
    transformer_param_names,
    init_transformer_state,
)
from .mx8 import mx8_dequant, mx8_quant  # noqa: F401,E402
from . import bf16x3  # noqa: F401,E402
