"""-m gpu: a FIXED budget of the random-shape drivers under tests/fuzz/ (fixed seeds: the same cases every run), so that the parity
evidence DESIGN.md cites from them is collected by the driver's `pytest -m gpu`.  Larger budgets / other seeds:
`python tests/fuzz/fuzz_gemm.py --cases 400 --seed 3`, `python tests/fuzz/fuzz_layer.py 60 1`."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(HERE, "fuzz", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("seed", [0, 1])
def test_fuzz_gemm_fixed_budget(seed):
    """NT GEMM kernels, bf16 and MX-FP8 operands, every fused epilogue, both output types: 150 random cases per seed against
    an fp32 product of the same rounded operands (the oracle's GELU / dequantiser as the checker)"""
    worst = _load("fuzz_gemm").run(cases=150, seed=seed)
    assert worst < 1.0


@pytest.mark.parametrize("seed", [0, 1])
def test_fuzz_layer_fixed_budget(seed):
    """whole stacks at random widths / token counts: the parity mode (both arithmetics) against the CPU oracle's autograd, the
    bf16 / bf16-residual / mx8 modes at their bounds - 12 configurations per seed"""
    _load("fuzz_layer").run(count=12, seed=seed, verbose=False)
