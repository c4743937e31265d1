"""-m gpu: the AU evaluation score accumulated on DEVICE tensors (no per-batch host synchronisation) against G12, the
values of the reference's metrics/accf1.py::MultiLabelAccF1 on the same seeded batches."""
import pytest
import torch

from test_metrics_cpu import G12_CASES, _run_g12

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", G12_CASES)
def test_metric_on_device_matches_reference_fixture(case):
    (acc, f1), (acc_ref, f1_ref) = _run_g12(case, "cuda")
    assert abs(acc - acc_ref) < 1e-12 and abs(f1 - f1_ref) < 1e-12, (case, acc, acc_ref, f1, f1_ref)


def test_metric_from_model_logits_end_to_end():
    """logits of the HIP model -> metric, against the same logits scored on the host"""
    import avformer_amd as A
    torch.manual_seed(0)
    model = A.build_model("avformer", task="AU", compute_dtype="f32").cuda().eval()
    g = torch.Generator().manual_seed(1)
    dev_m, host_m = A.metrics.MultiLabelAccF1(), A.metrics.MultiLabelAccF1()
    for _ in range(3):
        x = {"clip": torch.randn(16, 512, generator=g).cuda(), "audio_features": torch.randn(16, 512, generator=g).cuda()}
        y = (torch.rand(16, 12, generator=g) > 0.5).float()
        y[torch.rand(16, 12, generator=g) < 0.1] = -1
        with torch.no_grad():
            out = model(x)
        dev_m.update_from_logits(out, y.cuda())
        host_m.update_from_logits(out.cpu(), y)
    assert dev_m.get() == host_m.get()
