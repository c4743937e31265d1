"""AU evaluation metric of the reference: per-AU binary F1 + accuracy.

Reference: ``MultiLabelAccF1`` (metrics/accf1.py:47-77) fed with ``round(sigmoid(logits))`` per batch
(train.py:155) and scored as ``0.5*F1 + 0.5*acc`` (train.py:163).  The reference stacks every prediction on the host
and calls sklearn per AU at the end; here the sufficient statistics (true/false positives, false negatives, correct
and labelled counts per AU) accumulate on the device, so an evaluation loop never synchronises per batch.  Results
equal sklearn's ``f1_score(average='binary')`` / ``accuracy_score(normalize=False)`` path (F1 := 0 where a class has
no positive label and no positive prediction, sklearn's zero_division default).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch


class MultiLabelAccF1:
    def __init__(self, ignore_index: Optional[float] = -1, num_labels: int = 12):
        self.ignore_index = ignore_index
        self.num_labels = num_labels
        self._stats: Optional[torch.Tensor] = None  # [5, num_labels]: tp, fp, fn, correct, labelled

    def clear(self):
        self._stats = None

    @torch.no_grad()
    def update(self, y_pred: torch.Tensor, y_true: torch.Tensor):
        """y_pred: hard 0/1 predictions [B, num_labels] (``update_from_logits`` applies the reference's
        round(sigmoid(.))); y_true: labels in {0, 1, ignore_index}."""
        y_pred = y_pred.reshape(-1, self.num_labels)
        y_true = y_true.reshape(-1, self.num_labels).to(y_pred.device)
        keep = torch.ones_like(y_true, dtype=torch.bool) if self.ignore_index is None else (y_true != self.ignore_index)
        pos_p, pos_t = (y_pred == 1) & keep, (y_true == 1) & keep
        s = torch.stack([(pos_p & pos_t).sum(0), (pos_p & ~pos_t).sum(0), (~pos_p & pos_t & keep).sum(0),
                         ((y_pred == y_true) & keep).sum(0), keep.sum(0)]).to(torch.float64)
        self._stats = s if self._stats is None else self._stats + s

    def update_from_logits(self, logits: torch.Tensor, y_true: torch.Tensor):
        self.update(torch.round(torch.sigmoid(logits[:, :self.num_labels])), y_true)

    def get(self) -> Tuple[float, float]:
        """(accuracy over all labelled entries, mean of the per-AU binary F1 scores) - accf1.py:60-77"""
        if self._stats is None:
            raise RuntimeError("no samples")
        tp, fp, fn, correct, labelled = self._stats.cpu()
        denom = 2 * tp + fp + fn
        f1 = torch.where(denom > 0, 2 * tp / denom.clamp(min=1), torch.zeros_like(denom))
        return float(correct.sum() / labelled.sum()), float(f1.mean())

    def score(self) -> float:
        acc, f1 = self.get()
        return 0.5 * f1 + 0.5 * acc  # train.py:163
