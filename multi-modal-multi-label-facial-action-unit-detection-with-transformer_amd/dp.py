"""Per-clip data parallelism: one process per GPU, replicated weights, batch sharded across ranks,
one SUM all-reduce of the parameter gradients per step (RCCL over xGMI on the GPUs; gloo in CPU tests).

The reference has no distributed code at all (single process); the hot path shards
naturally because every op is per-clip (LayerNorm, attention within a clip, per-token MLP).

Overlap: each ``Transformer`` calls a hook right after a layer's backward has been enqueued (reverse layer
order) with that layer's flat fp32 gradient bucket (11 tensors, 2.1 M floats at d=512).  The buckets of a stack
are consecutive slices of one allocation; the wrapper collects adjacent ones up to ``bucket_bytes`` (default 16 MiB: two layers,
16.8 MB, at d=512) and launches ONE asynchronous all-reduce for the merged range: the process group runs it on its own
communication stream, ordered after the producing kernels, so it runs under the backward of the earlier layers - half the collectives
of a per-layer scheme (each costs the host ~0.1 ms and the links a latency-bound ring).  ``finish()`` reduces the
few parameters outside the transformer stacks in one extra bucket and makes the compute stream wait for
everything.  Gradients are averaged (sum / world) so that the update equals a single-process step on the global
batch.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class _GlobalMeanFn(torch.autograd.Function):
    """loss = (sum_r s_r) / (sum_r k_r) from each rank's (s_r, k_r).  The wrapper AVERAGES parameter gradients over the
    ranks, so each rank back-propagates world / K times its own numerator: the average is then d loss / d theta."""

    @staticmethod
    def forward(ctx, s, k, world, group):
        # s and k as AULoss hands them over are the two elements of ONE float32 2-vector (ops.au_loss_sum): reduce that
        # vector where it lies instead of stacking a copy (each tiny kernel here sits between forward and backward)
        pack = None
        if (s.dtype == torch.float32 and k.dtype == torch.float32 and s.dim() == 0 and k.dim() == 0
                and s.untyped_storage().data_ptr() == k.untyped_storage().data_ptr()
                and k.storage_offset() == s.storage_offset() + 1):
            pack = torch.empty(0, dtype=torch.float32, device=s.device).set_(s.untyped_storage(), s.storage_offset(), (2,))
            # ... which CONSUMES the inputs: after the call s and k hold the GLOBAL sum and count.  The alias is invisible to
            # autograd, so their version counters are bumped by hand - a backward that saved the local values, or any other
            # version-checked use, then fails loudly instead of reading the global ones
            torch.autograd.graph.increment_version((s, k))
        if pack is None:
            pack = torch.stack([s.detach().to(torch.float32), k.detach().to(torch.float32)])
        dist.all_reduce(pack, op=dist.ReduceOp.SUM, group=group)
        inv = torch.reciprocal(pack[1])  # a tensor: no host synchronisation (K = 0 on every rank -> NaN, as the reference)
        ctx.scale = inv * float(world)
        return pack[0] * inv

    @staticmethod
    def backward(ctx, g):
        return g * ctx.scale, None, None, None


class DataParallel:
    def __init__(self, model: torch.nn.Module, process_group=None, broadcast_parameters: bool = True,
                 bucket_layers: Optional[int] = None, bucket_bytes: int = 16 << 20):
        if not dist.is_initialized():
            raise RuntimeError("DataParallel needs torch.distributed to be initialised (one process per GPU)")
        self.model = model
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        # A collective is launched once the held layers amount to ``bucket_bytes`` (default 16 MiB: two layers of 8.4 MB at
        # d = 512, one of 18.9 MB at d = 768, a whole 12-token head stack at once): a ring all-reduce over the 7 x 153 GB/s
        # xGMI links moves 2 (n - 1) / n of the bucket per GPU, ~0.1 ms for 16 MiB at 8 GPUs - several times the ~30-50 us
        # latency floor of a collective, yet small enough that the LAST bucket (the bottom layers, whose reduction nothing
        # overlaps) stays a fraction of a step.  ``bucket_layers`` (a layer count) overrides the byte rule.
        self.bucket_layers = None if bucket_layers is None else max(1, int(bucket_layers))
        self.bucket_bytes = max(1, int(bucket_bytes))
        self._stacks = [m for m in model.modules() if hasattr(m, "set_grad_hook") and hasattr(m, "flat_parameters")]
        self._owned = set()
        for st in self._stacks:
            st.set_grad_hook(self._on_layer_grads)
            self._owned.update(id(p) for p in st.flat_parameters())
        self._rest_params = [p for p in model.parameters() if id(p) not in self._owned]
        self._cuda = any(p.is_cuda for p in model.parameters())
        # ncclAvg exists in RCCL only: a gloo group over GPU tensors (tests: two ranks on ONE GPU) pre-divides and sums
        self._avg = self._cuda and str(dist.get_backend(process_group)).lower() == "nccl"
        self.stats_reset()
        self._pending: List = []
        self._held: List = []  # (layer, flat) handed over but not launched yet
        # losses that are ratios over the kept rows (AULoss, loss.py:85-102) reduce numerator and denominator over the ranks
        for m in model.modules():
            if hasattr(m, "global_mean"):
                m.global_mean = self.global_mean
        if broadcast_parameters:
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t.data, src=0, group=process_group)
            # the broadcast wrote through ``.data`` (no version bump): stacks that cached bf16 / MX-FP8 weight images
            # (frozen pretrained branches, cache_weights) must rebuild them
            for st in self._stacks:
                if hasattr(st, "refresh_weights"):
                    st.refresh_weights()
        # rank-distinct dropout streams: ranks that seed torch identically would otherwise draw identical masks
        rank = dist.get_rank(process_group)
        for st in self._stacks:
            if hasattr(st, "set_seed_rank"):
                st.set_seed_rank(rank)

    # -- called from Transformer backward, once per layer ------------------------------------------
    def _on_layer_grads(self, layer: int, flat: torch.Tensor):
        self._held.append((layer, flat))
        full = (len(self._held) >= self.bucket_layers) if self.bucket_layers is not None else \
            (sum(f.numel() * f.element_size() for _, f in self._held) >= self.bucket_bytes)
        if full or layer == 0:
            works = self._flush()
            return lambda: [w.wait() for w in works]
        return self._flush_and_wait  # accumulation path: launch what is held now, make the current stream wait

    def _flush_and_wait(self):
        for w in self._flush():
            w.wait()

    @staticmethod
    def _merge(flats: List[torch.Tensor]) -> List[torch.Tensor]:
        """adjacent slices of one allocation -> one tensor over the whole range"""
        out: List[torch.Tensor] = []
        for f in flats:
            if out:
                g = out[-1]
                same = (g.untyped_storage().data_ptr() == f.untyped_storage().data_ptr() and g.dtype == f.dtype
                        and g.is_contiguous() and f.is_contiguous()
                        and g.storage_offset() + g.numel() == f.storage_offset())
                if same:
                    out[-1] = torch.empty(0, dtype=g.dtype, device=g.device).set_(
                        g.untyped_storage(), g.storage_offset(), (g.numel() + f.numel(),))
                    continue
            out.append(f)
        return out

    def _flush(self):
        if not self._held:
            return []
        flats = [f for _, f in sorted(self._held, key=lambda t: t[0])]
        self._held = []
        return [self._launch(t) for t in self._merge(flats)]

    def stats_reset(self):
        # bucket_bytes: the sizes of the gradient collectives of the LAST COMPLETED step (finish() publishes the list it
        # gathered during the step and starts a new one) - bounded however long an eager run lasts; the rest are counters
        self._stats = {"collectives": 0, "loss_collectives": 0, "bytes_total": 0, "bucket_bytes": []}
        self._step_bytes = []

    def stats(self):
        """``collectives`` / ``loss_collectives`` / ``bytes_total``: counters since stats_reset().  ``bucket_bytes``: the sizes of
        the gradient collectives of the LAST step that went through ``finish()`` - empty until one has, and unchanged by steps
        that bypass the Python ``finish()`` (a replayed hipGraph issues the same collectives without touching this object).
        A caller that sizes something from it must have run one eager step since stats_reset() (bench.py::dp_report asserts)."""
        return {k: (list(v) if isinstance(v, list) else v) for k, v in self._stats.items()}

    def _launch(self, flat: torch.Tensor):
        self._stats["collectives"] += 1
        nb = flat.numel() * flat.element_size()
        self._stats["bytes_total"] += nb
        self._step_bytes.append(nb)
        if self._avg:
            # Issued from the compute stream with async_op=True: the process group makes ITS communication stream wait for
            # what the compute stream has enqueued so far (the producing backward kernels), runs the collective there and
            # hands back a Work whose wait() makes the compute stream wait for it - the overlap with the remaining backward
            # comes from that stream, so no second stream / event / record_stream of our own (0.15 ms of host time per step).
            # RCCL averages inside the collective (ncclAvg): no separate scaling pass over the bucket.
            work = dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        else:
            flat.div_(self.world)
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append((work, flat))
        return work

    def global_mean(self, local_sum: torch.Tensor, local_count: torch.Tensor) -> torch.Tensor:
        """mean over the GLOBAL batch of a per-row quantity from each rank's (sum over its kept rows, number of kept rows):
        equals the single-process loss on the concatenated batch for any split of the ignored rows over the ranks.
        CONSUMES its inputs when they are the two elements of one fp32 2-vector (what AULoss passes): they hold the global
        sum / count afterwards (their autograd version counters are bumped)"""
        # on the GPU the "blocking" all-reduce only makes the current stream wait for the group's communication stream (no host
        # block), so it is recorded by a hipGraph capture like the gradient collectives (graphs.GraphedTrainStep(dp=...))
        self._stats["loss_collectives"] += 1
        return _GlobalMeanFn.apply(local_sum, local_count, self.world, self.group)

    # -- called once per step, after loss.backward() and before optimizer.step() -------------------
    def finish(self):
        self._flush()
        rest = [p for p in self._rest_params if p.grad is not None]
        bucket: Optional[torch.Tensor] = None
        if rest:
            bucket = torch.cat([p.grad.reshape(-1) for p in rest])
            self._launch(bucket)
        for work, _ in self._pending:
            work.wait()  # CUDA: the current stream waits (no host block); gloo: blocks until done
        if bucket is not None:
            # the reduced gradients stay where the collective left them: every .grad becomes a view of the bucket (no copy
            # back; the optimizers read .grad through its data pointer)
            off = 0
            for p in rest:
                n = p.numel()
                p.grad = bucket[off:off + n].view_as(p)
                off += n
        self._pending.clear()
        self._stats["bucket_bytes"], self._step_bytes = self._step_bytes, []

    def __call__(self, *a, **k):
        return self.model(*a, **k)
