#!/usr/bin/env python3
"""bench.py - clips/sec (fwd+bwd) of the AV-former transformer hot path on synthetic (B,T,d) AV sequences.

    python bench.py --gpus N --steps K --warmup W

N>1: either under `python -m torch.distributed.run --nproc-per-node N ...` (RANK / WORLD_SIZE in the environment), or plain
`python bench.py --gpus N`: with no WORLD_SIZE in the environment the script becomes the launcher - it starts N rank
processes itself, before it has imported torch or touched a GPU - and rank 0 prints the line.

One "step" = one training pass of the hot path over one resident synthetic batch on every rank:
zero_grad -> SyntheticAVFormer forward (pos-emb + Transformer stack + AU logits) -> AULoss -> backward (hand-written
HIP kernels) -> gradient all-reduce (RCCL, overlapped with backward; N>1 only) -> Adam step (as reference train.py:206-237).
Workload at N=1: BASELINE.json configs[1] ("C2"): d=512, 6 layers, 8 heads x 64, mlp 1024, T_v=196 + T_a=128 = 324
tokens, B=32 per GPU, bf16 MFMA compute with fp32 accumulate/residual.  Weak scaling: B=32 per GPU at every N.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel class, timed
live with HIP events on the launch stream during the timed steps) and `cpu_baseline` (the CPU oracle - a port of the
reference's math - timed on this host, rank 0, N=1 only, bounded sample).  Beside the contract's fields the line carries
`north_star_shape` (a second timed region, same step, at B=32 T=512 d=512 - the shape the >=30 % MFMA target is quoted
on - with per-kernel-class TFLOP/s and fraction of the bf16 MFMA peak) and `f32_parity_clips_per_s` (the fp32 parity mode
on the main workload, N=1).  `roofline.traffic` comes from the committed PMC profile only while that profile was
collected from the kernel sources in this tree (hash stamp), else null.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # name: dim, depth, heads, dim_head, mlp, T_v, T_a, per-GPU batch
    "c1": dict(dim=128, depth=2, heads=8, dim_head=32, mlp_dim=256, t_video=32, t_audio=32, batch=4),
    "c2": dict(dim=512, depth=6, heads=8, dim_head=64, mlp_dim=1024, t_video=196, t_audio=128, batch=32),
    "c3": dict(dim=512, depth=6, heads=8, dim_head=64, mlp_dim=1024, t_video=384, t_audio=128, batch=32),
    "c4": dict(dim=768, depth=12, heads=12, dim_head=64, mlp_dim=1536, t_video=768, t_audio=256, batch=16),
    "c5": dict(dim=512, depth=6, heads=8, dim_head=64, mlp_dim=1024, t_video=384, t_audio=128, batch=64),  # with --dtype mx8
}
PKG_DIR = "multi-modal-multi-label-facial-action-unit-detection-with-transformer_amd"
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3, "mx8": 2500.0, "mx8-fwd": 2500.0}  # dense, /opt/skills/guides/MI355X_MICROARCH.md
MX8_PEAK_TFLOPS = 5000.0  # the MX-scaled fp8 MFMA (kernel class gemm_mx8_nt only; attention, dW and dqkv -> dh1 stay bf16)
HBM_PEAK_GBS = 8000.0


def stack_flops_fwd(c, B):
    D, L, H, dh, M = c["dim"], c["depth"], c["heads"], c["dim_head"], c["mlp_dim"]
    N, I = c["t_video"] + c["t_audio"], c["heads"] * c["dim_head"]
    return L * (2.0 * B * N * (3 * D * I + I * D + 2 * D * M) + 4.0 * B * N * N * I)


def cpu_baseline(c, steps, threads):
    """The oracle (a port of the reference's op sequence) fwd+bwd on the host CPU, same shapes."""
    import torch
    import oracle
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(123)
    D, L, H, dh, M = c["dim"], c["depth"], c["heads"], c["dim_head"], c["mlp_dim"]
    B, N = c["batch"], c["t_video"] + c["t_audio"]
    sd = oracle.init_transformer_state(D, L, H, dh, M, generator=g)
    sd = {k: v.requires_grad_(True) for k, v in sd.items()}
    pos = (torch.randn(1, N, D, generator=g) * 0.02).requires_grad_(True)
    w = (torch.randn(12, D, generator=g) * 0.04).requires_grad_(True)
    b = torch.zeros(12, requires_grad=True)
    x = torch.randn(B, N, D, generator=g)
    labels = (torch.rand(B, 12, generator=g) > 0.5).float()

    def one():
        for t in list(sd.values()) + [pos, w, b]:
            t.grad = None
        y = oracle.transformer_forward(x + pos, sd, L, H)
        loss = oracle.au_loss(y.mean(1) @ w.t() + b, labels)
        loss.backward()

    one()  # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    dt = (time.perf_counter() - t0) / steps
    return B / dt, dt


def kernel_source_hash():
    """sha256 over the kernel sources the built library comes from: a PMC traffic file is only quoted while it was
    collected from these very kernels"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, PKG_DIR, "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_class, workload):
    """HBM bytes per launch of a kernel class from the newest committed PMC profile of `workload` (profiles/rNN_traffic.json,
    produced by tools/pmc_traffic.py from separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes with the gfx950
    correction).  The file is stamped with the hash of the kernel sources it was collected from; when that is not the hash
    of the sources in this tree the number is STALE and None is returned (bench.py cannot read PMCs itself).
    -> (bytes or None, note)"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True)
    cur = kernel_source_hash()
    for path in files:
        try:
            j = json.load(open(path))
        except Exception:
            continue
        if j.get("workload", "c2") != workload:
            continue
        v = j.get("per_class", {}).get(kernel_class)
        if j.get("kernel_sources_sha") != cur:
            return None, f"{os.path.basename(path)} was collected from other kernel sources ({j.get('kernel_sources_sha')} != {cur})"
        return (None if v is None else round(float(v))), f"{os.path.basename(path)} @ {j.get('commit', '?')}"
    return None, "no PMC profile committed for this workload"


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N child processes (one rank per GPU) BEFORE this process has
    touched the GPU or imported torch, hand them the rendezvous through the environment, and exit with their status.
    Rank 0 prints the JSON line on the stdout it inherits."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                r = p.poll()
                if r is None:
                    continue
                procs.remove(p)
                if r != 0:
                    rc = rc or r
                    for q in procs:  # one rank failed: the others would wait in a collective for ever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for q in procs:
            q.kill()
    return rc


SETTLE_S = float(os.environ.get("AVF_BENCH_SETTLE_S", "0.3"))  # untimed busy time in front of every timed region


class Region:
    """one configuration on this rank: model, resident synthetic batch, optimizer, the step closure"""

    def __init__(self, A, torch, dist, name, dtype, args, dev, rank, world, use_dist):
        self.A, self.torch, self.dist, self.use_dist, self.world, self.dtype = A, torch, dist, use_dist, world, dtype
        self.rank = rank
        c = dict(CONFIGS[name])
        if args.batch and name == args.config:
            c["batch"] = args.batch
        self.c, self.B = c, c["batch"]
        B, Tv, Ta = c["batch"], c["t_video"], c["t_audio"]
        torch.manual_seed(123)  # identical weights on every rank (reference default seed, opts.py:19)
        self.residual = args.residual if dtype != "f32" else "f32"
        model = A.SyntheticAVFormer(c["dim"], c["depth"], c["heads"], c["dim_head"], c["mlp_dim"], Tv, Ta, task="AU",
                                    compute_dtype=dtype, residual_dtype=self.residual, dropout=args.dropout).to(dev)
        g = torch.Generator().manual_seed(123 + rank)  # rank-distinct synthetic clips
        clip = torch.randn(B, Tv, c["dim"], generator=g).to(dev)
        audio = torch.randn(B, Ta, c["dim"], generator=g).to(dev)
        labels = (torch.rand(B, 12, generator=g) > 0.5).float()
        labels[::16] = -1  # 1/16 of the clips carry the ignore label
        labels = labels.to(dev)
        batch = {"clip": clip, "audio_features": audio}
        opt = None
        if not args.no_optimizer:
            # Adam(lr, weight_decay) as the reference's loop (train.py:318-322)
            if args.torch_adam:
                try:
                    opt = torch.optim.Adam(model.parameters(), lr=5e-4, weight_decay=5e-5, fused=True)
                except Exception:
                    opt = torch.optim.Adam(model.parameters(), lr=5e-4, weight_decay=5e-5)
            else:
                opt = A.optim.FusedAdam(model, lr=5e-4, weight_decay=5e-5)
        dp = A.dp.DataParallel(model) if use_dist else None
        self.model, self.opt, self.dp, self.dev = model, opt, dp, dev
        self.optimizer = None if opt is None else type(opt).__name__

        def step():
            # optimizer.zero_grad() as the reference (train.py:206); Module.zero_grad walks the whole module tree (0.7 ms)
            (opt if opt is not None else model).zero_grad(set_to_none=True)
            out = model(batch)
            loss = model.get_au_loss(out, labels)
            loss.backward()
            if dp is not None:
                dp.finish()
            if opt is not None:
                opt.step()
            return loss

        self.step = step

        def fwd_bwd():
            model.zero_grad(set_to_none=True)  # (drops the .grad views; the flat gradient buckets are overwritten, not cleared)
            loss = model.get_au_loss(model(batch), labels)
            loss.backward()
            if dp is not None:
                dp.finish()
            return loss

        self.fwd_bwd = fwd_bwd if opt is not None else None

    def agree(self, flag: bool) -> bool:
        """`flag` on one rank; under data parallelism: true on EVERY rank if it is true on any (one tiny all-reduce)"""
        if not self.use_dist:
            return bool(flag)
        t = self.torch.tensor([1.0 if flag else 0.0], device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return t.item() > 0.5

    def fence(self):
        self.torch.cuda.synchronize()
        if self.use_dist:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def timed(self, steps, warmup, graph=False, eager_too=True):
        """W untimed steps, then exactly K steps between two fences, no instrumentation; MAX over ranks"""
        torch = self.torch
        for _ in range(warmup):
            self.step()
        # the caching allocator may still be growing after W steps of a large configuration (C4: 59 device allocations over
        # the first ~10 steps, each a multi-millisecond hipMalloc that would land inside the timed region): keep stepping,
        # untimed, until two consecutive steps allocate nothing new (at most 12 more)
        # Every step issues collectives under data parallelism, so the decision to take another one must be the SAME on every
        # rank (a rank that left such a loop on its own clock / its own allocator counter would meet its peers' all-reduces
        # with a barrier): the local verdict is MAX-reduced over the ranks before it is acted on (self.agree).
        quiet, extra = 0, 0
        while quiet < 2 and extra < 12:
            before = torch.cuda.memory_stats(self.dev).get("num_device_alloc", 0)
            self.step()
            extra += 1
            grew = torch.cuda.memory_stats(self.dev).get("num_device_alloc", 0) != before
            quiet = 0 if self.agree(grew) else quiet + 1
        # ... and until the device has been busy for SETTLE_S seconds, so that the clock / power state the timed steps see is
        # the steady one of this workload and not whatever the few warm-up steps left (a 20-step region is 45 ms long)
        t_settle = time.perf_counter()
        while self.agree(time.perf_counter() - t_settle < SETTLE_S):
            for _ in range(5):
                self.step()
                extra += 1
            torch.cuda.synchronize()
        self.extra_warmup = extra
        self.fence()
        run = self.step
        self.launch = "eager"
        self.eager_ms = None
        # graph == "auto_dp" (the multi-rank default, --launch auto): FIRST the K eager steps, timed like any region - a valid
        # measurement whatever happens next; THEN the captured step is attempted under a watchdog.  It becomes `value` only if
        # every rank captured, the first replay's loss is finite, the same on every rank and within 10 % of the last eager
        # loss; otherwise the eager measurement stands.  A replay that never returns (a multi-rank capture has never run on
        # this build's hardware) is ended by the watchdog: rank 0 prints the eager line, every rank exits - nothing re-execs.
        auto_dp = graph == "auto_dp"
        crash_armed = False
        eager_dt = None
        watchdog = None
        if auto_dp:
            t0 = time.perf_counter()
            for _ in range(steps):
                loss = self.step()
            self.fence()
            te = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=self.dev)
            if self.use_dist:
                self.dist.all_reduce(te, op=self.dist.ReduceOp.MAX)
            eager_dt = te.item()
            self.eager_ms = eager_dt / steps * 1e3
            eager_loss = float(loss.item())
            del loss  # a live loss keeps the last eager step's autograd graph (its AccumulateGrad nodes, bound to the stream they
            #           were created on) alive into the capture: torch warns "may break CUDA graph capture" - here it segfaults
            #           in capture_end
            watchdog = self._arm_watchdog(float(os.environ.get("AVF_BENCH_GRAPH_WATCHDOG_S", "90")), steps)
            # ... and a crash INSIDE the runtime during the attempt (capture_end / instantiate of a multi-rank graph has never run
            # on this build's hardware; a stale autograd graph made it segfault on one rank during development) cannot be caught
            # as an exception: while the attempt runs, a fatal signal writes the eager line (rank 0) and leaves with _exit(0)
            crash_armed = False
            try:
                line = b""
                if self.rank == 0 and getattr(self, "hang_line", None) is not None:
                    d = self.hang_line(self.eager_ms, steps)
                    d["launch"] = "eager (the captured data-parallel step crashed inside the runtime; this is the eager region timed before the attempt)"
                    line = (json.dumps(d) + "\n").encode()
                self.A._lib.check(self.A._lib.load().avf_crash_line_arm(line, _REAL_STDOUT if line else -1), "crash_line_arm")
                crash_armed = True
            except Exception:
                pass
            graph, eager_too = True, False
        if graph and eager_too:
            # the same K steps launched from Python, timed the same way: reported beside the graph-replay number so that the
            # N = 1 line can be compared like for like with lines that cannot replay a graph
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step()
            self.fence()
            self.eager_ms = (time.perf_counter() - t0) / steps * 1e3
        if graph:
            # the whole step (zero_grad, forward, loss, backward, [bucketed RCCL all-reduces, finish()], optimizer) captured
            # once and replayed: the ~130 launches of a step no longer depend on the host keeping up (eager C2 needs ~1.8 ms
            # of host time per 2.3 ms step; under data parallelism 1.7-2.65 ms).  The collectives are stream operations of
            # torch's process group and are recorded like kernels.
            err = None
            side = torch.cuda.Stream()
            cap = torch.cuda.Stream()          # the capture stream (ours, so that a failed capture can be closed by hand)
            home = torch.cuda.current_stream()
            backend = str(self.dist.get_backend()).lower() if self.use_dist else "none"
            try:
                if self.use_dist and backend != "nccl":
                    # only RCCL's collectives are stream operations that a hipGraph can record (gloo synchronises the host)
                    raise RuntimeError(f"the {backend} backend's collectives cannot be recorded into a hipGraph")
                if self.opt is not None:
                    for gdict in self.opt.param_groups:
                        gdict["capturable"] = True
                    self.step()
                gr = torch.cuda.CUDAGraph()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    self.step()
                torch.cuda.current_stream().wait_stream(side)
                with torch.cuda.graph(gr, stream=cap):
                    if os.environ.get("AVF_BENCH_FAIL_CAPTURE") == "1":  # test aid: an operation no capture can record
                        torch.cuda.synchronize()
                    if os.environ.get("AVF_BENCH_FAIL_CAPTURE") == "segv":  # test aid: the process dies inside the attempt
                        import signal
                        os.kill(os.getpid(), signal.SIGSEGV)
                    static_loss = self.step()
            except Exception as e:  # capture is an optimisation of the launch path, never a requirement
                err = f"{type(e).__name__}: {str(e)[:120]}"
                # A failed capture can leave three things behind: torch's current stream still pointing at the capture stream
                # (torch.cuda.graph.__exit__ raises before it restores it), the capture itself still open in the invalidated
                # state, and the runtime's sticky last-error - each of which would fail the next, unrelated HIP call.
                import ctypes as _C
                torch.cuda.set_stream(home)
                for st in (cap, side, home):
                    try:
                        self.A._lib.load().avf_hip_error_reset(_C.c_void_p(st.cuda_stream))
                    except Exception:
                        pass
                try:
                    torch.cuda.synchronize()
                except Exception:
                    self.A._lib.load().avf_hip_error_reset(_C.c_void_p(home.cuda_stream))
            # Recording executes nothing, so the ranks are still in step here.  They must also AGREE on the launch mode before
            # the first replay: a rank that failed to capture would issue eager collectives against its peers' replayed ones.
            ok = torch.tensor([0.0 if err else 1.0], device=self.dev)
            if self.use_dist:
                self.dist.all_reduce(ok, op=self.dist.ReduceOp.MIN)
            why = None if ok.item() > 0.5 else f"graph capture failed: {err or 'on another rank'}"
            if why is None and auto_dp:
                # validate the first replay before trusting it with `value`
                gr.replay()
                self.fence()
                lv = static_loss.detach().float().reshape(1).clone()
                lo, hi = lv.clone(), lv.clone()
                if self.use_dist:
                    self.dist.all_reduce(lo, op=self.dist.ReduceOp.MIN)
                    self.dist.all_reduce(hi, op=self.dist.ReduceOp.MAX)
                l0, l1 = float(lo.item()), float(hi.item())
                good = (l0 == l0 and l1 == l1 and abs(l1 - l0) <= 1e-6 * max(1.0, abs(l1))
                        and abs(l1 - eager_loss) <= 0.1 * max(abs(eager_loss), 1e-3))
                if not good:
                    why = f"first replayed loss {l0:.6g}..{l1:.6g} over the ranks against the eager loss {eager_loss:.6g}"
            if why is None:
                run = lambda: (gr.replay(), static_loss)[1]
                for _ in range(3):
                    run()
                self.fence()
                self.launch = "hipGraph replay" + (" (RCCL all-reduces captured)" if self.dp is not None else "")
            else:
                run = self.step
                self.launch = f"eager ({why})"
                for _ in range(2):
                    self.step()
                self.fence()
        if auto_dp and crash_armed and run is self.step:
            self.A._lib.load().avf_crash_line_disarm()
        if auto_dp and run is self.step:
            # the attempt fell back to eager launches: the eager region timed above IS the measurement
            if watchdog is not None:
                watchdog.cancel()
            self.steps, self.ms = steps, eager_dt / steps * 1e3
            self.clips_per_s = self.B * self.world * steps / eager_dt
            self.loss = eager_loss
            self._run = run
            self._finish_timed()
            self.fwd_bwd_ms = None
            return
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = run()
        self.fence()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device=self.dev)
        if self.use_dist:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        dt = t.item()
        if watchdog is not None:
            watchdog.cancel()
        if auto_dp and crash_armed:
            self.A._lib.load().avf_crash_line_disarm()
        # forward + loss + backward (+ the gradient all-reduces) alone - what the metric names; the step above also clears the
        # gradients and applies Adam.  Eager launches, same fences (the step is GPU-bound either way: `eager_ms_per_step`).
        self.fwd_bwd_ms = None
        if getattr(self, "fwd_bwd", None) is not None:
            # no optimizer step between these forwards: the stacks keep their bf16 weight images (cache_weights) instead of
            # re-deriving them per forward as they must when nothing vouches for the masters being unchanged
            stacks = [m for m in self.model.modules() if hasattr(m, "cache_weights")]
            keep = [m.cache_weights for m in stacks]
            for m in stacks:
                m.cache_weights = True
            for _ in range(3):
                self.fwd_bwd()
            self.fence()
            t1 = time.perf_counter()
            for _ in range(steps):
                self.fwd_bwd()
            self.fence()
            tfb = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=self.dev)
            if self.use_dist:
                self.dist.all_reduce(tfb, op=self.dist.ReduceOp.MAX)
            self.fwd_bwd_ms = tfb.item() / steps * 1e3
            for m, k in zip(stacks, keep):
                m.cache_weights = k
                m.refresh_weights()
        self._run = run
        self.steps = steps
        self.ms = dt / steps * 1e3
        self.clips_per_s = self.B * self.world * steps / dt
        self.loss = float(loss.item())
        self._finish_timed()

    def _finish_timed(self):
        self.tflops = 3.0 * stack_flops_fwd(self.c, self.B) / (self.ms * 1e-3) / 1e12
        self.peak = MFMA_PEAK_TFLOPS[self.dtype]
        if self.dtype == "f32" and self.A._lib.get_f32_arithmetic() == "bf16x3":
            self.peak = MFMA_PEAK_TFLOPS["bf16"] / 3.0  # three bf16 MFMA products per fp32 product
        self.frac = self.tflops / self.peak

    def _arm_watchdog(self, seconds, steps):
        """A timer thread for the multi-rank graph attempt: if capture / validation / the replayed region has not finished after
        `seconds`, the replay is taken for hung (a collective waiting for a peer that will never issue it).  The eager
        measurement taken before the attempt is then the result: rank 0 prints the contract's line from it (self.hang_line,
        set by main()), every rank leaves with os._exit - no re-exec, no second attempt."""
        import threading
        if seconds <= 0:   # AVF_BENCH_GRAPH_WATCHDOG_S=0: no watchdog
            return None

        def fire():
            try:
                if self.rank == 0 and getattr(self, "hang_line", None) is not None:
                    line = self.hang_line(self.eager_ms, steps)
                    os.write(_REAL_STDOUT, (json.dumps(line) + "\n").encode())
                sys.stderr.write(f"bench.py rank {self.rank}: graph replay did not finish within {seconds:.0f} s - "
                                 f"reported the eager measurement, leaving\n")
                sys.stderr.flush()
            finally:
                os._exit(0 if getattr(self, "hang_line", None) is not None or self.rank != 0 else 3)

        t = threading.Timer(seconds, fire)
        t.daemon = True
        t.start()
        return t

    def clock_power(self, seconds=1.5):
        """Shader clock and package power while the timed step keeps running (rocm-smi sampled from a thread, AFTER the timed
        region; one rank, one GPU).  The step runs against the package power limit: `frac` is against
        the NOMINAL roof at 2.4 GHz, this says what the box allowed.  None when rocm-smi is missing or prints something else."""
        import re
        import shutil
        import statistics
        import subprocess
        import threading
        if self.use_dist or shutil.which("rocm-smi") is None:
            return None
        stop, out = threading.Event(), []

        def sample():
            while not stop.is_set():
                try:
                    txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
                    m = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", txt)
                    w = re.search(r"Package Power \(W\): ([0-9.]+)", txt)
                    if m and w:
                        out.append((int(m.group(1)), float(w.group(1))))
                except Exception:
                    return
                time.sleep(0.15)

        th = threading.Thread(target=sample, daemon=True)
        t0 = time.perf_counter()
        th.start()
        while time.perf_counter() - t0 < seconds:
            for _ in range(20):
                self._run()
            self.torch.cuda.synchronize()
        stop.set()
        th.join(timeout=10)
        if not out:
            return None
        return {"sclk_mhz": statistics.median(c for c, _ in out), "package_w": statistics.median(w for _, w in out),
                "samples": len(out), "nominal_sclk_mhz": 2400, "how": "rocm-smi sampled while the timed step keeps replaying, after the timed region"}

    def instrumented(self, steps):
        """K more steps with a HIP-event pair ATTACHED to every hot-path dispatch (hipExtLaunchKernelGGL start/stop events =
        the kernel's own begin / end timestamps, on the launch stream): per-kernel-class durations.  Run after ALL timed
        regions: the ~220 event records per step cost 15-20 % of a step, and once a stream has carried such dispatches the
        launches after them stay slow (a timed region placed behind an instrumented pass measured 3x its real time)."""
        A = self.A
        A._lib.timing_enable(True)
        t1 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.fence()
        self.ms_events = (time.perf_counter() - t1) / steps * 1e3
        tm = A._lib.timing_read()
        A._lib.timing_enable(False)
        peak = self.peak
        self.timing = tm
        self.classes = {
            k: {"ms_per_step": round(v["ms"] / steps, 4), "launches_per_step": v["launches"] / steps,
                "TFLOPs": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["ms"] > 0 and v["flops"] > 0 else None,
                "frac_of_mfma_peak": (round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / (MX8_PEAK_TFLOPS if k == "gemm_mx8_nt" else peak), 4)
                                      if v["ms"] > 0 and v["flops"] > 0 else None),
                "GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else None}
            for k, v in tm.items() if v["launches"] > 0}


def dp_report(torch, dist, region, dev, world):
    """What the data-parallel step did on the wire, for the record of a multi-GPU run: the ranks every rank saw, the
    collectives of one step and their sizes, and the time of the same all-reduces issued ALONE (back to back, nothing to
    overlap with) - an upper bound of the communication time per step; the overlap is the difference to `ms_per_step`."""
    ones = torch.ones(1, device=dev)
    dist.all_reduce(ones)
    seen = int(round(ones.item()))
    mism = torch.tensor([0.0 if seen == world else 1.0], device=dev)
    dist.all_reduce(mism, op=dist.ReduceOp.MAX)
    if mism.item() > 0.5:
        raise SystemExit(f"bench.py: a rank saw {seen} ranks in the process group, expected {world}")
    dp = region.dp
    dp.stats_reset()
    region.step()  # ONE EAGER step: bucket_bytes is published by the wrapper's Python finish(), which a graph replay bypasses
    torch.cuda.synchronize()
    st = dp.stats()
    sizes = st["bucket_bytes"]
    if not (len(sizes) == st["collectives"] > 0):
        raise SystemExit(f"bench.py: the data-parallel step published {len(sizes)} bucket sizes for {st['collectives']} "
                         f"collectives - the step did not go through DataParallel.finish()")
    bufs = [torch.zeros(max(1, b // 4), device=dev) for b in sizes]
    for b in bufs:
        dist.all_reduce(b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        for b in bufs:
            dist.all_reduce(b)  # synchronous form: the calling stream waits for the group's stream, so the events bracket it
    e1.record()
    torch.cuda.synchronize()
    alone = torch.tensor([e0.elapsed_time(e1) / reps], device=dev)
    dist.all_reduce(alone, op=dist.ReduceOp.MAX)
    return {"rccl_ranks": seen, "backend": dist.get_backend(), "gradient_collectives_per_step": st["collectives"],
            "loss_collectives_per_step": st["loss_collectives"], "bucket_bytes": sizes,
            "gradient_bytes_per_step": int(sum(sizes)), "bucket_rule_bytes": dp.bucket_bytes,
            "allreduce_alone_ms_per_step": round(alone.item(), 4),
            "note": "allreduce_alone = the step's gradient all-reduces issued back to back with nothing to overlap (max over "
                    "ranks); ms_per_step minus the one-rank ms_per_step is what the collectives cost beside the backward"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120)  # 0.27 s of device time at C2
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "mx8", "mx8-fwd"],
                    help="mx8: bf16 path with MX-FP8 operands on the forward qkv / out / mlp GEMMs and the backward dX GEMMs of "
                         "the mlp and out-projection (BASELINE config 5); mx8-fwd: forward operands only (A/B aid)")
    ap.add_argument("--residual", default="bf16", choices=["f32", "bf16"],
                    help="storage type of the forward residual stream in the bf16 / mx8 modes (Transformer(residual_dtype=...)); "
                         "bf16 since round 3 (statistics, accumulation and the add stay fp32; tolerance: tests/test_gpu_resid16.py), "
                         "the fp32-stream step is timed beside it (`residual_f32`)")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch override")
    ap.add_argument("--dropout", type=float, default=0.0,
                    help="dropout of the stack (BASELINE's configs: 0; the reference's real stacks train at 0.2, heads.py:277)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--no-optimizer", action="store_true", help="time fwd+bwd(+all-reduce) only")
    ap.add_argument("--torch-adam", action="store_true",
                    help="step torch.optim.Adam(fused=True) instead of the library's Adam (same arithmetic; the library's "
                         "also rewrites the bf16 weight copies in its pass)")
    ap.add_argument("--cpu-steps", type=int, default=8)
    ap.add_argument("--graph", action="store_true", help="same as --launch graph")
    ap.add_argument("--launch", default="auto", choices=["auto", "graph", "eager"],
                    help="how the timed steps are issued: graph = the step captured once into a HIP graph and replayed; eager = "
                         "launched from Python every step; auto = graph on one rank (falls back to eager if capture fails) and eager "
                         "on several ranks (graph there captures the RCCL all-reduces with the step: opt-in, see AVF_BENCH_DP_GRAPH)")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the extra timed regions (north-star shape C3, fp32 parity mode); the contract line's own "
                         "fields are unaffected")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher.  Nothing above imported torch or touched the GPU.
        sys.exit(spawn_ranks(args.gpus))
    os.dup2(2, 1)  # worker: everything but the result line (written to the saved descriptor) goes to stderr
    if os.environ.get("AVF_BENCH_DRYRUN") == "1":
        # launcher rehearsal without GPUs (tests/test_bench_cpu.py): report the rendezvous this rank was handed, touch nothing
        if int(os.environ.get("RANK", "0")) == 0:
            os.write(_REAL_STDOUT, (json.dumps({"dryrun": True, "world": int(os.environ.get("WORLD_SIZE", "1")),
                                                "master": os.environ.get("MASTER_ADDR"), "n_gpus": args.gpus}) + "\n").encode())
        return

    import torch
    import torch.distributed as dist
    import avformer_amd as A

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    # AVF_BENCH_ONE_DEVICE=1 + AVF_BENCH_BACKEND=gloo: rehearsal of the N > 1 control flow (launcher, rendezvous, collective
    # decisions of the untimed loops, data-parallel report) on a ONE-GPU box - every rank on cuda:0, the collectives through gloo
    # (RCCL refuses two ranks on one device).  Its numbers mean nothing; what it shows is that the multi-rank path runs to its line.
    if os.environ.get("AVF_BENCH_ONE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    A._lib.load()  # no fallback: fails here if the HIP library is missing
    if os.environ.get("AVF_F32_ARITH"):  # "bf16x3" (the library's default) / "f32": the parity mode's arithmetic for this run
        A._lib.set_f32_arithmetic(os.environ["AVF_F32_ARITH"])
    # AVF_BENCH_FORCE_DP=1: take the multi-GPU code path (process group, data-parallel wrapper, barriers) even with one
    # rank - the only way to rehearse it on a one-GPU box
    use_dist = world > 1 or os.environ.get("AVF_BENCH_FORCE_DP") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        backend = os.environ.get("AVF_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    events = not args.no_kernel_events
    mk = lambda name, dtype, dist_on: Region(A, torch, dist, name, dtype, args, dev, rank, world, dist_on)
    # ---- timed regions first, all un-instrumented: the contract's own, then (same process) the shape the north-star
    # target is quoted on (C3: B=32/GPU, T=512, d=512 = BASELINE configs[2], which at N=8 is exactly its global batch of
    # 256) and, at N=1, the fp32 parity mode on the main workload (the mode that meets north_star's logits rtol 1e-3).
    main_r = mk(args.config, args.dtype, use_dist)
    # Data parallelism: the captured step (RCCL all-reduces recorded with it) is tested bit-equal to eager on a 1-rank group
    # (tests/test_gpu_dp.py) and is what the one-rank rehearsal (AVF_BENCH_FORCE_DP=1) replays; with MORE than one rank the
    # timed steps stay eager unless AVF_BENCH_DP_GRAPH=1 (or --launch graph): no multi-GPU box was available to this build to
    # run a multi-rank replay even once, and the scaling line must not depend on an unexercised path.  AVF_BENCH_DP_GRAPH=0
    # keeps the one-rank rehearsal eager as well.
    dp_graph_env = os.environ.get("AVF_BENCH_DP_GRAPH", "")
    dp_graph_ok = (dp_graph_env != "0") if world == 1 else (dp_graph_env == "1" or args.launch == "graph")
    use_graph = ((args.graph or args.launch in ("auto", "graph")) and args.launch != "eager" and (not use_dist or dp_graph_ok))
    # --launch auto with more than one rank: eager steps first (the safe measurement), then the captured step under a watchdog
    # and only if its first replay validates (Region.timed, "auto_dp"); the other regions of a multi-rank run stay eager
    main_graph = use_graph
    # (AVF_BENCH_AUTO_DP=1: take that path with ONE rank too - with AVF_BENCH_FORCE_DP=1 the rehearsal of its success branch on a
    # one-GPU box: eager region, capture with the 1-rank RCCL all-reduces, validated first replay, replayed region)
    if (use_dist and (world > 1 or os.environ.get("AVF_BENCH_AUTO_DP") == "1") and args.launch == "auto" and not args.graph
            and dp_graph_env not in ("0", "1")):
        main_graph = "auto_dp"
        B0, c0 = main_r.B, main_r.c
        main_r.hang_line = lambda ms, k: {
            "metric": "clips/sec (fwd+bwd) on synthetic (B,T,d) AV sequences", "value": round(B0 * world / (ms * 1e-3), 2),
            "unit": "clips/s", "n_gpus": world, "steps": k, "warmup": args.warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[{'1' if args.config == 'c2' else args.config}]: avformer transformer "
                                   f"stack d={c0['dim']} L={c0['depth']}, B={B0}/GPU", "global_batch": B0 * world,
                       "seq_len": c0["t_video"] + c0["t_audio"], "parallelism": f"dp{world}"},
            "launch": "eager (the captured data-parallel step did not return: watchdog; this is the eager region timed before the attempt)",
            "roofline": None, "cpu_baseline": None}
    try:
        main_r.timed(args.steps, args.warmup, main_graph)
    finally:
        A._lib.load().avf_crash_line_disarm()  # (armed only around the multi-rank graph attempt; never past this region)
    main_clock = main_r.clock_power() if (world == 1 and not args.no_extra) else None
    c, B = main_r.c, main_r.B
    Tv, Ta = c["t_video"], c["t_audio"]
    result = {
        "metric": "clips/sec (fwd+bwd) on synthetic (B,T,d) AV sequences",
        "value": round(main_r.clips_per_s, 2), "unit": "clips/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(main_r.ms, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"BASELINE.json configs[{'1' if args.config == 'c2' else args.config}]: avformer "
                               f"transformer stack d={c['dim']} L={c['depth']} H={c['heads']}x{c['dim_head']} "
                               f"mlp={c['mlp_dim']}, T_v={Tv}+T_a={Ta} tokens, B={B}/GPU",
                   "global_batch": B * world, "seq_len": Tv + Ta, "parallelism": f"dp{world}",
                   "step": "zero_grad+fwd+AULoss+bwd" + ("+allreduce" if use_dist else "") + ("+adam" if main_r.optimizer else ""),
                   "optimizer": main_r.optimizer, "residual_stream": main_r.residual, "dropout": args.dropout, "loss": main_r.loss},
        # what the box allowed while this step ran: the step is package-power-limited
        "clock_power": main_clock,
        "untimed_steps_beyond_warmup": main_r.extra_warmup,  # until the caching allocator stopped growing (Region.timed)
        "launch": main_r.launch,
        # the same step launched eagerly from Python (None when `value` itself is the eager number)
        "eager_ms_per_step": None if main_r.eager_ms is None else round(main_r.eager_ms, 4),
        "eager_clips_per_s": None if main_r.eager_ms is None else round(B * world / (main_r.eager_ms * 1e-3), 2),
        # forward + AULoss + backward (+ all-reduce) only, eager launches: the part of the step the metric names
        "fwd_bwd_ms_per_step": None if main_r.fwd_bwd_ms is None else round(main_r.fwd_bwd_ms, 4),
        "fwd_bwd_clips_per_s": None if main_r.fwd_bwd_ms is None else round(B * world / (main_r.fwd_bwd_ms * 1e-3), 2),
    }
    if use_dist:
        result["data_parallel"] = dp_report(torch, dist, main_r, dev, world)
    c3_r = f32_r = f32m_r = r32_r = None
    if not args.no_extra and args.dtype != "f32" and args.residual == "bf16":
        # the same workload with the fp32 forward residual stream (the round-1/2 default): reported beside `value`
        keep = args.residual
        args.residual = "f32"
        r32_r = mk(args.config, args.dtype, use_dist)
        args.residual = keep
        r32_r.timed(args.steps, args.warmup, use_graph, eager_too=False)
        r32 = {"ms_per_step": round(r32_r.ms, 4), "clips_per_s": round(r32_r.clips_per_s, 2), "launch": r32_r.launch}
        r32_r.model = r32_r.opt = r32_r.step = r32_r.dp = None
        result["residual_f32"] = r32
    if not args.no_extra and args.config == "c2" and args.dtype == "bf16":
        c3_r = mk("c3", "bf16", use_dist)
        c3_r.timed(args.steps, args.warmup, use_graph)
        c3_clock = c3_r.clock_power() if world == 1 else None
        if world == 1:
            # the parity mode (compute_dtype "f32"), in its default arithmetic (bf16x3: three bf16 products per fp32 product on the
            # bf16 matrix pipe) and, beside it, on the f32-input MFMA (avf_set_f32_arith(0)) - same weights, same inputs
            f32_r = mk(args.config, "f32", False)
            f32_r.timed(max(2, min(20, args.steps // 6)), 1)
            f32_steps = f32_r.steps
            f32_r.model = f32_r.opt = f32_r.step = None  # free it before the instrumented passes
            prev_arith = A._lib.set_f32_arithmetic("f32")
            try:
                f32m_r = mk(args.config, "f32", False)
                f32m_r.timed(max(2, min(10, args.steps // 12)), 1)
                f32m_r.model = f32m_r.opt = f32m_r.step = None
            finally:
                A._lib.set_f32_arithmetic(prev_arith)
    # ---- then the instrumented passes (per-kernel-class HIP events) of the regions that report classes
    if events:
        main_r.instrumented(args.steps)
        if c3_r is not None:
            c3_r.instrumented(args.steps)
    if rank == 0:
        result["stack_tflops_per_gpu"] = round(main_r.tflops, 2)
        result["stack_frac_of_mfma_peak"] = round(main_r.frac, 4)
        tm = None
        if events:
            result["ms_per_step_with_kernel_events"] = round(main_r.ms_events, 4)
            tm = main_r.timing
        if tm is not None:
            mfma = {k: v for k, v in tm.items() if k.startswith("gemm") or k.startswith("attn")}
            dom = max(mfma, key=lambda k: mfma[k]["ms"])
            d = mfma[dom]
            if d["launches"] > 0 and d["ms"] > 0:
                ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
                peak = MX8_PEAK_TFLOPS if dom == "gemm_mx8_nt" else main_r.peak
                traffic, note = pmc_traffic(dom, args.config)
                result["roofline"] = {
                    "kernel": dom, "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": note,
                    "avg_launch_us": round(d["ms"] * 1e3 / d["launches"], 2), "launches": d["launches"],
                    "flops_per_launch": d["flops"] / d["launches"],
                    "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
                    "algorithmic_GBps": round(d["bytes"] / (d["ms"] * 1e-3) / 1e9, 1),
                    "measured": "HIP events attached to every dispatch of the hot-path kernel classes on the launch stream "
                                "(hipExtLaunchKernelGGL start/stop events: the dispatch's own begin/end timestamps), over "
                                "K instrumented steps run right after the timed region (same process, same inputs)",
                }
            result["kernel_classes"] = main_r.classes
        if c3_r is not None:
            cc = c3_r.c
            result["north_star_shape"] = {
                "workload": f"BASELINE.json configs[2] per GPU: d={cc['dim']} L={cc['depth']} T={cc['t_video'] + cc['t_audio']} "
                            f"B={c3_r.B}/GPU (global {c3_r.B * world}), same step as `value`",
                "clips_per_s": round(c3_r.clips_per_s, 2), "ms_per_step": round(c3_r.ms, 4),
                "steps": args.steps, "warmup": args.warmup, "launch": c3_r.launch,
                "eager_ms_per_step": None if c3_r.eager_ms is None else round(c3_r.eager_ms, 4),
                "clock_power": c3_clock,
                "stack_tflops_per_gpu": round(c3_r.tflops, 2),
                "stack_frac_of_mfma_peak": round(c3_r.frac, 4),
                "target_frac": 0.30, "kernel_classes": c3_r.classes if events else None}
            # north_star's own statement is about "the attention + MLP GEMMs": algorithmic FLOPs of the matrix-pipe kernel classes
            # (every GEMM and both attention kernels) over THEIR measured time (HIP events per dispatch), against the bf16 MFMA peak
            if events and getattr(c3_r, "timing", None):
                mf = {k: v for k, v in c3_r.timing.items() if (k.startswith("gemm") or k.startswith("attn")) and v["ms"] > 0}
                fl, ms = sum(v["flops"] for v in mf.values()), sum(v["ms"] for v in mf.values())
                if ms > 0:
                    result["north_star_shape"]["gemm_attention_kernels"] = {
                        "algorithmic_tflops": round(fl / (ms * 1e-3) / 1e12, 1), "frac_of_mfma_peak": round(fl / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS["bf16"], 4),
                        "ms_per_step": round(ms / args.steps, 4), "classes": sorted(mf),
                        "note": "north_star target: >= 0.30 on the attention + MLP GEMMs at (B=32, T=512, d=512); the whole step (LayerNorm, "
                                "folds, Adam, glue included) is stack_frac_of_mfma_peak"}
        if f32_r is not None:
            result["f32_parity_clips_per_s"] = round(f32_r.clips_per_s, 2)
            x3 = A._lib.get_f32_arithmetic() == "bf16x3"
            # bf16x3 executes three bf16 MFMA products per algorithmic product: its roof is the bf16 MFMA peak / 3
            roof = MFMA_PEAK_TFLOPS["bf16"] / 3.0 if x3 else MFMA_PEAK_TFLOPS["f32"]
            result["f32_parity"] = {"ms_per_step": round(f32_r.ms, 3), "steps": f32_steps,
                                    "arithmetic": A._lib.get_f32_arithmetic(),
                                    "stack_tflops_per_gpu": round(f32_r.tflops, 2),
                                    "peak": round(roof, 1),
                                    "peak_is": ("dense bf16 MFMA peak 2500 TFLOP/s / 3 products per fp32 product" if x3
                                                else "dense f32-input MFMA peak"),
                                    "frac": round(f32_r.tflops / roof, 4),
                                    "executed_bf16_mfma_tflops": round(3.0 * f32_r.tflops, 1) if x3 else None,
                                    "f32_mfma_ab": None if f32m_r is None else {
                                        "ms_per_step": round(f32m_r.ms, 3), "clips_per_s": round(f32m_r.clips_per_s, 2),
                                        "steps": f32m_r.steps, "peak": MFMA_PEAK_TFLOPS["f32"],
                                        "frac": round(f32m_r.tflops / MFMA_PEAK_TFLOPS["f32"], 4),
                                        "note": "the same step with avf_set_f32_arith(0): v_mfma_f32_16x16x4_f32 GEMMs and attention"},
                                    "note": "compute_dtype='f32': the mode held to logits rtol 1e-3 against the fp32 CPU reference"}
        if world == 1 and not args.no_cpu_baseline:
            threads = min(os.cpu_count() or 1, 16)
            try:
                threads = min(threads, len(os.sched_getaffinity(0)))
            except Exception:
                pass
            cps, sec = cpu_baseline(c, args.cpu_steps, threads)
            result["cpu_baseline"] = {
                "value": round(cps, 3), "unit": "clips/s", "cores": threads, "kind": "port",
                "sample": f"{args.cpu_steps} fwd+bwd steps (after 1 warm-up) of the same workload (B={B}) through "
                          f"oracle/ (fp32 eager PyTorch ops, un-fused, as the reference), {sec:.2f} s/step"}
        sys.stdout.flush()
        os.write(_REAL_STDOUT, (json.dumps(result) + "\n").encode())  # the ONE line of the contract
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    # Libraries under us write banners to file descriptor 1 (RCCL prints its version block there at communicator creation):
    # everything but the result line goes to stderr.
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    main()
