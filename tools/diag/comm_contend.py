"""What does the training step lose while a collective's kernels are RESIDENT on some CUs?  (one GPU, no second rank)

    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/diag/comm_contend.hip -o tools/diag/bin/libcomm_contend.so
    python tools/diag/comm_contend.py [--blocks 32] [--usec 150] [--per-step 3]

Runs the C2 step eagerly on a compute stream while a side stream keeps launching `per-step` stand-in kernels per step
(`blocks` workgroups x 256 threads resident for `usec` us each: RCCL's ring kernels hold one workgroup per channel for the
whole collective).  Reported: ms per step alone, and with the stand-ins resident.  Run it once per library setting
(AVF_NT_WS=0, AVF_NT_WS_GRID=224, default) - the switches are read once per process.
"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=32)
    ap.add_argument("--usec", type=float, default=150.0)
    ap.add_argument("--per-step", type=int, default=3)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--config", default="c2")
    a = ap.parse_args()
    import torch
    import avformer_amd as A
    import bench
    lib = C.CDLL(os.path.join(ROOT, "tools", "diag", "bin", "libcomm_contend.so"))
    lib.comm_contend_launch.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
    dev = torch.device("cuda:0")
    args = argparse.Namespace(batch=0, config=a.config, residual="bf16", no_optimizer=False, torch_adam=False)
    r = bench.Region(A, torch, None, a.config, "bf16", args, dev, 0, 1, False)
    comp, side = torch.cuda.Stream(), torch.cuda.Stream()
    mhz = 100.0  # first guess of the counter's rate; calibrated below

    def run(contend):
        nonlocal mhz
        with torch.cuda.stream(comp):
            for _ in range(5):
                r.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                if contend:
                    # paced: this step's stand-ins start when the device reaches this step (not when the host issues them)
                    ev = torch.cuda.Event()
                    ev.record(comp)
                    side.wait_event(ev)
                    for _ in range(a.per_step):
                        lib.comm_contend_launch(C.c_void_p(side.cuda_stream), a.blocks, a.usec, mhz)
                r.step()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / a.steps * 1e3
    # calibrate the stand-in's duration (the cycle counter's rate is not documented for this part)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def one():
        with torch.cuda.stream(side):
            lib.comm_contend_launch(C.c_void_p(side.cuda_stream), a.blocks, a.usec, mhz)
            torch.cuda.synchronize()
            e0.record()
            lib.comm_contend_launch(C.c_void_p(side.cuda_stream), a.blocks, a.usec, mhz)
            e1.record()
            torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3
    got = one()
    mhz *= a.usec / max(got - 5.0, 1.0)
    print(f"stand-in kernel: {a.blocks} workgroups resident {one():.0f} us (asked {a.usec:.0f}; counter at ~{mhz:.0f} MHz)")
    run(True)
    sw = {k: os.environ.get(k) for k in ("AVF_NT_WS", "AVF_NT_WS_GRID")}
    for rep in range(3):
        alone, both = run(False), run(True)
        print(f"{sw}: step alone {alone:.3f} ms, with {a.per_step} x {a.blocks} workgroups x {a.usec:.0f} us resident per step {both:.3f} ms "
              f"(+{(both / alone - 1) * 100:.1f} %)")


if __name__ == "__main__":
    main()
