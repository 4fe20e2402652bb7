"""Developer probe: what clock and socket power does the chip hold while the encoder's library GEMMs run?

Question behind it: the 256-chunk encode sits at ~0.51 of the 2.5 PFLOP/s dense bf16 peak, and that
peak is quoted at the 2.4 GHz boost clock.  If the matrix pipes hold a lower clock under a sustained GEMM load (power
management), the ceiling of ANY bf16 GEMM on this part is 2.5 PF x (held clock / 2.4 GHz), whatever its schedule.
The probe samples amdsmi (gfx clock, socket power) every ~20 ms from a side thread while the GPU runs
  idle / the gate|up GEMM alone / the whole 256-chunk forward / the HBM-bound SwiGLU pass alone / the search scan,
and prints the median and the range of the samples of every phase plus the achieved rate of the phase.
"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

samples = []
stop = False


def sampler():
    try:
        import amdsmi
        amdsmi.amdsmi_init()
        h = amdsmi.amdsmi_get_processor_handles()[0]
    except Exception as e:  # noqa: BLE001
        print("amdsmi unavailable:", repr(e), flush=True)
        return
    while not stop:
        rec = {"t": time.perf_counter()}
        try:
            m = amdsmi.amdsmi_get_gpu_metrics_info(h)
            for key in ("current_gfxclk", "average_gfxclk_frequency", "current_socket_power", "average_socket_power",
                        "current_uclk", "temperature_hotspot", "throttle_status", "indep_throttle_status"):
                if key in m:
                    rec[key] = m[key]
            if "current_gfxclks" in m:
                v = [x for x in m["current_gfxclks"] if isinstance(x, (int, float)) and 0 < x < 60000]
                if v:
                    rec["gfxclks_mean"] = float(np.mean(v))
                    rec["gfxclks_min"] = float(np.min(v))
        except Exception as e:  # noqa: BLE001
            rec["metrics_err"] = repr(e)[:80]
        try:
            c = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)
            rec["clk_info"] = c.get("clk", c.get("cur_clk"))
        except Exception as e:  # noqa: BLE001
            rec["clk_err"] = repr(e)[:80]
        try:
            p = amdsmi.amdsmi_get_power_info(h)
            for key in ("current_socket_power", "average_socket_power", "socket_power"):
                if key in p:
                    rec["pw_" + key] = p[key]
        except Exception as e:  # noqa: BLE001
            rec["pw_err"] = repr(e)[:80]
        samples.append(rec)
        time.sleep(0.02)


def phase(name, fn, seconds, work_per_call=None, unit=""):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    i0 = len(samples)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        fn()
        n += 1
        if n % 4 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    recs = samples[i0 + len(samples[i0:]) // 4:]   # drop the first quarter (ramp)
    line = f"{name}: {n} calls in {dt:.2f} s"
    if work_per_call:
        line += f", {work_per_call * n / dt:.1f} {unit}"
    keys = sorted({k for r in recs for k in r if k != "t" and not k.endswith("err")})
    for k in keys:
        v = [r[k] for r in recs if isinstance(r.get(k), (int, float))]
        if v:
            line += f" | {k} med {np.median(v):.0f} [{min(v):.0f}..{max(v):.0f}]"
    errs = {r[k] for r in recs for k in r if k.endswith("err")}
    if errs:
        line += f" | errors {sorted(errs)[:2]}"
    print(line, flush=True)


def main():
    global stop
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    phase("idle", lambda: time.sleep(0.05), 2.0)

    m, k, n = 65536, 2560, 19456
    a = torch.randn(m, k, device=dev, dtype=torch.bfloat16) * 0.5
    w = torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    phase("gate|up GEMM 65536x2560x19456 (hipBLASLt)", lambda: torch.matmul(a, w.t(), out=out), 6.0,
          2.0 * m * k * n / 1e12, "TFLOP/s")
    # the same GEMM in short bursts with idle gaps (what an isolated micro-benchmark sees)
    def burst():
        torch.matmul(a, w.t(), out=out)
        torch.cuda.synchronize()
        time.sleep(0.02)
    phase("gate|up GEMM, one call per 20 ms pause", burst, 4.0)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); time.sleep(0.5)
    ev0.record(); torch.matmul(a, w.t(), out=out); ev1.record(); torch.cuda.synchronize()
    print(f"  one cold-start call after 0.5 s idle: {2.0 * m * k * n / 1e9 / ev0.elapsed_time(ev1):.1f} TFLOP/s", flush=True)
    az = torch.zeros_like(a); wz = torch.zeros_like(w)
    phase("gate|up GEMM on ZERO operands", lambda: torch.matmul(az, wz.t(), out=out), 4.0,
          2.0 * m * k * n / 1e12, "TFLOP/s")
    del az, wz

    from cadence_rag_amd.encoder import ops
    act = torch.empty(m, n // 2, device=dev, dtype=torch.bfloat16)
    phase("swiglu pass (HBM bound)", lambda: ops.swiglu(out, act), 3.0, (m * n * 2 + m * n) / 1e12, "TB/s")
    del a, w, out, act

    from cadence_rag_amd.encoder.qwen3 import PackedBatch, Qwen3Config, Qwen3Encoder
    cfg = Qwen3Config()
    enc = Qwen3Encoder.random_init(cfg, seed=1, device=dev)
    rng = np.random.default_rng(7)
    lens = np.clip(rng.normal(256, 64, size=256).astype(np.int64), 8, 1024)
    batch = PackedBatch.build(lens.tolist(), dev)
    ids = torch.randint(0, cfg.vocab_size, (int(lens.sum()),), device=dev, dtype=torch.int32)
    ctx = float((lens.astype(np.float64) ** 2).sum() / lens.sum())
    fl = cfg.flops_per_token(ctx) * float(lens.sum()) / 1e12
    phase("256-chunk forward (36 layers)", lambda: enc.forward_packed(ids, batch), 12.0, fl, "TFLOP/s (algorithmic)")
    stop = True
    th.join(timeout=1.0)


if __name__ == "__main__":
    main()
