#!/usr/bin/env python
"""bench.py -- predicted frames / s of the iVideoGPT prediction hot path on N MI355X of one node.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one pass of the hot path over one batch of synthetic clips already resident in HBM:
    encode the context frames -> autoregressive rollout (17*F - 1 tokens) -> decode all T frames -> clamp(0, 1)
(BASELINE.json configs[1]: ivideogpt-oxe-64-act-free shapes, synthetic 64x64 bf16 pixels, 64 trajectories per GPU,
2 context + 14 predicted frames; seeded random weights of the real architecture -- no checkpoints exist offline).
Four batches are kept in flight per GPU by default (``--lanes 4``: engine instances over one copy of the weights, each with its own KV
cache, workspace, HIP stream and host thread, decode GEMMs planned under a 40 KiB LDS budget PER ENGINE -- ``ivg_config.decode_lds_kb``;
the K timed steps are dealt round-robin to the lanes): the latency- and HBM-bound rollouts of the batches overlap one another.
``value`` counts the frames of exactly K steps over the wall clock; ``single_lane`` in the line is the same pipeline with one batch in
flight (the latency of a batch); ``roofline_in_flight`` the aggregate HBM rate of the rollout phase with all lanes running.
Multi-GPU: independent trajectories shard by batch rows (weak scaling: 64 per GPU), no data-path collective; the
per-sample metric rows are all-gathered over RCCL once per step (the reference's accelerator.gather, train_gpt.py:476-479).

``python bench.py --gpus N`` without a torchrun environment re-executes itself under ``torch.distributed.run`` (one rank per GPU,
127.0.0.1 rendezvous); ``--config {2,3,4,5}`` selects the per-GPU shapes of BASELINE.json's configs; ``--scaling strong`` keeps the
GLOBAL batch fixed and splits it over the ranks (default: weak, ``--batch`` trajectories per GPU).

Prints ONE JSON line on rank 0 (see the task contract): value = predicted frames / s over all GPUs, measured over a CLEAN timed
loop (no measurement hooks); a separate, untimed profiled pass then yields
  "roofline"     : the kernel class with the most time per step (decode attention / decode GEMMs: HBM; conv3x3 / igemm: MFMA),
                   the others in "roofline_other"
  "cpu_baseline" : the oracle's restatement of the reference algorithm timed on this box's host cores (N = 1 only)
  "fp32_mode"    : the same step with fp32 decode + fp32 rollout, i.e. the arithmetic that meets the 1e-3 parity bar (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver: RCCL needs it (multi-process runs)
# HIP deals its streams round-robin over GPU_MAX_HW_QUEUES hardware queues (default 4): the null stream, this benchmark's main stream,
# the lane streams, the gatherer's stream and RCCL's own must not share one -- two lanes on one hardware queue run one after the other
# (4,987 instead of 5,355 frames/s at four lanes with 4 queues; with a process group 5,069 at 8 queues, 5,617 at 16:
# profiles/r04_lanes.txt).  Has to be in the environment before the runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from ivideogpt_amd import CompressiveVQModel, LlamaForCausalLM, weights as W  # noqa: E402
from ivideogpt_amd import _lib, parallel  # noqa: E402
from ivideogpt_amd.pipeline import frame_metrics, predict_frames  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
# What a PURE stream of v_mfma_f32_16x16x32_bf16 sustains on this part with random operands: the socket reaches its 1,400 W limit and
# the shader clock falls to ~2.0 GHz (tools/ubench/mfma_power.hip, profiles/r05_mfma_power.txt: 1,946-1,999 TFLOP/s; 2,397-2,417 on
# all-zero operands).  Printed beside `frac` as `frac_of_sustained`; `frac` keeps the nominal peak as its denominator.
SUSTAINED_BF16_TFLOPS = 1950.0
PEAK_F32_TFLOPS = 157.3     # f32-input MFMA peak (same guide)
PEAK_HBM_GBS = 8000.0


def build_models(device, res, medium, enc, dec, llm, action_dim=0, ctx=None, frames=16):
    tcfg = W.tokenizer_config(**(W.CTX_VAE64 if res == 64 else dict(W.CTX_VAE256, resolution=256, max_att_resolution=32)))
    lcfg = W.LLAMA_MEDIUM if medium else W.LLAMA_SMALL
    tsd = W.random_tokenizer_state_dict(tcfg, seed=0, codebook_std=0.4)
    tok = CompressiveVQModel(tcfg, tsd, encode_dtype=enc, decode_dtype=dec).to(device)
    if ctx is not None and ctx != tcfg["context_length"]:
        tok.set_context_length(ctx)      # e.g. the BAIR checkpoints run with 1 context frame
    c = tok.context_length
    if action_dim:
        from ivideogpt_amd import HeadModelWithAction
        lsd = W.random_llama_state_dict(lcfg, seed=0, action_dim=action_dim)
        model = HeadModelWithAction(LlamaForCausalLM(lcfg, None, dtype=llm), action_dim, 257 * c - 1, 16, c, frames)
        model.load_state_dict(lsd, strict=True)
        model.to(device)
    else:
        lsd = W.random_llama_state_dict(lcfg, seed=0)
        model = LlamaForCausalLM(lcfg, lsd, dtype=llm).to(device)
    return tcfg, lcfg, tsd, lsd, tok, model


def _cpu_threads():
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return avail


def cpu_baseline_worker(res, medium, ctx, T, sample_b, threads):
    """Runs in a subprocess: reference algorithm on the host cores (oracle = CPU port of the reference's op sequence)."""
    from oracle.llama import LlamaRef
    from oracle.pipeline import predict_reference_algorithm
    from oracle.vq_tokenizer import CompressiveVQRef
    torch.set_num_threads(threads)
    tcfg = W.tokenizer_config(**(W.CTX_VAE64 if res == 64 else dict(W.CTX_VAE256, resolution=256, max_att_resolution=32)))
    lcfg = W.LLAMA_MEDIUM if medium else W.LLAMA_SMALL
    tok = CompressiveVQRef(**tcfg).eval()
    tok.load_state_dict(W.random_tokenizer_state_dict(tcfg, seed=0, codebook_std=0.4), strict=True)
    lsd = W.random_llama_state_dict(lcfg, seed=0)
    llm = LlamaRef(lsd, lcfg["num_hidden_layers"], lcfg["num_attention_heads"], lcfg["rms_norm_eps"], lcfg["rope_theta"],
                   lcfg["max_position_embeddings"])
    g = torch.Generator().manual_seed(123)
    F = T - ctx
    px = torch.rand(sample_b, T, 3, res, res, generator=g)
    u = torch.rand(sample_b, 17 * F - 1, generator=g)
    predict_reference_algorithm(tok, llm, px[:1, :ctx + 1], ctx, uniforms=u[:1, :16], top_k=100)   # warm-up: threads, allocator, kernels
    t0 = time.perf_counter()
    frames, _ = predict_reference_algorithm(tok, llm, px, ctx, uniforms=u, top_k=100)
    dt = time.perf_counter() - t0
    assert torch.isfinite(frames).all()
    print(json.dumps({"value": sample_b * F / dt, "unit": "predicted frames/s", "cores": threads, "kind": "port",
                      "sample": f"{sample_b} trajectories x ({ctx} context + {F} predicted) frames {res}x{res}, fp32, whole-clip tokenize + "
                                f"top-k 100 sampling rollout + detokenize, one pass of {dt:.1f} s on {threads} threads after a 1-frame warm-up; "
                                f"indicative only (the oracle is a port, not the target)"}), flush=True)


def cpu_baseline(res, medium, ctx, T, sample_b, threads, budget_s=240):
    """Bounded: the worker runs in a subprocess under a wall-clock limit so the bench line always prints."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--res", str(res), "--frames", str(T),
           "--cpu-sample", str(sample_b), "--cpu-threads", str(threads)] + (["--medium"] if medium else [])
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=budget_s)
        for line in reversed(out.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "unit": "predicted frames/s", "cores": threads, "kind": "port",
                "sample": "worker failed: " + (out.stderr.strip().splitlines() or ["?"])[-1][:200]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "predicted frames/s", "cores": threads, "kind": "port",
                "sample": f"{sample_b} trajectories did not finish within the {budget_s} s budget"}


KERNEL_NAMES = {
    "decode_attn": "ivg::decode_attn_kernel (RoPE + KV append + single-query attention over the KV cache)",
    "decode_gemm": "ivg::dgemm_kernel (decode-step GEMMs: q/k/v, o-proj, gate/up, down, lm_head; weights streamed once per launch)",
    "conv3x3": "ivg::conv3x3_kernel (LDS-halo 3x3 convolution, MFMA)",
    "igemm": "ivg::gemm256l_kernel + ivg::igemm_kernel<128,128,64> (dense GEMMs / implicit-GEMM convs other than 3x3, MFMA)",
}
PMC_FILES = ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json")
TRACE_FILES = ("r06_kernel_trace_classes.json", "r05_kernel_trace_classes.json", "r04_kernel_trace_classes.json", "r03_kernel_trace_classes.json")
# the same two evidence files for BASELINE config 4 (256 x 256, B = 16): rocprofv3 passes of `bench.py --config 4 --lanes 1 ...`
PMC_FILES_C4 = ("r06_c4_pmc_traffic.json",)
TRACE_FILES_C4 = ("r06_c4_kernel_trace_classes.json",)


def _pmc_traffic(name, files=None):
    """HBM bytes per launch from the committed PMC passes (profiles/rNN_pmc_traffic.json: separate rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE runs of this same command, gfx950 corrections applied as the file states) -- NOT measured in this run."""
    for fn in (files or PMC_FILES):
        path = os.path.join(ROOT, "profiles", fn)
        try:
            with open(path) as f:
                v = json.load(f)["per_launch_bytes"].get(name)
            if v is not None:
                return v, "profiles/" + fn
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def _trace_mean_us(name, files=None):
    """Mean launch duration of a kernel class on the PROFILER's clock (rocprofv3 --kernel-trace of this command, committed under
    profiles/; includes the dispatch the kernels' own stamps do not see) -- NOT measured in this run."""
    for fn in (files or TRACE_FILES):
        try:
            with open(os.path.join(ROOT, "profiles", fn)) as f:
                v = json.load(f)["classes"].get(name)
            if v:
                return v["mean_us"], "profiles/" + fn
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def rooflines(kstats, a, config4=False):
    """One roofline object per measured kernel class, the one with the most kernel time per step first.
    decode_attn / decode_gemm are HBM-bound (algorithmic bytes = the K and V rows, resp. the weight matrix, one launch reads);
    the conv / GEMM classes are MFMA-bound (2 * M * N * K flops per launch).
    `frac` / `achieved` are THIS RUN's figures (HIP events on the launching stream for the conv / GEMM classes; the kernels' own
    wall-clock stamps for the decode classes, which run inside replayed graphs / back to back without host access).  The committed
    profiler figure of the same command -- rocprofv3 --kernel-trace mean duration, dispatch included -- sits beside them as
    `frac_profiler` with its source file: the two must agree; a regression moves `frac` even when the committed file is stale."""
    trace_files, pmc_files = (TRACE_FILES_C4, PMC_FILES_C4) if config4 else (None, None)
    peak_f = PEAK_BF16_TFLOPS if a.decode_dtype == "bf16" else PEAK_F32_TFLOPS
    out = []
    for name, s in kstats.items():
        if not s["launches"] or s["total_ms"] <= 0:
            continue
        sec = s["total_ms"] * 1e-3
        common = {"kernel": KERNEL_NAMES[name], "launches_per_step": s["launches"], "avg_launch_ms": s["total_ms"] / s["launches"],
                  "kernel_ms_per_step": s["total_ms"]}
        if name in ("decode_attn", "decode_gemm"):
            # Two clocks.  The kernels stamp their own launch window (first workgroup start -> last end on the 100 MHz wall clock: no
            # dispatch) in THIS run: that is `frac`.  The profiler's mean duration of the class (rocprofv3 --kernel-trace of this
            # command, committed under profiles/, dispatch included -- what the class costs the step) gives `frac_profiler`.
            per_launch = s["total_bytes"] / s["launches"]
            ach_st = s["total_bytes"] / sec / 1e9
            r = {"bound": "hbm", "achieved": ach_st, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach_st / PEAK_HBM_GBS,
                 "frac_clock": "this run: the kernels' own launch-window stamps (100 MHz wall clock, first workgroup start -> last end)",
                 "algorithmic_bytes_per_launch": per_launch}
            if "mean_launch_us_by_kind" in s:
                r["mean_launch_us_by_kind"] = s["mean_launch_us_by_kind"]
            mean_us, src = _trace_mean_us(name, trace_files)
            if mean_us:
                r["achieved_profiler"] = per_launch / (mean_us * 1e-6) / 1e9
                r["frac_profiler"] = r["achieved_profiler"] / PEAK_HBM_GBS
                r["avg_launch_ms_profiler"] = mean_us * 1e-3
                r["kernel_ms_per_step_profiler"] = mean_us * 1e-3 * s["launches"]
                r["frac_profiler_source"] = src + " (rocprofv3 --kernel-trace mean duration of this class for this command, dispatch included; committed, not this run)"
        else:
            ach = s["total_flops"] / sec / 1e12
            r = {"bound": "mfma", "achieved": ach, "peak": peak_f, "unit": "TFLOP/s", "frac": ach / peak_f,
                 "frac_clock": "this run: HIP events around every launch of the class on the launching stream",
                 "algorithmic_flops_per_launch": s["total_flops"] / s["launches"],
                 "algorithmic_flops_note": "the REFERENCE algorithm's multiplies (an upsampling convolution counts its nine taps over the upsampled "
                                           "grid; the sub-pixel form that runs does 2.25 x fewer)"}
            mean_us, src = _trace_mean_us(name, trace_files)
            if mean_us:
                r["achieved_profiler"] = s["total_flops"] / s["launches"] / (mean_us * 1e-6) / 1e12
                r["frac_profiler"] = r["achieved_profiler"] / peak_f
                r["frac_profiler_source"] = src + " (rocprofv3 --kernel-trace mean duration of this class for this command; committed, not this run)"
            if a.decode_dtype == "bf16":
                r["peak_sustained"] = SUSTAINED_BF16_TFLOPS
                r["frac_of_sustained"] = ach / SUSTAINED_BF16_TFLOPS
                r["peak_sustained_source"] = ("profiles/r05_mfma_power.txt: a pure bf16 MFMA stream on random operands at the socket's 1,400 W limit "
                                              "(committed micro-benchmark, not this run)")
        r["traffic"], src = _pmc_traffic(name, pmc_files)
        if src:
            r["traffic_source"] = src + " (committed rocprofv3 --pmc passes, not this run)"
        r.update(common)
        out.append(r)
    out.sort(key=lambda r: -r["kernel_ms_per_step"])
    return out or [{"bound": "hbm", "achieved": 0.0, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": 0.0, "traffic": None}]


# per-GPU shapes of BASELINE.json's configs; config 1 is the CPU-plumbing case and is not a bench line
CONFIGS = {
    2: dict(batch=64, frames=16, res=64, medium=False, action_dim=0, ctx=0),
    3: dict(batch=32, frames=16, res=64, medium=False, action_dim=4, ctx=1),     # bair-64-act-cond: 256 trajectories over 8 GPUs
    4: dict(batch=16, frames=16, res=256, medium=False, action_dim=0, ctx=0),
    5: dict(batch=64, frames=30, res=64, medium=True, action_dim=0, ctx=0),      # 512 trajectories over 8 GPUs, 989-token sequences
}


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def torchrun_command(gpus, argv, port=None):
    """The launch line of an N-GPU run: one process per GPU under torch.distributed.run, rendezvous on 127.0.0.1."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port or _free_port()), os.path.abspath(__file__)] + list(argv)


def measure(tok, model, pixels, actions, ctx, F, greedy, gen, steps, warmup, per_step=False):
    """-> (seconds of `steps` timed passes, last frames, last gathered rows, the step function[, per-step seconds]); barrier +
    synchronize on both sides.  per_step: every step is also closed by a synchronize and timed on its own (the median is reported
    next to the mean; one host sync per ~200 ms step)."""
    def step():
        frames, rows = predict_frames(tok, model, pixels, ctx, F, actions=actions, do_sample=not greedy, top_k=100, generator=gen,
                                      metrics_of=pixels)   # rows: (mse, psnr, ssim) per trajectory over the predicted frames, on the device
        return frames, parallel.gather_metric_rows_even(rows)
    for _ in range(max(1, warmup)):   # also builds the engines / captures the decode-step graph
        frames, rows = step()
    parallel.barrier()
    torch.cuda.synchronize()
    times = []
    t0 = time.perf_counter()
    for _ in range(steps):
        ts = time.perf_counter()
        frames, rows = step()
        if per_step:
            torch.cuda.synchronize()
            times.append(time.perf_counter() - ts)
    torch.cuda.synchronize()
    parallel.barrier()
    el = time.perf_counter() - t0
    return (el, frames, rows, step, times) if per_step else (el, frames, rows, step)


def measure_lanes(lanes, ctx, F, greedy, steps, warmup, gate=None, gatherer=None):
    """Several batches in flight on one GPU: lane i = its own engines (KV cache, workspace), its own resident batch, its own HIP
    stream and host thread; the `steps` timed steps are dealt round-robin to the lanes (step g -> lane g % L) and run concurrently --
    the MFMA-bound convolutions of one batch's encode / decode fill the matrix pipes the latency- and HBM-bound rollout of the other
    leaves idle.  The metric all-gathers are issued in global step order on every rank (parallel.Turnstile).
    -> (seconds for exactly `steps` steps: barrier + synchronize on both sides, last frames per lane, last rows per lane)."""
    import threading
    L = len(lanes)
    turn = parallel.Turnstile()
    last = [None] * L
    errors = []

    def lane_step(i, g):
        ln = lanes[i]
        frames, rows = predict_frames(ln["tok"], ln["model"], ln["pixels"], ctx, F, actions=ln["actions"], do_sample=not greedy, top_k=100,
                                      generator=ln["gen"], conv_gate=gate, metrics_of=ln["pixels"], rollout_stream=ln.get("rollout_stream"))
        if gatherer is not None:     # the collective is issued by the gatherer's thread on its own stream, in step order
            gatherer.submit(g, rows)
        else:
            rows = turn.run(g, lambda: parallel.gather_metric_rows_even(rows))
        last[i] = (frames, rows)

    def lane_body(i, first, n):
        try:
            torch.cuda.set_device(lanes[i]["stream"].device)   # a new host thread starts on device 0: this rank's GPU, explicitly
            with torch.cuda.stream(lanes[i]["stream"]):
                for k in range(n):
                    lane_step(i, first + k * L)
        except Exception as e:   # surface in the main thread; the other lanes must not wait for this lane's tickets
            errors.append(e)
            turn.abort()

    gathered = {}

    def run(n_steps):
        turn.reset(0)
        if gatherer is not None:
            gatherer.start(0)
        ths = [threading.Thread(target=lane_body, args=(i, i, len(range(i, n_steps, L)))) for i in range(L)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if errors:
            if gatherer is not None:
                gatherer.abort()
            raise errors[0]
        if gatherer is not None:
            gathered.clear()
            gathered.update(gatherer.finish(n_steps))

    for i in range(L):            # warm-up lane by lane on the main thread (engine build, one-time attribute setup), then together
        for _ in range(max(1, warmup)):
            with torch.cuda.stream(lanes[i]["stream"]):
                turn.reset(0)
                if gatherer is not None:
                    gatherer.start(0)
                lane_step(i, 0)
                if gatherer is not None:
                    gatherer.finish(1)
    torch.cuda.synchronize()
    run(L)
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    parallel.barrier()
    el = time.perf_counter() - t0
    if gatherer is not None:      # the gathered rows of the lanes' last steps
        rows_last = [gathered[max(g for g in gathered if g % L == i)] if any(g % L == i for g in gathered) else last[i][1] for i in range(L)]
        return el, [x[0] for x in last], rows_last
    return el, [x[0] for x in last], [x[1] for x in last]


def in_flight_pass(lanes, ctx, F, greedy, with_actions):
    """The dominant kernel IN THE MODE `value` IS MEASURED IN: one more pass of all lanes (one step each, concurrently, as in the timed
    loop -- after it, with the decode kernels' own launch stamps on in every lane's engine).  Reported:
      * per lane the mean launch window of the decode attention / the decode GEMMs while the other lanes' kernels share the chip,
      * over the ROLLOUT PHASE of the pass (the union of the lanes' rollout intervals, from events on the lane streams) the aggregate
        rate at which the K / V rows and the weight matrices were read: all lanes' algorithmic bytes / that time.
    -> the `roofline_in_flight` object of the line."""
    import threading
    L = len(lanes)
    engines = [(ln["model"].llm if with_actions else ln["model"])._engine for ln in lanes]
    for e in engines:
        for k in (_lib.IVG_K_DECODE_ATTN, _lib.IVG_K_DECODE_GEMM):
            e.profile_read(k)
            e.profile_enable(k, True)
    ref = torch.cuda.Event(enable_timing=True)
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(L)]
    errors, host_ms = [], [0.0] * L

    def body(i, record):
        try:
            ln = lanes[i]
            torch.cuda.set_device(ln["stream"].device)
            with torch.cuda.stream(ln["stream"]):
                kw = {"action": ln["actions"]} if ln["actions"] is not None else {}
                if record:
                    evs[i][0].record()
                prompt = ln["tok"].encode_context(ln["pixels"], ctx)
                if record:
                    evs[i][1].record()
                th0 = time.perf_counter()
                toks = ln["model"].generate(prompt, do_sample=not greedy, top_k=100, max_new_tokens=17 * F - 1, generator=ln["gen"], **kw)
                host_ms[i] = (time.perf_counter() - th0) * 1e3   # the host thread's time to ENQUEUE the rollout (no sync inside)
                if record:
                    evs[i][2].record()
                ln["tok"].detokenize(toks, ctx, clamp=True)
                if record:
                    evs[i][3].record()
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    def run(record):
        ths = [threading.Thread(target=body, args=(i, record)) for i in range(L)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        torch.cuda.synchronize()
        if errors:
            raise errors[0]

    run(False)                      # stamps on: the first pass re-captures / warms whatever the switch touches
    for e in engines:
        for k in (_lib.IVG_K_DECODE_ATTN, _lib.IVG_K_DECODE_GEMM):
            e.profile_read(k)
    torch.cuda.synchronize()
    ref.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    run(True)
    per_lane, tot = [], {"attn": 0.0, "gemm": 0.0}
    for i, e in enumerate(engines):
        sa, sg = e.profile_read(_lib.IVG_K_DECODE_ATTN), e.profile_read(_lib.IVG_K_DECODE_GEMM)
        kinds = {k: round(v[0], 2) for k, v in e.profile_gemm_kinds().items()}
        for k in (_lib.IVG_K_DECODE_ATTN, _lib.IVG_K_DECODE_GEMM):
            e.profile_enable(k, False)
        t = [ref.elapsed_time(ev) for ev in evs[i]]
        per_lane.append({"rollout_interval_ms": [t[1], t[2]], "rollout_host_enqueue_ms": host_ms[i], "decode_attn_mean_launch_us": 1e3 * sa["total_ms"] / max(1, sa["launches"]),
                         "decode_gemm_mean_launch_us": 1e3 * sg["total_ms"] / max(1, sg["launches"]), "decode_gemm_mean_launch_us_by_kind": kinds,
                         "decode_attn_bytes": sa["total_bytes"], "decode_gemm_bytes": sg["total_bytes"]})
        tot["attn"] += sa["total_bytes"]
        tot["gemm"] += sg["total_bytes"]
    iv = sorted(p["rollout_interval_ms"] for p in per_lane)
    union, cur0, cur1 = 0.0, iv[0][0], iv[0][1]
    for b0, b1 in iv[1:]:
        if b0 > cur1:
            union += cur1 - cur0
            cur0, cur1 = b0, b1
        else:
            cur1 = max(cur1, b1)
    union += cur1 - cur0
    ach = (tot["attn"] + tot["gemm"]) / (union * 1e-3) / 1e9
    mean_attn = sum(p["decode_attn_mean_launch_us"] for p in per_lane) / L
    return {"bound": "hbm", "unit": "GB/s", "peak": PEAK_HBM_GBS, "lanes": L,
            "achieved": ach, "frac": ach / PEAK_HBM_GBS,
            "what": "all lanes' decode-attention K / V rows + decode-GEMM weight bytes of one pass (one step per lane, concurrent, launch stamps on in "
                    "every lane's engine, run after the timed loop) / the union of the lanes' rollout intervals (events on the lane streams)",
            "rollout_phase_ms": union, "decode_attn_GB": tot["attn"] / 1e9, "decode_gemm_weight_GB": tot["gemm"] / 1e9,
            "decode_attn_only": {"achieved": tot["attn"] / (union * 1e-3) / 1e9, "frac": tot["attn"] / (union * 1e-3) / 1e9 / PEAK_HBM_GBS},
            "decode_attn_mean_launch_us_in_flight": mean_attn,
            "per_lane": per_lane,
            "note": "the launches of one lane are slower beside the other lanes' kernels than alone (roofline: one batch alone); the chip as a whole "
                    "moves this many bytes per second while the rollouts overlap -- the figure the headline mode is bound by",
            "evidence": "timing only: these stamps and one rocprofv3 --kernel-trace of the lanes-only command (profiles/r06_lanes4_overlap.txt); a "
                        "--pmc pass of this mode does not exist -- rocprofv3 serialises kernels under --pmc (device-wide counters), so its per-launch "
                        "traffic would be the one-lane figure by construction"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=0, choices=[0, 2, 3, 4, 5], help="BASELINE.json config preset (per-GPU shapes); 0: the flags below")
    ap.add_argument("--batch", type=int, default=64, help="trajectories per GPU (weak scaling) / in total (--scaling strong)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--res", type=int, default=64, choices=[64, 256])
    ap.add_argument("--medium", action="store_true", help="436 M transformer (config_medium)")
    ap.add_argument("--encode-dtype", default="fp32")
    ap.add_argument("--decode-dtype", default="bf16")
    ap.add_argument("--llm-dtype", default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32-mode", action="store_true", help="skip the extra fp32-arithmetic measurement")
    ap.add_argument("--x3-lanes", type=int, default=4, help="batches in flight of the compliant_mode's second figure (round 6, 24-bit K / V cache: "
                    "2 / 3 / 4 lanes = 2,599 / 2,718 / 2,742 frames/s, profiles/r06_x3_lanes.txt; with the fp32 cache two was the optimum)")
    ap.add_argument("--no-profile", action="store_true", help="skip the profiled pass (rooflines)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of BASELINE configs 3, 4, 5 (default config, N = 1 only)")
    ap.add_argument("--cpu-sample", type=int, default=8)
    ap.add_argument("--cpu-threads", type=int, default=0, help="host threads of the CPU baseline (0: min(32, available cores) -- measured fastest)")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--greedy", action="store_true")
    ap.add_argument("--action-dim", type=int, default=0, help=">0: action-conditioned HeadModelWithAction (BASELINE config 3: 4)")
    ap.add_argument("--ctx", type=int, default=0, help="context frames (0: the tokenizer's pretrained context_length)")
    ap.add_argument("--lane-lds-kb", type=int, default=LlamaForCausalLM.BATCHES_IN_FLIGHT_LDS_KB,
                    help="decode-GEMM LDS budget (KiB) of the engines of the lanes while several batches are in flight (per engine: "
                         "ivg_config.decode_lds_kb / set_decode_lds_kb); 0: the process default, i.e. a whole CU per workgroup (A/B)")
    ap.add_argument("--only-lanes", action="store_true", help="skip the per-stage and one-batch-in-flight passes (kernel traces of the lanes mode alone)")
    ap.add_argument("--lanes", type=int, default=4, help="batches in flight per GPU: engine instances on their own HIP streams and host threads "
                                                         "(1: one batch at a time, the per-batch latency case)")
    ap.add_argument("--conv-gate", type=int, default=0, help="1: at most one lane's convolution phase (encode / decode) on the device at a time "
                                                              "(parallel.PhaseGate), rollouts of the other lanes beside it")
    ap.add_argument("--gather-mode", default="thread", choices=["thread", "lanes"],
                    help="several lanes + a process group: the per-step metric all-gather is issued by one thread on its own stream in step "
                         "order (parallel.OrderedGatherer; default) or by the lanes themselves in ticket order (parallel.Turnstile; A/B)")
    ap.add_argument("--cu-split", type=int, default=0, help="experiment: CUs (of 256) given to the convolution phases of all lanes; the rollouts "
                                                             "run on the others (CU-masked streams); 0: off")
    a = ap.parse_args()
    default_run = not a.config and not (a.medium or a.action_dim or a.res != 64 or a.frames != 16 or a.ctx or a.batch != 64)
    if a.config:
        for k, v in CONFIGS[a.config].items():
            setattr(a, k, v)
        if a.scaling == "strong":
            a.batch *= 8          # the config's global batch (quoted on 8 GPUs)

    if a.cpu_baseline_worker:
        ctx = (W.CTX_VAE64 if a.res == 64 else W.CTX_VAE256)["context_length"]
        return cpu_baseline_worker(a.res, a.medium, ctx, a.frames, a.cpu_sample, a.cpu_threads)

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched bare: become the torchrun parent (one rank per GPU, RCCL over xGMI, loopback rendezvous)
        import subprocess
        sys.exit(subprocess.call(torchrun_command(a.gpus, sys.argv[1:]), env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")))

    rank, world, local = parallel.init_from_env("nccl")
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU path)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    if a.scaling == "strong":
        lo, hi = parallel.shard_rows(a.batch, rank, world)
        B, global_b = hi - lo, a.batch
        assert a.batch % world == 0, "strong scaling: the global batch must divide by the number of GPUs (even all-gather)"
    else:
        B, global_b = a.batch, a.batch * world
    tcfg, lcfg, tsd, lsd, tok, model = build_models(dev, a.res, a.medium, a.encode_dtype, a.decode_dtype, a.llm_dtype, a.action_dim,
                                                    a.ctx or None, a.frames)
    ctx, T = tok.context_length, a.frames
    F = T - ctx
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    pixels = torch.rand(B, T, 3, a.res, a.res, device=dev, generator=g).to(torch.bfloat16)   # resident in HBM before timing
    sample_gen = torch.Generator(device=dev).manual_seed(2000 + rank)
    actions = torch.randn(B, T, a.action_dim, device=dev, generator=g) if a.action_dim else None

    kw = {"action": actions} if a.action_dim else {}
    # Every pass of this benchmark runs on an explicit stream of its own: the engines then launch on the caller's stream and never
    # create their dedicated one.  Streams are dealt round-robin over the (4) hardware queues in creation order -- a stream nobody
    # uses still takes a slot, and two lanes whose streams share a hardware queue run one after the other.
    main_stream = torch.cuda.Stream(device=dev)
    lane_streams = [main_stream] + [torch.cuda.Stream(device=dev) for _ in range(max(0, a.lanes - 1))]   # all of them NOW, next to each other
    gatherer = parallel.OrderedGatherer(dev) if (a.lanes > 1 and torch.distributed.is_initialized()) else None   # (+ its stream)
    main_stream.wait_stream(torch.cuda.current_stream(dev))   # (the resident inputs were written on the default stream)
    torch.cuda.set_stream(main_stream)

    def stage_pass():
        """One pass of the three stages with events on the current stream (lane 0 alone, nothing else on the chip)."""
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        prompt = tok.encode_context(pixels, ctx)
        ev[1].record()
        toks = model.generate(prompt, do_sample=not a.greedy, top_k=100, max_new_tokens=17 * F - 1, generator=sample_gen, **kw)
        ev[2].record()
        tok.detokenize(toks, ctx, clamp=True)
        ev[3].record()
        torch.cuda.synchronize()
        return [ev[i].elapsed_time(ev[i + 1]) for i in range(3)]

    def median(v):
        v = sorted(v)
        return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])

    # ---- per-stage split: median of 3 passes, taken BEFORE anything else runs (one warm-up pass builds the engines)
    stage_pass()
    n_sp = 1 if a.only_lanes else 3
    sp = [stage_pass() for _ in range(n_sp)]
    stage = {"encode_ms": median([p[0] for p in sp]), "rollout_ms": median([p[1] for p in sp]), "decode_ms": median([p[2] for p in sp]),
             "passes": n_sp, "note": f"median of {n_sp} passes on one batch alone, before the timed loops"}

    # ---- one batch in flight (the latency of a batch): >= 10 timed steps, mean and median
    n1 = a.steps if a.lanes <= 1 else (1 if a.only_lanes else max(10, min(a.steps, 12)))
    e1, frames, rows, step, t1 = measure(tok, model, pixels, actions, ctx, F, a.greedy, sample_gen, n1, a.warmup if a.lanes <= 1 else 1, per_step=True)
    assert torch.isfinite(frames).all() and rows.shape == (global_b, 3) and torch.isfinite(rows).all()
    my_elapsed = e1
    single = {"value": global_b * F * n1 / parallel.max_over_ranks(e1, dev), "unit": "predicted frames/s", "ms_per_step": e1 / n1 * 1e3,
              "ms_per_step_median": median(t1) * 1e3, "value_at_median": B * F * world / median(t1), "steps": n1,
              "note": "one batch in flight (lane 0 alone): the latency of a batch through encode -> rollout -> decode"}
    steps_timed = n1
    in_flight = None

    # ---- the headline: `lanes` batches in flight per GPU, clean loop, no hooks
    if a.lanes > 1:
        lanes = [dict(tok=tok, model=model, pixels=pixels, actions=actions, gen=sample_gen, stream=main_stream)]
        for i in range(1, a.lanes):   # further lanes: their own engines over the SAME weights in HBM, and their own resident batch
            tok_i, model_i = tok.replica(), model.replica()
            gi = torch.Generator(device=dev).manual_seed(1000 + rank + 7919 * i)
            lanes.append(dict(tok=tok_i, model=model_i, pixels=torch.rand(B, T, 3, a.res, a.res, device=dev, generator=gi).to(torch.bfloat16),
                              actions=torch.randn(B, T, a.action_dim, device=dev, generator=gi) if a.action_dim else None,
                              gen=torch.Generator(device=dev).manual_seed(2000 + rank + 7919 * i), stream=lane_streams[i]))
        gate = parallel.PhaseGate() if a.conv_gate else None
        if a.cu_split:   # experiment: convolution phases on `cu_split` CUs, rollouts on the other 256 - cu_split (CU-masked streams)
            inter = os.environ.get("IVG_CU_SPLIT_MODE", "block") == "interleave"
            n_conv = a.cu_split
            if inter:    # spread both sets over all XCDs: every (256 / gcd)-th ... take CU i for conv when (i * n_conv) % 256 < n_conv
                conv_bits = [i for i in range(256) if (i * n_conv) % 256 < n_conv]
            else:
                conv_bits = list(range(n_conv))
            roll_bits = [i for i in range(256) if i not in set(conv_bits)]
            for ln in lanes:
                ln["stream"] = parallel.cu_masked_stream(dev, conv_bits)
                ln["rollout_stream"] = parallel.cu_masked_stream(dev, roll_bits)
        # several batches in flight: decode GEMMs with a small LDS footprint, so that the kernels of the other batches fit beside them
        # -- a property of the lanes' ENGINES (ivg_config.decode_lds_kb), not of the process; lane 0's engine gets its default back below
        for ln in lanes:
            ln["model"].set_decode_lds_kb(a.lane_lds_kb)
        my_elapsed, lane_frames, lane_rows = measure_lanes(lanes, ctx, F, a.greedy, a.steps, a.warmup, gate,
                                                           gatherer if a.gather_mode == "thread" else None)
        for fr, rw in zip(lane_frames, lane_rows):
            assert torch.isfinite(fr).all() and rw.shape == (global_b, 3) and torch.isfinite(rw).all()
        steps_timed = a.steps
        if not a.no_profile:
            in_flight = in_flight_pass(lanes, ctx, F, a.greedy, actions is not None)
        model.set_decode_lds_kb(0)
        del lanes[1:]
    elapsed = parallel.max_over_ranks(my_elapsed, dev)
    per_rank = parallel.gather_metric_rows_even(torch.tensor([[B * F * steps_timed / my_elapsed]], device=dev, dtype=torch.float32)).flatten().tolist()

    # ---- profiled pass (untimed): per-kernel-class durations for the rooflines
    def profile_classes(tok_engine, lm_engine, step_fn):
        """-> (kstats per kernel class, line fit of the decode attention): HIP events around every conv / GEMM launch of the two engines,
        the decode kernels' own launch-window stamps; one capture step, one profiled step."""
        ks = {}
        prof_engines = [tok_engine, lm_engine]
        bf = a.decode_dtype == "bf16"
        ev_classes = {"igemm": _lib.IVG_K_IGEMM_BF16 if bf else _lib.IVG_K_IGEMM_F32,
                      "conv3x3": _lib.IVG_K_CONV3X3_BF16 if bf else _lib.IVG_K_CONV3X3_F32}
        for e in prof_engines:
            for k in ev_classes.values():
                e.profile_read(k)
                e.profile_enable(k, True)
        lm_engine.profile_enable(_lib.IVG_K_DECODE_ATTN, True)   # the step graph is re-captured with the stamps on
        lm_engine.profile_enable(_lib.IVG_K_DECODE_GEMM, True)
        step_fn()       # capture
        for e in prof_engines:
            for k in ev_classes.values():
                e.profile_read(k)
        step_fn()       # the profiled step
        for name, k in ev_classes.items():
            st = [e.profile_read(k) for e in prof_engines]
            for e in prof_engines:
                e.profile_enable(k, False)
            ks[name] = {key: sum(x[key] for x in st) for key in ("launches", "total_ms", "total_flops", "total_bytes")}
        ks["decode_attn"] = lm_engine.profile_read(_lib.IVG_K_DECODE_ATTN)
        fit = lm_engine.profile_attn_fit()
        ks["decode_gemm"] = lm_engine.profile_read(_lib.IVG_K_DECODE_GEMM)
        ks["decode_gemm"]["mean_launch_us_by_kind"] = {k: round(v[0], 2) for k, v in lm_engine.profile_gemm_kinds().items()}
        lm_engine.profile_enable(_lib.IVG_K_DECODE_ATTN, False)
        lm_engine.profile_enable(_lib.IVG_K_DECODE_GEMM, False)
        return ks, fit

    kstats, attn_fit = {}, (0.0, 0.0)
    llm_engine = (model.llm if a.action_dim else model)._engine
    if not a.no_profile:
        kstats, attn_fit = profile_classes(tok._engine, llm_engine, step)

    # not part of a prediction step: the eval path's full-clip tokenize (train_gpt.py:356: context encoder on the ctx frames +
    # CONDITIONAL encoder with cross-attention on all F future frames of every trajectory, SURVEY.md rows a1 / a3 at batch)
    tok.tokenize(pixels, ctx)   # warm-up (workspace plan)
    ev_t = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev_t[0].record()
    tok.tokenize(pixels, ctx)
    ev_t[1].record()
    torch.cuda.synchronize()
    stage["tokenize_full_ms"] = ev_t[0].elapsed_time(ev_t[1])
    stage["tokenize_full_note"] = (f"full-clip tokenize of {B} x {T} frames (eval path, outside the timed step): "
                                   f"{B * T / max(stage['tokenize_full_ms'], 1e-9) * 1e3:.0f} frames/s")

    # ---- the arithmetic that meets the 1e-3 parity bar, same workload, short runs (N = 1 only):
    #   fp32_mode       fp32 decode + fp32 rollout on f32-input MFMAs (1/16 of the bf16 matrix rate)
    #   compliant_mode  "x3": fp32 tensors, every matrix product of decode and prompt pass in split-bf16 arithmetic (bf16 hi + lo per
    #                   operand, fp32 accumulate; conv3x3.hip / igemm.hip / gemm256.hip X3), K / V cache at 24 bits per element in two
    #                   planes (round 6, llama_ops.hip) -- the same 1e-3 bars (tests/test_gpu_x3.py)
    alt = {}
    if world == 1 and not a.no_fp32_mode and (a.decode_dtype, a.llm_dtype) != ("fp32", "fp32"):
        del model, tok
        torch.cuda.empty_cache()
        for key, dec, llm, note in (("fp32_mode", "fp32", "fp32", "pixels / logits within 1e-3 of the fp32 reference, token-identical rollouts (tests/test_gpu_models.py)"),
                                    ("compliant_mode", "x3", "x3", "split-bf16 arithmetic on fp32 tensors, 24-bit K / V cache: pixels / logits within 1e-3 of the fp32 reference, "
                                                                   "token-identical greedy rollouts (tests/test_gpu_x3.py)")):
            _, _, _, _, tok_a, model_a = build_models(dev, a.res, a.medium, a.encode_dtype, dec, llm, a.action_dim, a.ctx or None, a.frames)
            n_a = max(1, min(5, a.steps))
            e_a, fr_a, _, _, t_a = measure(tok_a, model_a, pixels, actions, ctx, F, a.greedy, sample_gen, n_a, 1, per_step=True)
            assert torch.isfinite(fr_a).all()
            alt[key] = {"value": B * F * n_a / e_a, "unit": "predicted frames/s", "ms_per_step": e_a / n_a * 1e3, "ms_per_step_median": median(t_a) * 1e3,
                        "steps": n_a, "lanes": 1, "arith": {"encode": a.encode_dtype, "rollout": llm, "decode": dec}, "note": note}
            if key == "compliant_mode" and a.lanes > 1:   # the same mode with batches in flight, as the headline
                n_x3 = min(a.lanes, a.x3_lanes)
                lanes_a = [dict(tok=tok_a, model=model_a, pixels=pixels, actions=actions, gen=sample_gen, stream=main_stream)]
                for i in range(1, n_x3):
                    gi = torch.Generator(device=dev).manual_seed(1000 + rank + 7919 * i)
                    lanes_a.append(dict(tok=tok_a.replica(), model=model_a.replica(),
                                        pixels=torch.rand(B, T, 3, a.res, a.res, device=dev, generator=gi).to(torch.bfloat16),
                                        actions=torch.randn(B, T, a.action_dim, device=dev, generator=gi) if a.action_dim else None,
                                        gen=torch.Generator(device=dev).manual_seed(2000 + rank + 7919 * i), stream=lane_streams[i]))
                n_l = 4 * n_x3
                for ln in lanes_a:
                    ln["model"].set_decode_lds_kb(a.lane_lds_kb)
                e_l, fl, _ = measure_lanes(lanes_a, ctx, F, a.greedy, n_l, 1, parallel.PhaseGate() if a.conv_gate else None)
                assert all(torch.isfinite(x).all() for x in fl)
                alt[key]["lanes_in_flight"] = {"lanes": n_x3, "value": B * F * n_l / e_l, "ms_per_step": e_l / n_l * 1e3, "steps": n_l}
                del lanes_a
            del model_a, tok_a
            torch.cuda.empty_cache()

    # ---- the other per-GPU shapes of BASELINE.json (configs 3, 4, 5) inside the same driver-observed line: one batch in flight over
    # >= 10 steps (mean and median), then `lanes` batches in flight over 2 * lanes steps -- the two figures config 2 is reported with
    other = {}
    if world == 1 and default_run and not a.no_other_configs:
        for k in (3, 4, 5):
            c = CONFIGS[k]
            try:
                tc, lc, _, _, tok_o, model_o = build_models(dev, c["res"], c["medium"], a.encode_dtype, a.decode_dtype, a.llm_dtype, c["action_dim"],
                                                            c["ctx"] or None, c["frames"])
                ctx_o, Fo = tok_o.context_length, c["frames"] - tok_o.context_length
                go = torch.Generator(device=dev).manual_seed(3000 + k)

                def inputs(gen):
                    return (torch.rand(c["batch"], c["frames"], 3, c["res"], c["res"], device=dev, generator=gen).to(torch.bfloat16),
                            torch.randn(c["batch"], c["frames"], c["action_dim"], device=dev, generator=gen) if c["action_dim"] else None)
                px_o, act_o = inputs(go)
                n_o = 10
                e_o, fr_o, _, _, t_o = measure(tok_o, model_o, px_o, act_o, ctx_o, Fo, a.greedy, go, n_o, 1, per_step=True)
                assert torch.isfinite(fr_o).all()
                other[f"config_{k}"] = {"value": c["batch"] * Fo * n_o / e_o, "unit": "predicted frames/s", "ms_per_step": e_o / n_o * 1e3,
                                        "ms_per_step_median": median(t_o) * 1e3, "steps": n_o, "lanes": 1,
                                        "workload": f"{c['batch']} trajectories per GPU, {ctx_o} context + {Fo} predicted frames, {c['res']}x{c['res']}, "
                                                    f"{'medium (436 M)' if c['medium'] else 'small (138 M)'} transformer"
                                                    + (f", {c['action_dim']}-dim actions" if c["action_dim"] else "")}
                if k == 4 and not a.no_profile:
                    # BASELINE config 4 is labelled "HBM-bound conv decode": its stage split and its dominant kernel class, measured here
                    # (events / stamps of this run) with the committed rocprofv3 evidence of `bench.py --config 4 --lanes 1` beside it
                    ev4 = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                    sp4 = []
                    for _ in range(3):
                        ev4[0].record()
                        pr4 = tok_o.encode_context(px_o, ctx_o)
                        ev4[1].record()
                        tk4 = model_o.generate(pr4, do_sample=not a.greedy, top_k=100, max_new_tokens=17 * Fo - 1, generator=go)
                        ev4[2].record()
                        tok_o.detokenize(tk4, ctx_o, clamp=True)
                        ev4[3].record()
                        torch.cuda.synchronize()
                        sp4.append([ev4[i].elapsed_time(ev4[i + 1]) for i in range(3)])
                    other["config_4"]["stage_ms"] = {"encode_ms": median([p[0] for p in sp4]), "rollout_ms": median([p[1] for p in sp4]),
                                                     "decode_ms": median([p[2] for p in sp4]), "passes": 3}
                    ks4, _ = profile_classes(tok_o._engine, model_o._engine,
                                             lambda: predict_frames(tok_o, model_o, px_o, ctx_o, Fo, do_sample=not a.greedy, top_k=100, generator=go))
                    rl4 = rooflines(ks4, a, config4=True)
                    other["config_4"]["roofline"], other["config_4"]["roofline_other"] = rl4[0], rl4[1:]
                if a.lanes > 1:
                    lanes_o = [dict(tok=tok_o, model=model_o, pixels=px_o, actions=act_o, gen=go, stream=main_stream)]
                    for i in range(1, a.lanes):
                        gi = torch.Generator(device=dev).manual_seed(3000 + k + 7919 * i)
                        px_i, act_i = inputs(gi)
                        lanes_o.append(dict(tok=tok_o.replica(), model=model_o.replica(), pixels=px_i, actions=act_i,
                                            gen=torch.Generator(device=dev).manual_seed(4000 + k + 7919 * i), stream=lane_streams[i]))
                    for ln in lanes_o:
                        ln["model"].set_decode_lds_kb(a.lane_lds_kb)
                    n_l = 2 * a.lanes
                    e_l, fl, _ = measure_lanes(lanes_o, ctx_o, Fo, a.greedy, n_l, 1)
                    assert all(torch.isfinite(x).all() for x in fl)
                    other[f"config_{k}"]["lanes_in_flight"] = {"lanes": a.lanes, "value": c["batch"] * Fo * n_l / e_l, "ms_per_step": e_l / n_l * 1e3,
                                                               "steps": n_l}
                    del lanes_o, fl
                del tok_o, model_o, px_o, fr_o
            except Exception as ex:   # never lose the headline to a side measurement
                other[f"config_{k}"] = {"value": None, "error": repr(ex)[:200]}
            torch.cuda.empty_cache()

    # ---- the reference's MULTI-SAMPLE callers (round 6): one clip's context, t rollouts.  BASELINE config 1 = inference/predict.py
    # (--repeat_times 5 over one clip) and VP2's planner (vp/ivideogpt_interface.py: 200 candidate action sequences over the same two
    # frames, generate_max_batchsize 100 / decode_max_batchsize 67 of vp/ivideogpt.yaml), each timed twice: the callers' own flow on
    # the plain entries (every row prefills / stores / streams / decodes its copy of the context) and with shared_context (once per clip)
    shared = {}
    if world == 1 and default_run and not a.no_other_configs:
        for key, spec in (("config_1", dict(t=5, frames=16, action_dim=0, gmax=5, dmax=5, steps=8,
                                            what="inference/predict.py: 1 clip, --repeat_times 5, 2 context + 14 predicted frames, 64x64, small transformer")),
                          ("vp2", dict(t=200, frames=12, action_dim=5, gmax=100, dmax=67, steps=4,
                                       what="vp/ivideogpt_interface.py: 200 candidate action sequences (5-dim) over ONE 2-frame context, 10 predicted "
                                            "frames, generate_max_batchsize 100, decode_max_batchsize 67 (vp/ivideogpt.yaml)"))):
            try:
                _, lc, _, _, tok_s, model_s = build_models(dev, 64, False, a.encode_dtype, a.decode_dtype, a.llm_dtype, spec["action_dim"], None, spec["frames"])
                ctx_s, t_s = tok_s.context_length, spec["t"]
                Fs = spec["frames"] - ctx_s
                gs = torch.Generator(device=dev).manual_seed(5000 + t_s)
                clip = torch.rand(1, spec["frames"], 3, 64, 64, device=dev, generator=gs).to(torch.bfloat16)
                acts = torch.randn(t_s, spec["frames"], spec["action_dim"], device=dev, generator=gs) if spec["action_dim"] else None
                n_new = 17 * Fs - 1

                def flow(share):
                    # the caller's own sequence (predict.py:47-73 / ivideogpt_interface.py:155-202), chunked as the caller chunks it
                    outs = []
                    for s0 in range(0, t_s, spec["gmax"]):
                        n = min(spec["gmax"], t_s - s0)
                        if share or key == "config_1":      # predict.py tokenizes the ONE clip and repeats the tokens (:53-65)
                            prompt = tok_s.encode_context(clip, ctx_s).repeat(n, 1)
                        else:                               # VP2 tokenizes every copy of the observation (:155-169)
                            prompt = tok_s.encode_context(clip.expand(n, -1, -1, -1, -1).contiguous(), ctx_s)
                        kw_s = {"action": acts[s0:s0 + n]} if acts is not None else {}
                        toks = model_s.generate(prompt, do_sample=True, top_k=100, max_new_tokens=n_new, generator=gs,
                                                shared_context=n if share and n > 1 else None, **kw_s)
                        for d0 in range(0, n, spec["dmax"]):
                            ch = toks[d0:d0 + spec["dmax"]]
                            outs.append(tok_s.detokenize(ch, ctx_s, clamp=True, shared_context=ch.shape[0] if share and ch.shape[0] > 1 else None))
                    return outs
                res_s = {}
                for mode, share in (("shared_context", True), ("plain", False)):
                    flow(share)                             # warm-up: engines, workspace plans
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(spec["steps"]):
                        outs = flow(share)
                    torch.cuda.synchronize()
                    el = time.perf_counter() - t0
                    assert all(torch.isfinite(o).all() for o in outs)
                    res_s[mode] = {"value": t_s * Fs * spec["steps"] / el, "unit": "predicted frames/s", "ms_per_call": el / spec["steps"] * 1e3, "steps": spec["steps"]}
                # K / V bytes a decode step asks HBM for (algorithmic; bf16 K + V of every layer = kv_row bytes per cached position), mean over
                # the rollout: every row streams all its positions (plain) vs the shared prefix ONCE per generate chunk + the rows' own tails
                kv_row = 2 * lc["num_hidden_layers"] * lc["hidden_size"] * (2 if a.llm_dtype == "bf16" else 4)
                P = 257 * ctx_s - 1
                mean_len = P + 1 + (n_new - 1) / 2.0
                chunks = -(-t_s // spec["gmax"])
                shared[key] = dict(res_s, workload=spec["what"], speedup=res_s["shared_context"]["value"] / res_s["plain"]["value"],
                                   kv_bytes_per_decode_step={"plain": t_s * mean_len * kv_row, "shared_context": (chunks * P + t_s * (mean_len - P)) * kv_row,
                                                             "note": "algorithmic HBM bytes of the K / V rows one decode step reads, mean over the rollout; "
                                                                     "shared: the prompt's rows once per generate chunk (the group's other rows hit L2 / Infinity Cache)"},
                                   prompt_rows_prefilled={"plain": t_s, "shared_context": chunks}, context_frames_decoded={"plain": t_s * ctx_s,
                                                           "shared_context": ctx_s * sum(-(-min(spec["gmax"], t_s - s0) // spec["dmax"]) for s0 in range(0, t_s, spec["gmax"]))})
                del tok_s, model_s, outs
            except Exception as ex:   # never lose the headline to a side measurement
                shared[key] = {"value": None, "error": repr(ex)[:300]}
            torch.cuda.empty_cache()

    if rank == 0:
        units = global_b * F * a.steps
        rl = rooflines(kstats, a)
        for r in rl:
            if r.get("kernel", "").startswith("ivg::decode_attn") and attn_fit[1] > 0:   # launch duration = fixed + bytes / rate over the cache lengths
                r["fit"] = {"fixed_us_per_launch": attn_fit[0], "streaming_GBps": attn_fit[1]}
        name = f"ivideogpt-{'bair' if a.action_dim else 'oxe'}-{a.res}-{'act-cond' if a.action_dim else 'act-free'}{'-medium' if a.medium else ''}"
        default_shapes = not (a.medium or a.action_dim or a.res != 64 or T != 16 or a.ctx)
        cfg_label = a.config or (2 if default_shapes else "custom shapes")
        out = {
            "metric": f"predicted frames/sec (encode+GPT rollout+decode), {a.res}x{a.res}x{T}f",
            "value": units / elapsed,
            "unit": "predicted frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
            "dtype": "bf16" if "bf16" in (a.decode_dtype, a.llm_dtype) else "f32", "data": "synthetic",
            "config": {"workload": f"{name} (BASELINE config {cfg_label}): "
                                   f"synthetic {a.res}x{a.res} bf16 clips, {B} trajectories per GPU, {ctx} context + {F} predicted frames, "
                                   f"top-k 100 sampling, seeded random weights",
                       "global_batch": global_b, "frames": T, "resolution": a.res,
                       "arith": {"encode": a.encode_dtype, "rollout": a.llm_dtype, "decode": a.decode_dtype},
                       "parallelism": f"batch-shard x{world} (no data-path collective; 1 RCCL all-gather of [B,3] metric rows (mse, psnr, ssim) per step)"
                                      + (f"; {a.lanes} batches in flight per GPU (engine instances on their own HIP streams and host threads: steps dealt "
                                         f"round-robin, a step = one batch through encode -> rollout -> decode)" if a.lanes > 1 else ""),
                       "lanes": a.lanes},
            "per_rank_frames_per_s": per_rank,
            "roofline": rl[0], "roofline_other": rl[1:],
            "stage_ms": stage,
        }
        if in_flight:
            out["roofline_in_flight"] = in_flight
        if single:
            out["single_lane"] = single
        out.update(alt)
        if other:
            out["other_configs"] = other
        if shared:
            out["shared_context"] = shared
        if world == 1 and not a.no_cpu_baseline:
            # 32 threads: MORE threads make this port slower on the GPU box's host (4 trajectories: 4.35 frames/s on 32 threads, 2.11 on
            # 64, no result within 200 s on all 256 -- profiles/r04_cpu_baseline_threads.txt); --cpu-threads N overrides
            threads = a.cpu_threads or min(32, _cpu_threads())
            out["cpu_baseline"] = cpu_baseline(a.res, a.medium, ctx, T, a.cpu_sample, threads, budget_s=150)
            out["cpu_baseline"]["host_cores_available"] = _cpu_threads()
            out["cpu_baseline"]["threads_note"] = ("32 of the host's threads: measured fastest (64 threads 0.49x, 256 threads no result in 200 s: "
                                                   "profiles/r04_cpu_baseline_threads.txt)")
            if out["cpu_baseline"]["value"]:
                out["speedup_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    parallel.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
