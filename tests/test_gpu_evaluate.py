"""``train_gpt.evaluate`` (mirror of /root/reference/train_gpt.py:152-195,321-512) on the MI355X engine against the same loop over
the CPU oracle (fp32 engine mode, t = 2 samples per trajectory, B = 3, chunked generation and decoding), the RCCL collectives on
one GPU, and the eval CLI."""
import math
import os
import subprocess
import sys

import pytest
import torch

from helpers import oracle_llama, oracle_tokenizer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TOK_CFG = dict(block_out_channels=(64, 128, 128), layers_per_block=1, latent_channels=64, num_vq_embeddings=512, num_dyn_embeddings=512,
               norm_num_groups=32, mid_block_add_attention=False, context_length=2, resolution=64, max_att_resolution=16)
LLM_CFG = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2, rms_norm_eps=1e-6,
               rope_theta=10000.0, max_position_embeddings=1024, vocab_size=1026)


def test_evaluate_matches_the_oracle_loop():
    import train_gpt
    from eval_standins import OracleEvaluator, OracleLM
    from ivideogpt_amd import CompressiveVQModel, LlamaForCausalLM, weights as W
    from ivideogpt_amd.metrics import Evaluator
    from ivideogpt_amd.parallel import LocalAccelerator
    tcfg = W.tokenizer_config(**TOK_CFG)
    tsd = W.random_tokenizer_state_dict(tcfg, 81, codebook_std=0.4)
    lsd = W.random_llama_state_dict(LLM_CFG, 82)
    g = torch.Generator().manual_seed(83)
    T, ctx, B, t = 5, 2, 3, 2
    batches = [torch.rand(B, T, 3, 64, 64, generator=g) for _ in range(2)]
    args = train_gpt.eval_args(context_length=ctx, segment_length=T, eval_generate_times=t, max_generate_batchsize=B, max_decode_batchsize=4,
                               log_gif_interval=1000)
    n_new = 17 * (T - ctx) - 1
    # the engine draws its uniforms with torch.rand on the GPU, one call per generate: replay the same stream for the oracle
    torch.manual_seed(1234)
    draws = [torch.rand(B, n_new, device=DEV).cpu() for _ in range(len(batches) * t)]
    tok = CompressiveVQModel(tcfg, tsd, encode_dtype="fp32", decode_dtype="fp32").to(DEV)
    llm = LlamaForCausalLM(dict(LLM_CFG), lsd, dtype="fp32").to(DEV)
    torch.manual_seed(1234)
    acc = LocalAccelerator(DEV)
    logs = train_gpt.evaluate(args, acc, tok, llm, batches, Evaluator(), 7)
    assert acc.logged and acc.logged[0][0] == 7
    it = iter(draws)
    ref = train_gpt.evaluate(args, LocalAccelerator("cpu"), oracle_tokenizer(tcfg, tsd, ctx), OracleLM(oracle_llama(LLM_CFG, lsd), lambda b, n: next(it)),
                             batches, OracleEvaluator(), 7)
    print("engine", logs, "\noracle", ref)
    assert abs(logs["eval/eval_loss"] - ref["eval/eval_loss"]) < 1e-3
    assert abs(logs["eval/perplexity"] - ref["eval/perplexity"]) < 2e-3 * ref["eval/perplexity"]
    assert abs(logs["eval/mse"] - ref["eval/mse"]) < 1e-4 and abs(logs["eval/psnr"] - ref["eval/psnr"]) < 2e-2
    assert abs(logs["eval/ssim"] - ref["eval/ssim"]) < 1e-3
    assert math.isnan(logs["eval/lpips"])


def test_rccl_collectives_on_one_gpu():
    """The RCCL path of the multi-GPU run, with what a 1-GPU lease allows: ``nccl`` backend initialised through
    ``parallel.init_from_env`` with WORLD_SIZE = 1, ``all_gather_into_tensor`` of the [B, 3] metric rows, ``all_reduce(MAX)``,
    ``barrier``, teardown -- in a subprocess (a process group cannot be re-initialised inside the pytest process)."""
    code = r"""
import os, sys, torch
sys.path.insert(0, sys.argv[1])
os.environ["IVG_FORCE_COLLECTIVE"] = "1"
from ivideogpt_amd import parallel
import torch.distributed as dist
rank, world, local = parallel.init_from_env("nccl")
assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
dev = torch.device("cuda", local)
rows = torch.arange(64 * 3, dtype=torch.float32, device=dev).view(64, 3)
out = parallel.gather_metric_rows_even(rows)                      # ncclAllGather through RCCL
assert out.data_ptr() != rows.data_ptr() and torch.equal(out, rows)
acc = parallel.LocalAccelerator(dev, force_collective=True)
assert torch.equal(acc.gather(rows[:, 0]), rows[:, 0])
assert parallel.max_over_ranks(3.5, dev, force_collective=True) == 3.5   # ncclAllReduce(MAX)
acc.wait_for_everyone()
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL-OK", torch.cuda.get_device_name(0))
"""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(24000 + os.getpid() % 4000),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", code, ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL-OK" in p.stdout, p.stdout + p.stderr


def test_eval_cli_runs_full_width_with_forced_collectives():
    """``python train_gpt.py`` (seeded random weights, ivideogpt-oxe-64-act-free shapes, bf16): 2 batches of 4 clips, 2 samples each,
    gathers issued through RCCL (IVG_FORCE_COLLECTIVE=1, 1 rank)."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(25000 + os.getpid() % 4000),
               IVG_FORCE_COLLECTIVE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "train_gpt.py"), "--batch", "4", "--iters", "2", "--segment_length", "6",
                        "--eval_generate_times", "2", "--max_generate_batchsize", "4", "--max_decode_batchsize", "6"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout + p.stderr
    import json
    logs = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert math.isfinite(logs["eval/eval_loss"]) and 0 < logs["eval/ssim"] <= 1 and logs["eval/psnr"] > 0


@pytest.mark.parametrize("profile", ["default", "batches_in_flight"])
def test_two_engines_in_flight_equal_sequential_runs(profile):
    """bench.py --lanes keeps several batches in flight on one GPU: one engine set, host thread and HIP stream each.  libivg has no
    global mutable state on the data path (an engine handle is not thread-safe, two handles are independent): the frames and tokens
    of two pipelines running CONCURRENTLY must equal, bit for bit, those of the same pipelines run one after the other -- with the
    library's default switches and with the small decode-GEMM footprint bench.py sets while its lanes run
    (ivideogpt_amd.switches.BATCHES_IN_FLIGHT; the second engine of each pair is a replica() over the first one's weights in HBM)."""
    import contextlib
    import threading
    from ivideogpt_amd import CompressiveVQModel, LlamaForCausalLM, switches, weights as W
    from ivideogpt_amd.pipeline import predict_frames
    with (switches.override(**switches.BATCHES_IN_FLIGHT) if profile == "batches_in_flight" else contextlib.nullcontext()):
        _two_engines_in_flight(CompressiveVQModel, LlamaForCausalLM, W, predict_frames, threading, replica=profile == "batches_in_flight")


def _two_engines_in_flight(CompressiveVQModel, LlamaForCausalLM, W, predict_frames, threading, replica):
    tcfg = W.tokenizer_config(**TOK_CFG)
    tsd = W.random_tokenizer_state_dict(tcfg, 91, codebook_std=0.4)
    lsd = W.random_llama_state_dict(LLM_CFG, 92)
    ctx, F_ = 2, 4
    g = torch.Generator().manual_seed(93)
    sets = []
    for i in range(2):
        if replica and sets:
            tok, llm = sets[0][0].replica(), sets[0][1].replica()
        else:
            tok = CompressiveVQModel(tcfg, tsd, encode_dtype="fp32", decode_dtype="bf16").to(DEV)
            llm = LlamaForCausalLM(dict(LLM_CFG), lsd, dtype="bf16").to(DEV)
        px = torch.rand(6, ctx + F_, 3, 64, 64, generator=g).to(DEV)
        u = torch.rand(6, 17 * F_ - 1, generator=g).to(DEV)
        sets.append((tok, llm, px, u, torch.cuda.Stream(device=DEV)))

    def run(i, out):
        tok, llm, px, u, st = sets[i]
        with torch.cuda.stream(st):
            for _ in range(3):
                frames, tokens = predict_frames(tok, llm, px, ctx, F_, uniforms=u, return_tokens=True)
            st.synchronize()
        out[i] = (frames.clone(), tokens.clone())

    seq, par = {}, {}
    for i in range(2):
        run(i, seq)
    ths = [threading.Thread(target=run, args=(i, par)) for i in range(2)]
    [t.start() for t in ths]; [t.join() for t in ths]
    torch.cuda.synchronize()
    for i in range(2):
        assert torch.equal(par[i][1], seq[i][1]), f"lane {i}: tokens differ between concurrent and sequential runs"
        assert torch.equal(par[i][0], seq[i][0]), f"lane {i}: frames differ between concurrent and sequential runs"


def test_decode_lds_budget_is_a_property_of_the_engine(switches):
    """ivg_config.decode_lds_kb / set_decode_lds_kb: the LDS budget of the decode GEMMs belongs to ONE engine.  Two models in one
    process, one under 40 KiB (what bench.py's lanes set), one with the default: the first produces exactly what a process-wide
    IVG_DECODE_LDS_KB=40 produces and its q/k/v / gate-up / down GEMMs leave the third-generation kernel; the second is untouched."""
    from ivideogpt_amd import LlamaForCausalLM, _lib, weights as W
    l = _lib.load()
    cfg = dict(W.LLAMA_SMALL, num_hidden_layers=2)
    sd = W.random_llama_state_dict(cfg, 97)
    g = torch.Generator().manual_seed(98)
    prompt = torch.randint(0, cfg["vocab_size"], (8, 20), generator=g).to(DEV)

    def roll(model):
        n3, n2 = l.ivg_debug_counter(b"decode_gemm_gen3"), l.ivg_debug_counter(b"decode_gemm_gen2")
        out = model.generate(prompt, do_sample=False, max_new_tokens=12)
        torch.cuda.synchronize()
        return out, l.ivg_debug_counter(b"decode_gemm_gen3") - n3, l.ivg_debug_counter(b"decode_gemm_gen2") - n2

    switches(IVG_DECODE_LDS_KB=None)
    base = LlamaForCausalLM(cfg, sd, dtype="bf16").to(DEV)
    t_default, g3_default, _ = roll(base)
    assert g3_default > 0, "with a whole CU's LDS the small transformer's decode GEMMs run on the third-generation kernel"
    switches(IVG_DECODE_LDS_KB=40)
    t_proc40, g3_proc40, g2_proc40 = roll(LlamaForCausalLM(cfg, sd, dtype="bf16").to(DEV))
    assert g3_proc40 < g3_default and g2_proc40 > 0
    switches(IVG_DECODE_LDS_KB=None)
    small = base.replica().set_decode_lds_kb(LlamaForCausalLM.BATCHES_IN_FLIGHT_LDS_KB)      # policy set before the engine exists
    t_eng40, g3_eng40, g2_eng40 = roll(small)
    assert (g3_eng40, g2_eng40) == (g3_proc40, g2_proc40) and torch.equal(t_eng40, t_proc40)
    t_again, g3_again, _ = roll(base)                                                            # the neighbour keeps its own policy
    assert g3_again == g3_default and torch.equal(t_again, t_default)
    small.set_decode_lds_kb(0)                                                                   # and a live engine can be switched back
    t_back, g3_back, _ = roll(small)
    assert g3_back == g3_default and torch.equal(t_back, t_default)
    with pytest.raises(Exception):
        small.set_decode_lds_kb(7)


def test_bench_lanes_small_run_with_the_in_flight_roofline():
    """the profiled passes of bench.py at a small shape: `roofline_in_flight` (all lanes' decode bytes over the union of their rollout
    intervals) beside `roofline` (one batch alone)"""
    import json
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--lanes", "2", "--batch", "4", "--frames", "6", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-fp32-mode"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    r = d["roofline_in_flight"]
    assert r["lanes"] == 2 and r["achieved"] > 0 and 0 < r["frac"] < 1 and len(r["per_lane"]) == 2
    assert all(pl["rollout_interval_ms"][1] > pl["rollout_interval_ms"][0] for pl in r["per_lane"]) and r["rollout_phase_ms"] > 0
    assert d["roofline"]["frac"] > 0


def test_bench_two_lanes_small_run():
    """``bench.py --lanes 2`` end to end at a small shape: one JSON line with both figures (two batches in flight / one)."""
    # (custom shapes: no other_configs block; the fp32 / x3 modes are skipped explicitly)
    import json
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--lanes", "2", "--batch", "4", "--frames", "6", "--steps", "4", "--warmup", "1",
                        "--no-cpu-baseline", "--no-fp32-mode", "--no-profile"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["lanes"] == 2 and d["steps"] == 4 and d["value"] > 0 and d["single_lane"]["value"] > 0
    assert abs(d["value"] - 4 * 4 * 4 / (d["ms_per_step"] * 4e-3)) < 1e-6 * d["value"]      # frames of exactly 4 steps over the wall clock


def test_engine_runs_on_the_callers_stream_and_orders_a_switch_of_streams():
    """engine.Engine._On: under a stream the caller made current the launches go to THAT stream; a call on another stream than the
    previous call's waits for it (the engine's workspace belongs to one stream at a time).  Same ids on every route."""
    from ivideogpt_amd import CompressiveVQModel, weights as W
    tcfg = W.tokenizer_config(**TOK_CFG)
    tok = CompressiveVQModel(tcfg, W.random_tokenizer_state_dict(tcfg, 95, codebook_std=0.4), encode_dtype="fp32", decode_dtype="bf16").to(DEV)
    ctx = tok.context_length
    px = torch.rand(8, ctx + 5, 3, 64, 64, generator=torch.Generator().manual_seed(96)).to(DEV)
    ref = tok.tokenize(px, ctx)[0].clone()                       # legacy default stream -> the engine's dedicated stream
    torch.cuda.synchronize()
    A, B = torch.cuda.Stream(device=DEV), torch.cuda.Stream(device=DEV)
    got = []
    for s in (A, B, A, None, B):                                 # back-to-back on different streams, no synchronisation in between
        if s is None:
            got.append(tok.tokenize(px, ctx)[0])
        else:
            with torch.cuda.stream(s):
                got.append(tok.tokenize(px, ctx)[0])
                assert tok._engine._run == s
    torch.cuda.synchronize()
    for i, g in enumerate(got):
        assert torch.equal(g, ref), f"call {i}"


def test_bench_two_lanes_gathers_through_rccl():
    """The N-GPU run of ``bench.py`` issues its per-step all-gathers from the two lane threads of every rank, in ticket order
    (parallel.Turnstile).  What a 1-GPU lease can execute of that: a 1-rank ``nccl`` group with IVG_FORCE_COLLECTIVE=1 -- the collectives
    are really issued (RCCL), from two host threads with different current streams."""
    import json
    env = dict(os.environ, IVG_FORCE_COLLECTIVE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29633", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--lanes", "2", "--batch", "4", "--frames", "6", "--steps", "6",
                        "--warmup", "1", "--no-cpu-baseline", "--no-fp32-mode", "--no-profile"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["lanes"] == 2 and d["steps"] == 6 and d["value"] > 0


def test_replica_shares_the_weights_and_reproduces_the_original():
    """``replica()``: a second model object with its own engine over the SAME packed weights in HBM (what bench.py's second lane
    runs on) -- same tokens and frames as the original, bit for bit, and no second copy of the weights."""
    from ivideogpt_amd import CompressiveVQModel, LlamaForCausalLM, weights as W
    from ivideogpt_amd.pipeline import predict_frames
    tcfg = W.tokenizer_config(**TOK_CFG)
    tok = CompressiveVQModel(tcfg, W.random_tokenizer_state_dict(tcfg, 97, codebook_std=0.4), encode_dtype="fp32", decode_dtype="bf16").to(DEV)
    llm = LlamaForCausalLM(dict(LLM_CFG), W.random_llama_state_dict(LLM_CFG, 98), dtype="bf16").to(DEV)
    ctx, F_ = 2, 3
    g = torch.Generator().manual_seed(99)
    px = torch.rand(5, ctx + F_, 3, 64, 64, generator=g).to(DEV)
    u = torch.rand(5, 17 * F_ - 1, generator=g).to(DEV)
    f0, t0 = predict_frames(tok, llm, px, ctx, F_, uniforms=u, return_tokens=True)
    tok2, llm2 = tok.replica(), llm.replica()
    f1, t1 = predict_frames(tok2, llm2, px, ctx, F_, uniforms=u, return_tokens=True)
    torch.cuda.synchronize()
    assert torch.equal(t0, t1) and torch.equal(f0, f1)
    assert llm2._engine is not llm._engine and llm2._engine.tensors is llm._engine.tensors
    assert tok2._engine is not tok._engine and tok2._engine.tensors is tok._engine.tensors
