"""``train_gpt.evaluate`` (mirror of /root/reference/train_gpt.py:152-195,321-512) on CPU stand-ins: the host logic -- full-clip
tokenize, loss gather, t x B repeated generation chunked by ``max_generate_batchsize``, chunked detokenize, best-of-t metrics,
gathers -- against a direct restatement, and a world_size-2 gloo run whose gathered logs equal the single-process run."""
import json
import math
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

WORKER = r"""
import json, os, sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
torch.set_num_threads(2)
from ivideogpt_amd import parallel, weights as W
from helpers import oracle_llama, oracle_tokenizer
from eval_standins import OracleEvaluator, OracleLM
import train_gpt

rank, world, local = parallel.init_from_env("gloo")
tcfg = W.tokenizer_config(block_out_channels=(32, 32, 64), layers_per_block=1, latent_channels=64, num_vq_embeddings=256, num_dyn_embeddings=256,
                          norm_num_groups=32, mid_block_add_attention=False, context_length=2, resolution=64, max_att_resolution=16)
lcfg = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=1, num_key_value_heads=1, rms_norm_eps=1e-6,
            rope_theta=10000.0, max_position_embeddings=1024, vocab_size=514)
tok = oracle_tokenizer(tcfg, W.random_tokenizer_state_dict(tcfg, 5, codebook_std=0.5), 2)
llm = oracle_llama(lcfg, W.random_llama_state_dict(lcfg, 6))
g = torch.Generator().manual_seed(7)
batches = [torch.rand(2, 4, 3, 64, 64, generator=g) for _ in range(4)]

def uniforms_for(prompt_sum_box):
    # draws keyed by the prompt content and the repeat index of that prompt: identical no matter which rank / order evaluates the batch
    seen = {}
    def u(B, n):
        key = prompt_sum_box[0]
        k = seen.get(key, 0); seen[key] = k + 1
        return torch.rand(B, n, generator=torch.Generator().manual_seed(key * 16 + k))
    return u

class KeyedLM(OracleLM):
    def generate(self, input_ids, **kw):
        self.box[0] = int(input_ids[0].sum()) % 100003
        return super().generate(input_ids, **kw)

box = [0]
model = KeyedLM(llm, uniforms_for(box)); model.box = box
args = train_gpt.eval_args(context_length=2, segment_length=4, eval_generate_times=4, max_generate_batchsize=4, max_decode_batchsize=3,
                           max_eval_iters=100, log_gif_interval=1000)
mine = batches[rank::world]
acc = parallel.LocalAccelerator("cpu")
logs = train_gpt.evaluate(args, acc, tok, model, mine, OracleEvaluator(), 0)
if rank == 0:
    print("LOGS " + json.dumps(logs))
parallel.barrier()
"""


def run_world(tmp_path, world):
    script = tmp_path / f"w{world}.py"
    script.write_text(WORKER)
    port = 23000 + (os.getpid() * 7 + world) % 4000
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    line = [l for l in outs[0].splitlines() if l.startswith("LOGS ")][0]
    return json.loads(line[5:])


def test_evaluate_world2_gloo_equals_single_process(tmp_path):
    one = run_world(tmp_path, 1)
    two = run_world(tmp_path, 2)
    assert set(one) == {"eval/eval_loss", "eval/perplexity", "eval/mse", "eval/psnr", "eval/ssim", "eval/lpips"}
    for k in one:
        if k == "eval/lpips":
            assert math.isnan(one[k]) and math.isnan(two[k])     # LPIPS weights do not ship: explicit NaN, never a silent 0
        else:
            assert abs(one[k] - two[k]) <= 1e-6 * max(1.0, abs(one[k])), (k, one[k], two[k])
    assert math.isfinite(one["eval/eval_loss"]) and abs(one["eval/perplexity"] - math.exp(one["eval/eval_loss"])) < 1e-6 * one["eval/perplexity"]


def test_generate_multiple_times_layout_and_chunking():
    """t samples per prompt: row k * B + b is sample k of trajectory b; chunked by max_batch_size (train_gpt.py:152-191)."""
    import train_gpt
    from ivideogpt_amd.parallel import LocalAccelerator

    class Echo:
        calls = []

        def generate(self, ids, max_new_tokens=None, pad_token_id=None, action=None, **kw):
            Echo.calls.append((ids.shape[0], None if action is None else action.shape[0]))
            tag = torch.full((ids.shape[0], max_new_tokens), len(Echo.calls), dtype=ids.dtype)
            return torch.cat([ids, tag], 1)

    prompt = torch.arange(3)[:, None].repeat(1, 5)
    act = torch.zeros(3, 4, 2)
    out = train_gpt.generate_multiple_times(4, LocalAccelerator("cpu"), Echo(), prompt, act, {"max_new_tokens": 2}, max_batch_size=6)
    assert Echo.calls == [(6, 6), (6, 6)] and out.shape == (12, 7)
    assert torch.equal(out[:, 0], torch.arange(3).repeat(4)) and out[:6, -1].eq(1).all() and out[6:, -1].eq(2).all()
    parts = train_gpt.batch_forward(5, out, lambda x: x[:, :1] * 2)
    assert torch.equal(parts[:, 0], out[:, 0] * 2)
