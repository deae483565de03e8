"""Step-wise MBRL imagination (mbrl/video_predictor.py: VideoPredictor.rollout) on full-width models: environment steps / s
with the KV cache kept across steps vs re-prefilling the grown prompt every step (what the reference's generate does).
Usage: python tools/mbrl_bench.py [batch] [horizon]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ivideogpt_amd import CompressiveVQModel, HeadModelWithAction, LlamaForCausalLM, weights as W  # noqa: E402
from mbrl.video_predictor import VideoPredictor  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
horizon = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = "cuda:0"
tcfg = W.tokenizer_config(**W.CTX_VAE64)
tok = CompressiveVQModel(tcfg, W.random_tokenizer_state_dict(tcfg, 1, codebook_std=0.4), encode_dtype="fp32", decode_dtype="bf16").to(dev)
lcfg = dict(W.LLAMA_SMALL)
head = HeadModelWithAction(LlamaForCausalLM(lcfg, None, dtype="bf16"), 4, 513, 16, 2, 2 + horizon, reward_prediction=True)
head.load_state_dict(W.random_llama_state_dict(lcfg, 2, action_dim=4, reward_prediction=True), strict=True)
head.to(dev)
obs = torch.randint(0, 256, (B, 9, 64, 64)).float()
policy = lambda o, t: torch.zeros(B, 4)  # noqa: E731
for reuse in (False, True):
    vp = VideoPredictor.from_models(tok, head, context_length=2, reuse_cache=reuse)
    for _ in range(2):
        vp.rollout(obs, policy, horizon)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        vp.rollout(obs, policy, horizon)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"reuse_cache={reuse}: B={B} horizon={horizon}: {dt * 1e3:.1f} ms per rollout, {B * horizon / dt:.0f} imagined steps/s", flush=True)
