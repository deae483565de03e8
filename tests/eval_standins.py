"""CPU stand-ins with the reference's object API, built on the oracle: what ``train_gpt.evaluate`` is checked against
(tests only -- the product never imports the oracle)."""
from types import SimpleNamespace

import torch

from oracle import metrics as OM
from oracle.llama import eval_forward, generate_cached


class OracleLM:
    """``model(input_ids=, labels=).loss`` and ``model.generate(...)`` (HF convention: prompt included) on ``oracle.llama.LlamaRef``.
    ``uniforms``: callable (B, n) -> float32 [B, n] giving the draws of the next ``generate`` call (explicit-uniform inverse-CDF
    sampling, the engine's documented sampler)."""

    def __init__(self, llama_ref, uniforms):
        self.m, self.uniforms = llama_ref, uniforms

    def __call__(self, input_ids=None, labels=None, **unused):
        r = eval_forward(self.m, input_ids.cpu(), labels.cpu())
        return SimpleNamespace(loss=r["loss"], sample_loss=r["sample_loss"])

    def generate(self, input_ids, do_sample=True, temperature=1.0, top_k=100, max_new_tokens=None, pad_token_id=None, **unused):
        assert do_sample and temperature == 1.0
        ids = input_ids.cpu()
        return generate_cached(self.m, ids, max_new_tokens, top_k=top_k, uniforms=self.uniforms(ids.shape[0], max_new_tokens))


class OracleEvaluator:
    """``Evaluator.forward`` (ivideogpt/utils/video_metric.py:63-100) without LPIPS: (mse, psnr, ssim, nan)."""

    def __call__(self, video_1, video_2):
        m = OM.frame_metric_rows(video_1.cpu(), video_2.cpu()).mean(0)
        return m[0], m[1], m[2], torch.tensor(float("nan"))
