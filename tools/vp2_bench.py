"""VP2-shaped call (vp/ivideogpt_interface.py:155-202): N candidate action sequences over ONE two-frame context, generate in chunks of
generate_max_batchsize, decode in chunks of decode_max_batchsize -- timed with and without the shared-context path (development aid;
bench.py's `shared_context.vp2` entry is the contract).   python tools/vp2_bench.py [candidates=200] [iters=4]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ivideogpt_amd import CompressiveVQModel, HeadModelWithAction, LlamaForCausalLM, weights as W  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    gmax, dmax, T, ctx, adim = 100, 67, 12, 2, 5
    dev = torch.device("cuda:0")
    tcfg = W.tokenizer_config(**W.CTX_VAE64)
    tok = CompressiveVQModel(tcfg, W.random_tokenizer_state_dict(tcfg, 0, 0.4), encode_dtype="fp32", decode_dtype="bf16").to(dev)
    lsd = W.random_llama_state_dict(W.LLAMA_SMALL, seed=0, action_dim=adim)
    model = HeadModelWithAction(LlamaForCausalLM(W.LLAMA_SMALL, None, dtype="bf16"), adim, 257 * ctx - 1, 16, ctx, T)
    model.load_state_dict(lsd, strict=True)
    model.to(dev)
    g = torch.Generator(device=dev).manual_seed(1)
    clip = torch.rand(1, T, 3, 64, 64, device=dev, generator=g).to(torch.bfloat16)
    acts = torch.randn(n, T, adim, device=dev, generator=g)
    n_new = 17 * (T - ctx) - 1
    eng = model.llm

    def flow(share):
        outs = []
        for s0 in range(0, n, gmax):
            k = min(gmax, n - s0)
            prompt = tok.encode_context(clip, ctx).repeat(k, 1) if share else tok.encode_context(clip.expand(k, -1, -1, -1, -1).contiguous(), ctx)
            toks = model.generate(prompt, do_sample=True, top_k=100, max_new_tokens=n_new, generator=g, action=acts[s0:s0 + k],
                                  shared_context=k if share and k > 1 else None)
            for d0 in range(0, k, dmax):
                ch = toks[d0:d0 + dmax]
                outs.append(tok.detokenize(ch, ctx, clamp=True, shared_context=ch.shape[0] if share and ch.shape[0] > 1 else None))
        return outs
    res = {}
    for mode, share in (("shared_context", True), ("plain", False)):
        flow(share)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            outs = flow(share)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / iters
        # decode attention alone (launch-window stamps of the last generate chunk)
        e = eng._engine
        e.profile_enable(4, True)
        flow(share)
        st = e.profile_read(4)
        e.profile_enable(4, False)
        res[mode] = {"ms_per_call": round(el * 1e3, 2), "frames_per_s": round(n * (T - ctx) / el, 1),
                     "decode_attn_mean_launch_us": round(1e3 * st["total_ms"] / max(1, st["launches"]), 2), "finite": bool(all(torch.isfinite(o).all() for o in outs))}
    res["candidates"] = n
    print(json.dumps(res))


if __name__ == "__main__":
    main()
