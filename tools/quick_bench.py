"""Quick stage timing at BASELINE config-2 shapes (development aid; bench.py is the contract)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ivideogpt_amd import CompressiveVQModel, LlamaForCausalLM, weights as W  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--res", type=int, default=64)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--dec", default="bf16")
    ap.add_argument("--llm", default="bf16")
    ap.add_argument("--enc", default="fp32")
    ap.add_argument("--decode-only", action="store_true", help="time detokenize alone on random valid token ids (no rollout)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    t0 = time.time()
    tcfg = W.tokenizer_config(**(W.CTX_VAE64 if a.res == 64 else dict(W.CTX_VAE256, resolution=256, max_att_resolution=32)))
    tok = CompressiveVQModel(tcfg, W.random_tokenizer_state_dict(tcfg, 0, 0.4), encode_dtype=a.enc, decode_dtype=a.dec).to(dev)
    llm = LlamaForCausalLM(W.LLAMA_SMALL, W.random_llama_state_dict(W.LLAMA_SMALL, 0), dtype=a.llm).to(dev)
    print(f"weights ready in {time.time() - t0:.1f}s", flush=True)
    ctx, B, T = tcfg["context_length"], a.batch, a.frames
    F = T - ctx
    px = torch.rand(B, T, 3, a.res, a.res, device=dev).to(torch.bfloat16)
    n_new = 17 * F - 1
    res = {}
    if a.decode_only:
        g = torch.Generator().manual_seed(1)
        ids = torch.randint(0, 8192, (B, 257 * ctx - 1 + 17 * F), generator=g)
        ids[:, 257 * ctx:] += 8192
        for f in range(ctx):
            ids[:, 257 * f + 256 if f < ctx - 1 else 257 * ctx - 1] = 16384 if f < ctx - 1 else 16385
        for f in range(F - 1):
            ids[:, 257 * ctx + 17 * f + 16] = 16385
        ids = ids.to(dev)
        ts = []
        for it in range(a.iters + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            frames = tok.detokenize(ids, ctx)
            e1.record()
            torch.cuda.synchronize()
            if it:
                ts.append(e0.elapsed_time(e1))
        print(json.dumps(dict(decode_ms=sorted(ts)[len(ts) // 2], decode_ms_all=[round(t, 2) for t in ts], batch=B, frames=T, res=a.res, dec=a.dec,
                              finite=bool(torch.isfinite(frames.float()).all()))))
        return
    for it in range(a.iters + 1):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        torch.cuda.synchronize()
        ev[0].record()
        prompt = tok.encode_context(px, ctx)
        ev[1].record()
        ids = llm.generate(prompt, do_sample=True, top_k=100, max_new_tokens=n_new)
        ev[2].record()
        frames = tok.detokenize(ids, ctx).clamp_(0, 1)
        ev[3].record()
        torch.cuda.synchronize()
        if it == 0:
            continue  # warm-up (engine creation, graph capture)
        for k, (i, j) in {"encode_ms": (0, 1), "generate_ms": (1, 2), "decode_ms": (2, 3), "total_ms": (0, 3)}.items():
            res.setdefault(k, []).append(ev[i].elapsed_time(ev[j]))
    out = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    out["pred_frames_per_s"] = B * F / (out["total_ms"] / 1e3)
    out.update(batch=B, frames=T, res=a.res, dec=a.dec, llm=a.llm, enc=a.enc, finite=bool(torch.isfinite(frames).all()))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
