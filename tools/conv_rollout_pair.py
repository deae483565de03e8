"""A/B of the round-3 review's proposal, measured directly: one lane loops ONLY the convolution phase (detokenize of a 64-trajectory batch:
45 ms of MFMA-bound grids), another lane loops ONLY rollouts (146 ms of 14.7 k short HBM- / latency-bound launches), each on its own stream
and host thread.  Reported: iteration time of each loop alone and beside the other, for the uncapped conv3x3 (two workgroups per CU: all of
a CU's LDS and vector registers) and the occupancy-capped one (IVG_CONV_CAP=1: one workgroup per CU), with full-LDS and small-footprint
decode GEMMs.  Usage: python tools/conv_rollout_pair.py [seconds per measurement]"""
import os
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ivideogpt_amd import CompressiveVQModel, LlamaForCausalLM, switches, weights as W  # noqa: E402

T_MEAS = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
dev = torch.device("cuda:0")
tcfg = W.tokenizer_config(**W.CTX_VAE64)
tok = CompressiveVQModel(tcfg, W.random_tokenizer_state_dict(tcfg, 0, codebook_std=0.4), encode_dtype="fp32", decode_dtype="bf16").to(dev)
lcfg = dict(W.LLAMA_SMALL)
llm = LlamaForCausalLM(lcfg, W.random_llama_state_dict(lcfg, 0), dtype="bf16").to(dev)
B, ctx, F = 64, 2, 14
s_conv, s_roll = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
g = torch.Generator(device=dev).manual_seed(1)
px = torch.rand(B, ctx + F, 3, 64, 64, device=dev, generator=g).to(torch.bfloat16)
with torch.cuda.stream(s_conv):
    prompt = tok.encode_context(px, ctx)
    tokens = llm.generate(prompt, do_sample=True, top_k=100, max_new_tokens=17 * F - 1, generator=g)
    tok.detokenize(tokens, ctx, clamp=True)
torch.cuda.synchronize()
u = torch.rand(B, 17 * F - 1, device=dev)


def loop(kind, stop, out):
    st = s_conv if kind == "conv" else s_roll
    torch.cuda.set_device(dev)
    n, t0 = 0, None
    with torch.cuda.stream(st):
        while not stop.is_set():
            if kind == "conv":
                tok.detokenize(tokens, ctx, clamp=True)
            else:
                llm.generate(prompt, do_sample=True, top_k=100, max_new_tokens=17 * F - 1, uniforms=u)
            st.synchronize()             # one iteration in flight per lane: the iteration time is what is measured
            if t0 is None:
                t0 = time.perf_counter()  # (the first iteration is warm-up)
            else:
                n += 1
        out[kind] = (time.perf_counter() - t0) / max(n, 1) * 1e3


def measure(kinds):
    stop, out = threading.Event(), {}
    ths = [threading.Thread(target=loop, args=(k, stop, out)) for k in kinds]
    [t.start() for t in ths]
    time.sleep(T_MEAS)
    stop.set()
    [t.join() for t in ths]
    return out


for name, sw in (("uncapped conv, full-LDS decode GEMMs", {}),
                 ("uncapped conv, 40 KiB decode GEMMs", dict(IVG_DECODE_LDS_KB=40)),
                 ("capped conv (1 workgroup / CU), full-LDS decode GEMMs", dict(IVG_CONV_CAP=1)),
                 ("capped conv (1 workgroup / CU), 40 KiB decode GEMMs", dict(IVG_CONV_CAP=1, IVG_DECODE_LDS_KB=40))):
    with switches.override(**sw):
        a = measure(["conv"])["conv"]
        b = measure(["roll"])["roll"]
        both = measure(["conv", "roll"])
    # if the two loops shared nothing they would keep their own iteration times; if they time-slice the chip, 1/ta' + ... :
    util = a / both["conv"] + b / both["roll"]      # fraction of "alone" work rates achieved together (1.0 = pure time slicing, 2.0 = free overlap)
    print(f"{name}:\n   alone: decode {a:6.1f} ms, rollout {b:6.1f} ms | side by side: decode {both['conv']:6.1f} ms, rollout {both['roll']:6.1f} ms"
          f" | combined work rate {util:.2f} x of one lane", flush=True)
