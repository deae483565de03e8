"""How long does a host thread need to ENQUEUE a rollout (ivg_generate issues ~14.7 k launches eagerly and returns without a sync), alone
and beside 1 / 3 other threads doing the same -- against the time the device needs for it?  Separates a launch-rate bound from a
device bound in the batches-in-flight mode (development aid).   python tools/host_enqueue.py [batch=64]"""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ivideogpt_amd import LlamaForCausalLM, weights as W  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dev = torch.device("cuda:0")
    base = LlamaForCausalLM(W.LLAMA_SMALL, W.random_llama_state_dict(W.LLAMA_SMALL, 0), dtype="bf16").to(dev)
    for lanes in (1, 2, 4):
        models = [base] + [base.replica() for _ in range(lanes - 1)]
        for m in models:
            m.set_decode_lds_kb(40 if lanes > 1 else 0)
        streams = [torch.cuda.Stream(device=dev) for _ in range(lanes)]
        prompts = [torch.randint(0, 16384, (B, 514), device=dev) for _ in range(lanes)]
        res = [None] * lanes

        def body(i, rec):
            torch.cuda.set_device(dev)
            with torch.cuda.stream(streams[i]):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                t0 = time.perf_counter()
                models[i].generate(prompts[i], do_sample=True, top_k=100, max_new_tokens=237)
                host = (time.perf_counter() - t0) * 1e3
                e1.record()
                streams[i].synchronize()
                if rec:
                    res[i] = (host, e0.elapsed_time(e1))

        for rec in (False, True):
            ths = [threading.Thread(target=body, args=(i, rec)) for i in range(lanes)]
            [t.start() for t in ths]
            [t.join() for t in ths]
        print(f"B={B} lanes={lanes}: " + " | ".join(f"host enqueue {h:6.1f} ms, device {d:6.1f} ms" for h, d in res), flush=True)
        del models[1:]


if __name__ == "__main__":
    main()
