"""Shared helpers for the parity tests: rebuild the seeded weights a golden fixture was made with."""
import json
import os

import numpy as np
import torch

from ivideogpt_amd import weights as W

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name))
    return {k: d[k] for k in d.files}


def tokenizer_fixture(name):
    """-> (cfg, state dict, ctx, pixels [B,T,3,H,W] fp32, golden arrays)"""
    g = load_golden(name)
    cfg = W.tokenizer_config(**json.loads(str(g["config"])))
    sd = W.random_tokenizer_state_dict(cfg, int(g["seed"]), float(g["codebook_std"]))
    px = torch.from_numpy(g["pixels_u8"]).float() / 255.0
    return cfg, sd, int(g["context_length"]), px, g


def oracle_tokenizer(cfg, sd, ctx):
    from oracle.vq_tokenizer import CompressiveVQRef
    m = CompressiveVQRef(**cfg).eval()
    m.load_state_dict(sd, strict=True)
    if ctx != cfg["context_length"]:
        m.set_context_length(ctx)
    return m


def llama_fixture(name):
    g = load_golden(name)
    cfg = json.loads(str(g["config"]))
    adim = int(g["action_dim"]) if "action_dim" in g else None
    sd = W.random_llama_state_dict(cfg, int(g["seed"]), action_dim=adim)
    return cfg, sd, g


def oracle_llama(cfg, sd, prefix="model."):
    from oracle.llama import LlamaRef
    return LlamaRef(sd, cfg["num_hidden_layers"], cfg["num_attention_heads"], cfg["rms_norm_eps"],
                    cfg["rope_theta"], cfg["max_position_embeddings"], prefix=prefix)
