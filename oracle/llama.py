"""CPU restatement of the autoregressive transformer stage.  TEST INFRASTRUCTURE.

Restates (fp32, plain torch) what the reference obtains from ``transformers==4.38.2``:

  * ``LlamaForCausalLM.forward``  -- called at inference/predict.py:64, train_gpt.py:181,
    ivideogpt/transformer/action_model.py:101  (RMSNorm eps 1e-6, RoPE theta 1e4 with rotate_half,
    MHA, SiLU-GLU MLP, no biases, untied lm_head)
  * ``GenerationMixin._sample``   -- temperature 1, TopKLogitsWarper(100), softmax, draw
    (SURVEY.md Appendix A.5).  ``torch.multinomial`` streams are not reproducible across devices
    even inside the reference, so the draw is restated as an explicit-uniform inverse CDF over the
    kept tokens in ascending-id order; ``uniforms=None`` means greedy (argmax, lowest id on ties).
  * ``HeadModelWithAction.generate``  ivideogpt/transformer/action_model.py:56-121  (per-frame
    action injection on the ``sdf`` slot, forced ``sdf`` after every 16 tokens, per-frame
    re-prefill -- the re-prefill is kept in ``generate_reference_algorithm`` because it is what
    the CPU baseline times).

Pinned by oracle/pin/pin_against_reference.py against HF ``LlamaForCausalLM`` (transformers 5.15
in the build image; same Llama maths as 4.38.2) and the reference's ``HeadModelWithAction``.
Weights are a plain dict in the HF key schema.
"""
import math
import torch
import torch.nn.functional as F


def rope_table(max_pos, head_dim, theta=10000.0):
    """cos/sin [max_pos, head_dim/2] in fp32, exactly as HF builds them (inv_freq fp32, outer product)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))
    freqs = torch.arange(max_pos, dtype=torch.float)[:, None] * inv_freq[None, :]
    return freqs.cos(), freqs.sin()


def rms_norm(x, w, eps):
    xf = x.float()
    return w * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).to(x.dtype)


def apply_rope(x, cos, sin):
    """x [B,H,L,D]; cos/sin [L, D/2].  q' = q*cos + rotate_half(q)*sin, rotate_half = cat(-x2, x1)."""
    d2 = x.shape[-1] // 2
    x1, x2 = x[..., :d2], x[..., d2:]
    return torch.cat([x1 * cos - x2 * sin, x2 * cos + x1 * sin], -1)


class LlamaRef:
    def __init__(self, weights, n_layers, n_heads, rms_eps=1e-6, theta=10000.0, max_pos=1024, prefix="model."):
        self.w, self.n_layers, self.n_heads, self.eps, self.p = weights, n_layers, n_heads, rms_eps, prefix
        self.hidden = weights[prefix + "embed_tokens.weight"].shape[1]
        self.head_dim = self.hidden // n_heads
        self.cos, self.sin = rope_table(max_pos, self.head_dim, theta)
        self.lm_head = weights[("lm_head.weight" if prefix == "model." else prefix[:-len("model.")] + "lm_head.weight")]

    def embed(self, ids):
        return F.embedding(ids, self.w[self.p + "embed_tokens.weight"])

    @torch.no_grad()
    def forward_embeds(self, x, past=None, return_hidden=False):
        """x [B,L,hidden]; past = list of (k,v) [B,H,Lp,D] or None -> (logits fp32 [B,L,V], new past)."""
        B, L, _ = x.shape
        p0 = 0 if past is None else past[0][0].shape[2]
        cos, sin = self.cos[p0:p0 + L], self.sin[p0:p0 + L]
        new_past = []
        for l in range(self.n_layers):
            pre = f"{self.p}layers.{l}."
            h = rms_norm(x, self.w[pre + "input_layernorm.weight"], self.eps)
            q = F.linear(h, self.w[pre + "self_attn.q_proj.weight"]).view(B, L, self.n_heads, -1).transpose(1, 2)
            k = F.linear(h, self.w[pre + "self_attn.k_proj.weight"]).view(B, L, self.n_heads, -1).transpose(1, 2)
            v = F.linear(h, self.w[pre + "self_attn.v_proj.weight"]).view(B, L, self.n_heads, -1).transpose(1, 2)
            q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
            if past is not None:
                k, v = torch.cat([past[l][0], k], 2), torch.cat([past[l][1], v], 2)
            new_past.append((k, v))
            s = (q @ k.transpose(-1, -2)) / math.sqrt(self.head_dim)
            Lk = k.shape[2]
            mask = torch.ones(L, Lk, dtype=torch.bool).tril(diagonal=Lk - L)
            s = s.masked_fill(~mask, float("-inf"))
            a = torch.softmax(s.float(), -1).to(x.dtype) @ v
            x = x + F.linear(a.transpose(1, 2).reshape(B, L, -1), self.w[pre + "self_attn.o_proj.weight"])
            h = rms_norm(x, self.w[pre + "post_attention_layernorm.weight"], self.eps)
            g = F.silu(F.linear(h, self.w[pre + "mlp.gate_proj.weight"])) * F.linear(h, self.w[pre + "mlp.up_proj.weight"])
            x = x + F.linear(g, self.w[pre + "mlp.down_proj.weight"])
        hid = rms_norm(x, self.w[self.p + "norm.weight"], self.eps)
        logits = F.linear(hid, self.lm_head).float()
        if return_hidden:
            return logits, new_past, hid
        return logits, new_past

    def logits(self, ids=None, embeds=None):
        """teacher-forced logits [B,L,V] fp32."""
        return self.forward_embeds(self.embed(ids) if embeds is None else embeds)[0]


def sample_from_logits(logits, top_k, u, temperature=1.0):
    """logits [B,V] fp32; u [B] in [0,1) or None (greedy).  Restates TemperatureLogitsWarper (``scores / temperature`` in fp32,
    applied first: HF builds the warper list temperature -> top-k) + TopKLogitsWarper + softmax +
    one draw: keep every token whose logit >= the k-th largest (ties at the threshold are all
    kept), p = softmax over the kept set, pick the first kept token, in ascending id order, whose
    cumulative probability exceeds u * sum(p)."""
    if u is None:
        return torch.argmax(logits, -1)
    if temperature != 1.0:
        logits = logits / torch.tensor(temperature, dtype=torch.float32)
    kth = torch.topk(logits, min(top_k, logits.shape[-1]), dim=-1).values[..., -1:]
    keep = logits >= kth
    m = logits.max(-1, keepdim=True).values
    e = torch.where(keep, torch.exp((logits - m).double()), torch.zeros((), dtype=torch.double))
    cdf = torch.cumsum(e, -1)
    target = u.double().view(-1, 1) * cdf[:, -1:]
    tok = (cdf > target).double().argmax(-1)
    return tok


@torch.no_grad()
def generate_cached(model, input_ids, n_new, top_k=100, uniforms=None, action_embeds=None, ctx=None,
                    tokens_per_dyn=16, sdf_token=None, return_last_hidden=False, temperature=1.0):
    """One prefill + cached single-token steps (the engine's algorithm; SURVEY.md 3.3 shows it is
    token-identical to the reference's per-frame re-prefill).

    action-free  (action_embeds None): every new token is sampled           (inference/predict.py:57-69)
    action-cond  : new token j (1-based) is the forced ``sdf`` when j % 17 == 0; the embedding of
                   the i-th ``sdf`` slot (i = 0 is the prompt's last token) gets action_embeds[:, i+ctx-1].
    uniforms [B, n_new] or None (greedy).  Returns ids [B, L0+n_new] (prompt included)."""
    B, L0 = input_ids.shape
    x = model.embed(input_ids)
    per = tokens_per_dyn + 1
    slot0 = 0
    if action_embeds is not None:
        # every sdf slot already inside the prompt carries its action (slot i at position 257*ctx - 1 + 17*i reads row
        # i + ctx - 1): one slot for a plain 257*ctx prompt (action_model.py:80-81); t + 1 slots for the step-wise MBRL
        # rollout, whose embeddings keep the actions added at earlier steps (mbrl/video_predictor.py:295-296, 315-316)
        x = x.clone()
        slot0 = (L0 - 257 * ctx) // per
        for i in range(slot0 + 1):
            x[:, 257 * ctx - 1 + per * i] += action_embeds[:, i + ctx - 1]
    out_h = model.forward_embeds(x, return_hidden=True)
    logits, past, hid = out_h
    out = [input_ids]
    last = logits[:, -1]
    last_hidden = hid[:, -1]
    for j in range(1, n_new + 1):
        forced = action_embeds is not None and j % per == 0
        if forced:
            tok = torch.full((B,), sdf_token, dtype=input_ids.dtype)
        else:
            tok = sample_from_logits(last, top_k, None if uniforms is None else uniforms[:, j - 1], temperature).to(input_ids.dtype)
        out.append(tok[:, None])
        if j == n_new:
            break
        e = model.embed(tok[:, None])
        if forced:
            e = e + action_embeds[:, slot0 + j // per + ctx - 1][:, None]
        logits, past, hid = model.forward_embeds(e, past, return_hidden=True)
        last = logits[:, -1]
        last_hidden = hid[:, -1]
    ids = torch.cat(out, 1)
    return (ids, last_hidden) if return_last_hidden else ids


@torch.no_grad()
def generate_reference_algorithm(model, input_ids, n_new, top_k=100, uniforms=None, action_embeds=None,
                                 ctx=None, tokens_per_dyn=16, sdf_token=None):
    """The reference's op sequence (what the CPU baseline times).  Action-free: one prefill + cached
    steps (HF generate).  Action-conditioned: for every future frame re-prefill the whole prefix,
    then 15 cached steps, then append the forced ``sdf``  (action_model.py:78-114)."""
    if action_embeds is None:
        return generate_cached(model, input_ids, n_new, top_k, uniforms)
    B = input_ids.shape[0]
    per = tokens_per_dyn + 1
    n_frames = (n_new + 1) // per
    embeds = model.embed(input_ids).clone()
    tokens = input_ids
    prelude = input_ids.shape[1] - 1
    ucol = 0
    for i in range(n_frames):
        embeds[:, prelude + i * per] += action_embeds[:, i + ctx - 1]
        logits, past = model.forward_embeds(embeds)
        new = []
        for s in range(tokens_per_dyn):
            u = None if uniforms is None else uniforms[:, ucol]
            tok = sample_from_logits(logits[:, -1], top_k, u).to(input_ids.dtype)
            ucol += 1
            new.append(tok[:, None])
            if s + 1 < tokens_per_dyn:
                logits, past = model.forward_embeds(model.embed(tok[:, None]), past)
        ucol += 1  # the uniform column of the forced sdf slot is unused
        new.append(torch.full((B, 1), sdf_token, dtype=input_ids.dtype))
        new = torch.cat(new, 1)
        embeds = torch.cat([embeds, model.embed(new)], 1)
        tokens = torch.cat([tokens, new], 1)
    return tokens[:, :-1]


@torch.no_grad()
def eval_forward(model, input_ids, labels, action_embeds=None, ctx=None, n_future=None, tokens_per_dyn=16):
    """Teacher-forced eval forward with labels: ``LlamaForCausalLM.forward(labels=...)`` (HF shifted cross-entropy, ignore_index
    -100, mean over the batch's valid targets) and ``HeadModelWithAction.forward`` (action embeddings
    ``action_embeds[:, ctx-1:-1]`` added on the sdf slots, ivideogpt/transformer/action_model.py:170-178).
    -> dict(loss, token_nll [B, L] (0 where ignored), sample_loss [B], hidden [B, L, H] post final norm)."""
    x = model.embed(input_ids)
    if action_embeds is not None:
        x = x.clone()
        start = (257 * ctx - 1) + torch.arange(n_future) * (tokens_per_dyn + 1)
        x[:, start] += action_embeds[:, ctx - 1:-1]
    logits, _, hid = model.forward_embeds(x, return_hidden=True)
    B, L, V = logits.shape
    tgt = labels[:, 1:]
    nll = F.cross_entropy(logits[:, :-1].reshape(-1, V), tgt.reshape(-1), ignore_index=-100, reduction="none").view(B, L - 1)
    valid = (tgt != -100)
    token_nll = torch.cat([nll * valid, torch.zeros(B, 1)], 1)
    counts = valid.sum(1).clamp_min(1)
    return dict(loss=token_nll.sum() / valid.sum().clamp_min(1), token_nll=token_nll, sample_loss=token_nll.sum(1) / counts, hidden=hid)
