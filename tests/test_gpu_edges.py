"""Edge cases of the prediction path on the MI355X (-m gpu): shortest / longest rollouts, ragged and single-row batches,
input dtypes and layouts, context-length switching, error behaviour (the reference asserts; so does the mirror)."""
import numpy as np
import pytest
import torch

from helpers import llama_fixture, oracle_llama, oracle_tokenizer, tokenizer_fixture

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def make_tok(cfg, sd, ctx, enc="fp32", dec="fp32"):
    from ivideogpt_amd import CompressiveVQModel
    m = CompressiveVQModel(cfg, sd, encode_dtype=enc, decode_dtype=dec).to(DEV)
    if ctx != cfg["context_length"]:
        m.set_context_length(ctx)
    return m


def make_llm(cfg, sd, dtype="fp32"):
    from ivideogpt_amd import LlamaForCausalLM
    return LlamaForCausalLM(cfg, sd, dtype=dtype).to(DEV)


@pytest.mark.parametrize("n_new", [1, 2, 3, 17, 18])
def test_shortest_rollouts_match_oracle(n_new):
    """1 new token = prefill + one decision (no cached step); 2 = one eager step; >= 3 enters the replayed step graph."""
    from oracle.llama import generate_cached
    cfg, sd, g = llama_fixture("llama_tiny_ctx1_free.npz")
    m, ora = make_llm(cfg, sd), oracle_llama(cfg, sd)
    prompt = torch.from_numpy(g["prompt"])
    out = m.generate(prompt.to(DEV), do_sample=False, max_new_tokens=n_new).cpu()
    assert torch.equal(out, generate_cached(ora, prompt, n_new, uniforms=None))
    u = torch.rand(prompt.shape[0], n_new, generator=torch.Generator().manual_seed(n_new))
    out = m.generate(prompt.to(DEV), do_sample=True, top_k=100, max_new_tokens=n_new, uniforms=u.to(DEV)).cpu()
    assert torch.equal(out, generate_cached(ora, prompt, n_new, top_k=100, uniforms=u))


def test_rollout_to_the_last_position_and_beyond():
    """max_position_embeddings = 1024: a rollout may fill the KV cache to the last slot; one more token is an error, not a
    silent overrun."""
    from oracle.llama import generate_cached
    cfg, sd, g = llama_fixture("llama_tiny_ctx1_free.npz")
    m = make_llm(cfg, sd)
    Lmax = cfg["max_position_embeddings"]
    prompt = torch.from_numpy(g["prompt"])[:2]
    L0 = prompt.shape[1]
    out = m.generate(prompt.to(DEV), do_sample=False, max_new_tokens=Lmax - L0).cpu()
    assert out.shape == (2, Lmax)
    ref = generate_cached(oracle_llama(cfg, sd), prompt[:1], Lmax - L0, uniforms=None)
    assert torch.equal(out[:1], ref)
    with pytest.raises((AssertionError, RuntimeError)):
        m.generate(prompt.to(DEV), do_sample=False, max_new_tokens=Lmax - L0 + 2)


def test_long_teacher_forced_sequence_matches_oracle():
    """logits over a full 1024-token sequence (every RoPE position, the longest causal attention)."""
    cfg, sd, g = llama_fixture("llama_tiny_ctx1_free.npz")
    ids = torch.randint(0, cfg["vocab_size"], (1, cfg["max_position_embeddings"]), generator=torch.Generator().manual_seed(5))
    ref = oracle_llama(cfg, sd).logits(ids)
    lg = make_llm(cfg, sd).logits(ids.to(DEV)).cpu()
    err = (lg - ref).abs().max().item()
    assert err < 1e-3, f"max abs err {err:.2e}"


@pytest.mark.parametrize("B", [1, 3])
def test_single_and_odd_batches_match_reference_vectors(B):
    cfg, sd, ctx, px, g = tokenizer_fixture("tok_mini64_ctx2.npz")
    m = make_tok(cfg, sd, ctx)
    rows = [0] if B == 1 else [1, 0, 1]
    ids, labels = m.tokenize(px[rows].to(DEV), ctx)
    assert np.array_equal(ids.cpu().numpy(), g["indices"][rows]) and np.array_equal(labels.cpu().numpy(), g["labels"][rows])
    rec = m.detokenize(ids, ctx).cpu().numpy()
    assert np.abs(rec - g["recon"][rows]).max() < 1e-3


def test_pixel_dtypes_and_layouts():
    """bf16 pixels are read as such (the benchmark keeps clips in bf16): same tokens as the rounded values in fp32;
    uint8-derived, non-contiguous and CPU tensors are accepted like any torch module would."""
    cfg, sd, ctx, px, g = tokenizer_fixture("tok_mini64_ctx2.npz")
    m = make_tok(cfg, sd, ctx)
    base, _ = m.tokenize(px.to(DEV), ctx)
    p16 = px.to(torch.bfloat16)
    a, _ = m.tokenize(p16.to(DEV), ctx)
    b, _ = m.tokenize(p16.float().to(DEV), ctx)
    assert torch.equal(a, b)
    wide = torch.zeros(px.shape[0], px.shape[1], 3, 64, 128)
    wide[..., ::2] = px
    c, _ = m.tokenize(wide[..., ::2].to(DEV), ctx)        # strided view
    d, _ = m.tokenize(px, ctx)                            # host tensor: moved by the mirror
    e, _ = m.tokenize(px.double().to(DEV), ctx)           # other float types are converted to fp32
    assert torch.equal(c, base) and torch.equal(d, base) and torch.equal(e, base)


def test_context_length_switching_is_stateless():
    """set_context_length (compressive_vq_model.py:154-158) re-slices the position embeddings; going 2 -> 1 -> 2 gives the
    first answer again and the ctx = 1 answer equals the oracle switched the same way."""
    cfg, sd, ctx, px, g = tokenizer_fixture("tok_mini64_ctx2.npz")
    m = make_tok(cfg, sd, ctx)
    first, _ = m.tokenize(px.to(DEV), 2)
    m.set_context_length(1)
    one, _ = m.tokenize(px.to(DEV), 1)
    want, _ = oracle_tokenizer(cfg, sd, 1).tokenize(px, 1)
    assert torch.equal(one.cpu(), want)
    with pytest.raises(AssertionError):
        m.tokenize(px.to(DEV), 2)                          # stale context_length (:166)
    m.set_context_length(2)
    again, _ = m.tokenize(px.to(DEV), 2)
    assert torch.equal(first, again)


def test_error_behaviour():
    cfg, sd, ctx, px, g = tokenizer_fixture("tok_mini64_ctx2.npz")
    m = make_tok(cfg, sd, ctx)
    with pytest.raises(AssertionError):
        m.tokenize(px[:, :ctx].to(DEV), ctx)               # no future frame: the reference cannot reshape an empty batch either
    with pytest.raises(AssertionError):
        m.tokenize(px[:, :, :, :32].to(DEV), ctx)          # wrong resolution
    with pytest.raises(AssertionError):
        m.tokenize(px[:, :, :2].to(DEV), ctx)              # not RGB
    ids, _ = m.tokenize(px.to(DEV), ctx)
    with pytest.raises(AssertionError):
        m.detokenize(ids[:, :-1], ctx)                     # (L + 1 - 257 ctx) % 17 != 0 (:230)
    rec, cache = m.detokenize(ids, ctx, return_cache=True)
    with pytest.raises(AssertionError):
        m.detokenize(ids[:1], ctx, cache=cache)            # cache of another batch size
    assert torch.equal(m.encode_context(px[:, :ctx].to(DEV), ctx), ids[:, :257 * ctx])   # context only needs the context frames
    # the engine stays usable after every rejected call
    assert torch.equal(m.detokenize(ids, ctx), rec)
    cfg2, sd2, g2 = llama_fixture("llama_tiny_ctx1_free.npz")
    llm = make_llm(cfg2, sd2)
    bad = torch.full((1, 257), cfg2["vocab_size"] + 7, dtype=torch.int64)
    out = llm.generate(bad.to(DEV), do_sample=False, max_new_tokens=4)   # ids outside the vocabulary: no out-of-bounds gather
    torch.cuda.synchronize()
    assert (out[:, 257:] >= 0).all() and (out[:, 257:] < cfg2["vocab_size"]).all()
    with pytest.raises((AssertionError, RuntimeError)):
        llm.generate(torch.zeros(1, 1020, dtype=torch.int64).to(DEV), do_sample=False, max_new_tokens=10)   # past the KV cache


def test_out_of_range_dynamics_tokens_are_clamped_like_the_reference():
    """a rollout may emit any id < vocab in a dynamics slot; detokenize maps it with (id - n_vq).clamp(0, n_dyn - 1)
    (compressive_vq_model.py:236-237)."""
    cfg, sd, ctx, px, g = tokenizer_fixture("tok_mini64_ctx2.npz")
    m, ora = make_tok(cfg, sd, ctx), oracle_tokenizer(cfg, sd, ctx)
    ids = torch.from_numpy(g["indices"]).clone()
    start = 257 * ctx
    ids[:, start + 0] = 3                                            # a context code in a dynamics slot -> clamps to 0
    ids[:, start + 5] = cfg["num_vq_embeddings"] + cfg["num_dyn_embeddings"] + 1   # separator id -> clamps to n_dyn - 1
    want = ora.detokenize(ids, ctx)
    got = m.detokenize(ids.to(DEV), ctx).cpu()
    assert (got - want).abs().max().item() < 1e-3


def test_tokenize_many_clips_bit_exact_vs_oracle():
    """24 more clips through the mini tokenizer (13,128 token ids): uniform noise, smooth ramps, constant and saturated frames,
    zero-padded futures (the reference's callers pad with zero frames: vp/ivideogpt_interface.py:158-169) -- every id equal
    to the CPU oracle's, and the decoded frames within 1e-3."""
    cfg, sd, ctx, _, _ = tokenizer_fixture("tok_mini64_ctx2.npz")
    m, ora = make_tok(cfg, sd, ctx), oracle_tokenizer(cfg, sd, ctx)
    g = torch.Generator().manual_seed(77)
    T = 4
    clips = []
    clips.append(torch.rand(6, T, 3, 64, 64, generator=g))                                   # noise
    ramp = torch.linspace(0, 1, 64)[None, :].expand(64, 64)
    smooth = torch.stack([ramp, ramp.T, 1 - ramp], 0)[None, None].expand(4, T, 3, 64, 64).clone()
    smooth += 0.05 * torch.rand(4, T, 3, 64, 64, generator=g)
    clips.append(smooth.clamp(0, 1))                                                          # ramps
    const = torch.rand(4, 1, 3, 1, 1, generator=g).expand(4, T, 3, 64, 64).clone()
    clips.append(const)                                                                       # flat colour frames
    sat = (torch.rand(4, T, 3, 64, 64, generator=g) > 0.5).float()
    clips.append(sat)                                                                         # saturated 0 / 1
    padded = torch.rand(6, T, 3, 64, 64, generator=g)
    padded[:, ctx:] = 0
    clips.append(padded)                                                                      # zero-padded future frames
    px = torch.cat(clips, 0)
    assert px.shape[0] == 24
    want, _ = ora.tokenize(px, ctx)
    got, _ = m.tokenize(px.to(DEV), ctx)
    bad = (got.cpu() != want).nonzero()
    assert len(bad) == 0, f"{len(bad)} of {want.numel()} ids differ, first at {bad[:3].tolist()}"
    err = (m.detokenize(got, ctx).cpu() - ora.detokenize(want, ctx)).abs().max().item()
    assert err < 1e-3, f"decoded pixels max abs err {err:.2e}"


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_sampled_rollouts_many_rows_match_oracle(seed):
    """top-k 100 sampling with explicit uniforms, 8 trajectories x 120 new tokens per seed, action-free and action-conditioned:
    every token equal to the oracle's (fp32 engine mode)."""
    from oracle.llama import generate_cached
    from ivideogpt_amd import HeadModelWithAction, LlamaForCausalLM
    g = torch.Generator().manual_seed(100 + seed)
    cfg, sd, fx = llama_fixture("llama_tiny_ctx1_free.npz")
    prompt = torch.randint(0, 8192, (8, 257), generator=g)
    prompt[:, -1] = cfg["vocab_size"] - 1
    u = torch.rand(8, 120, generator=g)
    out = make_llm(cfg, sd).generate(prompt.to(DEV), do_sample=True, top_k=100, max_new_tokens=120, uniforms=u.to(DEV)).cpu()
    ref = generate_cached(oracle_llama(cfg, sd), prompt, 120, top_k=100, uniforms=u)
    assert torch.equal(out, ref), f"action-free: {(out != ref).sum().item()} tokens differ"
    cfg, sd, fx = llama_fixture("llama_tiny_ctx1_act.npz")
    ctx, adim = int(fx["ctx"]), int(fx["action_dim"])
    head = HeadModelWithAction(LlamaForCausalLM(cfg, None, dtype="fp32"), adim, 257 * ctx - 1, 16, ctx, 9)
    head.load_state_dict(sd, strict=True)
    head.to(DEV)
    action = torch.randn(8, 9, adim, generator=g)
    n_new = 17 * 7 - 1
    u = torch.rand(8, n_new, generator=g)
    out = head.generate(prompt.to(DEV), do_sample=True, top_k=100, max_new_tokens=n_new, action=action.to(DEV), uniforms=u.to(DEV)).cpu()
    ae = torch.nn.functional.linear(action, sd["action_linear.weight"], sd["action_linear.bias"])
    ref = generate_cached(oracle_llama(cfg, sd, prefix="llm.model."), prompt, n_new, top_k=100, uniforms=u, action_embeds=ae, ctx=ctx,
                          sdf_token=cfg["vocab_size"] - 1)
    assert torch.equal(out, ref), f"action-conditioned: {(out != ref).sum().item()} tokens differ"


def test_generate_without_action_matches_the_reference_loop():
    """``HeadModelWithAction.generate_without_action`` (action_model.py:123-152, dead code in the reference but part of its class):
    per future frame 16 sampled tokens from a FRESH prefill of the grown prompt, then the forced ``sdf``; the engine runs it as one
    prefill + cached steps.  Oracle = the reference's loop restated with per-frame re-prefill; uniform column j - 1 drives new token j
    (the columns of the forced slots are not consumed)."""
    from oracle.llama import generate_cached
    from ivideogpt_amd import HeadModelWithAction, LlamaForCausalLM
    cfg, sd, fx = llama_fixture("llama_tiny_ctx1_act.npz")
    ctx, adim = int(fx["ctx"]), int(fx["action_dim"])
    F_ = 4
    head = HeadModelWithAction(LlamaForCausalLM(cfg, None, dtype="fp32"), adim, 257 * ctx - 1, 16, ctx, ctx + F_)
    head.load_state_dict(sd, strict=True)
    head.to(DEV)
    g = torch.Generator().manual_seed(321)
    sdf = cfg["vocab_size"] - 1
    prompt = torch.randint(0, 8192, (5, 257 * ctx), generator=g)
    prompt[:, -1] = sdf
    n_new = 17 * F_ - 1
    u = torch.rand(5, n_new, generator=g)
    out = head.generate_without_action(prompt.to(DEV), do_sample=True, top_k=100, max_new_tokens=n_new, uniforms=u.to(DEV)).cpu()
    ora = oracle_llama(cfg, sd, prefix="llm.model.")
    toks = prompt
    for i in range(F_):
        toks = generate_cached(ora, toks, 16, top_k=100, uniforms=u[:, 17 * i:17 * i + 16])
        toks = torch.cat([toks, torch.full((5, 1), sdf, dtype=toks.dtype)], 1)
    ref = toks[:, :-1]
    assert out.shape == ref.shape and torch.equal(out, ref), f"{(out != ref).sum().item()} tokens differ"
    assert bool((out[:, 257 * ctx + 16::17] == sdf).all())
    with pytest.raises(ValueError):                                   # the gpt2 branch of the reference (action_model.py:30-33) is not a Llama: refused
        HeadModelWithAction(LlamaForCausalLM(cfg, None, dtype="fp32"), adim, 257 * ctx - 1, 16, ctx, ctx + F_, model_type="gpt2")


def test_detokenize_bf16_output_and_cache_element_type():
    """``ivg_detokenize_to`` / ``detokenize(out_dtype=torch.bfloat16)``: the bf16 decode path hands the clip back in bfloat16 (what the
    reference's callers get under autocast, vp/ivideogpt_interface.py:180) -- equal to the float32 result rounded to bf16, cache paths
    included; the fp32 decode path refuses it, and a cache filled with one element type refuses reuse with the other."""
    from helpers import tokenizer_fixture
    from ivideogpt_amd import CompressiveVQModel
    cfg, sd, ctx, px, g = tokenizer_fixture("tok_mini64_ctx2.npz")
    ids = torch.from_numpy(g["indices"]).to("cuda:0")
    m = CompressiveVQModel(cfg, sd, encode_dtype="fp32", decode_dtype="bf16").to("cuda:0")
    f32 = m.detokenize(ids, ctx)
    b16 = m.detokenize(ids, ctx, out_dtype=torch.bfloat16)
    assert b16.dtype == torch.bfloat16 and b16.shape == f32.shape
    assert torch.equal(b16, f32.to(torch.bfloat16))
    c16 = m.detokenize(ids, ctx, out_dtype=torch.bfloat16, clamp=True)
    assert torch.equal(c16, f32.clamp(0, 1).to(torch.bfloat16))
    a, cache = m.detokenize(ids, ctx, return_cache=True, out_dtype=torch.bfloat16)
    b = m.detokenize(ids, ctx, cache=cache, out_dtype=torch.bfloat16)
    assert torch.equal(a, b16) and torch.equal(b, b16)
    with pytest.raises(AssertionError):
        m.detokenize(ids, ctx, cache=cache)            # float32 call on a cache that holds bfloat16 context pixels
    m32 = CompressiveVQModel(cfg, sd, encode_dtype="fp32", decode_dtype="fp32").to("cuda:0")
    with pytest.raises(AssertionError):
        m32.detokenize(ids, ctx, out_dtype=torch.bfloat16)


def test_detokenize_cache_remembers_the_clamp_mode():
    """A cache filled with clamp off and reused with clamp on (or the reverse) would hand back context frames in one mode beside
    predicted frames in the other: refused (IVG_ERR_INVALID -> AssertionError), same mode works."""
    from helpers import tokenizer_fixture
    from ivideogpt_amd import CompressiveVQModel
    cfg, sd, ctx, px, g = tokenizer_fixture("tok_mini64_ctx2.npz")
    ids = torch.from_numpy(g["indices"]).to("cuda:0")
    m = CompressiveVQModel(cfg, sd, encode_dtype="fp32", decode_dtype="fp32").to("cuda:0")
    raw, cache = m.detokenize(ids, ctx, return_cache=True)
    assert torch.equal(m.detokenize(ids, ctx, cache=cache), raw)
    with pytest.raises(AssertionError):
        m.detokenize(ids, ctx, cache=cache, clamp=True)
    cl, cache2 = m.detokenize(ids, ctx, return_cache=True, clamp=True)
    assert torch.equal(cl, raw.clamp(0, 1)) and torch.equal(m.detokenize(ids, ctx, cache=cache2, clamp=True), cl)
