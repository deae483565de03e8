"""Shared-context rollouts and decodes (round 6; -m gpu, all through the C ABI: ivg_generate_shared / ivg_detokenize_shared).

Three of the reference's four callers hand the path rows whose prompt is ONE clip's context repeated -- inference/predict.py:57-73
(``gen_input.repeat(repeat_times, 1)``), train_gpt.py:152-195 (generate_multiple_times), vp/ivideogpt_interface.py:155-202 (VP2: every
candidate action sequence over the same two frames).  The engine then prefills the prompt once per distinct row, keeps its K / V rows
once and decodes the context frames once.  Held to: the REFERENCE's own vectors (tests/golden: HF generate / HeadModelWithAction.generate
/ CompressiveVQModel.detokenize outputs), the un-shared engine with the same uniforms, and the CPU oracle."""
import numpy as np
import pytest
import torch

from helpers import assert_sampled_rollout_matches, llama_fixture, oracle_llama, tokenizer_fixture

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def make_llm(cfg, sd, dtype="fp32", lds_kb=0):
    from ivideogpt_amd import LlamaForCausalLM
    return LlamaForCausalLM(cfg, sd, dtype=dtype, decode_lds_kb=lds_kb).to(DEV)


@pytest.mark.parametrize("lds_kb", [0, 40], ids=["one_batch", "batches_in_flight"])
@pytest.mark.parametrize("name", ["llama_tiny_ctx2_free.npz", "llama_tiny_ctx1_free.npz"])
def test_shared_greedy_equals_hf_vectors_for_every_sample(name, lds_kb):
    """``prompt.repeat(t, 1)`` through the shared-context entry: every one of the t greedy samples of a prompt equals the tokens HF
    generate produced for it (tests/golden), for t = 1 (plain entry), 3 and 5; ``"auto"`` finds the repetition by itself; an explicit t
    that does not describe the rows is refused."""
    cfg, sd, g = llama_fixture(name)
    m = make_llm(cfg, sd, lds_kb=lds_kb)
    prompt = torch.from_numpy(g["prompt"]).to(DEV)
    n_new = g["greedy"].shape[1] - prompt.shape[1]
    ref = torch.from_numpy(g["greedy"])
    for t in (1, 3, 5):
        out = m.generate(prompt.repeat(t, 1), do_sample=False, max_new_tokens=n_new, shared_context=t).cpu()
        assert torch.equal(out, ref.repeat(t, 1)), f"t={t}: {(out != ref.repeat(t, 1)).sum().item()} greedy tokens differ from HF generate"
    out = m.generate(prompt.repeat(4, 1), do_sample=False, max_new_tokens=n_new, shared_context="auto").cpu()
    assert torch.equal(out, ref.repeat(4, 1))
    with pytest.raises(ValueError):
        m.generate(torch.cat([prompt, prompt.flip(0)]), do_sample=False, max_new_tokens=4, shared_context=2)


def test_shared_sampled_rollouts_equal_the_unshared_engine_and_the_oracle():
    """t = 6 samples of 2 prompts with explicit uniforms: row k * 2 + b of the shared call == the same row of the plain call on the
    repeated prompt (same uniforms) == the oracle's rollout, near-ties of the inverse CDF excepted (the prompt's last position goes
    through the decode-step kernels in the shared call -- the rule of tests/helpers.py assert_sampled_rollout_matches)."""
    from oracle.llama import generate_cached
    cfg, sd, g = llama_fixture("llama_tiny_ctx2_free.npz")
    m = make_llm(cfg, sd)
    prompt = torch.from_numpy(g["prompt"])
    t, n_new = 6, 60
    rep = prompt.repeat(t, 1)
    u = torch.rand(rep.shape[0], n_new, generator=torch.Generator().manual_seed(5))
    shared = m.generate(rep.to(DEV), do_sample=True, top_k=100, max_new_tokens=n_new, uniforms=u.to(DEV), shared_context=t).cpu()
    plain = m.generate(rep.to(DEV), do_sample=True, top_k=100, max_new_tokens=n_new, uniforms=u.to(DEV)).cpu()
    ora = oracle_llama(cfg, sd)
    ref = generate_cached(ora, rep, n_new, top_k=100, uniforms=u)
    assert_sampled_rollout_matches(shared, ref, ora, u, 100, prompt.shape[1], what="shared-context sampled rollout")
    assert_sampled_rollout_matches(plain, ref, ora, u, 100, prompt.shape[1], what="plain sampled rollout")
    assert (shared[:2] != shared[2:4]).any(), "different uniforms must give different samples"


@pytest.mark.parametrize("name", ["llama_tiny_ctx2_act.npz", "llama_tiny_ctx1_act.npz"])
def test_shared_action_conditioned_matches_reference_and_keeps_actions_per_row(name):
    """HeadModelWithAction.generate with a shared context: (a) the reference's layout ``inputs.repeat(t, 1)`` + ``action.repeat(t, 1, 1)``
    (train_gpt.py:170-172) -> every sample equals the reference class's tokens (tests/golden); (b) VP2's layout -- ONE context, every
    row its own action sequence -- equals the plain entry row for row (greedy), and rows with different actions differ; rewards too."""
    from ivideogpt_amd import HeadModelWithAction, LlamaForCausalLM
    cfg, sd, g = llama_fixture(name)
    ctx, adim = int(g["ctx"]), int(g["action_dim"])
    prompt, action = torch.from_numpy(g["prompt"]).to(DEV), torch.from_numpy(g["action"]).to(DEV)
    head = HeadModelWithAction(LlamaForCausalLM(cfg, None, dtype="fp32"), adim, 257 * ctx - 1, 16, ctx, action.shape[1])
    head.load_state_dict(sd, strict=True)
    head.to(DEV)
    n_new = g["greedy"].shape[1] - prompt.shape[1]
    ref = torch.from_numpy(g["greedy"])
    out = head.generate(prompt.repeat(3, 1), do_sample=False, max_new_tokens=n_new, action=action.repeat(3, 1, 1), shared_context=3).cpu()
    assert torch.equal(out, ref.repeat(3, 1)), f"{(out != ref.repeat(3, 1)).sum().item()} tokens differ from HeadModelWithAction.generate"
    # VP2: 7 candidate action sequences over the context of trajectory 0
    gen = torch.Generator().manual_seed(3)
    acts = torch.randn(7, action.shape[1], adim, generator=gen).to(DEV)
    acts[0] = action[0]
    one = prompt[:1].repeat(7, 1)
    shared = head.generate(one, do_sample=False, max_new_tokens=n_new, action=acts, shared_context="auto").cpu()
    plain = head.generate(one, do_sample=False, max_new_tokens=n_new, action=acts).cpu()
    assert torch.equal(shared, plain), f"{(shared != plain).sum().item()} tokens differ between the shared and the plain entry"
    assert torch.equal(shared[0], ref[0]) and (shared[1] != shared[2]).any()


def test_shared_group_spanning_two_cache_chunks_and_ragged_row_tiles():
    """70 samples of 2 prompts = 140 rows: the engine rolls out in chunks of 128 rows, so the second prompt's group spans two chunks
    (both prefill it) and the last chunk has 12 rows; greedy, every sample equals HF's tokens."""
    cfg, sd, g = llama_fixture("llama_tiny_ctx1_free.npz")
    m = make_llm(cfg, sd)
    prompt = torch.from_numpy(g["prompt"]).to(DEV)
    n_new = 24
    out = m.generate(prompt.repeat(70, 1), do_sample=False, max_new_tokens=n_new, shared_context=70).cpu()
    ref = torch.from_numpy(g["greedy"])[:, :prompt.shape[1] + n_new].repeat(70, 1)
    assert torch.equal(out, ref)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_shared_full_width_small_llama(dtype):
    """12 layers / 768 wide, 514-token prompt, 5 samples (BASELINE config 1's shape: predict.py --repeat_times 5): fp32 -- greedy and
    sampled rows equal the oracle's (near-ties excepted); bf16 (the benchmarked arithmetic) -- bf16 rounding of the one re-fed position
    may flip a near-tie against any fp32 reference, so that arm asserts self-consistency (same uniforms -> same row), the prompt copy
    and in-range tokens."""
    from oracle.llama import generate_cached
    from ivideogpt_amd import weights as W
    cfg = dict(W.LLAMA_SMALL)
    sd = W.random_llama_state_dict(cfg, 41)
    gen = torch.Generator().manual_seed(9)
    prompt = torch.randint(0, 8192, (1, 514), generator=gen)
    prompt[:, 256], prompt[:, -1] = cfg["vocab_size"] - 2, cfg["vocab_size"] - 1
    t, n_new = 5, 20
    u = torch.rand(t, n_new, generator=gen)
    u[3] = u[1]
    m = make_llm(cfg, sd, dtype)
    out = m.generate(prompt.repeat(t, 1).to(DEV), do_sample=True, top_k=100, max_new_tokens=n_new, uniforms=u.to(DEV), shared_context=t).cpu()
    assert torch.equal(out[1], out[3]), "rows of a group with the same uniforms must be identical"
    if dtype == "bf16":   # a larger group over several 16-row tiles of the decode GEMMs
        t2 = 24
        u2 = torch.rand(t2, n_new, generator=gen)
        u2[11], u2[17] = u2[2], u2[2]
        o2 = m.generate(prompt.repeat(t2, 1).to(DEV), do_sample=True, top_k=100, max_new_tokens=n_new, uniforms=u2.to(DEV), shared_context=t2).cpu()
        assert torch.equal(o2[2], o2[11]) and torch.equal(o2[2], o2[17]) and (o2[0] != o2[1]).any()
        assert ((o2 >= 0) & (o2 < cfg["vocab_size"])).all() and torch.equal(o2[:, :514], prompt.repeat(t2, 1))
    assert ((out >= 0) & (out < cfg["vocab_size"])).all() and torch.equal(out[:, :514], prompt.repeat(t, 1))
    if dtype == "fp32":
        ora = oracle_llama(cfg, sd)
        ref = generate_cached(ora, prompt.repeat(t, 1), n_new, top_k=100, uniforms=u)
        assert_sampled_rollout_matches(out, ref, ora, u, 100, 514, what="full-width shared rollout")
        og = m.generate(prompt.repeat(2, 1).to(DEV), do_sample=False, max_new_tokens=n_new, shared_context=2).cpu()
        assert torch.equal(og, generate_cached(ora, prompt.repeat(2, 1), n_new))


@pytest.mark.parametrize("dt", ["bf16", "fp32"])
@pytest.mark.parametrize("B,G,row0,P,pos", [(40, 40, 0, 513, 600), (37, 16, -5, 256, 256), (100, 100, 0, 513, 513), (7, 3, 0, 65, 97), (16, 200, -150, 513, 750),
                                            (9, 1, 0, 0, 300)])
def test_shared_decode_attention_step_vs_fp64(B, G, row0, P, pos, dt):
    """One decode-attention step of a shared-context rollout at op level (ivg_op_shared_decode_attn -> decode_attn_kernel SHARED) against
    softmax(q k^T / 8) v in fp64 over the keys each trajectory logically sees -- its group's prompt rows [0, P) (stored ONCE, in cache
    row `slot`), its own rows [P, pos) and the token being fed -- with RoPE at `pos`.  Groups that start before / end after the chunk
    (row0 < 0, ragged last group), a prefix that is not a multiple of the row block, pos == P (no own rows yet), G = 1 (the plain
    kernel), and the k / v append."""
    import ctypes as C
    from ivideogpt_amd import _lib
    l = _lib.load()
    heads, hd, Lmax = 12, 64, 1024
    tdt = torch.bfloat16 if dt == "bf16" else torch.float32
    gen = torch.Generator().manual_seed(B + G + pos)
    n_slots = (B - 1 - row0) // G + 1
    rows = max(B, n_slots)
    kc = (torch.randn(rows, heads, Lmax, hd, generator=gen) * 1.2).to(tdt)
    vc = torch.randn(rows, heads, Lmax, hd, generator=gen).to(tdt)
    qkv = (torch.randn(B, 3 * heads * hd, generator=gen) * 1.5).to(tdt)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    fr = torch.arange(Lmax, dtype=torch.float32)[:, None] * inv[None, :]
    cos, sin = fr.cos().to(tdt).float(), fr.sin().to(tdt).float()          # tables as the engine holds them (rounded through the model dtype)

    def rope(x):                                                             # x (B, heads, hd) in T -> roped, rounded to T
        x = x.float()
        a, b = x[..., :hd // 2], x[..., hd // 2:]
        c, s_ = cos[pos], sin[pos]
        return torch.cat([a * c - b * s_, b * c + a * s_], -1).to(tdt)
    q = rope(qkv[:, :heads * hd].view(B, heads, hd))
    kn = rope(qkv[:, heads * hd:2 * heads * hd].view(B, heads, hd))
    vn = qkv[:, 2 * heads * hd:].view(B, heads, hd)
    ref = torch.empty(B, heads, hd, dtype=torch.float64)
    for b in range(B):
        s_ = (b - row0) // G
        K = torch.cat([kc[s_, :, :P], kc[b, :, P:pos], kn[b][:, None]], 1).double()     # (heads, pos + 1, hd)
        V = torch.cat([vc[s_, :, :P], vc[b, :, P:pos], vn[b][:, None]], 1).double()
        w = torch.softmax(torch.einsum("hd,hkd->hk", q[b].double(), K) / 8.0, -1)
        ref[b] = torch.einsum("hk,hkd->hd", w, V)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    cd, sd = cos.to(DEV), sin.to(DEV)
    kd, vd, qd = kc.to(DEV), vc.to(DEV), qkv.to(DEV)
    out = torch.full((B, heads * hd), float("nan"), dtype=tdt, device=DEV)
    rc = l.ivg_op_shared_decode_attn(C.c_void_p(qd.data_ptr()), C.c_void_p(kd.data_ptr()), C.c_void_p(vd.data_ptr()), C.c_void_p(out.data_ptr()),
                                     C.c_void_p(cd.data_ptr()), C.c_void_p(sd.data_ptr()), B, heads, hd, Lmax, pos, P, G, row0, 1 if dt == "bf16" else 0, st)
    assert rc == 0, rc
    torch.cuda.synchronize()
    got = out.float().view(B, heads, hd).cpu().double()
    assert torch.isfinite(got).all()
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    assert err < (1.5e-2 if dt == "bf16" else 2e-5), f"rel err {err:.3e}"
    # the append: position `pos` of every trajectory's own cache row holds the roped k (up to one rounding: the kernel contracts the
    # rotation into fused multiply-adds) and, bit for bit, the v of the fed token
    kgot = kd[:B, :, pos].float().cpu()
    assert (kgot - kn.float()).abs().max().item() <= (2.0 ** -7 if dt == "bf16" else 1e-5) * kn.float().abs().max().item()
    assert torch.equal(vd[:B, :, pos].cpu(), vn)
    assert torch.equal(kd[:, :, :pos].cpu(), kc[:, :, :pos]), "cached rows must not be touched"


# ------------------------------------------------------------------------------------------------ detokenize
def make_tok(cfg, sd, ctx, dec="fp32"):
    from ivideogpt_amd import CompressiveVQModel
    m = CompressiveVQModel(cfg, sd, encode_dtype="fp32", decode_dtype=dec).to(DEV)
    if ctx != cfg["context_length"]:
        m.set_context_length(ctx)
    return m


@pytest.mark.parametrize("name", ["tok_mini64_ctx2.npz", "tok_mini64_ctx1.npz", "tok_mini256_ctx2.npz"])
def test_shared_detokenize_equals_reference_pixels(name):
    """t samples per clip with DIFFERENT predicted-frame tokens over the same context tokens, in the reference's ``repeat(t, 1)`` row
    order: (a) the rows that carry the fixture's own tokens reproduce the REFERENCE's decoded pixels within 1e-3 (fp32 mode);
    (b) every row is bit-identical to the plain (per-row) detokenize of the same ids -- context decoded / projected once per clip
    changes no arithmetic; (c) ``"auto"`` finds the grouping; an explicit t over rows whose context tokens differ is refused."""
    cfg, sd, ctx, px, g = tokenizer_fixture(name)
    m = make_tok(cfg, sd, ctx)
    ids = torch.from_numpy(g["indices"])                       # (B0, L)
    B0, L = ids.shape
    t = 3
    rep = ids.repeat(t, 1)
    gen = torch.Generator().manual_seed(1)
    dyn = torch.zeros(L, dtype=torch.bool)
    for f in range((L + 1 - 257 * ctx) // 17):
        dyn[257 * ctx + 17 * f:257 * ctx + 17 * f + 16] = True
    noise = torch.randint(0, cfg["num_dyn_embeddings"], rep.shape, generator=gen) + cfg["num_vq_embeddings"]
    rep[B0:, dyn] = noise[B0:, dyn]                             # samples 1, 2: other predicted-frame tokens, same context
    s = int(g["subsample"])
    shared = m.detokenize(rep.to(DEV), ctx, shared_context=t)
    plain = m.detokenize(rep.to(DEV), ctx)
    err = np.abs(shared[:B0].cpu().numpy()[..., ::s, ::s] - g["recon"]).max()
    assert err < 1e-3, f"shared-context decode vs reference pixels: {err:.2e}"
    assert torch.equal(shared, plain), f"max |shared - plain| = {(shared - plain).abs().max().item():.3e}"
    assert torch.equal(m.detokenize(rep.to(DEV), ctx, shared_context="auto"), plain)
    bad = rep.clone()
    bad[B0, 3] = (bad[B0, 3] + 1) % cfg["num_vq_embeddings"]
    with pytest.raises(ValueError):
        m.detokenize(bad.to(DEV), ctx, shared_context=t)


def test_shared_detokenize_bf16_and_clamped_output():
    """the benchmarked arithmetic (bf16 decode, clamp in the epilogue, bfloat16 pixels): shared == plain bit for bit, VP2's layout
    (one clip, 9 candidates)."""
    cfg, sd, ctx, px, g = tokenizer_fixture("tok_mini64_ctx2.npz")
    m = make_tok(cfg, sd, ctx, dec="bf16")
    ids = torch.from_numpy(g["indices"])[:1]
    L = ids.shape[1]
    rep = ids.repeat(9, 1)
    gen = torch.Generator().manual_seed(2)
    for f in range((L + 1 - 257 * ctx) // 17):
        c0 = 257 * ctx + 17 * f
        rep[1:, c0:c0 + 16] = torch.randint(0, cfg["num_dyn_embeddings"], (8, 16), generator=gen) + cfg["num_vq_embeddings"]
    a = m.detokenize(rep.to(DEV), ctx, clamp=True, out_dtype=torch.bfloat16, shared_context=9)
    b = m.detokenize(rep.to(DEV), ctx, clamp=True, out_dtype=torch.bfloat16)
    assert a.dtype == torch.bfloat16 and torch.equal(a, b) and float(a.float().min()) >= 0.0 and float(a.float().max()) <= 1.0
    assert (a[1] != a[2]).any()
