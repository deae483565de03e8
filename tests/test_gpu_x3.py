"""The 1e-3-compliant arithmetic that is not fp32-rate ("x3": IVG_F32X3, include/ivg.h): fp32 tensors in HBM, every matrix product on
the bf16 MFMA path with both operands split into bf16 hi + lo (conv3x3.hip / igemm.hip X3 instances), fp32 accumulation.  Held to the
SAME bars as the fp32 engine mode -- decoded pixels and logits within 1e-3 of the reference's vectors and of the oracle at full width,
greedy rollouts token-identical -- and shown to really be another arithmetic than the f32-input MFMA path (IVG_X3=0 switches it off)."""
import numpy as np
import pytest
import torch

from helpers import assert_sampled_rollout_matches, llama_fixture, oracle_llama, oracle_tokenizer, tokenizer_fixture

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RECORD = {}


def make_tok(cfg, sd, ctx, dec):
    from ivideogpt_amd import CompressiveVQModel
    m = CompressiveVQModel(cfg, sd, encode_dtype="fp32", decode_dtype=dec).to(DEV)
    if ctx != cfg["context_length"]:
        m.set_context_length(ctx)
    return m


def make_llm(cfg, sd, dtype, lds_kb=0):
    from ivideogpt_amd import LlamaForCausalLM
    return LlamaForCausalLM(cfg, sd, dtype=dtype, decode_lds_kb=lds_kb).to(DEV)


@pytest.mark.parametrize("name", ["tok_mini64_ctx2.npz", "tok_mini64_ctx1.npz", "tok_mini256_ctx2.npz"])
def test_x3_detokenize_within_1e3_of_reference_vectors(name):
    """Pixels the REFERENCE CompressiveVQModel.detokenize produced (tests/golden): x3 decode within 1e-3, tokenize untouched (the
    encoder never runs in x3: ids stay bit-exact)."""
    cfg, sd, ctx, px, g = tokenizer_fixture(name)
    m = make_tok(cfg, sd, ctx, "x3")
    ids, _ = m.tokenize(px.to(DEV), ctx)
    assert np.array_equal(ids.cpu().numpy(), g["indices"]), "tokenize must not be affected by the decode arithmetic"
    s = int(g["subsample"])
    for key_i, key_o in (("indices", "recon"), ("indices_perturbed", "recon_perturbed")):
        rec = m.detokenize(torch.from_numpy(g[key_i]).to(DEV), ctx).cpu().numpy()[..., ::s, ::s]
        err = np.abs(rec - g[key_o]).max()
        assert err < 1e-3, f"{name} {key_o}: x3 decoded pixels max abs err {err:.2e} vs reference"
    m32 = make_tok(cfg, sd, ctx, "fp32")
    a = m.detokenize(torch.from_numpy(g["indices"]).to(DEV), ctx)
    b = m32.detokenize(torch.from_numpy(g["indices"]).to(DEV), ctx)
    d = (a - b).abs().max().item()
    assert 0 < d < 1e-3, f"x3 vs fp32 engine decode: max abs difference {d:.2e} (0 would mean the x3 kernels did not run)"


@pytest.mark.parametrize("res", [64, 256])
def test_x3_full_width_decode_vs_oracle(res):
    """ctx_vae64 (114 M) / ctx_vae256 (310 M) decoders at full width: x3 pixels within 1e-3 of the CPU oracle's fp32 decode; the
    deviation is recorded next to the fp32 engine mode's."""
    from ivideogpt_amd import weights as W
    cfg = W.tokenizer_config(**(W.CTX_VAE64 if res == 64 else W.CTX_VAE256))
    sd = W.random_tokenizer_state_dict(cfg, 31 if res == 64 else 33, codebook_std=0.4)
    T = 4 if res == 64 else 3
    px = torch.randint(0, 256, (1, T, 3, res, res), generator=torch.Generator().manual_seed(2)).float() / 255
    ora = oracle_tokenizer(cfg, sd, 2)
    m = make_tok(cfg, sd, 2, "x3")
    ids, _ = m.tokenize(px.to(DEV), 2)
    ref = ora.detokenize(ids.cpu(), 2)
    e3 = (m.detokenize(ids, 2).cpu() - ref).abs()
    e32 = (make_tok(cfg, sd, 2, "fp32").detokenize(ids, 2).cpu() - ref).abs()
    msg = f"{res}x{res} full-width decode vs fp32 oracle: x3 max {e3.max():.2e} mean {e3.mean():.2e}; fp32 engine max {e32.max():.2e} mean {e32.mean():.2e}"
    print(msg)
    RECORD[f"decode_{res}"] = msg
    assert e3.max().item() < 1e-3, msg


@pytest.mark.parametrize("name", ["llama_tiny_ctx2_free.npz", "llama_tiny_ctx1_free.npz"])
def test_x3_llama_logits_and_greedy_rollout_vs_hf_vectors(name):
    cfg, sd, g = llama_fixture(name)
    m = make_llm(cfg, sd, "x3")
    lg = m.logits(torch.from_numpy(g["teacher_ids"]).to(DEV)).cpu().numpy()
    e = max(np.abs(lg[:, -2:] - g["teacher_logits_last"]).max(), np.abs(lg[:, ::37, ::101] - g["teacher_logits_sub"]).max())
    assert e < 1e-3, f"x3 teacher-forced logits: max abs err {e:.2e} vs HF"
    prompt = torch.from_numpy(g["prompt"]).to(DEV)
    out = m.generate(prompt, do_sample=False, max_new_tokens=g["greedy"].shape[1] - prompt.shape[1]).cpu().numpy()
    assert np.array_equal(out, g["greedy"]), f"{(out != g['greedy']).sum()} greedy tokens differ from HF generate"


def test_x3_full_width_llama_logits_vs_oracle():
    """12 layers / 768 wide (138 M): teacher-forced logits of the x3 prompt pass within 1e-3 of the fp32 oracle, beside the fp32 engine."""
    from ivideogpt_amd import weights as W
    cfg = dict(W.LLAMA_SMALL)
    sd = W.random_llama_state_dict(cfg, 41)
    ids = torch.randint(0, 16386, (2, 300), generator=torch.Generator().manual_seed(3))
    ref = oracle_llama(cfg, sd).logits(ids)
    e3 = (make_llm(cfg, sd, "x3").logits(ids.to(DEV)).cpu() - ref).abs()
    e32 = (make_llm(cfg, sd, "fp32").logits(ids.to(DEV)).cpu() - ref).abs()
    msg = (f"12-layer logits vs fp32 oracle (scale {ref.abs().max():.1f}): x3 max {e3.max():.2e} mean {e3.mean():.2e}; "
           f"fp32 engine max {e32.max():.2e} mean {e32.mean():.2e}")
    print(msg)
    RECORD["logits_small"] = msg
    assert e3.max().item() < 1e-3, msg
    assert e3.max().item() > 0.0


@pytest.mark.parametrize("lds_kb", [0, 40], ids=["one_batch", "batches_in_flight"])
@pytest.mark.parametrize("width", ["small", "medium"])
def test_x3_decode_path_at_released_widths_vs_oracle(width, lds_kb):
    """(Both launch profiles of an engine -- include/ivg.h ivg_config.decode_lds_kb: under the batches-in-flight budget the q/k/v,
    gate/up and down GEMMs run the second-generation plans, which multiply fp32 tensors on f32-input MFMAs.)
    The x3 DECODE path (dgemm3.hip X3: both fragments split in registers) at the released transformer widths: greedy rollouts from
    a 514-token prompt through ``generate`` equal the oracle's token for token; sampled ones with the same uniforms equal it up to
    near-ties of the inverse CDF (a draw within what the 1e-3 logits bar allows of a boundary may fall to the neighbouring kept
    token: tests/helpers.py assert_sampled_rollout_matches, the rule the fp32 mode's long rollouts are held to); rows do not depend
    on their batch-mates (64-row batch vs shards)."""
    from oracle.llama import generate_cached
    from ivideogpt_amd import weights as W
    cfg = dict(W.LLAMA_SMALL if width == "small" else W.LLAMA_MEDIUM)
    cfg["num_hidden_layers"] = 4 if width == "small" else 3          # (keeps the CPU oracle quick: the GEMM shapes are what matters)
    sd = W.random_llama_state_dict(cfg, 49)
    g = torch.Generator().manual_seed(13)
    prompt = torch.randint(0, 8192, (2, 514), generator=g)
    prompt[:, 256], prompt[:, -1] = cfg["vocab_size"] - 2, cfg["vocab_size"] - 1
    n_new = 40
    u = torch.rand(2, n_new, generator=g)
    ora = oracle_llama(cfg, sd)
    m = make_llm(cfg, sd, "x3", lds_kb)
    from ivideogpt_amd import _lib
    a24 = _lib.load().ivg_debug_counter(b"decode_attn24")
    out_g = m.generate(prompt.to(DEV), do_sample=False, max_new_tokens=n_new).cpu()
    assert _lib.load().ivg_debug_counter(b"decode_attn24") - a24 == cfg["num_hidden_layers"] * (n_new - 1), "the x3 rollout did not run over the 24-bit K / V cache"
    ref_g = generate_cached(ora, prompt, n_new)
    assert torch.equal(out_g, ref_g), f"greedy: {(out_g != ref_g).sum().item()} of {2 * n_new} tokens differ from the oracle"
    out_s = m.generate(prompt.to(DEV), do_sample=True, top_k=100, max_new_tokens=n_new, uniforms=u.to(DEV)).cpu()
    ref_s = generate_cached(ora, prompt, n_new, top_k=100, uniforms=u)
    diverged = assert_sampled_rollout_matches(out_s, ref_s, ora, u, 100, prompt.shape[1], what=f"x3 {width} sampled rollout")
    RECORD[f"sampled_{width}_lds{lds_kb}"] = f"{diverged} of 2 rows left the oracle's rollout at a near-tie of the inverse CDF (margin < 3e-3)"
    p64 = torch.randint(0, 16384, (64, 40), generator=g)
    u64 = torch.rand(64, 12, generator=g)
    full = m.generate(p64.to(DEV), do_sample=True, top_k=100, max_new_tokens=12, uniforms=u64.to(DEV)).cpu()
    for rows in (slice(0, 16), slice(63, 64)):
        part = m.generate(p64[rows].to(DEV), do_sample=True, top_k=100, max_new_tokens=12, uniforms=u64[rows].to(DEV)).cpu()
        assert torch.equal(part, full[rows]), f"{width}: rows {rows} differ between the 64-row batch and the shard"


def test_x3_shared_context_rollout_over_the_24_bit_cache(switches):
    """x3 rollout of t samples over ONE prompt (shared-context path: the prompt's rows live once in the 24-bit cache, SHARED instance of
    decode_attn24_kernel): greedy rows equal the oracle's tokens, sampled rows equal the oracle's up to near-ties, rows with the same
    uniforms are identical; and the same engine configuration with the fp32 cache (IVG_KV24=0) decodes the same greedy tokens."""
    from oracle.llama import generate_cached
    from ivideogpt_amd import weights as W
    cfg = dict(W.LLAMA_SMALL)
    cfg["num_hidden_layers"] = 4
    sd = W.random_llama_state_dict(cfg, 43)
    gen = torch.Generator().manual_seed(19)
    prompt = torch.randint(0, 8192, (1, 514), generator=gen)
    prompt[:, 256], prompt[:, -1] = cfg["vocab_size"] - 2, cfg["vocab_size"] - 1
    t, n_new = 5, 24
    u = torch.rand(t, n_new, generator=gen)
    u[3] = u[1]
    ora = oracle_llama(cfg, sd)
    m = make_llm(cfg, sd, "x3")
    out = m.generate(prompt.repeat(t, 1).to(DEV), do_sample=True, top_k=100, max_new_tokens=n_new, uniforms=u.to(DEV), shared_context=t).cpu()
    assert torch.equal(out[1], out[3])
    ref = generate_cached(ora, prompt.repeat(t, 1), n_new, top_k=100, uniforms=u)
    assert_sampled_rollout_matches(out, ref, ora, u, 100, 514, what="x3 shared rollout over the 24-bit cache")
    og = m.generate(prompt.repeat(2, 1).to(DEV), do_sample=False, max_new_tokens=n_new, shared_context=2).cpu()
    ref_g = generate_cached(ora, prompt.repeat(2, 1), n_new)
    assert torch.equal(og, ref_g)
    switches(IVG_KV24="0")
    m32 = make_llm(cfg, sd, "x3")
    from ivideogpt_amd import _lib
    a24 = _lib.load().ivg_debug_counter(b"decode_attn24")
    assert torch.equal(m32.generate(prompt.repeat(2, 1).to(DEV), do_sample=False, max_new_tokens=n_new, shared_context=2).cpu(), ref_g)
    assert _lib.load().ivg_debug_counter(b"decode_attn24") == a24, "IVG_KV24=0 must keep the fp32 cache"


def test_x3_switch_off_is_the_fp32_path(switches):
    """IVG_X3=0: an engine created in x3 mode runs the f32-input MFMA kernels -- bit-identical to the fp32 mode."""
    cfg, sd, ctx, px, g = tokenizer_fixture("tok_mini64_ctx2.npz")
    ids = torch.from_numpy(g["indices"]).to(DEV)
    ref = make_tok(cfg, sd, ctx, "fp32").detokenize(ids, ctx)
    switches(IVG_X3="0")
    off = make_tok(cfg, sd, ctx, "x3").detokenize(ids, ctx)
    assert torch.equal(off, ref)


def test_zz_record_x3_margins():
    import json
    import os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "r04_x3_margins.json"), "w") as f:
        json.dump(RECORD, f, indent=1)


# ------------------------------------------------------------------------------------------------ 24-bit K / V cache (round 6)
def _round24(x):
    """fp32 -> the nearest value with 24 of the 32 bits (sign, exponent, 15 mantissa bits; ties to even): what the cache keeps"""
    u = x.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    u = (u + 0x7F + ((u >> 8) & 1)) & 0xFFFFFF00
    return (u - ((u >> 31) << 32)).to(torch.int32).view(torch.float32)


@pytest.mark.parametrize("B,G,row0,P,pos", [(40, 1, 0, 0, 600), (9, 1, 0, 0, 255), (5, 1, 0, 0, 256), (3, 1, 0, 0, 1), (37, 16, -5, 256, 256),
                                            (40, 40, 0, 513, 700), (7, 3, 0, 65, 97)])
def test_decode_attention_over_the_24_bit_cache_vs_fp64(B, G, row0, P, pos):
    """The K / V cache of the x3 rollout keeps 24 bits per element in two planes (csrc/llama_ops.hip: decode_attn24_kernel,
    kv24_pack_kernel).  (a) the pack kernel's planes hold exactly round-to-nearest-even of the fp32 rows (decoded here from the bytes);
    (b) one decode-attention step equals softmax(q k^T / 8) v in fp64 over THOSE rounded rows within 2e-5 -- and the fp64 result over
    the un-rounded fp32 rows within 4e-5: the cache format costs 2^-17 per element, the error class of the mode's arithmetic;
    (c) the append: position `pos` holds the rounded roped k and the rounded v of the fed token, earlier rows untouched;
    (d) shared-context groups (G > 1) read rows < P from the group's cache row.  Cache lengths around the 256-row block edges."""
    import ctypes as C
    from ivideogpt_amd import _lib
    l = _lib.load()
    heads, hd, Lmax = 12, 64, 1024
    gen = torch.Generator().manual_seed(B + G + pos)
    n_slots = (B - 1 - row0) // G + 1
    rows = max(B, n_slots)
    k32 = torch.randn(rows, heads, Lmax, hd, generator=gen) * 1.2
    v32 = torch.randn(rows, heads, Lmax, hd, generator=gen)
    qkv = torch.randn(B, 3 * heads * hd, generator=gen) * 1.5
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    fr = torch.arange(Lmax, dtype=torch.float32)[:, None] * inv[None, :]
    cos, sin = fr.cos(), fr.sin()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    kd32, vd32 = k32.to(DEV), v32.to(DEV)
    kc = torch.zeros(rows, heads, Lmax * 192, dtype=torch.uint8, device=DEV)
    vc = torch.zeros_like(kc)
    rc = l.ivg_op_kv24_pack(C.c_void_p(kd32.data_ptr()), C.c_void_p(vd32.data_ptr()), C.c_void_p(kc.data_ptr()), C.c_void_p(vc.data_ptr()),
                            rows * heads, pos, Lmax, st)
    assert rc == 0, rc
    torch.cuda.synchronize()

    def planes_to_f32(c):                    # (rows, heads, Lmax * 192) uint8 -> (rows, heads, Lmax, 64) fp32
        c = c.cpu()
        hi = c[..., :Lmax * 128].contiguous().view(torch.int16).view(rows, heads, Lmax, hd).to(torch.int64) & 0xFFFF
        lo = c[..., Lmax * 128:].view(rows, heads, Lmax, hd).to(torch.int64)
        u = (hi << 16) | (lo << 8)
        return (u - ((u >> 31) << 32)).to(torch.int32).view(torch.float32)
    k24, v24 = _round24(k32), _round24(v32)
    assert torch.equal(planes_to_f32(kc)[:, :, :pos], k24[:, :, :pos]) and torch.equal(planes_to_f32(vc)[:, :, :pos], v24[:, :, :pos])
    assert (planes_to_f32(kc)[:, :, pos:] == 0).all(), "rows beyond L must not be written"
    assert ((k24 - k32).abs() <= 2.0 ** -16 * k32.abs()).all()

    def rope(x):
        a, b = x[..., :hd // 2], x[..., hd // 2:]
        c, s_ = cos[pos], sin[pos]
        return torch.cat([a * c - b * s_, b * c + a * s_], -1)
    q = rope(qkv[:, :heads * hd].view(B, heads, hd))
    kn = rope(qkv[:, heads * hd:2 * heads * hd].view(B, heads, hd))
    vn = qkv[:, 2 * heads * hd:].view(B, heads, hd)
    ref24 = torch.empty(B, heads, hd, dtype=torch.float64)
    ref32 = torch.empty_like(ref24)
    for b in range(B):
        s_ = (b - row0) // G
        for ref, kk, vv, knn, vnn in ((ref24, k24, v24, _round24(kn), _round24(vn)), (ref32, k32, v32, kn, vn)):
            K = torch.cat([kk[s_, :, :P], kk[b, :, P:pos], knn[b][:, None]], 1).double()
            V = torch.cat([vv[s_, :, :P], vv[b, :, P:pos], vnn[b][:, None]], 1).double()
            w = torch.softmax(torch.einsum("hd,hkd->hk", q[b].double(), K) / 8.0, -1)
            ref[b] = torch.einsum("hk,hkd->hd", w, V)
    cd, sd, qd = cos.to(DEV), sin.to(DEV), qkv.to(DEV)
    out = torch.full((B, heads * hd), float("nan"), device=DEV)
    before_k = kc.clone()
    rc = l.ivg_op_decode_attn24(C.c_void_p(qd.data_ptr()), C.c_void_p(kc.data_ptr()), C.c_void_p(vc.data_ptr()), C.c_void_p(out.data_ptr()),
                                C.c_void_p(cd.data_ptr()), C.c_void_p(sd.data_ptr()), B, heads, Lmax, pos, P, G, row0, st)
    assert rc == 0, rc
    torch.cuda.synchronize()
    got = out.view(B, heads, hd).cpu().double()
    assert torch.isfinite(got).all()
    e24 = ((got - ref24).abs().max() / ref24.abs().max()).item()
    e32 = ((got - ref32).abs().max() / ref32.abs().max()).item()
    assert e24 < 2e-5 and e32 < 4e-5, (e24, e32)
    kgot, vgot = planes_to_f32(kc), planes_to_f32(vc)
    assert (kgot[:B, :, pos] - kn).abs().max().item() <= 2.0 ** -15 * kn.abs().max().item()       # one rounding to 24 bits + the rotation's fma contraction
    assert torch.equal(vgot[:B, :, pos], _round24(vn))
    assert torch.equal(kgot[:, :, :pos], k24[:, :, :pos]), "cached rows must not be touched"
    assert torch.equal(kc[B:], before_k[B:]), "rows of other trajectories must not be touched"
