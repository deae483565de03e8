"""The oracle against the committed golden vectors (outputs of the REFERENCE classes, produced by
oracle/pin/pin_against_reference.py in the build container).  CPU only."""
import numpy as np
import pytest
import torch

from helpers import llama_fixture, oracle_llama, oracle_tokenizer, tokenizer_fixture


@pytest.mark.parametrize("name", ["tok_mini64_ctx2.npz", "tok_mini64_ctx1.npz", "tok_mini256_ctx2.npz"])
def test_tokenizer_oracle_matches_reference_vectors(name):
    cfg, sd, ctx, px, g = tokenizer_fixture(name)
    m = oracle_tokenizer(cfg, sd, ctx)
    ids, labels = m.tokenize(px, ctx)
    assert np.array_equal(ids.numpy(), g["indices"]), "VQ indices must be bit-exact"
    assert np.array_equal(labels.numpy(), g["labels"])
    s = int(g["subsample"])
    rec = m.detokenize(torch.from_numpy(g["indices"]), ctx)[..., ::s, ::s]
    rec2 = m.detokenize(torch.from_numpy(g["indices_perturbed"]), ctx)[..., ::s, ::s]
    # same torch build on the same ISA reproduces bit-for-bit; allow fp32 round-off for other hosts
    assert np.abs(rec.numpy() - g["recon"]).max() < 1e-4
    assert np.abs(rec2.numpy() - g["recon_perturbed"]).max() < 1e-4
    st = m.encode_stages(px, ctx)
    assert np.abs(st["hq"].numpy() - g["latent_ctx"]).max() < 1e-4
    assert np.abs(st["dq"].numpy() - g["latent_dyn"]).max() < 1e-4


def test_token_layout_and_labels():
    """compressive_vq_model.py:205-218: [256 ctx][scf][256 ctx][sdf][16 dyn]...; labels -100 over context."""
    cfg, sd, ctx, px, g = tokenizer_fixture("tok_mini64_ctx2.npz")
    ids, labels = g["indices"], g["labels"]
    nvq, ndyn = cfg["num_vq_embeddings"], cfg["num_dyn_embeddings"]
    scf, sdf = nvq + ndyn, nvq + ndyn + 1
    F = px.shape[1] - ctx
    assert ids.shape[1] == 257 * ctx - 1 + 17 * F
    assert (ids[:, 256] == scf).all() and (ids[:, 513] == sdf).all() and (ids[:, 530] == sdf).all()
    assert ((ids[:, :256] >= 0) & (ids[:, :256] < nvq)).all()
    assert ((ids[:, 514:530] >= nvq) & (ids[:, 514:530] < nvq + ndyn)).all()
    assert (labels[:, :514] == -100).all() and np.array_equal(labels[:, 514:], ids[:, 514:])


@pytest.mark.parametrize("name", ["llama_tiny_ctx2_free.npz", "llama_tiny_ctx1_free.npz"])
def test_llama_oracle_matches_hf_vectors(name):
    from oracle.llama import generate_cached
    cfg, sd, g = llama_fixture(name)
    m = oracle_llama(cfg, sd)
    lg = m.logits(torch.from_numpy(g["teacher_ids"]))
    assert np.abs(lg[:, -2:].numpy() - g["teacher_logits_last"]).max() < 2e-4
    assert np.abs(lg[:, ::37, ::101].numpy() - g["teacher_logits_sub"]).max() < 2e-4
    prompt = torch.from_numpy(g["prompt"])
    out = generate_cached(m, prompt, g["greedy"].shape[1] - prompt.shape[1])
    assert np.array_equal(out.numpy(), g["greedy"])


@pytest.mark.parametrize("name", ["llama_tiny_ctx2_act.npz", "llama_tiny_ctx1_act.npz"])
def test_action_conditioned_oracle_matches_reference_vectors(name):
    from oracle.llama import generate_cached, generate_reference_algorithm
    cfg, sd, g = llama_fixture(name)
    m = oracle_llama(cfg, sd, prefix="llm.model.")
    prompt, action, ctx = torch.from_numpy(g["prompt"]), torch.from_numpy(g["action"]), int(g["ctx"])
    ae = torch.nn.functional.linear(action, sd["action_linear.weight"], sd["action_linear.bias"])
    n_new = g["greedy"].shape[1] - prompt.shape[1]
    sdf = cfg["vocab_size"] - 1
    a = generate_reference_algorithm(m, prompt, n_new, action_embeds=ae, ctx=ctx, sdf_token=sdf)
    b = generate_cached(m, prompt, n_new, action_embeds=ae, ctx=ctx, sdf_token=sdf)
    assert np.array_equal(a.numpy(), g["greedy"]) and np.array_equal(b.numpy(), g["greedy"])
    # teacher-forced logits of the reference's HeadModelWithAction.forward (action_model.py:154-185) on the finished sequence
    ids = torch.from_numpy(g["greedy"])
    F = (ids.shape[1] + 1 - 257 * ctx) // 17
    x = m.embed(ids).clone()
    x[:, 257 * ctx - 1 + 17 * torch.arange(F)] += ae[:, ctx - 1:-1]
    lg = m.logits(embeds=x).numpy()
    err = max(np.abs(lg[:, -2:] - g["forward_logits_last"]).max(), np.abs(lg[:, ::37, ::101] - g["forward_logits_sub"]).max())
    assert err < 2e-4, f"forward logits: max abs err {err:.2e}"


def test_mbrl_step_oracle_matches_reference_vectors():
    """mbrl/video_predictor.py:293-317 run step by step with HF generate(inputs_embeds, output_hidden_states) + reward_linear
    (oracle/pin/pin_against_reference.py: pin_mbrl_step): the oracle's growing-prompt generate gives the same 16 tokens per step
    and the same reward (hidden state of the LAST generation step)."""
    from oracle.llama import generate_cached
    cfg, _, g = llama_fixture("llama_tiny_ctx2_mbrl.npz")
    import ivideogpt_amd.weights as W
    sd = W.random_llama_state_dict(cfg, int(g["seed"]), action_dim=int(g["action_dim"]), reward_prediction=True)
    m = oracle_llama(cfg, sd, prefix="llm.model.")
    ctx, V = int(g["ctx"]), cfg["vocab_size"]
    tokens = torch.from_numpy(g["prompt"])
    actions = torch.from_numpy(g["actions"])
    n_steps, B = actions.shape[0], tokens.shape[0]
    table = torch.zeros(B, ctx - 1 + n_steps + 1, actions.shape[-1])
    for t in range(n_steps):
        table[:, ctx - 1 + t] = actions[t]
        ae = torch.nn.functional.linear(table, sd["action_linear.weight"], sd["action_linear.bias"])
        out, hid = generate_cached(m, tokens, 17, uniforms=None, action_embeds=ae, ctx=ctx, sdf_token=V - 1, return_last_hidden=True)
        pred = out[:, tokens.shape[1]:tokens.shape[1] + 16]
        assert np.array_equal(pred.numpy(), g["step_tokens"][t])
        r = torch.nn.functional.linear(hid, sd["reward_linear.weight"], sd["reward_linear.bias"]).reshape(B)
        assert np.abs(r.numpy() - g["step_rewards"][t]).max() < 1e-4
        tokens = torch.cat([tokens, pred, torch.full((B, 1), V - 1, dtype=tokens.dtype)], 1)


def test_sampler_restatement_properties():
    """explicit-uniform top-k sampler: greedy == argmax; u->0 picks the lowest kept id; only kept ids drawn."""
    from oracle.llama import sample_from_logits
    g = torch.Generator().manual_seed(0)
    lg = torch.randn(8, 16386, generator=g) * 3
    assert torch.equal(sample_from_logits(lg, 100, None), lg.argmax(-1))
    kth = lg.topk(100, -1).values[:, -1:]
    lowest_kept = (lg >= kth).float().argmax(-1)
    assert torch.equal(sample_from_logits(lg, 100, torch.zeros(8)), lowest_kept)
    for _ in range(5):
        t = sample_from_logits(lg, 100, torch.rand(8, generator=g))
        assert (lg.gather(1, t[:, None]) >= kth).all()
    # empirical frequencies follow softmax over the kept set
    row = lg[:1].expand(4000, -1)
    t = sample_from_logits(row, 100, torch.rand(4000, generator=g))
    p = torch.softmax(lg[0].masked_fill(lg[0] < kth[0], float("-inf")), -1)
    top = p.argmax()
    assert abs((t == top).float().mean().item() - p[top].item()) < 0.03


def test_eval_forward_oracle_matches_reference_vectors():
    """oracle.llama.eval_forward against the REFERENCE's losses (tests/golden/llama_tiny_ctx2_eval.npz: HF
    ``LlamaForCausalLM(input_ids, labels).loss`` and ``HeadModelWithAction(reward_prediction, action_recon=0.5).forward``)."""
    import json
    from helpers import load_golden
    from oracle.llama import eval_forward
    import ivideogpt_amd.weights as W
    g = load_golden("llama_tiny_ctx2_eval.npz")
    cfg = json.loads(str(g["config"]))
    seed, adim, ctx, F = int(g["seed"]), int(g["action_dim"]), int(g["ctx"]), int(g["n_future"])
    ids, labels, action = torch.from_numpy(g["ids"]), torch.from_numpy(g["labels"]), torch.from_numpy(g["action"])
    o = eval_forward(oracle_llama(cfg, W.random_llama_state_dict(cfg, seed)), ids, labels)
    assert abs(o["loss"].item() - float(g["loss_free"])) < 1e-4
    sda = W.random_llama_state_dict(cfg, seed + 1, action_dim=adim, reward_prediction=True, action_recon=True)
    ae = torch.nn.functional.linear(action, sda["action_linear.weight"], sda["action_linear.bias"])
    o = eval_forward(oracle_llama(cfg, sda, prefix="llm.model."), ids, labels, action_embeds=ae, ctx=ctx, n_future=F)
    hid = o["hidden"]
    rec = torch.nn.functional.linear(hid[:, 257 * ctx - 1:], sda["action_recon_linear.weight"], sda["action_recon_linear.bias"])
    tgt = action[:, ctx - 1:-1].unsqueeze(-2).repeat(1, 1, 17, 1)
    ar = torch.nn.functional.mse_loss(rec.reshape(-1, F, 17, adim), tgt)
    assert abs(ar.item() - float(g["action_recon_loss"])) < 1e-5
    assert abs((o["loss"] + float(g["action_recon_weight"]) * ar).item() - float(g["loss_act"])) < 1e-4
    start = (257 * ctx - 1) + torch.arange(F) * 17
    rp = torch.nn.functional.linear(hid[:, start + 16], sda["reward_linear.weight"], sda["reward_linear.bias"])
    assert np.abs(rp.numpy() - g["reward_pred"]).max() < 1e-4


def test_clip_ingest_matches_reference_parser_output():
    """ivideogpt_amd.data.NPZParser (host path) on the reference's sample episode, seed 0, == the clip the REFERENCE's own
    inference/utils.py NPZParser produced (tests/golden/fractal_clip_seed0.npz), bit for bit."""
    import os
    from helpers import GOLDEN
    from ivideogpt_amd.data import NPZParser, frame_stride
    ref = torch.from_numpy(np.load(os.path.join(GOLDEN, "fractal_clip_seed0.npz"))["clip"])
    np.random.seed(0)
    clip, actions = NPZParser(16, 64).parse(os.path.join(GOLDEN, "fractal_sample.npz"), "fractal20220817_data")
    assert actions is None and clip.shape == (16, 3, 64, 64) and torch.equal(clip, ref)
    assert frame_stride("fractal20220817_data") == 1 and frame_stride("kuka") == 3 and frame_stride("toto") == 10
    assert frame_stride("asu_table_top_converted_externally_to_rlds") == 4 and frame_stride("unknown_dataset") == 1


def test_frame_metric_oracle_known_answers():
    """oracle/metrics.py (restatement of Evaluator.forward, ivideogpt/utils/video_metric.py:63-100): closed-form cases.
    Constant frames x = a, y = b: every filtered moment is the constant itself, all variances vanish, so
    ssim = (2ab + c1) / (a^2 + b^2 + c1); mse = (a - b)^2; psnr = 10 log10(1 / (mse + 1e-8)); best-of-t keeps the closest sample."""
    from oracle.metrics import frame_metric_rows, gaussian_window
    w = gaussian_window()
    assert abs(w.sum().item() - 1) < 1e-6 and torch.allclose(w, w.flip(0)) and w.argmax().item() == 5
    a, b1, b2 = 0.6, 0.5, 0.2
    gt = torch.full((2, 3, 3, 32, 32), a)
    pred = torch.cat([torch.full((2, 3, 3, 32, 32), b2), torch.full((2, 3, 3, 32, 32), b1)], 0)   # sample 0: far, sample 1: close
    rows = frame_metric_rows(gt, pred)
    c1 = 0.01 ** 2
    assert rows.shape == (2, 3)
    assert torch.allclose(rows[:, 0], torch.tensor((a - b1) ** 2), atol=1e-7)
    assert torch.allclose(rows[:, 1], torch.tensor(10 * np.log10(1 / ((a - b1) ** 2 + 1e-8)), dtype=torch.float32), atol=1e-3)
    assert torch.allclose(rows[:, 2], torch.tensor((2 * a * b1 + c1) / (a * a + b1 * b1 + c1), dtype=torch.float32), atol=2e-4)
    same = frame_metric_rows(gt, gt.clone())
    assert same[:, 0].abs().max().item() == 0 and torch.allclose(same[:, 1], torch.tensor(80.0), atol=1e-3) and torch.allclose(same[:, 2], torch.tensor(1.0))


def test_sampler_with_temperature_vs_hf_processor_golden():
    """oracle.llama.sample_from_logits against tokens drawn through HF's OWN TemperatureLogitsWarper -> TopKLogitsWarper -> softmax
    (tests/golden/sampler_temperature.npz, written by oracle/pin/pin_against_reference.py: pin_sampler) at T = 0.7 / 1.0 / 1.3."""
    from helpers import load_golden
    from oracle.llama import sample_from_logits
    g = load_golden("sampler_temperature.npz")
    B, V, k = int(g["B"]), int(g["V"]), int(g["top_k"])
    gen = torch.Generator().manual_seed(int(g["seed"]))
    logits = torch.randn(B, V, generator=gen) * 3
    u = torch.from_numpy(g["u"])
    assert torch.equal(torch.rand(B, generator=gen), u)
    for T in (0.7, 1.0, 1.3):
        assert torch.equal(sample_from_logits(logits, k, u, temperature=T), torch.from_numpy(g[f"tok_T{T}"])), T
