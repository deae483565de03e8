"""The boundary callers on the MI355X (-m gpu): the drop-ins for inference/predict.py, vp/ivideogpt_interface.py and
mbrl/video_predictor.py::rollout run end to end on seeded random checkpoints in the reference's on-disk layout and are
checked against the CPU oracle fed the same uniforms (fp32 engine mode: tokens identical, pixels within 1e-3)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from helpers import oracle_llama, oracle_tokenizer

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"

TOK_CFG = dict(block_out_channels=(64, 128, 128), layers_per_block=1, latent_channels=64, num_vq_embeddings=512,
               num_dyn_embeddings=512, mid_block_add_attention=False, context_length=2, resolution=64, max_att_resolution=16)
LLM_CFG = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2,
               rms_norm_eps=1e-6, rope_theta=10000.0, max_position_embeddings=1024, vocab_size=1026)


def write_checkpoint(path, action_dim=None, reward=False):
    from ivideogpt_amd import weights as W
    tcfg = W.tokenizer_config(**TOK_CFG)
    tsd = W.random_tokenizer_state_dict(tcfg, 51, codebook_std=0.4)
    lsd = W.random_llama_state_dict(LLM_CFG, 52, action_dim=action_dim, reward_prediction=reward)
    W.save_tokenizer_checkpoint(path, tcfg, tsd, "tokenizer")
    W.save_transformer_checkpoint(path, LLM_CFG, lsd, "transformer")
    return tcfg, tsd, lsd


def test_predict_cli_matches_oracle(tmp_path, monkeypatch):
    sys.path.insert(0, os.path.join(ROOT, "inference"))
    import importlib
    predict = importlib.import_module("predict")
    from oracle.llama import generate_cached
    ck = str(tmp_path / "ckpt")
    tcfg, tsd, lsd = write_checkpoint(ck)
    rng = np.random.default_rng(1)
    ep = rng.integers(0, 256, (12, 64, 64, 3), dtype=np.uint8)
    np.savez(tmp_path / "clip.npz", image=ep)
    out_dir = str(tmp_path / "out")
    argv = ["--pretrained_model_name_or_path", ck, "--input_path", str(tmp_path / "clip.npz"), "--dataset_name", "fractal20220817_data",
            "--output_path", out_dir, "--context_length", "2", "--segment_length", "5", "--repeat_times", "3", "--seed", "7", "--dtype", "fp32"]
    rec = predict.main(argv).cpu()
    saved = np.load(os.path.join(out_dir, "pred-samples.npz"))
    assert saved["frames"].shape == (3, 5, 64, 128, 3) and rec.shape == (3, 5, 3, 64, 64)
    # replicate: same seed -> same segment start (np.random) and the same uniforms (torch.rand on the GPU)
    predict.set_seed(7)
    from ivideogpt_amd.data import NPZParser
    frames, _ = NPZParser(5, 64).parse(str(tmp_path / "clip.npz"), "fractal20220817_data")
    u = torch.rand(3, 17 * 3 - 1, device=DEV).cpu()
    tok, llm = oracle_tokenizer(tcfg, tsd, 2), oracle_llama(LLM_CFG, lsd)
    ids_ref, _ = tok.tokenize(frames[None], 2)
    out_ref = generate_cached(llm, ids_ref[:, :514].repeat(3, 1), 17 * 3 - 1, top_k=100, uniforms=u)
    assert np.array_equal(saved["tokens"], out_ref.numpy()), "CLI tokens differ from the oracle rollout"
    rec_ref = tok.detokenize(out_ref, 2).clamp(0, 1)
    assert (rec - rec_ref).abs().max().item() < 1e-3


def test_vp2_interface(tmp_path):
    sys.path.insert(0, ROOT)
    from vp.ivideogpt_interface import iVideoGPTPredictor
    from oracle.llama import generate_cached
    ck = str(tmp_path / "ckpt")
    tcfg, tsd, lsd = write_checkpoint(ck, action_dim=4)
    with open(os.path.join(ck, "llama_cfg.json"), "w") as f:
        json.dump(LLM_CFG, f)
    pred = iVideoGPTPredictor(os.path.join(ck, "llama_cfg.json"), seed=3, vqgan_type="ctx_vqgan",
                              pretrained_vqgan_name_or_path=os.path.join(ck, "tokenizer"),
                              pretrained_transformer_path=os.path.join(ck, "transformer"), action_dim=4,
                              generate_max_batchsize=2, decode_max_batchsize=2, action_recon=None, lora=False, lora_r=8, lora_alpha=32,
                              lora_dropout=0.0, dtype="fp32")
    assert pred.num_context == 2 and pred.base_prediction_modality == "rgb"
    g = torch.Generator().manual_seed(4)
    batch = {"video": torch.rand(3, 2, 64, 64, 3, generator=g).numpy(), "actions": torch.randn(3, 11, 4, generator=g).numpy()}
    torch.manual_seed(11)
    out = pred(batch)["rgb"]
    assert out.shape == (3, 11, 64, 64, 3) and out.dtype == np.float32 and out.min() >= 0 and out.max() <= 1
    # oracle with the same uniforms: two generate chunks (2 + 1 rows) draw torch.rand in order
    torch.manual_seed(11)
    u = torch.cat([torch.rand(2, 169, device=DEV), torch.rand(1, 169, device=DEV)]).cpu()
    tok = oracle_tokenizer(tcfg, tsd, 2)
    llm = oracle_llama(LLM_CFG, lsd, prefix="llm.model.")
    px = torch.from_numpy(batch["video"]).permute(0, 1, 4, 2, 3)
    ids, _ = tok.tokenize(torch.cat([px, torch.zeros_like(px[:, 1:])], 1), 2)     # the reference's zero-padded tokenize (:158-167)
    act = torch.from_numpy(batch["actions"])
    ae = torch.nn.functional.linear(act, lsd["action_linear.weight"], lsd["action_linear.bias"])
    toks = generate_cached(llm, ids[:, :514], 169, top_k=100, uniforms=u, action_embeds=ae, ctx=2, sdf_token=1025)
    ref = tok.detokenize(toks, 2).clamp(0, 1)[:, 1:].permute(0, 1, 3, 4, 2).numpy()
    assert np.abs(out - ref).max() < 1e-3


def test_mbrl_rollout_matches_oracle():
    sys.path.insert(0, ROOT)
    from ivideogpt_amd import CompressiveVQModel, HeadModelWithAction, LlamaForCausalLM, weights as W
    from mbrl.video_predictor import VideoPredictor, symexp
    from oracle.llama import generate_cached
    tcfg = W.tokenizer_config(**TOK_CFG)
    tsd = W.random_tokenizer_state_dict(tcfg, 61, codebook_std=0.4)
    lsd = W.random_llama_state_dict(LLM_CFG, 62, action_dim=4, reward_prediction=True)
    tok = CompressiveVQModel(tcfg, tsd, encode_dtype="fp32", decode_dtype="fp32").to(DEV)
    head = HeadModelWithAction(LlamaForCausalLM(LLM_CFG, None, dtype="fp32"), 4, 513, 16, 2, 16, reward_prediction=True)
    head.load_state_dict(lsd, strict=True)
    head.to(DEV)
    vp = VideoPredictor.from_models(tok, head, context_length=2, symlog=True)
    g = torch.Generator().manual_seed(5)
    obs = torch.randint(0, 256, (2, 9, 64, 64), generator=g).float()
    acts = torch.randn(3, 2, 4, generator=g)
    horizon = 3
    torch.manual_seed(21)
    obss, actions, rewards = vp.rollout(obs, lambda o, t: acts[t], horizon)
    assert obss.shape == (2, horizon + 1, 9, 64, 64) and actions.shape == (2, horizon + 1, 4) and rewards.shape == (2, horizon + 1, 1)
    # oracle restatement of mbrl/video_predictor.py:267-339 with the same uniforms
    torch.manual_seed(21)
    otok, ollm = oracle_tokenizer(tcfg, tsd, 2), oracle_llama(LLM_CFG, lsd, prefix="llm.model.")
    o = obs / 255.
    frames = list(torch.chunk(o, 3, dim=1))
    ctx_frames = torch.stack(frames[-2:], 1)
    tokens, _ = otok.tokenize(torch.cat([ctx_frames, torch.zeros_like(ctx_frames)], 1), 2)
    tokens = tokens[:, :514]
    init = tokens
    act_tab = torch.zeros(2, 1 + horizon + 1, 4)
    for t in range(horizon):
        act_tab[:, 1 + t] = acts[t]
        ae = torch.nn.functional.linear(act_tab, lsd["action_linear.weight"], lsd["action_linear.bias"])
        u = torch.rand(2, 17, device=DEV).cpu()
        out, hid = generate_cached(ollm, tokens, 17, top_k=100, uniforms=u, action_embeds=ae, ctx=2, sdf_token=1025, return_last_hidden=True)
        pred16 = out[:, tokens.shape[1]:tokens.shape[1] + 16]
        r = torch.nn.functional.linear(hid, lsd["reward_linear.weight"], lsd["reward_linear.bias"])
        tokens = torch.cat([tokens, pred16, torch.full((2, 1), 1025, dtype=tokens.dtype)], 1)
        fmap = otok.detokenize(torch.cat([init, pred16], 1), 2).clamp(0, 1)
        frames.append(fmap[:, -1]); frames.pop(0)
        assert (obss[:, t + 1].cpu() - torch.cat(frames, 1)).abs().max().item() < 1e-3, f"step {t}: predicted observation differs"
        assert (rewards[:, t + 1].cpu() - symexp(r)).abs().max().item() < 1e-3, f"step {t}: reward differs"


def test_teacher_forced_logits_with_actions():
    """HeadModelWithAction.forward logits (action_model.py:154-185): action embeddings on every sdf slot."""
    from ivideogpt_amd import HeadModelWithAction, LlamaForCausalLM, weights as W
    lsd = W.random_llama_state_dict(LLM_CFG, 71, action_dim=3)
    head = HeadModelWithAction(LlamaForCausalLM(LLM_CFG, None, dtype="fp32"), 3, 513, 16, 2, 6)
    head.load_state_dict(lsd, strict=True)
    head.to(DEV)
    g = torch.Generator().manual_seed(6)
    ids = torch.randint(0, 1024, (2, 514 + 17 * 4 - 1), generator=g)
    action = torch.randn(2, 6, 3, generator=g)
    lg = head.logits(ids.to(DEV), action.to(DEV)).cpu()
    ollm = oracle_llama(LLM_CFG, lsd, prefix="llm.model.")
    ae = torch.nn.functional.linear(action, lsd["action_linear.weight"], lsd["action_linear.bias"])
    x = ollm.embed(ids).clone()
    start = 513 + torch.arange(4) * 17                       # start_index, action_model.py:176-178
    x[:, start] += ae[:, 1:-1]                               # action_embeds[:, context - 1 : -1]
    ref = ollm.logits(embeds=x)
    assert (lg - ref).abs().max().item() < 1e-3


def test_stepwise_rollout_with_kept_kv_cache_equals_re_prefill():
    """HeadModelWithAction.generate(reuse_cache=True): step t feeds only the last prompt token against the KV cache of step
    t - 1 (17 cached decode steps per environment step) -- same tokens as prefilling the grown prompt every step, rewards
    within 1e-3; a prompt the cache does not match is refused (AssertionError), never silently mis-decoded."""
    from ivideogpt_amd import HeadModelWithAction, LlamaForCausalLM, weights as W
    lsd = W.random_llama_state_dict(LLM_CFG, 81, action_dim=4, reward_prediction=True)
    head = HeadModelWithAction(LlamaForCausalLM(LLM_CFG, None, dtype="fp32"), 4, 513, 16, 2, 16, reward_prediction=True)
    head.load_state_dict(lsd, strict=True)
    head.to(DEV)
    g = torch.Generator().manual_seed(8)
    B, horizon = 5, 4
    prompt0 = torch.randint(0, 1024, (B, 514), generator=g)
    prompt0[:, -1] = 1025                                          # the sdf slot that receives the first action
    act = torch.randn(B, 1 + horizon + 1, 4, generator=g).to(DEV)
    us = [torch.rand(B, 17, generator=g).to(DEV) for _ in range(horizon)]
    sdf = torch.full((B, 1), 1025, dtype=torch.int64, device=DEV)

    def run(reuse):
        tokens, outs, rews = prompt0.to(DEV), [], []
        for t in range(horizon):
            out, r = head.generate(tokens, do_sample=True, top_k=100, max_new_tokens=17, action=act, uniforms=us[t],
                                   return_reward=True, reuse_cache=reuse and t > 0)
            outs.append(out.cpu()); rews.append(r.cpu())
            tokens = torch.cat([tokens, out[:, tokens.shape[1]:tokens.shape[1] + 16], sdf], 1)
        return outs, rews

    a, ra = run(False)
    b, rb = run(True)
    for t in range(horizon):
        assert torch.equal(a[t], b[t]), f"step {t}: {(a[t] != b[t]).sum().item()} tokens differ"
        assert (ra[t] - rb[t]).abs().max().item() < 1e-3
    assert (b[-1][:, 514 + 16::17] == 1025).all()                  # every 17th new token is the forced sdf
    with pytest.raises(AssertionError):                            # cache holds 514 + 17 * 4 - 1 positions, not 513
        head.generate(prompt0.to(DEV), do_sample=False, max_new_tokens=17, action=act, reuse_cache=True)
    with pytest.raises(AssertionError):                            # other batch size
        head.generate(torch.cat([prompt0, b[0][:, 514:530], sdf.cpu()], 1)[:3].to(DEV), do_sample=False, max_new_tokens=17,
                      action=act[:3], reuse_cache=True)


@pytest.mark.parametrize("reuse", [False, True])
def test_mbrl_step_matches_reference_vectors(reuse):
    """The engine, step by step, against the REFERENCE's own per-step outputs (tests/golden/llama_tiny_ctx2_mbrl.npz: HF
    ``generate(inputs_embeds, max_new_tokens=17, output_hidden_states=True)`` + ``reward_linear``, mbrl/video_predictor.py:293-317):
    same 16 tokens per step, reward within 1e-3 -- with the prompt re-prefilled every step and with the KV cache kept."""
    from helpers import llama_fixture
    from ivideogpt_amd import HeadModelWithAction, LlamaForCausalLM, weights as W
    cfg, _, g = llama_fixture("llama_tiny_ctx2_mbrl.npz")
    adim, ctx, V = int(g["action_dim"]), int(g["ctx"]), cfg["vocab_size"]
    sd = W.random_llama_state_dict(cfg, int(g["seed"]), action_dim=adim, reward_prediction=True)
    actions = torch.from_numpy(g["actions"])
    n_steps, B = actions.shape[0], g["prompt"].shape[0]
    head = HeadModelWithAction(LlamaForCausalLM(cfg, None, dtype="fp32"), adim, 257 * ctx - 1, 16, ctx, ctx + n_steps + 1,
                               reward_prediction=True)
    head.load_state_dict(sd, strict=True)
    head.to(DEV)
    tokens = torch.from_numpy(g["prompt"]).to(DEV)
    table = torch.zeros(B, ctx - 1 + n_steps + 1, adim, device=DEV)
    sdf = torch.full((B, 1), V - 1, dtype=torch.int64, device=DEV)
    for t in range(n_steps):
        table[:, ctx - 1 + t] = actions[t].to(DEV)
        out, r = head.generate(tokens, do_sample=False, max_new_tokens=17, action=table, return_reward=True, reuse_cache=reuse and t > 0)
        pred = out[:, tokens.shape[1]:tokens.shape[1] + 16]
        assert np.array_equal(pred.cpu().numpy(), g["step_tokens"][t]), f"step {t}: tokens differ from the reference"
        assert np.abs(r.cpu().numpy() - g["step_rewards"][t]).max() < 1e-3, f"step {t}: reward differs from the reference"
        tokens = torch.cat([tokens, pred, sdf], 1)


@pytest.mark.parametrize("use_cache", [True, False])
def test_mbrl_embeds_level_op_sequence_matches_reference_vectors(use_cache):
    """The op sequence of /root/reference/mbrl/video_predictor.py:286-317, line for line, against the mirror objects:
    ``get_input_embeddings`` -> ``action_linear`` added to the last embedding -> ``llm.generate(inputs_embeds=...,
    return_dict_in_generate=True, output_hidden_states=True)`` -> ``.sequences[:, :-1]`` / ``reward_linear(.hidden_states[-1][-1])``
    -> embeddings of the predicted tokens + sdf appended.  Tokens and rewards of every step must equal what the REFERENCE produced
    for the same weights and actions (tests/golden/llama_tiny_ctx2_mbrl.npz), with and without the kept KV cache; with it, every
    step after the first must actually have reused the cache."""
    from helpers import llama_fixture
    from ivideogpt_amd import HeadModelWithAction, LlamaForCausalLM, weights as W
    cfg, _, g = llama_fixture("llama_tiny_ctx2_mbrl.npz")
    adim, ctx, V = int(g["action_dim"]), int(g["ctx"]), cfg["vocab_size"]
    sd = W.random_llama_state_dict(cfg, int(g["seed"]), action_dim=adim, reward_prediction=True)
    actions = torch.from_numpy(g["actions"])
    n_steps, B = actions.shape[0], g["prompt"].shape[0]
    model = HeadModelWithAction(LlamaForCausalLM(cfg, None, dtype="fp32"), adim, 257 * ctx - 1, 16, ctx, ctx + n_steps + 1,
                                reward_prediction=True)
    model.load_state_dict(sd, strict=True)
    model.to(DEV)
    tokens = torch.from_numpy(g["prompt"]).to(DEV)
    embeds = model.get_input_embeddings(tokens)                                           # :285
    assert embeds.shape == (B, tokens.shape[1], cfg["hidden_size"]) and embeds.dtype == torch.float32
    reused = []
    for t in range(n_steps):
        action = actions[t].to(DEV)
        action_embeds = model.action_linear(action)                                        # :295
        embeds[:, -1] += action_embeds                                                     # :296
        result = model.llm.generate(inputs_embeds=embeds, do_sample=False, temperature=1.0, pad_token_id=50256, top_k=100,
                                    use_cache=use_cache, max_new_tokens=16 + 1, return_dict_in_generate=True,
                                    output_hidden_states=True)                             # :298-308
        reused.append(model.llm.last_generate_reused_cache)
        predicted_token = result.sequences[:, :-1]                                         # :310
        last_layer_hidden_states = result.hidden_states[-1]                                # :311
        last_token_states = last_layer_hidden_states[-1]                                   # :312
        reward = model.reward_linear(last_token_states).squeeze(-2)                        # :313
        assert result.sequences.shape == (B, 17) and last_token_states.shape == (B, 1, cfg["hidden_size"]) and reward.shape == (B, 1)
        cat_predicted_token = torch.concat([predicted_token, (torch.ones(B) * model.token_for_sdf).unsqueeze(1).to(DEV)],
                                           dim=1).to(predicted_token.dtype)               # :315
        embeds = torch.concat([embeds, model.get_input_embeddings(cat_predicted_token)], dim=1)   # :316
        assert np.array_equal(predicted_token.cpu().numpy(), g["step_tokens"][t]), f"step {t}: tokens differ from the reference"
        assert np.abs(reward[:, 0].cpu().numpy() - g["step_rewards"][t]).max() < 1e-3, f"step {t}: reward differs from the reference"
    assert reused == ([False] + [True] * (n_steps - 1) if use_cache else [False] * n_steps), reused


def test_kept_cache_is_verified_not_assumed():
    """A kept KV cache is reused only when it was built from exactly the presented prefix: an embedding changed in the middle
    of the prompt (same shapes) makes the embeds path fall back to a prefill -- with the same tokens as a fresh engine gives --
    and a token-level ``generate(reuse_cache=True)`` whose prefix was not what the cache holds is refused."""
    from ivideogpt_amd import HeadModelWithAction, LlamaForCausalLM, weights as W
    lsd = W.random_llama_state_dict(LLM_CFG, 91, action_dim=4, reward_prediction=True)

    def fresh():
        m = HeadModelWithAction(LlamaForCausalLM(LLM_CFG, None, dtype="fp32"), 4, 513, 16, 2, 16, reward_prediction=True)
        m.load_state_dict(lsd, strict=True)
        return m.to(DEV)
    g = torch.Generator().manual_seed(12)
    B = 3
    prompt = torch.randint(0, 1024, (B, 514), generator=g).to(DEV)
    prompt[:, -1] = 1025
    model = fresh()
    e0 = model.get_input_embeddings(prompt)
    r0 = model.llm.generate(inputs_embeds=e0, do_sample=False, max_new_tokens=17, return_dict_in_generate=True, output_hidden_states=True)
    sdf = torch.full((B, 1), 1025, dtype=torch.int64, device=DEV)
    e1 = torch.cat([e0, model.get_input_embeddings(torch.cat([r0.sequences[:, :-1], sdf], 1))], 1)
    tampered = e1.clone()
    tampered[1, 100] += 0.25                                   # same shape, one cached row differs
    r_t = model.llm.generate(inputs_embeds=tampered, do_sample=False, max_new_tokens=17, return_dict_in_generate=True)
    assert model.llm.last_generate_reused_cache is False
    r_ref = fresh().llm.generate(inputs_embeds=tampered, do_sample=False, max_new_tokens=17, return_dict_in_generate=True)
    assert torch.equal(r_t.sequences, r_ref.sequences)
    # ... and the untouched continuation right after a call that rebuilt the cache from something else is not reused either
    r1 = model.llm.generate(inputs_embeds=e1, do_sample=False, max_new_tokens=17, return_dict_in_generate=True)
    assert model.llm.last_generate_reused_cache is False
    assert torch.equal(r1.sequences, fresh().llm.generate(inputs_embeds=e1, do_sample=False, max_new_tokens=17))
    # token level: the cache now holds e1's 530 positions + 16; a 548-token prompt with another prefix must be refused
    act = torch.randn(B, 6, 4, generator=g).to(DEV)
    out = model.generate(prompt, do_sample=False, max_new_tokens=17, action=act)
    other = torch.randint(0, 1024, (B, 514), generator=g).to(DEV)
    other[:, -1] = 1025
    grown = torch.cat([other, out[:, 514:530], sdf], 1)
    with pytest.raises(AssertionError):
        model.generate(grown, do_sample=False, max_new_tokens=17, action=act, reuse_cache=True)
    good = torch.cat([prompt, out[:, 514:530], sdf], 1)
    act2 = act.clone()
    act2[:, 1] += 1.0                                          # the action of the cached first slot changed: refuse as well
    with pytest.raises(AssertionError):
        model.generate(good, do_sample=False, max_new_tokens=17, action=act2, reuse_cache=True)
    cont = model.generate(good, do_sample=False, max_new_tokens=17, action=act, reuse_cache=True)
    full = fresh().generate(good, do_sample=False, max_new_tokens=17, action=act)
    assert torch.equal(cont, full)


def test_config1_predict_cli_on_fractal_sample_full_width(tmp_path):
    """BASELINE config 1: ``predict.py`` on the reference's own sample episode (tests/golden/fractal_sample.npz, a data file
    of the reference: inference/samples/fractal_sample.npz) with ivideogpt-oxe-64-act-free shapes at FULL width (114 M tokenizer,
    138 M transformer with the real 16386-token vocabulary; seeded random weights), repeat_times 5,
    2 context + 14 predicted frames, fp32 -- tokens identical to the oracle fed the same uniforms, pixels within 1e-3; the clip
    the CLI fed the tokenizer is the one the REFERENCE's NPZParser produces (tests/golden/fractal_clip_seed0.npz)."""
    sys.path.insert(0, os.path.join(ROOT, "inference"))
    import importlib
    predict = importlib.import_module("predict")
    from oracle.llama import generate_cached
    from ivideogpt_amd import weights as W
    tcfg = W.tokenizer_config(**W.CTX_VAE64)               # 8192 + 8192 codes
    lcfg = dict(W.LLAMA_SMALL)                             # vocab 16386 = 8192 + 8192 + 2 (train_gpt.py:144-146)
    assert lcfg["vocab_size"] == 16386
    tsd = W.random_tokenizer_state_dict(tcfg, 71, codebook_std=0.4)
    lsd = W.random_llama_state_dict(lcfg, 72)
    ck = str(tmp_path / "ckpt")
    W.save_tokenizer_checkpoint(ck, tcfg, tsd, "tokenizer")
    W.save_transformer_checkpoint(ck, lcfg, lsd, "transformer")
    sample = os.path.join(ROOT, "tests", "golden", "fractal_sample.npz")
    out_dir = str(tmp_path / "out")
    argv = ["--pretrained_model_name_or_path", ck, "--input_path", sample, "--dataset_name", "fractal20220817_data", "--output_path", out_dir,
            "--context_length", "2", "--segment_length", "16", "--repeat_times", "5", "--seed", "0", "--dtype", "fp32"]
    rec = predict.main(argv).cpu()
    saved = np.load(os.path.join(out_dir, "pred-samples.npz"))
    assert rec.shape == (5, 16, 3, 64, 64)
    clip = torch.from_numpy(np.load(os.path.join(ROOT, "tests", "golden", "fractal_clip_seed0.npz"))["clip"])
    predict.set_seed(0)                                    # same seed -> the same uniforms from the GPU generator
    u = torch.rand(5, 17 * 14 - 1, device=DEV).cpu()
    tok, llm = oracle_tokenizer(tcfg, tsd, 2), oracle_llama(lcfg, lsd)
    ids_ref, _ = tok.tokenize(clip[None], 2)
    from helpers import assert_sampled_rollout_matches, vq_near_tie_audit
    toks = torch.from_numpy(saved["tokens"])
    assert all(torch.equal(toks[r, :514], toks[0, :514]) for r in range(5))
    hybrid = ids_ref.clone()
    hybrid[:, :514] = toks[:1, :514]                       # the context tokens the CLI produced; SURVEY 7 (iii) near-tie audit at 8192 codes
    vq_near_tie_audit(tok, clip[None], 2, hybrid, ids_ref, what="predict CLI, context tokens of fractal_sample.npz")
    out_ref = generate_cached(llm, toks[:, :514], 17 * 14 - 1, top_k=100, uniforms=u)
    n_tie = assert_sampled_rollout_matches(toks, out_ref, llm, u, 100, 514, what="predict CLI on fractal_sample.npz")
    assert n_tie <= 1, f"{n_tie} of 5 rows diverged at a sampling near-tie"
    # pixels: the oracle decodes the tokens the engine produced (rows that left the oracle's path at a near-tie included)
    rec_ref = tok.detokenize(toks, 2).clamp(0, 1)
    assert (rec - rec_ref).abs().max().item() < 1e-3


def test_video_predictor_from_hydra_config_equals_from_models(tmp_path):
    """``VideoPredictor('cuda', cfg.world_model)`` (reference mbrl/train_metaworld_mbpo.py:41-42): the constructor path -- checkpoint
    tree, ``load_internal_llm`` -- yields the same imagined rollout as wrapping the same weights by hand."""
    sys.path.insert(0, ROOT)
    from helpers import world_model_files
    from ivideogpt_amd import CompressiveVQModel, HeadModelWithAction, LlamaForCausalLM
    from mbrl.video_predictor import VideoPredictor
    args, tcfg, tsd, lcfg, full = world_model_files(tmp_path, True)
    args.update(encode_dtype="fp32", decode_dtype="fp32", llm_dtype="fp32")
    vp = VideoPredictor("cuda", args)
    lcfg = dict(lcfg, vocab_size=130)
    head = HeadModelWithAction(LlamaForCausalLM(lcfg, None, dtype="fp32"), 4, 513, 16, 2, 12, reward_prediction=True)
    head.load_state_dict(vp.model.state_dict(), strict=True)
    head.to(DEV)
    tok = CompressiveVQModel(tcfg, tsd, encode_dtype="fp32", decode_dtype="fp32").to(DEV)
    ref = VideoPredictor.from_models(tok, head, context_length=2, symlog=True)
    g = torch.Generator().manual_seed(8)
    obs = torch.randint(0, 256, (2, 9, 64, 64), generator=g).float()
    acts = torch.randn(2, 2, 4, generator=g)
    outs = []
    for p in (vp, ref):
        torch.manual_seed(33)
        outs.append(p.rollout(obs, lambda o, t: acts[t], 2))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert vp.steps_with_kept_cache == 1
