"""Host-side logic that needs no GPU: the C-ABI library loads and exports every symbol include/ivg.h declares,
checkpoint schema / packing, the loud failure without a GPU, the oracle pipeline used as cpu_baseline, and the
world_size-2 (gloo) batch shard + metric all-gather."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_library_exports_every_declared_symbol():
    from ivideogpt_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "libivg.so not built: run `python -c 'import __graft_entry__ as g; g.build()'`"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    hdr = open(os.path.join(ROOT, "include", "ivg.h")).read()
    declared = sorted(set(re.findall(r"\b(ivg_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 20
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, f"declared in include/ivg.h but not exported: {missing}"
    assert set(_lib.EXPORTS) == set(declared), "ctypes binding table and header disagree"
    assert _lib.load().ivg_version().startswith(b"libivg")


def test_param_counts_match_reference_readme():
    from ivideogpt_amd import weights as W
    assert abs(W.count_params(W.tokenizer_param_shapes(W.CTX_VAE64)) / 1e6 - 114.16) < 0.01       # README.md:35-37 "114M"
    assert abs(W.count_params(W.tokenizer_param_shapes(W.CTX_VAE256)) / 1e6 - 310.47) < 0.01      # README.md:38 "310M"
    assert abs(W.count_params(W.llama_param_shapes(W.LLAMA_SMALL)) / 1e6 - 138.43) < 0.01         # "138M"
    assert abs(W.count_params(W.llama_param_shapes(W.LLAMA_MEDIUM)) / 1e6 - 436.26) < 0.01        # "436M"


def test_checkpoint_roundtrip_and_legacy_key_remap(tmp_path):
    from ivideogpt_amd import weights as W
    cfg = W.tokenizer_config(block_out_channels=(64, 64, 64), layers_per_block=1, latent_channels=64, num_vq_embeddings=64,
                             num_dyn_embeddings=64, context_length=2, resolution=64, max_att_resolution=16,
                             mid_block_add_attention=False)
    sd = W.random_tokenizer_state_dict(cfg, 3)
    legacy = {}
    for k, v in sd.items():   # write the deprecated diffusers attention names; the loader must remap them
        for new, old in (("to_q", "query"), ("to_k", "key"), ("to_v", "value"), ("to_out.0", "proj_attn")):
            if f".attentions.0.{new}." in k:
                k = k.replace(f".attentions.0.{new}.", f".attentions.0.{old}.")
        legacy[k] = v
    W.save_tokenizer_checkpoint(str(tmp_path), cfg, legacy, "tokenizer")
    cfg2, sd2 = W.load_tokenizer_checkpoint(str(tmp_path), "tokenizer")
    assert cfg2["block_out_channels"] == (64, 64, 64) and set(sd2) == set(sd)
    assert all(torch.equal(sd[k], sd2[k]) for k in sd)
    lcfg = dict(W.LLAMA_SMALL, hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=1, num_key_value_heads=1)
    lsd = W.random_llama_state_dict(lcfg, 4)
    W.save_transformer_checkpoint(str(tmp_path), lcfg, lsd)
    lcfg2, lsd2 = W.load_transformer_checkpoint(str(tmp_path))
    assert lcfg2["hidden_size"] == 64 and all(torch.equal(lsd[k], lsd2[k]) for k in lsd)
    with pytest.raises(RuntimeError):
        W.validate_state_dict({k: v for k, v in list(sd.items())[1:]}, W.tokenizer_param_shapes(cfg), "tokenizer")


def test_packing_layouts_cpu():
    from ivideogpt_amd import weights as W
    from ivideogpt_amd.packing import pack_llama, pack_tokenizer
    cfg = W.tokenizer_config(block_out_channels=(64, 64, 64), layers_per_block=1, latent_channels=64, num_vq_embeddings=64,
                             num_dyn_embeddings=64, context_length=2, resolution=64, max_att_resolution=16,
                             mid_block_add_attention=False)
    sd = W.random_tokenizer_state_dict(cfg, 3)
    t = pack_tokenizer(sd, cfg, "cpu", 0, 1)
    w = sd["decoder.up_blocks.0.resnets.0.conv1.weight"]
    p = t["decoder.up_blocks.0.resnets.0.conv1.weight"]
    assert p.dtype == torch.bfloat16 and p.shape == (64, 9 * 64)
    assert torch.equal(p.view(64, 3, 3, 64)[5, 1, 2], w[5, :, 1, 2].to(torch.bfloat16))      # K ordered (kh, kw, c)
    assert t["encoder.conv_in.weight"].dtype == torch.float32 and t["encoder.conv_in.weight"].shape == (64, 3, 3, 3)
    assert t["encoder.down_blocks.0.resnets.0.conv1.weight"].dtype == torch.float32         # tokenize path stays fp32
    lcfg = dict(W.LLAMA_SMALL, hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=1, num_key_value_heads=1)
    lsd = W.random_llama_state_dict(lcfg, 4, action_dim=3)
    lt = pack_llama(lsd, lcfg, "cpu", 1, prefix="llm.")
    wgu = lt["llm.layers.0.wgu"]
    assert wgu.shape == (256, 64)
    ln2 = lsd["llm.model.layers.0.post_attention_layernorm.weight"][None, :]   # RMSNorm weight folded into the projection
    assert torch.equal(wgu[0:16], (lsd["llm.model.layers.0.mlp.gate_proj.weight"][0:16] * ln2).to(torch.bfloat16))
    assert torch.equal(wgu[16:32], (lsd["llm.model.layers.0.mlp.up_proj.weight"][0:16] * ln2).to(torch.bfloat16))
    assert torch.equal(wgu[32:48], (lsd["llm.model.layers.0.mlp.gate_proj.weight"][16:32] * ln2).to(torch.bfloat16))
    assert torch.equal(lt["llm.lm_head"], (lsd["llm.lm_head.weight"] * lsd["llm.model.norm.weight"][None, :]).to(torch.bfloat16))
    assert lt["llm.action_linear.weight"].shape == (64, 3) and lt["llm.rope_cos"].shape == (1024, 32)


def test_no_cpu_fallback():
    """The product path must fail loudly without a GPU: there is no eager / oracle fallback."""
    from ivideogpt_amd import CompressiveVQModel, LlamaForCausalLM, weights as W
    cfg = W.tokenizer_config(block_out_channels=(64, 64, 64), layers_per_block=1, latent_channels=64, num_vq_embeddings=64,
                             num_dyn_embeddings=64, context_length=2, resolution=64, max_att_resolution=16,
                             mid_block_add_attention=False)
    m = CompressiveVQModel.from_config(cfg, seed=1)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.tokenize(torch.zeros(1, 3, 3, 64, 64), 2)
    llm = LlamaForCausalLM(dict(W.LLAMA_SMALL, num_hidden_layers=1), None)
    with pytest.raises(RuntimeError):
        llm.generate(torch.zeros(1, 514, dtype=torch.int64), max_new_tokens=5)
    import ivideogpt_amd
    src = "".join(open(os.path.join(os.path.dirname(ivideogpt_amd.__file__), f)).read()
                  for f in os.listdir(os.path.dirname(ivideogpt_amd.__file__)) if f.endswith(".py"))
    assert "import oracle" not in src and "from oracle" not in src, "the product package must never import the oracle"


def test_cpu_baseline_pipeline_runs_reference_algorithm():
    """oracle/pipeline.py (what bench.py's cpu_baseline times): token-identical to tokenize -> generate_cached -> detokenize."""
    from helpers import llama_fixture, oracle_llama, oracle_tokenizer, tokenizer_fixture
    from oracle.llama import generate_cached
    from oracle.pipeline import predict_reference_algorithm
    cfg, sd, ctx, px, g = tokenizer_fixture("tok_mini64_ctx2.npz")
    lcfg, lsd, _ = llama_fixture("llama_tiny_ctx2_free.npz")
    tok, llm = oracle_tokenizer(cfg, sd, ctx), oracle_llama(lcfg, lsd)
    # mini tokenizer has a 1026-token vocabulary, the tiny llama 16386: rollout ids stay valid for detokenize via its clamp
    frames, ids = predict_reference_algorithm(tok, llm, px[:1], ctx, uniforms=None)
    assert frames.shape == (1, px.shape[1], 3, 64, 64) and float(frames.min()) >= 0 and float(frames.max()) <= 1
    prompt = torch.from_numpy(g["indices"])[:1, :257 * ctx]
    assert torch.equal(ids, generate_cached(llm, prompt, 17 * (px.shape[1] - ctx) - 1))


WORKER = r"""
import os, sys, torch
sys.path.insert(0, sys.argv[1])
from ivideogpt_amd import parallel
rank, world, local = parallel.init_from_env("gloo")
B, n = 7, 4                                   # uneven shard: 4 + 3 rows
lo, hi = parallel.shard_rows(B, rank, world)
full = torch.arange(B * n, dtype=torch.float32).view(B, n)
mine = full[lo:hi] * 2.0                      # 'metrics' of my trajectories
rows = parallel.gather_metric_rows(mine, total_rows=B)
assert torch.equal(rows, full * 2.0), (rank, rows)
even = parallel.gather_metric_rows_even(full[rank * 3:(rank + 1) * 3])
assert torch.equal(even, full[:6])
assert parallel.max_over_ranks(float(rank + 1), "cpu") == float(world)
parallel.barrier()
print("OK", rank, lo, hi)
"""


def test_batch_shard_and_metric_allgather_world2_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    port = 29500 + (os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "OK 0 0 4" in outs[0] and "OK 1 4 7" in outs[1]


def test_shard_rows_partition():
    from ivideogpt_amd.parallel import shard_rows
    for n in (1, 7, 64, 256, 513):
        for w in (1, 2, 4, 8):
            spans = [shard_rows(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_npz_ingest_and_resize(tmp_path):
    """datasets/oxe_data_converter.py:57-59 episode format through the NPZParser mirror (inference/utils.py:12-39)."""
    from ivideogpt_amd.data import NPZParser, resize_frames
    rng = np.random.default_rng(0)
    ep = rng.integers(0, 256, (22, 96, 120, 3), dtype=np.uint8)
    act = rng.normal(size=(22, 4))
    f = tmp_path / "ep.npz"
    np.savez(f, image=ep, action=act)
    np.random.seed(0)
    images, actions = NPZParser(16, 64).parse(str(f), "fractal20220817_data", load_action=True)
    assert images.shape == (16, 3, 64, 64) and actions.shape == (16, 4)
    assert float(images.min()) >= 0 and float(images.max()) <= 1
    # antialiased bilinear (triangle filter): constants and linear ramps are preserved, no crop (aspect ratio squashed)
    const = torch.full((1, 3, 96, 120), 0.37)
    assert torch.allclose(resize_frames(const, 64), torch.full((1, 3, 64, 64), 0.37), atol=1e-6)
    ramp = torch.linspace(0, 1, 128)[None, None, None, :].expand(1, 3, 128, 128).contiguous()
    out = resize_frames(ramp, 64)
    centres = (torch.arange(64) * 2 + 0.5) / 127
    assert torch.allclose(out[0, 0, 10, 4:-4], centres[4:-4], atol=1e-5)
    # bair-style key / dtype (int64 frames under 'aux1_image')
    np.savez(tmp_path / "b.npz", image=ep.astype(np.int64), aux1_image=ep[:, :64, :64].astype(np.int64), action=act)
    im2, _ = NPZParser(8, 64).parse(str(tmp_path / "b.npz"), "bair_robot_pushing")
    assert im2.shape == (8, 3, 64, 64)


def test_committed_bench_line_keeps_the_contract():
    """profiles/r02_bench_n1.json is the line bench.py printed on the MI355X at the end of round 2: it must carry every key of the
    driver's contract, the roofline of the kernel with the most time per step (with PMC traffic), the other measured kernel
    classes incl. the decode GEMMs, the fp32-mode throughput and the bounded host-CPU baseline."""
    import json
    path = os.path.join(ROOT, "profiles", "r02_bench_n1.json")
    d = json.load(open(path))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["global_batch"] * 14 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6   # frames / s of the whole job
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is None or r["traffic"] > 0
    assert all(o["kernel_ms_per_step"] <= r["kernel_ms_per_step"] for o in d.get("roofline_other", []))
    assert any("dgemm" in o["kernel"] for o in d["roofline_other"]) and all(0 < o["frac"] <= 1 for o in d["roofline_other"])
    assert d["fp32_mode"]["value"] > 0 and d["fp32_mode"]["value"] < d["value"] and len(d["per_rank_frames_per_s"]) == 1
    assert sum(d["stage_ms"].values()) <= d["ms_per_step"] * 1.02
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0


def test_committed_round5_bench_line_has_the_in_flight_roofline_and_all_configs():
    """profiles/r05_bench_n1.json (the driver's command at the end of round 5): the contract keys, the dominant kernel's roofline on
    the profiler's clock, `roofline_in_flight` measured in the mode `value` is measured in (all lanes' decode bytes over the union of
    their rollout intervals), the MFMA classes priced against the nominal AND the measured sustained peak, the 1e-3-compliant
    modes, and BASELINE configs 3 / 4 / 5 with one batch in flight over >= 10 steps and with the lanes."""
    import json
    d = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_n1.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "single_lane", "roofline_in_flight", "fp32_mode", "compliant_mode", "other_configs"):
        assert k in d, k
    assert d["steps"] == 20 and d["warmup"] == 5 and d["config"]["lanes"] == 4 and d["vs_baseline"] is None
    assert abs(d["value"] - d["config"]["global_batch"] * 14 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] == "hbm" and "decode_attn" in r["kernel"] and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["traffic"] > 0
    assert 0.95 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.05                       # PMC traffic ~ algorithmic bytes
    f = d["roofline_in_flight"]
    assert f["lanes"] == 4 and len(f["per_lane"]) == 4 and 0 < f["frac"] < 1 and abs(f["frac"] - f["achieved"] / f["peak"]) < 1e-9
    assert all(p["rollout_interval_ms"][1] > p["rollout_interval_ms"][0] and p["decode_attn_mean_launch_us"] > 0 for p in f["per_lane"])
    assert f["rollout_phase_ms"] >= max(p["rollout_interval_ms"][1] - p["rollout_interval_ms"][0] for p in f["per_lane"]) - 1e-6
    mf = [o for o in d["roofline_other"] if o["bound"] == "mfma"]
    assert mf and all(o["peak_sustained"] < o["peak"] and o["frac_of_sustained"] > o["frac"] for o in mf)
    assert d["single_lane"]["steps"] >= 10 and d["single_lane"]["value"] < d["value"]
    assert d["fp32_mode"]["value"] < d["compliant_mode"]["value"] < d["single_lane"]["value"]
    for k in ("config_3", "config_4", "config_5"):
        c = d["other_configs"][k]
        assert c["steps"] >= 10 and c["value"] > 0 and c["lanes_in_flight"]["lanes"] == 4 and c["lanes_in_flight"]["value"] > c["value"]


def test_bench_launch_line_and_config_presets():
    """bench.py --gpus N without a torchrun environment re-executes itself under torch.distributed.run on 127.0.0.1, one rank per
    GPU; --config presets carry the per-GPU shapes of BASELINE.json's configs."""
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.torchrun_command(8, ["--gpus", "8", "--steps", "3", "--config", "5"], port=29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "3", "--config", "5"] and cmd[-7].endswith("bench.py")
    c = bench.CONFIGS
    assert c[2] == dict(batch=64, frames=16, res=64, medium=False, action_dim=0, ctx=0)
    assert c[3]["batch"] * 8 == 256 and c[3]["action_dim"] == 4 and c[3]["ctx"] == 1            # bair-64-act-cond, 256 over 8 GPUs
    assert c[4]["res"] == 256 and c[4]["batch"] == 16
    assert c[5]["medium"] and c[5]["batch"] * 8 == 512 and 257 * 2 - 1 + 17 * (c[5]["frames"] - 2) + 1 == 990   # 989-token sequences


def test_lds_swizzle_keys_are_conflict_free_for_their_access_shapes():
    """csrc/conv3x3.hip reads its LDS tiles with ds_read_b128 under two XOR keys; tools/lds_swizzle_check.py restates the bank
    model of the microarchitecture guide.  The plain key serves 16 consecutive rows from any start row, the upsampling key both
    pair alignments -- and neither serves the other's shape (which is why there are two)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import lds_swizzle_check as L
    assert L.worst(L.key_plain) == (4, 4, 8)
    assert L.worst(L.key_ups) == (8, 4, 4)
    # the arithmetic form of the second column half of the 32-pixel-row upsampling fragments: address(hx + 8) = (address(hx) + 512) ^ 16
    for hx in range(24):
        for lg in range(4):
            a = hx * 64 + ((lg ^ L.key_ups(hx)) << 4)
            b = (hx + 8) * 64 + ((lg ^ L.key_ups(hx + 8)) << 4)
            assert ((a + 512) ^ 16) == b


def test_attention_vt_tile_layout_is_a_permutation_and_conflict_free():
    """csrc/llama_ops.hip VtTile: the V^T tile of the one-pass attention kernels is staged key-permuted so that a lane's P.V fragment is
    one ds_read_b128.  tools/lds_vt_layout_check.py restates the index functions: every (row, key) lands once, every lane reads the
    keys (32 pr + 4 lg + r, 32 pr + 16 + 4 lg + r) its P fragment enumerates, and both access shapes (fragment ds_read_b128, staging
    ds_write_b64) take the conflict-free 4 LDS cycles per wave instruction at every tile shape the kernels instantiate."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import lds_vt_layout_check as V
    for KT, HD in ((64, 64), (64, 128), (64, 192), (64, 32), (32, 512), (32, 768)):
        assert V.check(KT, HD) == (4, 4), (KT, HD)
    # the layout rounds 2-4 used (two 8-byte halves at a 144-byte pitch, merged into ds_read2_b64: 16 contiguous lanes, banks mod 32) was
    # 2-way conflicted: rows r and r + 8 of a fragment meet on the same banks
    pairs = [((row * 144) >> 3) & 15 for row in range(16)]
    assert len(set(pairs)) == 8


LANES_WORKER = r"""
import os, sys, threading, time, random, torch
sys.path.insert(0, sys.argv[1])
from ivideogpt_amd import parallel
rank, world, local = parallel.init_from_env("gloo")
L, steps = 2, 9
turn = parallel.Turnstile()
seen = {}
def lane(i):
    for g in range(i, steps, L):
        time.sleep(random.random() * 0.02 * (1 + (rank + i) % 2))        # lanes and ranks drift apart
        rows = torch.full((3, 2), float(100 * g + rank))
        out = turn.run(g, lambda: parallel.gather_metric_rows_even(rows))  # issued in global step order on every rank
        seen[g] = out
ths = [threading.Thread(target=lane, args=(i,)) for i in range(L)]
[t.start() for t in ths]; [t.join() for t in ths]
for g in range(steps):
    want = torch.cat([torch.full((3, 2), float(100 * g + r)) for r in range(world)], 0)
    assert torch.equal(seen[g], want), (rank, g, seen[g])
parallel.barrier()
print("LANES-OK", rank)
"""


def test_two_lanes_issue_their_gathers_in_step_order_world2_gloo(tmp_path):
    """bench.py --lanes 2: two batches in flight per GPU, one host thread per lane; the per-step metric all-gathers of the two lanes
    must reach the process group in the same order on every rank (parallel.Turnstile, ticket = global step index)."""
    script = tmp_path / "lanes.py"
    script.write_text(LANES_WORKER)
    port = 27000 + (os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "LANES-OK 0" in outs[0] and "LANES-OK 1" in outs[1]


GATHERER_WORKER = r"""
import os, sys, threading, time, random, torch
sys.path.insert(0, sys.argv[1])
from ivideogpt_amd import parallel
rank, world, local = parallel.init_from_env("gloo")
L, steps = 3, 11
gat = parallel.OrderedGatherer("cpu")
gat.start(0)
def lane(i):
    for g in range(i, steps, L):
        time.sleep(random.random() * 0.02 * (1 + (rank + i) % 3))        # lanes and ranks drift apart; a lane never waits for a collective
        gat.submit(g, torch.full((3, 2), float(100 * g + rank)))
ths = [threading.Thread(target=lane, args=(i,)) for i in range(L)]
[t.start() for t in ths]; [t.join() for t in ths]
seen = gat.finish(steps)
for g in range(steps):
    want = torch.cat([torch.full((3, 2), float(100 * g + r)) for r in range(world)], 0)
    assert torch.equal(seen[g], want), (rank, g, seen[g])
parallel.barrier()
print("GATHERER-OK", rank)
"""


def test_ordered_gatherer_issues_the_lanes_gathers_in_step_order_world2_gloo(tmp_path):
    """bench.py --lanes with a process group: the per-step metric all-gathers of all lanes are issued by ONE thread in global step
    order (parallel.OrderedGatherer) -- same order on every rank, and no lane waits for a collective."""
    script = tmp_path / "gatherer.py"
    script.write_text(GATHERER_WORKER)
    port = 29000 + (os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GATHERER-OK 0" in outs[0] and "GATHERER-OK 1" in outs[1]


def test_turnstile_orders_tickets_and_releases_waiters_when_a_lane_dies():
    import threading, time
    from ivideogpt_amd import parallel
    turn, order, errs = parallel.Turnstile(), [], []

    def lane(tickets, delay):
        try:
            for t in tickets:
                time.sleep(delay)
                turn.run(t, lambda: order.append(t))
        except RuntimeError as e:
            errs.append(str(e))

    ths = [threading.Thread(target=lane, args=([0, 2, 4], 0.02)), threading.Thread(target=lane, args=([1, 3, 5], 0.0))]   # the odd lane is the fast one
    [t.start() for t in ths]; [t.join(10) for t in ths]
    assert order == [0, 1, 2, 3, 4, 5] and not errs
    turn.reset(0)
    waiter = threading.Thread(target=lane, args=([1], 0.0))   # ticket 0 never comes: its lane "died"
    waiter.start(); time.sleep(0.05); turn.abort(); waiter.join(10)
    assert not waiter.is_alive() and errs and "aborted" in errs[0]


@pytest.mark.parametrize("internal", [True, False])
def test_video_predictor_constructor_follows_the_reference_loader(tmp_path, internal, capsys):
    """``VideoPredictor(device, cfg.world_model)`` (reference mbrl/video_predictor.py:40-110, mbrl/train_metaworld_mbpo.py:41-42):
    tokenizer from its checkpoint, vocabulary = codes + 2, Llama from ``config_name`` with that vocabulary, wrapped with
    prelude 257 * ctx - 1 / 16 tokens per frame / reward head; weights into ``model.llm`` only (load_internal_llm: zero action
    head, fresh reward head) or strictly into the whole wrapper; context-length mismatch -> the reference's warning + set_context_length."""
    sys.path.insert(0, ROOT)
    from types import SimpleNamespace
    from helpers import world_model_files
    from mbrl.video_predictor import VideoPredictor
    args, tcfg, tsd, lcfg, full = world_model_files(tmp_path, internal)
    vp = VideoPredictor("cpu", SimpleNamespace(**args))
    assert vp.tokenizer.context_length == 2 and vp.model.llm.config.vocab_size == 64 + 64 + 2
    assert vp.model.prelude_tokens_num == 513 and vp.model.tokens_num_per_dyna == 16 and vp.model.segment_length == 12
    assert vp.model.model_type == "llama" and vp.model.reward_prediction and vp.context_length == 2 and vp.symlog
    sd = vp.model.state_dict()
    for k, v in full.items():
        if k.startswith("llm."):
            assert torch.equal(sd[k], v), k
    if internal:
        assert not sd["action_linear.weight"].any() and not sd["action_linear.bias"].any()      # action_model.py:36-39
        assert sd["reward_linear.weight"].shape == (1, 128) and sd["reward_linear.weight"].abs().max() <= 128 ** -0.5
    else:
        assert torch.equal(sd["action_linear.weight"], full["action_linear.weight"]) and torch.equal(sd["reward_linear.bias"], full["reward_linear.bias"])
    # context-length mismatch: warning + set_context_length (video_predictor.py:50-53)
    vp1 = VideoPredictor("cpu", dict(args, context_length=1))
    assert vp1.tokenizer.context_length == 1 and vp1.model.prelude_tokens_num == 256 and "mismatch" in capsys.readouterr().out
    # load_pretrained_model off: random weights of the configured architectures (from_config)
    vp0 = VideoPredictor("cpu", dict(args, load_pretrained_model=False))
    assert set(vp0.model.state_dict()) == set(full) and vp0.tokenizer.state_dict().keys() == tsd.keys()
    with pytest.raises(Exception):   # strict loading: an action-free checkpoint into the whole wrapper (or the reverse) must fail
        VideoPredictor("cpu", dict(args, load_internal_llm=not internal))


def test_switch_table_set_override_and_restore():
    """ivideogpt_amd.switches: IVG_* variables are published to the loaded library (ivg_reload_switches); override() restores what was
    there before -- set or unset -- and only IVG_* names are accepted."""
    from ivideogpt_amd import switches
    os.environ.pop("IVG_DECODE_LDS_KB", None)
    os.environ["IVG_GRAPH"] = "1"
    try:
        with switches.override(**switches.BATCHES_IN_FLIGHT, IVG_GRAPH=None):
            assert os.environ["IVG_DECODE_LDS_KB"] == switches.BATCHES_IN_FLIGHT["IVG_DECODE_LDS_KB"] and "IVG_GRAPH" not in os.environ
        assert "IVG_DECODE_LDS_KB" not in os.environ and os.environ["IVG_GRAPH"] == "1"
        with pytest.raises(KeyError):
            switches.set(PATH="/tmp")
    finally:
        os.environ.pop("IVG_GRAPH", None)
        switches.set()


def test_pack_x3_slot_layout_and_split_accuracy():
    """packing.pack_x3: 4 consecutive K elements -> one 16-byte slot [hi(4) | lo(4)]; hi + lo restores the fp32 weight to 2^-16 relative
    (what the 1e-3 bars of the x3 mode rest on), and a dot product over split operands (all four partial products, fp32 accumulate --
    the arithmetic of the X3 kernels) stays within 1e-5 relative of the fp64 one at the K of the largest decoder convolution."""
    from ivideogpt_amd.packing import pack_x3
    g = torch.Generator().manual_seed(0)
    w = torch.randn(8, 9 * 512, generator=g) * 0.05
    p = pack_x3(w)
    assert p.dtype == torch.bfloat16 and p.shape == (8, 2 * 9 * 512)
    slots = p.view(8, -1, 2, 4)
    hi, lo = slots[:, :, 0].reshape(8, -1).float(), slots[:, :, 1].reshape(8, -1).float()
    assert torch.equal(hi, w.to(torch.bfloat16).float())
    assert ((hi + lo) - w).abs().max() <= 2.0 ** -16 * w.abs().max()
    a = torch.rand(9 * 512, generator=g)
    a_hi = a.to(torch.bfloat16).float(); a_lo = (a - a_hi).to(torch.bfloat16).float()
    y = (hi * a_hi + hi * a_lo + lo * a_hi + lo * a_lo).sum(1)
    ref = (w.double() * a.double()).sum(1)
    assert (y.double() - ref).abs().max() <= 1e-5 * (w.double().abs() * a.double()).sum(1).max()
    with pytest.raises(AssertionError):
        pack_x3(torch.zeros(4, 6))


def test_round5_host_fixes_policy_kwargs_gate_and_scratch_cache():
    """Host-side pieces of round 5 (no GPU): the per-engine launch policy travels with the model object and its replicas' constructor
    arguments; ``from_config`` rejects misspelt keyword arguments but ignores HF's inference-irrelevant ones; a PhaseGate whose
    ``wait_event`` raises releases its lock (the other lanes do not deadlock); the metric scratch cache is bounded."""
    import threading
    from ivideogpt_amd import LlamaForCausalLM, metrics, parallel, weights as W
    m = LlamaForCausalLM(W.LLAMA_SMALL, None, dtype="bf16", decode_lds_kb=40)
    assert m._decode_lds_kb == 40 and m.set_decode_lds_kb(0) is m and m._decode_lds_kb == 0        # no engine yet: stored for the next one
    assert LlamaForCausalLM.BATCHES_IN_FLIGHT_LDS_KB == 40
    cfg = dict(W.LLAMA_SMALL, num_hidden_layers=1)
    LlamaForCausalLM.from_config(cfg, trust_remote_code=True, attn_implementation="sdpa", attention_dropout=0.1)   # accepted, ignored
    with pytest.raises(TypeError, match="dtpye"):
        LlamaForCausalLM.from_config(cfg, dtpye="bf16")

    class Boom:
        def wait_event(self, ev):
            raise RuntimeError("stream died")
    gate = parallel.PhaseGate()
    gate._last = object()
    with pytest.raises(RuntimeError):
        with gate.phase(Boom()):
            pass
    got = threading.Event()
    t = threading.Thread(target=lambda: (gate._lock.acquire(), got.set(), gate._lock.release()))
    t.start()
    t.join(5)
    assert got.is_set(), "the gate's lock must be free again after a failed __enter__"
    assert metrics._WS_MAX == 16 and isinstance(metrics._WS, dict)


def test_pack_subpixel_is_the_upsampled_convolution():
    """packing.pack_subpixel (round 6): conv3x3(nearest_x2(x), padding 1) == the four 2 x 2 phase convolutions over x it packs -- phase
    (py, px) writes output pixels (2 iy + py, 2 ix + px), tap (kh2, kw2) reads input pixel (iy + py + kh2 - 1, ix + px + kw2 - 1), zero
    outside the image -- checked in fp64 with plain torch, incl. the borders (where the upsampled image's zero padding must coincide
    with the input's) and the [4 Cout, 4 Cin] layout with K ordered (kh2, kw2, c) the kernel reads (diffusers Upsample2D as built by
    /root/reference/ivideogpt/vq_model/vae.py:271-284)."""
    import torch.nn.functional as F
    from ivideogpt_amd.packing import pack_subpixel
    g = torch.Generator().manual_seed(3)
    Cout, Cin, H, W = 5, 4, 6, 7
    w = torch.randn(Cout, Cin, 3, 3, generator=g, dtype=torch.float64)
    x = torch.randn(2, Cin, H, W, generator=g, dtype=torch.float64)
    ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, padding=1)
    sub = pack_subpixel(w.float()).double()        # the sums are formed in fp32 by design: compare with fp32-sized tolerance
    assert sub.shape == (4 * Cout, 4 * Cin)
    out = torch.zeros_like(ref)
    xp = F.pad(x, (1, 1, 1, 1))
    for py in range(2):
        for px in range(2):
            wk = sub[(2 * py + px) * Cout:(2 * py + px + 1) * Cout].view(Cout, 2, 2, Cin).permute(0, 3, 1, 2)   # [Cout, Cin, kh2, kw2]
            # input rows iy + py + kh2 - 1 = padded rows iy + py + kh2: a 2 x 2 valid convolution over the padded image shifted by (py, px)
            out[:, :, py::2, px::2] = F.conv2d(xp[:, :, py:py + H + 1, px:px + W + 1], wk)
    assert (out - ref).abs().max().item() < 1e-5 * ref.abs().max().item()


def test_shared_context_row_order_helpers():
    """ivideogpt_amd.transformer: the reference's multi-sample callers build ``prompts.repeat(t, 1)`` (row k * B0 + b = sample k of
    prompt b: inference/predict.py:65, train_gpt.py:170); the engine keeps a prompt's samples in consecutive rows.  Detection of the
    repetition (explicit t is verified, "auto" finds the largest t) and the two row permutations are inverse to each other."""
    from ivideogpt_amd.transformer import _from_group_major, _to_group_major, shared_prompt_groups
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, 100, (3, 9), generator=g)
    rep = ids.repeat(4, 1)
    assert shared_prompt_groups(rep, "auto") == (4, 3) and shared_prompt_groups(rep, 4) == (4, 3) and shared_prompt_groups(rep, 2) == (2, 6)
    assert shared_prompt_groups(ids, "auto") == (1, 3) and shared_prompt_groups(ids, 1) == (1, 3)
    assert shared_prompt_groups(rep[:, :5], "auto") == (4, 3)          # a column slice (non-contiguous), as detokenize passes it
    with pytest.raises(ValueError):
        shared_prompt_groups(rep, 5)
    with pytest.raises(ValueError):
        shared_prompt_groups(torch.cat([ids, ids.flip(0)]), 2)
    x = torch.arange(12 * 2).view(12, 2)
    gm = _to_group_major(x, 4, 3)
    assert gm[:, 0].tolist() == [2 * r for r in (0, 3, 6, 9, 1, 4, 7, 10, 2, 5, 8, 11)]      # rows of prompt 0 first
    assert torch.equal(_from_group_major(gm, 4, 3), x) and _to_group_major(None, 4, 3) is None and _to_group_major(x, 12, 1) is x
