"""CPU: oracle/df_blocks.py (the restated diffusers blocks, "parity unpinned": diffusers is absent from the image) agrees with an
independent third-party implementation of the same taming-VQGAN blocks that IS installed -- transformers' Chameleon VQGAN encoder and
Janus VQGAN decoder -- block by block and over the whole encoder / decoder trunks; and the real-diffusers / real-piqa pin script runs (SKIPPED sections where the
wheels are missing, never a silent pass of a failed comparison).  oracle/pin/crosscheck_vqgan_blocks.py, oracle/pin/pin_df_blocks.py."""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(rel):
    spec = importlib.util.spec_from_file_location(os.path.basename(rel)[:-3], os.path.join(ROOT, rel))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_df_blocks_agree_with_chameleon_vqgan_blocks():
    pytest.importorskip("transformers.models.chameleon.modeling_chameleon")
    pytest.importorskip("transformers.models.janus.modeling_janus")
    cc = _load("oracle/pin/crosscheck_vqgan_blocks.py")
    res, bad = cc.run(verbose=False)
    assert not bad, bad
    # every block family was exercised
    for key in ("resnet_64_128", "downsample_9x11", "attention_128", "vq_8192_ids_differ", "encoder_trunk_mid_attention=0", "encoder_trunk_mid_attention=1",
                "upsample_5x7", "decoder_trunk_3_levels", "decoder_trunk_2_levels"):
        assert key in res
    assert res["vq_8192_ids_differ"] == 0 and res["vq_512_ids_differ"] == 0


def test_pin_df_blocks_script_runs_and_reports():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle/pin/pin_df_blocks.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    for lib in ("diffusers", "piqa"):
        have = importlib.util.find_spec(lib) is not None
        assert (f"{lib}: PINNED" in r.stdout) if have else (f"{lib}: SKIPPED" in r.stdout), r.stdout
