"""CPU oracle for the iVideoGPT prediction hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, in plain PyTorch fp32 on the CPU, the algorithm of the reference
(thuml/iVideoGPT) for the path  tokenize -> autoregressive generate -> detokenize.  It is the
checker the HIP engine is compared against; it is never the thing shipped or measured:

  * only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
    import anything from here;
  * the product package (``ivideogpt_amd``) never imports it and fails loudly if the HIP library
    is missing.

Pinning status ("Oracle"):

  * ``oracle.llama``       -- pinned against the reference's own ``HeadModelWithAction``
    (``ivideogpt/transformer/action_model.py``) and HuggingFace ``LlamaForCausalLM`` imported in
    the build container (teacher-forced logits, greedy rollouts).  Fixtures: tests/golden/llama_*.
  * ``oracle.vq_tokenizer`` -- the repo-owned classes (``Encoder``, ``Decoder``,
    ``ConditionalEncoder/Decoder``, ``CrossAttentionBlock``, ``CompressiveVQModel.tokenize /
    detokenize``) are pinned by executing the reference's unmodified ``ivideogpt/vq_model/*.py``
    over ``oracle.df_blocks`` (served as a throw-away ``diffusers`` shim in /tmp) on identical
    weights and inputs.  Fixtures: tests/golden/tok_*.
  * ``oracle.df_blocks``   -- restates ``diffusers==0.27.0`` blocks (ResnetBlock2D,
    Down/Upsample2D, Attention, UNetMidBlock2D, VectorQuantizer).  ``diffusers`` is NOT installed
    in the build image and the reference has no tests or golden vectors:  **parity unpinned** at
    this boundary; cross-checked only structurally (parameter counts 114.16 M / 310.47 M equal
    the reference README, state-dict key schema).

The generating script for every fixture is ``oracle/pin/pin_against_reference.py``.
"""
