"""CPU oracle: a plain-PyTorch fp32 restatement of the reference's denoising hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package, and only as the checker.
The product (``e3-invaraint-diffusion-model_amd/``) never imports it and has no CPU fallback.

Pinning status (see DESIGN.md "Oracle"):
  * everything except the ``relative_key`` attention term is pinned against the reference
    itself, imported in the build container (``tests/golden/make_fixtures.py`` ->
    ``tests/golden/*.pt``), plus the five docstring known answers the reference holds;
  * the ``relative_key`` term of transformers==4.38.2 ``BertSelfAttention`` (environment.yml:229)
    is NOT importable here (transformers 5.15 dropped the branch) and no reference test or
    fixture covers it: **parity unpinned** for that term.  It is restated from the published
    4.38.2 semantics and cross-checked against a literal einsum transcription.
"""
