"""CPU oracle for the qflux LoRA-training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker.  The product path
(``qwen-image-finetune_amd/``) never imports this package and fails loudly if
the HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * in-repo half of the reference (block wiring, RoPE tables, complex RoPE,
    modulation order, concat order, gating, model forward) is PINNED: the
    oracle is checked against the reference's own ``transformer_qwenimage.py``,
    ``transformer_qwen_custom.py``, ``transformer_flux.py``, ``transformer_flux_custom.py`` and its loss classes
    (``qflux/losses``) executed in the build container (``tests/golden/make_golden.py``, max |diff| = 0.0) and the
    resulting vectors are committed under ``tests/golden/``.
  * third-party half (diffusers primitives, peft LoRA layer) is restated from
    their published semantics (diffusers>=0.36, peft unpinned); neither
    package is installable offline -> for that half: **parity unpinned**.
  * ``oracle/prodigy.py`` restates the third-party ``prodigyopt.Prodigy`` step (unpinned requirement of the reference, not
    installable offline): **parity unpinned**; its formulas are pinned by hand-derived identities in ``tests/test_prodigy_cpu.py``.
"""
