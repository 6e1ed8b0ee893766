"""CPU oracle for the SYNT_ISIC DDPM sampling hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain-``torch`` fp32 CPU restatement of the arithmetic the
reference delegates to ``diffusers.UNet2DModel`` / ``diffusers.DDPMScheduler``
and ``torchvision.models.resnet18``, configured the
way the reference configures them (core/generator/model_manager.py:173-212).

It is NOT part of the product: only ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker.
``synt_isic_amd`` never imports it and has no CPU fallback.

PARITY STATUS: **parity unpinned** for the floating-point network outputs.
The reference holds no tests, golden vectors or fixtures,
and ``diffusers``/``torchvision`` are neither vendored under /root/reference
nor installed here, so the network restatement cannot be checked against the
real third-party code in this container.  What IS pinned (tests/test_oracle_anchors.py):
  * integer timestep grids, class seed offsets and ``noise_hash`` (bit-exact,
    core/generator/image_generator.py:383-389,586-592);
  * the parameter count 25 304 963 against the recorded checkpoint byte sizes
    (core/cache/metadata/cache_metadata.json:7..55);
  * the beta / alphas_cumprod table endpoints of SURVEY.md section 8c.
"""
