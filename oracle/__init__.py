"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the focused-attention ViT hot path.

Nothing under ``oracle/`` is part of the shipped product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and there only as the checker / the timed CPU baseline.  The product
path (``focused-attention-vit_amd``) never imports this package and fails loudly
when its HIP library is missing.
"""
