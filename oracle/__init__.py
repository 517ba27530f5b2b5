"""CPU oracle for the U-ResNet hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``ubresnet_amd/`` may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it,
and only as the checker / reported baseline, never as the thing shipped.
"""
