"""CPU oracle for the sai U/Q hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker (or the timed CPU baseline), never as the
thing being shipped.  ``sai_amd`` never imports this package; its compute path
is the HIP library and fails loudly when that library is missing.
"""
