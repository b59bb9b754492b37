"""sai_amd -- sai's sliding-window U/Q adaptive-introgression statistics on AMD MI355X.

A from-scratch, gfx950-native implementation of one hot path of xin-huang/sai (``sai score``
with the U and Q statistics), behind sai's own plugin surface: ``sai_amd.stats`` (registry,
``UStatistic``/``QStatistic``), ``sai_amd.generators``, ``sai_amd.preprocessors``,
``sai_amd.sai.score`` and the ``sai score`` CLI (``python -m sai_amd score ...``).
All statistics are computed by hand-written HIP kernels in ``libsaihip.so`` (C ABI in
``include/saihip.h``); there is no CPU compute path in this package.
"""

__version__ = "0.2.0"
