"""MI355X-native RFI flagging behind the katsdpsigproc operation API.

Sub-modules mirror the reference package layout for the hot path only:
``accel`` (arrays, slots, operations), ``abc`` / ``hip`` (backend seam and its HIP
implementation), ``transpose``, ``percentile``, ``maskedsum`` and ``rfi`` (``host``,
``device``). Device code is hand-written HIP for gfx950, compiled ahead of time into
``_native/libkatsdpsigproc_hip.so`` and reached through the C-ABI declared in
``include/katsdpsigproc_hip.h``.
"""

__version__ = "0.1.0"
