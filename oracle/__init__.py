"""CPU oracle for the DiffSplitting sampling hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``diffsplitting_amd/`` may import,
call or link anything in this package; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and
there only as the checker / reported baseline, never as the product path.

The oracle is a plain ``torch`` (fp32, CPU) functional restatement of the
reference algorithm, written over *state-dict keys* (no ``nn.Module`` copies).
Every function cites the reference file:line it follows (paths are relative to
the reference repo root).

Parity pinning: ``oracle/gen_golden.py`` imports the reference's own classes
from ``/root/reference`` (in the build container only), seeds weights and
inputs, and dumps small ``.npz`` fixtures into ``tests/golden/``.
``tests/test_oracle_golden.py`` checks this restatement against every one of
those fixtures plus the reference's own known-answer test for the tiling path
(``tests/test_tiling_setup.py``: stitch(tiles(arange)) == arange, exact).
"""
