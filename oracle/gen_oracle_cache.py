"""Cache of ONE long oracle run (test infrastructure, like everything under oracle/): the final images of the fp32
oracle's 2000-step SR3 loop on the tiny UNet of fixture loop_sr3_lin_8 (oracle/samplers.py::sr3_p_sample_loop, itself
pinned against the reference by the loop_sr3_* fixtures of gen_golden.py).  The GPU tests that compare the HIP loop with
this run (tests/test_gpu_parity.py::test_sr3_2000_steps_tiny, tests/test_gpu_fullsize.py::
test_reduced_precision_loop_psnr) used to recompute it on the GPU box's host: 1 - 3.5 minutes of a 5 - 10 minute suite.
The draws are not stored: they are re-drawn from the same seeded generator (tests/gpu_util.py::DrawRecorder).
tests/test_oracle_golden.py::test_cached_oracle_run_is_the_oracle recomputes the run on the CPU and compares bitwise.

    python oracle/gen_oracle_cache.py        # -> tests/golden/oracle_run_sr3_2000_tiny.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SCHED, SHAPE, COND_SEED, DRAW_SEED = "sr3_2000", (1, 3, 16, 16), 3, 77


def main():
    from tests.gpu_util import compute_oracle_sr3_loop_tiny
    _, _, _, _, draws, full = compute_oracle_sr3_loop_tiny(SCHED, SHAPE, COND_SEED, DRAW_SEED)
    out = os.path.join(ROOT, "tests", "golden", "oracle_run_sr3_2000_tiny.npz")
    np.savez_compressed(out, full=full.numpy(), n_draws=np.int64(len(draws)),
                        key=np.array([2000, *SHAPE, COND_SEED, DRAW_SEED], dtype=np.int64))
    print("wrote", out, "draws:", len(draws))


if __name__ == "__main__":
    main()
