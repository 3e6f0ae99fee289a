"""Randomised parity, driver-observed: a fixed set of tools/soak.py's random genomes (many short chromosomes down
to 36 bp, low-entropy alphabets, planted repeats, now and then a few Mbp) with random directory depth, slot table
on/off, -m, -b, -k, -L, every single-end record and every pair record compared with the oracle -- the same check
the builder runs for thousands of genomes (profiles/round*_soak*.log), here with seeds that never change."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_gpu_soak_fixed_seeds_pattern3():
    import soak
    msg = soak.run_soak(seconds=600, seed0=90001, pattern=3, max_genomes=40)
    assert msg.startswith("soak ok") and " 40 genomes" in msg, msg


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("pattern", [5, 7])
def test_gpu_soak_fixed_seeds_other_patterns(pattern):
    import refio
    import soak
    import walt_amd
    try:
        msg = soak.run_soak(seconds=300, seed0=91001, pattern=pattern, max_genomes=8)
    finally:
        refio.set_pattern(3)
        walt_amd.set_pattern(3)
    assert msg.startswith("soak ok") and " 8 genomes" in msg, msg
