"""Occupancy-critical resource limits of the hot kernels, pinned at compile time (no GPU: hipcc cross-compiles gfx950).

The traversal kernel's default instances run 5 waves per SIMD only while they stay within 96 VGPRs, 32 KB of LDS per block and
(measured: round 1, -11 %) without scratch; k_shade / k_raygen run 8 waves per SIMD only within 64 VGPRs.  A change that crosses
one of these lines still passes every parity test and silently costs 10-20 % of the headline: this test makes it loud."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_hot_kernel_instances_keep_their_occupancy():
    import resreport
    rows = {r["name"]: r for r in resreport.report()}

    def one(prefix):
        hits = [r for n, r in rows.items() if n.startswith(prefix)]
        assert len(hits) == 1, (prefix, [n for n in rows if prefix[:20] in n])
        return hits[0]

    for inst in ("k_traverse8_persistent<8, 5, false, false, true, false, false>",   # default instance (C2-C4, host-built C5)
                 "k_traverse8_persistent<8, 5, false, false, true, true, false>"):   # the same on compact primary rays
        r = one(inst)
        assert r["vgpr"] <= 96 and r["scratch"] == 0 and r["lds"] <= 32768 and r["occ"] >= 5, (inst, r)
    r = one("k_traverse8_persistent<12, 4, false, true, false, false, false>")        # two-level instance (C5I)
    assert r["vgpr"] <= 128 and r["scratch"] <= 16 and r["lds"] <= 40960 and r["occ"] >= 4, r
    for inst in ("k_shade<0, false, false, false, true>", "k_shade<0, false, false, false, false>",
                 "k_raygen<false, false, false, true>"):
        r = one(inst)
        assert r["vgpr"] <= 64 and r["scratch"] == 0 and r["occ"] >= 8, (inst, r)
