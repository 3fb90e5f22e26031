"""profiles/pmc_traffic.json against the kernel sources it was collected on (bench.py reports `roofline.traffic` from it only when the
source stamp matches): a stale entry is reported here as a skip naming the configuration to re-collect, never as a pass."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_pmc_summary_matches_the_kernel_sources():
    import bench
    d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    stale = []
    for e in d["entries"]:
        cfg = bench.CONFIGS[e["config"]]
        src = "kc_s2_best.hip" if e.get("kernel", "").startswith("kc_s2_best_kernel") else cfg["src"]  # (C4 at --s2-level 4 / 5: bench.py does the same)
        h = bench.kernel_source_hash(src)
        assert e["kernel_hbm_bytes"] > e["algorithmic_bytes"] > 0
        assert abs(e["ratio_to_algorithmic"] - e["kernel_hbm_bytes"] / e["algorithmic_bytes"]) < 0.02
        if e["kernel_source_sha16"] != h and h not in e.get("also_valid_for_sha16", []):
            stale.append(e["config"])
        if h in e.get("also_valid_for_sha16", []):
            assert e.get("note"), "an entry carried over to another source must say why"
    if stale:
        pytest.skip("PMC summary collected on other sources for %s: bench.py reports traffic null there; re-collect with "
                    "`python bench.py --config <C> --pmc` (tools/profile_round.sh)" % ", ".join(stale))
