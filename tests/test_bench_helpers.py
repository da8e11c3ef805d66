"""Host-side pieces of bench.py that feed the roofline objects (no GPU): the algorithmic bytes of one scan launch for
every dense plan the library reports, and the PMC-traffic lookup that must not hand out a figure measured on another
kernel."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import bench  # noqa: E402


def test_scan_algorithmic_bytes_per_plan():
    n, d, k = 10_000_000, 768, 10
    # 1-4 queries: the GEMV scan reads the matrix + the queries and writes k (score, id) pairs per query
    kern, hi, per, b = bench.scan_algorithmic_bytes("dense_scan_topk_kernel<NQ=4> grid=2048x1 + dense_merge_kernel", n, d, 4, k)
    assert (kern, hi, per) == ("dense_scan_topk_kernel", False, 4) and b == n * d * 4 + 4 * d * 4 + 4 * k * 8
    # exact two-level form: one maximum per 32-row tile and query
    plan = "dense_mfma_scores_kernel tile-maxima grid=256x1 queries_per_launch=32 two-level: top-10 of 312500 tile maxima"
    kern, hi, per, b = bench.scan_algorithmic_bytes(plan, n, d, 32, k)
    assert (kern, hi, per) == ("dense_mfma_scores_kernel", False, 32) and b == n * d * 4 + 32 * d * 4 + n / 32 * 32 * 4
    # fp16 first pass: 64 queries per launch whatever the batch, nothing per tile leaves the kernel
    plan = ("dense_hi_tilemax_kernel fp16 first pass queries_per_launch=64 two-level: top-33 of 312500 approximate tile "
            "maxima (width level 0; threshold from a sample of every 128-th tile)")
    kern, hi, per, b = bench.scan_algorithmic_bytes(plan, n, d, 128, k)
    assert (kern, hi, per) == ("dense_hi_tilemax_kernel", True, 64) and b == n * d * 4 + 64 * d * 4
    # full score matrix (panel / tile kernels): S[queries][rows] written once
    plan = "dense_panel_scores_kernel nb=6 parts=7 blocks=2044 queries_per_launch=37376 + scores_pair_topk_kernel"
    kern, hi, per, b = bench.scan_algorithmic_bytes(plan, 591, d, 37376, k)
    assert kern == "dense_panel_scores_kernel" and b == 591 * d * 4 + 37376 * d * 4 + 591.0 * 37376 * 4


def test_pmc_traffic_is_keyed_on_kernel_plan_and_sources(tmp_path, monkeypatch):
    """bench.pmc_traffic hands out an entry of profiles/pmc_traffic.json only when it was measured (scripts/pmc_traffic.py)
    on the kernel that ran, under the same plan string and on the same kernel SOURCES (fingerprint of the files the
    object's kernels live in): an entry of an earlier round, another kernel or another work cut reads as None."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("pmc_traffic", Path(bench.__file__).parent / "scripts" / "pmc_traffic.py")
    pm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pm)
    for key in ("ucc_hybrid", "dense_only_d384", "ucc_colbert", "full_hybrid_rerank", "synth10m_b64", "shard8_proxy_b64"):
        assert pm.sources_fingerprint(key) == bench.kernel_sources_fingerprint(key), key
    # a scratch profiles/ with one fresh and one stale entry
    root = tmp_path / "repo"
    (root / "profiles").mkdir(parents=True)
    fresh = {"kernel": "dense_hi_tilemax_kernel", "plan": "P1", "sources": bench.kernel_sources_fingerprint("synth10m_b64"),
             "bytes_per_launch": 123.0}
    stale = dict(fresh, sources="0" * 16)
    (root / "profiles" / "pmc_traffic.json").write_text(json.dumps({"synth10m_b64": fresh, "synth10m_b32": stale,
                                                                    "old_style": {"kernel": "k", "bytes_per_launch": 1}}))
    monkeypatch.setattr(bench, "ROOT", root)
    monkeypatch.setattr(bench, "kernel_sources_fingerprint", lambda key: fresh["sources"])
    assert bench.pmc_traffic("synth10m_b64", "dense_hi_tilemax_kernel") == 123.0
    assert bench.pmc_traffic("synth10m_b64", "dense_hi_tilemax_kernel<12, true>", "P1") == 123.0
    assert bench.pmc_traffic("synth10m_b64", "dense_hi_tilemax_kernel", "another plan") is None
    assert bench.pmc_traffic("synth10m_b64", "some_other_kernel") is None
    assert bench.pmc_traffic("synth10m_b32", "dense_hi_tilemax_kernel") is None      # measured on other sources
    assert bench.pmc_traffic("old_style", "k") is None                                # no fingerprint at all
    assert bench.pmc_traffic("no_such_entry", "k") is None
