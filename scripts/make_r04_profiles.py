#!/usr/bin/env python3
"""Assemble profiles/r04_* from the evidence runs merged back under gpurun_out/ (run on the build container):

    python scripts/make_r04_profiles.py gpurun_out/r4final [--sq gpurun_out/r4/sq_ms3 --sq-bm25 gpurun_out/r4/sq_bm25
                                                           --hs gpurun_out/r4]

Inputs: scripts/final_evidence.sh (bench line un-profiled / under rocprofv3 --kernel-trace --stats / 2-rank gloo
rehearsal), scripts/trace_hi_tail.sh (tail64, tail256), scripts/pmc_sq.sh (SQ counters), scripts/trace_hybrid_small.sh,
scripts/pmc_traffic.py (profiles/pmc_traffic.json).  The tables are the scripts' own summaries, copied; the headers say
which command produced them."""
import argparse
import json
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
P = ROOT / "profiles"


def last_json(path):
    for line in reversed(Path(path).read_text().strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit(f"no JSON line in {path}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("final")
    ap.add_argument("--sq", default=None)
    ap.add_argument("--sq-bm25", default=None)
    ap.add_argument("--hs", default=None)
    ap.add_argument("--sq-second", default=None)
    a = ap.parse_args()
    F = ROOT / a.final
    # ---- bench lines
    for src, dst in (("bench_unprofiled.json", "r04_bench_line_unprofiled.json"), ("bench_profiled.json", "r04_bench_line_profiled.json"),
                     ("bench_2rank.json", "r04_bench_line_2rank_gloo_rehearsal.json")):
        if (F / src).exists():
            (P / dst).write_text(json.dumps(last_json(F / src)) + "\n")
    un, pr = last_json(F / "bench_unprofiled.json"), last_json(F / "bench_profiled.json")
    # ---- kernel trace
    head = (
        "# Round 4 — kernel trace of the default bench run (1 x MI355X), final code of the round\n\n"
        f"Command: `rocprofv3 --kernel-trace --stats -d {a.final}/kt -o kt --output-format csv -- python3 bench.py --steps 20 "
        "--warmup 5 --no-cpu-baseline` (`scripts/final_evidence.sh`; the driver's command; 5 timed windows of 20 steps).  JSON line of this "
        f"run: `profiles/r04_bench_line_profiled.json` ({pr['value'] / 1e6:.1f} M queries/s, {pr['ms_per_step']:.4f} ms per step, roofline "
        f"{pr['roofline']['frac']:.3f}); the same command without the profiler, same box, run just before: "
        f"`profiles/r04_bench_line_unprofiled.json` (**{un['value'] / 1e6:.1f} M queries/s, {un['ms_per_step']:.4f} ms per 37 376-query step**, "
        f"windows {', '.join(f'{x:.4f}' for x in un['timing']['ms_per_step'])} ms after {un['warmup']} untimed warm-up steps; roofline "
        f"{un['roofline']['frac']:.3f} = HIP events over the launches of the median window, traffic {un['roofline']['traffic'] / 1e6:.1f} MB per "
        "launch from `profiles/pmc_traffic.json`).\n\n"
        "What the grids are: `dsh_split_queries_kernel` 299008 + `dsh_scores_kernel` 194560 + `dense_hi_select_fuse_kernel` 1196032 + "
        "`bm25_score_topk_kernel` 64x37376 = the timed UCC-en hybrid step in its two-pass form (DESIGN.md 4.11); "
        "`dense_panel_scores_kernel` 131072 + `dense_select_fuse_kernel` 1196032 = the same step's exact form (`exact_form`, "
        "AMDR_DENSE_SMALL_HI=0) and the d = 384 object's warm-up; `64000` / `64x1168` / `37376` = the un-tiled 1 168-query set; "
        "`hybrid_small_kernel` = the B = 1 latency loop (one launch per query); `37376x25` and `74752` (MaxSim) = `ucc_colbert` / "
        "`full_hybrid_rerank`; `dense_hi_tilemax_kernel<12|16, true|false>` + `dense_hi_*` + `dense_rescore_tiles` + `dense_final_topk` = "
        "the large scans (`hbm_scan`, `hbm_scan_d1024`, `shard8_proxy`); `shard_pack_kernel` / `shard_merge_kernel` = the exchange at "
        "world = 1.\n\n")
    (P / "r04_bench_kernel_trace.md").write_text(head + (F / "kernel_trace.md").read_text())
    # ---- the large scan's tail
    out = ["# Round 4 — what follows the large scan: launch timelines and per-kernel tables (1.25 M x 768, top-10)\n",
           "`scripts/trace_hi_tail.sh` (rocprofv3 --kernel-trace around `scripts/ab_hi_tail.py`, 6 searches per case).  `tail1` = round 4's chain "
           "(default), `tail0` = round 3's (`AMDR_DENSE_HI_TAIL=0`).  Timelines: the last 12 dispatches of the run in start order (start offset, "
           "duration, gap to the end of the previous dispatch, us).\n"]
    for case in ("tail64", "tail256"):
        d = F / case
        if not d.exists():
            continue
        for t in ("1", "0"):
            run = d / f"run{t}.json"
            line = ""
            if run.exists():
                for ln in run.read_text().splitlines():
                    if ln.startswith("{"):
                        o = json.loads(ln)
                        line = (f"B = {o['B']}: scan {o['scan_ms_per_search']} ms in {o['scan_launches_per_search']} launch(es), stream "
                                f"{o['stream_ms']} ms per search, behind the scan {o['tail_us']} us (under the profiler), counters {o['counters']}")
            out.append(f"\n## {case}, tail {t} ({'round 4' if t == '1' else 'round 3'})\n{line}\n")
            tl = d / f"timeline_tail{t}.md"
            if tl.exists():
                out.append("\n" + tl.read_text())
            kt = d / f"kt_tail{t}.md"
            if kt.exists():
                out.append("\n" + "\n".join(kt.read_text().splitlines()[:16]) + "\n")
    (P / "r04_dense_tail.md").write_text("\n".join(out))
    # ---- PMC
    rec = json.loads((P / "pmc_traffic.json").read_text())
    lines = ["# Round 4 — PMC passes (rocprofv3, separate runs, --kernel-trace only beside --pmc)\n",
             "## 1. HBM / fabric bytes per launch of every bench object (`scripts/pmc_traffic.py` -> `profiles/pmc_traffic.json`)",
             "One `--pmc FETCH_SIZE` pass and one `--pmc WRITE_SIZE` pass per object (together the two derived counters exceed the TCC counter "
             "slots of a pass: rocprofv3 error 38), program directly after `--`.  bytes = 2 x FETCH_SIZE KiB x 1024 (gfx950 counts 64 B per "
             "128-B request of a wide coalesced read, MI355X_MICROARCH.md) + WRITE_SIZE KiB x 1024, means over the dispatches of the object's "
             "step.  `sources` = sha256 over the kernel sources the object runs on: `bench.py` reports an entry as `roofline.traffic` only "
             "while it matches.\n",
             "| object | kernel | grid | dispatches | FETCH_SIZE KiB | WRITE_SIZE KiB | bytes per launch | sources |", "|---|---|---|---|---|---|---|---|"]
    for key, e in rec.items():
        if not isinstance(e, dict):
            continue
        for kn, kv in e["kernels"].items():
            lines.append(f"| {key} | `{kn}` | {kv['grid']} | {kv['dispatches']} | {kv['fetch_kib']:,.0f} | {kv['write_kib']:,.0f} | "
                         f"{kv['bytes_per_launch'] / 1e6:,.1f} MB | {e['sources']} |")
        if len(e["kernels"]) > 1:
            lines.append(f"| {key} | **sum** | | | | | **{e['bytes_per_launch'] / 1e6:,.1f} MB** | |")
    lines.append("\nReading: the scans read their matrix once (10 M x 768: 30.72 GB; 1.0000-1.0015 x, the excess is the query tile per block "
                 "and the candidate lists); the 1/8-shard proxy 4.09 GB for 3.84 (sample 0.13 + re-scoring 0.08 beside the scan's 3.875); "
                 "the 256-query search = four scans in one launch, 15.9 GB for 15.36.  MaxSim per UCC-en batch: pass 1 0.45 GB for a 24.3-MB "
                 "hi image read by 146 query groups (3.5 GB of LDS-DMA fills: 87 % L2 hits), re-scoring 0.35 GB (round 3: 0.93), split "
                 "0.04 GB (18 MB of query images written).  The headline's first pass (split + scores): 0.342 GB per launch — 115 MB of "
                 "queries read, 57 MB of fp16 fragments written and read back (once per query group: the chunk groups of a query "
                 "group run on one XCD), 91 MB of scores written; its second pass 0.33 GB (scores, queries and BM25 lists in, "
                 "dense lists and fusion outputs out).\n")
    for title, d, filt in (("## 2. SQ counters, MaxSim (`scripts/pmc_sq.sh ucc_colbert`, EXTRA=1; per launch, summed over the chip)", a.sq, "maxsim"),
                           ("## 3. SQ counters, BM25 in the headline step (`scripts/pmc_sq.sh headline r4/sq_bm25 bm25`)", a.sq_bm25, "bm25"),
                           ("## 4. SQ counters, the second pass of the headline's dense channel (`scripts/pmc_sq.sh headline r4/sq_b "
                            "dense_hi_select`; measured before the row length became a template parameter: 125 us)", a.sq_second,
                            "dense_hi_select")):
        if not d:
            continue
        lines.append(title)
        lines.append("Units: SQ_*_CYCLES and SQ_WAIT_* / SQ_ACTIVE_* in quad-cycles per wave summed over waves (x 4 = cycles), "
                     "SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over the 1 024 SIMDs, GRBM_GUI_ACTIVE in cycles summed over the 8 XCDs "
                     "(/ 8 / kernel duration = the shader clock).\n")
        lines.append("| kernel | grid | counter | dispatches | mean value |\n|---|---|---|---|---|")
        for md in sorted((ROOT / d).glob("sq*.md")):
            for ln in md.read_text().splitlines():
                if filt in ln and ln.startswith("| `"):
                    lines.append(ln)
        lines.append("")
    lines.append("Derived (MaxSim pass 1, `maxsim_hi2_ring_kernel<3>`, 1 168 UCC-en queries): SQ_INSTS_MFMA x 32 cycles = "
                 "SQ_VALU_MFMA_BUSY_CYCLES; / 1 024 SIMDs / (GRBM_GUI_ACTIVE / 8) = the matrix pipe's duty; GRBM_GUI_ACTIVE / 8 / duration = "
                 "the clock.  BM25: SQ_INSTS_VALU / SQ_WAVES = vector instructions per wave; x 4 cycles x waves / 1 024 SIMDs / clock = the "
                 "issue-bound floor (DESIGN.md 4.5, 4.6a).\n")
    (P / "r04_pmc.md").write_text("\n".join(lines))
    # ---- the one-launch serving step
    if a.hs:
        H = ROOT / a.hs
        out = ["# Round 4 — the serving call (dense + BM25 + fusion of one query) as one launch\n",
               "`scripts/trace_hybrid_small.sh` (rocprofv3 --kernel-trace around `scripts/ab_hybrid_small.py`: 591 x 384 and 1 260 x 768, 1 / 2 / 4 "
               "queries, `AMDR_HYBRID_SMALL=0` = the separate launches) and the same script un-profiled (p50 / p90 of engine call + synchronise, "
               "host cost of the call alone, hipGraph replay).\n"]
        if (H / "hs_kt.md").exists():
            out.append("## kernels (us)\n\n" + "\n".join((H / "hs_kt.md").read_text().splitlines()[:30]) + "\n")
        for name in sorted(H.glob("hs_ab4.log")):
            out.append(f"## {name.name}\n\n```\n" + "\n".join(ln for ln in name.read_text().splitlines() if ln.startswith("{")) + "\n```\n")
        (P / "r04_hybrid_small.md").write_text("\n".join(out))
    print("wrote", sorted(p.name for p in P.glob("r04_*")))


if __name__ == "__main__":
    main()
