#!/usr/bin/env python3
"""HBM / fabric bytes per launch of every bench object's dominant kernel(s), THIS round: one rocprofv3 `--pmc` pass per
counter per object (FETCH_SIZE and WRITE_SIZE never share a pass: together they exceed the TCC counter slots — rocprofv3
error 38 "Request exceeds the capabilities of the hardware to collect" — and kernel trace only beside them), corrected as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts 64 B per 128-B request of a wide coalesced
read: x 2; WRITE_SIZE as is; both in KiB).  Rewrites profiles/pmc_traffic.json; every entry carries the kernel name, the
dispatch grid, the object's plan string and a fingerprint of the kernel sources it was measured on — bench.py reports an
entry as `roofline.traffic` only when kernel, plan and fingerprint are those of the run at hand.

    python3 scripts/pmc_traffic.py --out gpurun_out/r4/pmc [--only ucc_hybrid,synth10m_b64]        (GPU box, via gpurun)

rocprofv3 is started with the program itself after `--` (python3 <script>): no env / bash -c hop between the profiler and
the process that owns the GPU."""
import argparse
import csv
import hashlib
import json
import os
import re
import statistics
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


# kernel sources an object's traffic depends on: its entry is reported only while THESE files are unchanged
SOURCES = {
    "ucc_hybrid": ("dense.hip", "dense_dot.hpp", "dense_panel.hip", "dense_small_hi.hip", "common.hpp"),
    "ucc_hybrid_second_pass": ("dense.hip", "dense_dot.hpp", "dense_panel.hip", "dense_small_hi.hip", "common.hpp"),
    "dense_only_d384": ("dense.hip", "dense_dot.hpp", "dense_panel.hip", "dense_small_hi.hip", "common.hpp"),
    "ucc_colbert": ("maxsim.hip", "topk.hpp", "common.hpp"),
    "full_hybrid_rerank": ("maxsim.hip", "topk.hpp", "common.hpp"),
}
SUM_GRIDS = ("full_hybrid_rerank",)
SCAN_SOURCES = ("dense.hip", "dense_dot.hpp", "dense_hi.hip", "dense_mfma.hip", "topk.hpp", "common.hpp")


def sources_fingerprint(key: str) -> str:
    h = hashlib.sha256()
    for name in SOURCES.get(key, SCAN_SOURCES):
        h.update(name.encode())
        h.update((ROOT / "legal-rag_amd" / "csrc" / name).read_bytes())
    return h.hexdigest()[:16]


# object -> (command after `python3`, kernels whose traffic is summed, key of the plan string in the command's JSON line)
OBJECTS = {
    "ucc_hybrid": (["bench.py", "--steps", "5", "--warmup", "1", "--windows", "1", "--no-cpu-baseline", "--no-extras"],
                   ["dsh_scores_kernel", "dsh_split_queries_kernel"], ("roofline", "plan")),
    "ucc_hybrid_second_pass": (["bench.py", "--steps", "5", "--warmup", "1", "--windows", "1", "--no-cpu-baseline", "--no-extras"],
                               ["dense_hi_select_fuse_kernel", "bm25_score_topk_kernel"], ("roofline", "plan")),
    "dense_only_d384": (["bench.py", "--only", "dense_only_d384", "--steps", "5"],
                        ["dsh_scores_kernel", "dsh_split_queries_kernel"],
                        ("dense_only_d384", "tiled_batch", "roofline", "plan")),
    "ucc_colbert": (["bench.py", "--only", "ucc_colbert", "--steps", "5"],
                    ["maxsim_hi2_ring_kernel", "maxsim_select_kernel", "maxsim_rescore_ring_kernel", "maxsim_rescore_kernel",
                     "maxsim_items_kernel", "maxsim_split_queries_kernel", "maxsim_final_topk_kernel"],
                    ("ucc_colbert", "roofline", "kernel")),
    "full_hybrid_rerank": (["bench.py", "--only", "full_hybrid_rerank", "--steps", "5"],
                           ["maxsim_hi2_ring_kernel", "maxsim_select_kernel", "maxsim_rescore_ring_kernel",
                            "maxsim_rescore_kernel", "maxsim_items_kernel", "maxsim_split_queries_kernel",
                            "maxsim_final_topk_kernel"], ("full_hybrid_rerank", "roofline", "kernel")),
    "synth10m_b1": (["scripts/run_dense_once.py", "10000000", "1", "768", "3"], ["dense_scan_topk_kernel"], None),
    "synth10m_b4": (["scripts/run_dense_once.py", "10000000", "4", "768", "3"], ["dense_scan_topk_kernel"], None),
    "synth10m_b8": (["scripts/run_dense_once.py", "10000000", "8", "768", "3"], ["dense_hi_tilemax_kernel<12, true>"], None),
    "synth10m_b32": (["scripts/run_dense_once.py", "10000000", "32", "768", "3"], ["dense_hi_tilemax_kernel<12, true>"], None),
    "synth10m_b64": (["scripts/run_dense_once.py", "10000000", "64", "768", "3"], ["dense_hi_tilemax_kernel<12, true>"], None),
    "synth7500k_d1024_b48": (["scripts/run_dense_once.py", "7500000", "48", "1024", "3"],
                             ["dense_hi_tilemax_kernel<16, true>"], None),
    "shard8_proxy_b64": (["scripts/run_dense_once.py", "1250000", "64", "768", "5"],
                         ["dense_hi_tilemax_kernel<12, true>", "dense_hi_tilemax_kernel<12, false>",
                          "dense_rescore_tiles_kernel"], None),
    "shard8_proxy_b256": (["scripts/run_dense_once.py", "1250000", "256", "768", "5"],
                          ["dense_hi_tilemax_kernel<12, true>", "dense_hi_tilemax_kernel<12, false>",
                           "dense_rescore_tiles_kernel"], None),
}


class GridKey(str):
    """A dispatch grid as rocprofv3 prints it, ordered by its thread count (equally frequent grids: the larger launch is
    the object's step, the smaller its warm-up or un-tiled twin)."""

    def _n(self):
        try:
            return int(self)
        except ValueError:
            return -1

    def __lt__(self, other):
        return self._n() < GridKey(other)._n()

    def __gt__(self, other):
        return self._n() > GridKey(other)._n()


def counter_rows(csv_path):
    rows = list(csv.DictReader(open(csv_path)))
    out = {}
    for r in rows:
        name = re.sub(r"^void ", "", r["Kernel_Name"])
        name = name.replace("amdr::", "")
        name = re.sub(r"\(.*", "", name)
        out.setdefault((name, r.get("Grid_Size", "")), []).append(float(r["Counter_Value"]))
    return out


def run_pass(cmd, counter, outdir, timeout):
    outdir.mkdir(parents=True, exist_ok=True)
    full = ["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", str(outdir), "--output-format", "csv", "--",
            sys.executable] + [str(ROOT / cmd[0])] + cmd[1:]
    env = dict(os.environ, TMPDIR="/tmp")
    p = subprocess.run(full, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
    (outdir / "stdout.log").write_text(p.stdout)
    (outdir / "stderr.log").write_text(p.stderr[-20000:])
    found = sorted(outdir.rglob("*counter_collection.csv"))
    return p.returncode, (found[0] if found else None), p.stdout


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=str(ROOT / "gpurun_out" / "r4" / "pmc"))
    ap.add_argument("--only", default="")
    ap.add_argument("--timeout", type=int, default=240)
    ap.add_argument("--round", type=int, default=4)
    a = ap.parse_args()
    out = Path(a.out).resolve()  # rocprofv3 runs with cwd=/tmp: absolute paths
    out.mkdir(parents=True, exist_ok=True)
    want = [k for k in OBJECTS if not a.only or k in a.only.split(",")]
    target = ROOT / "profiles" / "pmc_traffic.json"
    try:
        rec = json.loads(target.read_text())
    except Exception:  # noqa: BLE001
        rec = {}
    rec["_comment"] = ("HBM / fabric bytes per launch from rocprofv3 PMC passes (scripts/pmc_traffic.py: one --pmc pass per "
                       "counter per object; FETCH_SIZE x 2 per the gfx950 correction of MI355X_MICROARCH.md, + WRITE_SIZE); "
                       "bench.py reports an entry only when its kernel, plan and `sources` fingerprint are those of the run")
    report = []
    for key in want:
        cmd, kernels, plan_path = OBJECTS[key]
        per = {}
        plan = None
        ok = True
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            try:
                rc, path, stdout = run_pass(cmd, counter, out / key / counter.lower(), a.timeout)
            except subprocess.TimeoutExpired:
                rc, path, stdout = -9, None, ""
            print(f"[pmc] {key} {counter}: rc={rc} csv={'yes' if path else 'no'}", flush=True)
            if rc != 0 or path is None:
                ok = False
                break
            if plan_path and plan is None:
                for line in stdout.splitlines():
                    if line.startswith("{"):
                        o = json.loads(line)
                        for p_ in plan_path:
                            o = o.get(p_) if isinstance(o, dict) else None
                        plan = o
            rows = counter_rows(path)
            for (name, grid), vals in rows.items():
                for kname in kernels:
                    if name.startswith(kname.split("<")[0]) and (("<" not in kname) or name.startswith(kname)):
                        # the step's own launches: the most frequent grid of that kernel
                        per.setdefault(kname, {}).setdefault(counter, []).append(
                            (len(vals), GridKey(grid), statistics.mean(vals)))
        if not ok:
            report.append((key, "FAILED"))
            continue
        entry = {"round": a.round, "sources": sources_fingerprint(key), "plan": plan, "kernels": {}}
        total = 0.0
        steps = min((max(c[0] for c in cs["FETCH_SIZE"]) for cs in per.values() if "FETCH_SIZE" in cs), default=1)
        for kname, cs in per.items():
            if "FETCH_SIZE" not in cs:
                continue
            f = max(cs["FETCH_SIZE"])  # (dispatches, grid, mean KiB)
            w = max(cs.get("WRITE_SIZE", [(0, f[1], 0.0)]))
            if key in SUM_GRIDS:
                # a step of this object launches the kernel once per LANGUAGE — two grids equally often, or one grid twice
                # as often as the step count: per step = all dispatches' bytes / steps
                fs, ws = cs["FETCH_SIZE"], cs.get("WRITE_SIZE", [])
                f = (steps, "+".join(sorted(x[1] for x in fs)), sum(x[0] * x[2] for x in fs) / steps)
                w = (steps, f[1], sum(x[0] * x[2] for x in ws) / steps)
            b = f[2] * 1024 * 2 + w[2] * 1024
            entry["kernels"][kname] = {"grid": f[1], "dispatches": f[0], "fetch_kib": round(f[2], 2),
                                       "write_kib": round(w[2], 2), "bytes_per_launch": round(b)}
            total += b
        if not entry["kernels"]:
            report.append((key, "no matching kernel"))
            continue
        first = kernels[0] if kernels[0] in entry["kernels"] else next(iter(entry["kernels"]))
        entry["kernel"] = first.split("<")[0]
        entry["grid"] = entry["kernels"][first]["grid"]
        entry["bytes_per_launch"] = round(total)
        rec[key] = entry
        report.append((key, f"{total / 1e6:.1f} MB over {list(entry['kernels'])}"))
        target.write_text(json.dumps(rec, indent=1))
        (out / "pmc_traffic.json").write_text(json.dumps(rec, indent=1))  # gpurun merges gpurun_out/ back, not profiles/
    (out / "report.txt").write_text("\n".join(f"{k}: {v}" for k, v in report) + "\n")
    print("\n".join(f"{k}: {v}" for k, v in report))


if __name__ == "__main__":
    main()
