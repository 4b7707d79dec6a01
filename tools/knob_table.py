#!/usr/bin/env python3
"""The KBEST_* environment knobs: ONE list, checked against the sources, rendered into INTEGRATION.md section 7.

    python tools/knob_table.py            print the table
    python tools/knob_table.py --write    rewrite the block between the knobs:begin / knobs:end markers of INTEGRATION.md
    python tools/knob_table.py --check    exit 1 unless (a) every getenv("KBEST_*") of probabilisticsemslam_amd/csrc and every
                                          os.environ KBEST_* key of bench.py / engine.py is listed here and nothing listed is gone,
                                          (b) INTEGRATION.md holds exactly this table   (tests/test_abi.py runs this)

None of the knobs changes a result: they select code paths for A/B measurements and for the tests that pin every path."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (knobs, where they are read, effect) -- grouped as they appear in the table
KNOBS = [
    (["KBEST_NWAVES", "KBEST_SPEC"], "csrc", "64-row kernel: waves per matrix (4 / 8 / 12 / 16) and hypotheses split per round (default: by batch size, `choose_shape`)"),
    (["KBEST_EAGER"], "csrc", "64-row kernel: state slots per matrix for children kept in full (default 1024)"),
    (["KBEST_NO_T0"], "csrc", "64-row and general-size kernels: no a-priori thresholds (`=1`; per call: `KBEST_FLAG_NO_T0`)"),
    (["KBEST_EXACT_ROOT"], "csrc", "64-row kernel: root LAP by the reference's own sequence of augmentations (no column reduction; per call: `KBEST_FLAG_EXACT_ROOT`)"),
    (["KBEST_NO_REORDER"], "csrc", "64-row, lane-per-child and general-size kernels enumerate in the reference's column order instead of their own (per call: `KBEST_FLAG_NO_REORDER`)"),
    (["KBEST_NO_OPT", "KBEST_OPT_RHO0", "KBEST_OPT_RHO1", "KBEST_OPT_PHI", "KBEST_OPT_KAPPA", "KBEST_OPT_MINPOOL"], "csrc",
     "64-row kernel: no optimistic bounds (per call: `KBEST_FLAG_NO_OPT`); the quantile of the pool a node is split against (start, end, fraction of k over which it moves), the step of a re-split beyond its ticket's key, the pool size from which a quantile is used"),
    (["KBEST_RELAY", "KBEST_RELAY_FIRST", "KBEST_RELAY_STEP"], "csrc",
     "64-row kernel: never (0) / always (2 ... 8) enumerate a matrix in that many pieces by different workgroups (default: four pieces for batches of 1 ... 1.8 generations of resident workgroups, three up to 6.5, two up to 10 or with a cutoff; `NOTES.md` section 10.6); where the first piece hands over and how far apart the later ones (1 ... 1023, in 1/1024 of k)"),
    (["KBEST_SPLIT", "KBEST_NO_SPLIT"], "csrc", "64-row kernel: one matrix over 2 / 4 workgroups with a shared bound and a device merge (slower, kept for study; off by default) / never"),
    (["KBEST_LDS_PAD"], "csrc", "64-row kernel: extra bytes of LDS per workgroup (residency experiments)"),
    (["KBEST_SMALL_NW", "KBEST_FORCE_SMALL", "KBEST_NO_SMALL"], "csrc", "small-problem kernel: waves per problem (2 ... 16); route everything it can take to it / nothing"),
    (["KBEST_FORCE_LANE", "KBEST_NO_LANE", "KBEST_LANE_NW", "KBEST_LANE_SPEC"], "csrc", "lane-per-child kernel (<= 32 rows): take every plain batch / none; waves per problem (1 / 2 / 4); hypotheses split per round (1 ... 16)"),
    (["KBEST_FORCE_WIDE", "KBEST_WIDE_NW", "KBEST_WIDE_TILE", "KBEST_WIDE_SPEC", "KBEST_NO_WIDE_QUEUE"], "csrc",
     "general-size kernel: take every problem; waves per problem (8 / 16); cost copy in LDS (1) or HBM (0); hypotheses split per round; a batch larger than the grid strided over the workgroups instead of taken off a queue"),
    (["KBEST_NO_TINY", "KBEST_NO_BNB", "KBEST_BNB_SMALL_FROM"], "csrc", "association path: no exhaustive kernel / no bounded walk (frames then go to the fused enumeration kernel); batch size from which the bounded walk runs in 256-thread workgroups (default: CUs + 1)"),
    (["KBEST_NO_TIE"], "csrc", "no solution behind the k-th, no canonical order of exact ties (round 4's behaviour; per call: `KBEST_FLAG_NO_TIE_CHECK`)"),
    (["KBEST_PIECES", "KBEST_PIECE_PRIO"], "csrc", "`kbest_batch_f64`: pieces a large batch is sent through the GPU in (1 / 2 / 4); `=0`: all pieces' streams at one priority"),
    (["KBEST_NO_NARROW", "KBEST_HOST_THREADS"], "csrc", "`kbest_batch_f64`: int32 tables cross the link as they are (no narrow staging); host threads that widen the byte tables (default: up to 16)"),
    (["KBEST_ZC_COST"], "csrc", "`kbest_batch_f64`, registered cost blocks: `0` all pieces' uploads at once, `2` read in place by the kernels over the link (default: uploaded piece after piece on a copy stream)"),
    (["KBEST_ZC_LIMIT_KB", "KBEST_NO_POLL"], "csrc", "association host entries: largest call (bytes in + out) that runs on the pinned staging memory in place (default 65 536); wait with `hipStreamSynchronize` instead of polling the completion word"),
    (["KBEST_MULTI_WIDE", "KBEST_MULTI_WHOLE_LISTS"], "csrc", "multi-device entries: row4col travels as int32 even where bytes would do (round 5's exchange); subtree mode always exchanges the whole lists (no gains-first exchange)"),
    (["KBEST_EXACT_WAVES"], "csrc", "reference-order kernels up to 1 024 rows (`kbest_exact.hip`): waves per problem (`1`, `2`, `4`, `8`: a sweep's children side by side); unset: up to 64 rows eight where a problem has 16 columns or more or the batch has fewer than 512 problems, else one; 65 … 1 024 rows the child-solving waves of the eight: as many of 8 / 4 / 2 / 1 as a CU's LDS holds scratch for (A/B; same tables; read at every launch)"),
    (["KBEST_SHIM_REFERENCE_ORDER"], "csrc", "the reference-named C++ shims, exact ties: unset -- `kBest2D` / `kBest2DCutoff` answer as the reference does (the synchronous entry's default: a problem with an exact tie among its k + 1 best gains runs again on the reference-order kernel), `assignmentProb` / `bruteForceProb` weigh the engine's choice among a tied level; `=2` those too weigh the reference's own k best (`kbest_set_reference_order(ctx, 2)`); `=1` everything on the reference-order kernel (`KBEST_FLAG_REFERENCE_ORDER`); `=0` the engine's own rule everywhere (`KBEST_FLAG_CANONICAL_TIES`) -- the one knob that selects another (documented) answer on exact ties"),
    (["KBEST_LIB"], "python", "Python driver: file name of the library to load from the package directory (e.g. the `PROFILE=1` build)"),
    (["KBEST_BENCH_BACKEND", "KBEST_BENCH_FORCE_DIST", "KBEST_BENCH_SELF_LAUNCH", "KBEST_BENCH_WIDE_SLICES"], "python",
     "`bench.py`: `gloo` moves the ranks' slices through the host and deals the ranks to the GPUs there are (two ranks on one GPU: the one-GPU test of the world > 1 path); the distributed step with one rank; the launcher path with one rank; int32 slices instead of int8"),
]


def source_knobs():
    found = {"csrc": set(), "python": set()}
    csrc = os.path.join(ROOT, "probabilisticsemslam_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".cpp", ".hip", ".h")):
            found["csrc"].update(re.findall(r'getenv\("(KBEST_[A-Z0-9_]+)"\)', open(os.path.join(csrc, f)).read()))
    for f in (os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "probabilisticsemslam_amd", "engine.py")):
        found["python"].update(re.findall(r'environ[^\n]*?"(KBEST_[A-Z0-9_]+)"', open(f).read()))
    found["python"] -= found["csrc"]  # (bench.py SETS some of the library's knobs for its A/B legs)
    return found


def table():
    rows = ["| variable | effect |", "|---|---|"]
    for names, _, what in KNOBS:
        rows.append("| " + ", ".join(f"`{n}`" for n in names) + " | " + what + " |")
    return "\n".join(rows)


BEGIN, END = "<!-- knobs:begin (generated: python tools/knob_table.py --write) -->", "<!-- knobs:end -->"


def main():
    listed = {"csrc": set(), "python": set()}
    for names, where, _ in KNOBS:
        listed[where].update(names)
    found = source_knobs()
    ok = True
    for where in ("csrc", "python"):
        missing, stale = found[where] - listed[where], listed[where] - found[where]
        if missing:
            print(f"knobs read in the {where} sources but not listed in tools/knob_table.py: {sorted(missing)}", file=sys.stderr)
            ok = False
        if stale:
            print(f"knobs listed in tools/knob_table.py but read nowhere in the {where} sources: {sorted(stale)}", file=sys.stderr)
            ok = False
    path = os.path.join(ROOT, "INTEGRATION.md")
    txt = open(path).read()
    block = BEGIN + "\n" + table() + "\n" + END
    if "--write" in sys.argv:
        a, b = txt.index(BEGIN), txt.index(END) + len(END)
        open(path, "w").write(txt[:a] + block + txt[b:])
    elif "--check" in sys.argv:
        if block not in txt:
            print("INTEGRATION.md section 7 is not the generated table: python tools/knob_table.py --write", file=sys.stderr)
            ok = False
    else:
        print(table())
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
