#!/bin/bash
# HBM traffic of the stage-1 window-attention forward per launch, for the three-working-waves mapping (default at nH == 3) and the
# four-wave mapping: separate FETCH_SIZE / WRITE_SIZE passes over tools/attn_only.py (20 launches each).  Run on the GPU box.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_attn_wpb
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for wpb in 3 4; do
  for c in FETCH_SIZE WRITE_SIZE; do
    export SWIN_ATTN_FWD_WPB=$wpb
    (cd $REPO && rocprofv3 --pmc $c --output-format csv -d $OUT/w${wpb}_$c -- python3 tools/attn_only.py > $OUT/w${wpb}_$c.log 2>&1) || exit 1
  done
done
python3 - <<PY
import csv, glob, json
out = {}
for wpb in (3, 4):
    rec = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for f in glob.glob("$OUT/w%d_%s/**/*counter_collection.csv" % (wpb, c), recursive=True):
            for r in csv.DictReader(open(f)):
                if "win_attn_fwd_bf16" in r["Kernel_Name"] and r["Counter_Name"] == c:
                    vals.append(float(r["Counter_Value"]))
        rec[c + "_KB_avg"] = sum(vals) / max(1, len(vals)); rec["dispatches"] = len(vals)
    rec["hbm_bytes_per_launch_corrected"] = (2 * rec["FETCH_SIZE_KB_avg"] + rec["WRITE_SIZE_KB_avg"]) * 1024
    out["win_attn_fwd_bf16@stage1_wpb%d" % wpb] = rec
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
