#!/bin/bash
# counter passes for the fused MLP kernel (run on the GPU box): tools/pmc_mlp.sh C fwd|bwd tag
C=${1:-96}; W=${2:-fwd}; TAG=${3:-x}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_mlp_$TAG
mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/mlp_only.py $C $W 3 > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'ts_mlp' in r['Kernel_Name']:
            acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
with open("$OUT/summary.txt", "w") as o:
    for k, d in acc.items():
        o.write(k + "\n")
        for c, v in sorted(d.items()):
            o.write(f"  {c:32s} {sum(v)/len(v):16.1f}  (n={len(v)})\n")
print(open("$OUT/summary.txt").read())
PY
