#!/bin/bash
# Counter passes for any command, summarised per (kernel, grid, workgroup) -- run on the GPU box:
#   tools/pmc_run.sh TAG KERNEL_SUBSTRING -- python3 tools/microbench.py wgrad
# Passes are separate rocprofv3 runs (SQ has 8 slots; FETCH_SIZE / WRITE_SIZE do not fit one TCC pass together).
TAG=$1; PAT=$2; shift 3
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum"; do
  i=$((i+1))
  (cd $REPO && rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- "$@" > $OUT/p$i.log 2>&1)
done
(cd $REPO && rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- "$@" > $OUT/kt.log 2>&1)
python3 - <<PY
import csv, glob, collections
pat = "$PAT"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            key = (r['Kernel_Name'][:70], r['Grid_Size'], r['Workgroup_Size'])
            acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
dur = collections.defaultdict(list)
for f in glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            g = str(int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])) if 'Grid_Size_X' in r else r.get('Grid_Size', '?')
            w = str(int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z'])) if 'Workgroup_Size_X' in r else r.get('Workgroup_Size', '?')
            dur[(r['Kernel_Name'][:70], g, w)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
with open("$OUT/summary.txt", "w") as o:
    o.write("# command: $*\n# per launch averages; FETCH_SIZE/WRITE_SIZE in KiB as rocprofv3 reports them (gfx950: FETCH_SIZE counts 64 B per 128-B request -> double it for wide coalesced reads)\n")
    for k, d in sorted(acc.items()):
        t = dur.get(k)
        o.write(f"{k[0]}  grid={k[1]} wg={k[2]}" + (f"  avg_us={sum(t)/len(t):.1f} (n={len(t)})" if t else "") + "\n")
        for c, v in sorted(d.items()):
            o.write(f"  {c:28s} {sum(v)/len(v):16.1f}  (n={len(v)})\n")
print(open("$OUT/summary.txt").read())
PY
