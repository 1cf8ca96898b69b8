#!/bin/bash
# SQ issue/wait counters for one command (run on the GPU box):  tools/pmc_sq.sh TAG python tools/northstar_conv.py 10
set -e
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT -o sq1 -- "$@" > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT -o sq2 -- "$@" > $OUT/sq2.log 2>&1
python - "$OUT" <<'PY'
import csv, sys, collections
out = sys.argv[1]
for f in ('sq1', 'sq2'):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    try:
        rows = list(csv.DictReader(open(f'{out}/{f}_counter_collection.csv')))
    except Exception as e:
        print(f, 'missing', e); continue
    seen = set()
    for r in rows:
        k = r['Kernel_Name'][:60]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        key = (k, r['Dispatch_Id'])
        if key not in seen:
            seen.add(key); cnt[k] += 1
    for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:12]:
        print(f'{k:60s} n={cnt[k]:5d} ' + ' '.join(f'{c}={v / cnt[k]:.4g}' for c, v in sorted(d.items())))
PY
rm -f $OUT/*_kernel_trace.csv
