#!/bin/bash
# Memory-system counters for one command (run on the GPU box): tools/pmc_mem.sh TAG KERNEL_SUBSTR cmd...
set -e
TAG=$1; KSUB=$2; shift; shift
OUT=gpurun_out/pmcm_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for grp in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" \
           "TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum TCC_WRITEBACK_sum TCC_NORMAL_EVICT_sum" \
           "GRBM_GUI_ACTIVE TCC_BUSY_avr TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT -o g$i -- "$@" > $OUT/g$i.log 2>&1 || echo "group $i failed"
done
python - "$OUT" "$KSUB" <<'PY'
import csv, sys, collections, glob
out, ksub = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(f'{out}/g*_counter_collection.csv')):
    agg = collections.defaultdict(float); disp = set()
    for r in csv.DictReader(open(f)):
        if ksub in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value']); disp.add(r['Dispatch_Id'])
    n = max(1, len(disp))
    print(' '.join(f'{c}={v / n:.5g}' for c, v in sorted(agg.items())), f'(n={n})')
PY
rm -f $OUT/*_kernel_trace.csv
