#!/bin/bash
# counter passes for one GEMM shape; prints per-dispatch counter values of the GEMM kernel
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for ctr in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_WAVES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE"; do
  d=/tmp/pmc_$(echo $ctr | tr ' ' '_')
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $d -- python3 tools/gemm_one.py "$@" > /dev/null 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if 'gemm' in r['Kernel_Name']:
        acc[(r['Kernel_Name'][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(k[0], k[1], 'n=%d' % len(v), 'last=%.4g' % v[-1], 'mean=%.4g' % (sum(v) / len(v)))
PY
  else
    echo "no output for $ctr"
  fi
done
