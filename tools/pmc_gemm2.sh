#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" "SQ_ACTIVE_INST_SCA SQ_INSTS_SALU" "SQ_THREAD_CYCLES_VALU SQ_CYCLES"; do
  d=/tmp/pmc2_$(echo $ctr | tr ' ' '_')
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $d -- python3 tools/gemm_one.py "$@" > /tmp/pmc2.log 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if 'gemm' in r['Kernel_Name']:
        acc[(r['Kernel_Name'][:30], r['Counter_Name'])].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(k[0], k[1], 'n=%d' % len(v), 'mean=%.4g' % (sum(v) / len(v)))
PY
  else
    echo "no output for $ctr"; tail -3 /tmp/pmc2.log
  fi
done
