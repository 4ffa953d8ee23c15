#!/bin/bash
# counter passes over the bench for one kernel-name substring: tools/pmc_kernel.sh <substr> "<ctr ctr>" "<ctr ctr>" ...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
pat=$1; shift
for ctr in "$@"; do
  d=/tmp/pmck_$(echo $ctr | tr ' ' '_')
  rm -rf $d
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $d -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > /tmp/pmck.log 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python3 - "$f" "$pat" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if sys.argv[2] in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(k, 'n=%d' % len(v), 'mean=%.5g' % (sum(v) / len(v)))
PY
  else
    echo "no output for $ctr"; tail -2 /tmp/pmck.log
  fi
done
