#!/bin/bash
# counter passes over one bench invocation for a kernel-name substring:
#   tools/pmc_any.sh <kernel substr> "<bench args>" "<ctr ctr ...>" ["<ctr ...>" ...]     (one rocprofv3 --pmc run per group)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
pat=$1; shift
bargs=$1; shift
for ctr in "$@"; do
  d=/tmp/pmck_$(echo $ctr | tr ' ' '_' | cut -c1-60)
  rm -rf $d
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $d -- python3 bench.py --no-cpu-baseline --no-side $bargs > /tmp/pmck.log 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python3 - "$f" "$pat" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if sys.argv[2] in r['Kernel_Name']:
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        acc[(name.split('(')[0][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print(k[0], k[1], 'n=%d' % len(v), 'mean=%.6g' % (sum(v) / len(v)))
PY
  else
    echo "no output for $ctr"; tail -2 /tmp/pmck.log
  fi
done
