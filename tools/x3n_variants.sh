# builds of the panel GEMM probe, alternating (variant flags as arguments; "-" = as shipped)
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for d in "$@"; do
  if [ "$d" = "-" ]; then f=""; else f="$d"; fi
  echo "== build: ${d}"
  X3N_SHAPES=${X3N_SHAPES:-4} timeout -k 10 200 python3 tools/x3n_probe.py $f 2>&1 | grep "^G=" | cut -c1-105
done
done
