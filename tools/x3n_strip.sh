# strip-down of the whole-N panel GEMM probe: one build per stripped ingredient (wrong results in those; timing only)
cd $GRAFT_REPO_ROOT
for d in "" -DPANEL_PROBE_NOMFMA -DPANEL_PROBE_NOASTORE -DPANEL_PROBE_NOREAD -DPANEL_PROBE_NOCSTORE -DPANEL_PROBE_NOBLOAD -DPANEL_PROBE_NOALOAD "-DPANEL_PROBE_NOMFMA -DPANEL_PROBE_NOREAD" "-DPANEL_PROBE_NOASTORE -DPANEL_PROBE_NOREAD -DPANEL_PROBE_NOCSTORE -DPANEL_PROBE_NOBLOAD -DPANEL_PROBE_NOALOAD" $X3N_EXTRA; do
  echo "== build: ${d:-full}"
  X3N_SHAPES=${X3N_SHAPES:-3} timeout -k 10 200 python3 tools/x3n_probe.py $d 2>&1 | grep "^G=" | cut -c1-110
done
