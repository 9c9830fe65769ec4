for i in 1 2 3; do
python3 scripts/probe_e2e.py
BINF_LIB_OVERRIDE=$GRAFT_REPO_ROOT/scripts/variants/libbinf_prev.so python3 scripts/probe_e2e.py
done
