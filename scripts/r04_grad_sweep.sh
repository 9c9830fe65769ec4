# launch-shape sweep of the whole-tile MFMA gradient kernel (chain tiles per wave x target workgroup count)
R=$GRAFT_REPO_ROOT
for CT in 2 1; do for WGS in 512 640 768 1024; do
  BINF_POLY_GRAD_CT=$CT BINF_POLY_GRAD_WGS=$WGS python3 $R/scripts/probe_poly_grad.py
done; done
