for CT in 2 1; do for W in 512 768 1024 1536 2048 3072 4096; do
BINF_POLY_GRAD_CT=$CT BINF_POLY_GRAD_WGS=$W python3 scripts/probe_poly_grad.py 2>/dev/null
done; done
