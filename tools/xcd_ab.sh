# A/B of the XCD-aware tile order in ONE session (boxes differ by 10-40 %): bit 0 = fwd/dgrad, bit 1 = wgrad
for m in 0 1 2 3 0 3; do echo "== PSELD_GEMM_XCD=$m"; PSELD_GEMM_XCD=$m python tools/gemm_shapes.py 2>&1 | tail -17 | sed 's/hbm-floor.*//' ; done
