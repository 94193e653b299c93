bash scripts/canary_check.sh r07u_canary 6000
bash scripts/soak.sh r07u 200001 10
