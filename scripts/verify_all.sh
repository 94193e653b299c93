bash scripts/check.sh r07n || exit 1
bash scripts/canary_check.sh r07n_canary 2500
