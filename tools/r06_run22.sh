for KS in 0 2 4 1; do echo "== MINIGPT4_F16_KS=$KS"; MINIGPT4_F16_KS=$KS python bench_prefill.py --reps 3 2>/dev/null | tail -1 | cut -c1-300; done
