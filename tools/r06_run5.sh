# round 6: tile shapes of the batched image encode's GEMMs (M = B x 257), micro-benchmark only
SK="$((2 | (4 << 8)))"
for M in 514 771 1028 2056; do
  A=""
  for arm in 0 32 34; do A="$A $M 1408 6144 0 $((SK | (arm << 16)))"; done          # fc2, 4 K slices + reduce
  for arm in 0 34 35 38 39; do A="$A $M 6144 1408 5 $arm"; done                       # fc1 (GELU, fp16 output)
  for arm in 0 34 35; do A="$A $M 4224 1408 0 $arm"; done                             # qkv
  for arm in 0 31 23 32 34; do A="$A $M 1408 1408 2 $arm"; done                       # proj (+ residual)
  python tools/timeline_gemm.py $A 2>&1 | grep "us per launch"
done
