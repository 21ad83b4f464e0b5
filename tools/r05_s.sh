#!/bin/bash
# round-5 run S: dress rehearsal of bench.py's N = 2 path on a one-GPU box (MG4_BENCH_REHEARSAL=1: ranks share the GPU, gloo collectives): self-launch and the driver's launcher form
set -u
export TMPDIR=/tmp
cd /root/repo
OUT=gpurun_out/r05s
mkdir -p $OUT
MG4_BENCH_REHEARSAL=1 timeout 900 python bench.py --gpus 2 --steps 64 --warmup 4 > $OUT/self_launch_n2.json 2> $OUT/self_launch_n2.err; echo "self-launch rc=$?"; tail -3 $OUT/self_launch_n2.err | cut -c1-300
python -c "
import json;d=json.loads(open('$OUT/self_launch_n2.json').read().strip().splitlines()[-1]);print({k:d.get(k) for k in ['value','n_gpus','ms_per_step','scaling','rccl_ranks','collective_backend','launcher','load_mode','recv_load_s','weight_bcast_ms']}); print(d.get('weight_bcast')); print(str(d.get('per_rank'))[:600]); print(d.get('rehearsal'))"
MG4_BENCH_REHEARSAL=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 32 --warmup 4 --no-cpu-baseline > $OUT/external_n2.json 2> $OUT/external_n2.err; echo "external rc=$?"; tail -2 $OUT/external_n2.err | cut -c1-300
python -c "
import json;d=json.loads(open('$OUT/external_n2.json').read().strip().splitlines()[-1]);print({k:d.get(k) for k in ['value','n_gpus','ms_per_step','launcher','collective_backend','weight_bcast_ms']})"
