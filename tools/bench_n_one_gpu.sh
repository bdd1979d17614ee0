#!/bin/bash
# Development helper: the N > 1 code paths of bench.py (bands + gather, stripes + reduce, batched offsets) on a ONE-GPU box, all N ranks on GPU 0
# over gloo (DTOF_BENCH_SHARE_GPU=1); the throughputs mean nothing, the image checksums must equal the N = 1 ones.
#   tools/bench_n_one_gpu.sh [N]      N = 2 (default) .. 6: a GPU box admits at most six processes on its card, so world 8 is rehearsed on the CPU
#                                     (tests/test_distributed_cpu.py, gloo, both exchanges) and with N = 4 / 6 here
N=${1:-2}
if [ "$N" -lt 2 ] || [ "$N" -gt 6 ]; then echo "N must be 2 .. 6"; exit 1; fi
export DTOF_BENCH_SHARE_GPU=1
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['config']['image_checksum'], d['config']['sharding'], d['scaling'])"; }
run() { port=$1; shift; python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $port bench.py --gpus $N "$@" 2>/dev/null; }
# a height that neither N nor N x stripe rows divides for the Domino frames (--res sets width = height: 250 = 2 * 5^3)
python bench.py --config c4 --res 250 --spp 16 --steps 2 --warmup 1 --no-cpu-baseline | line "N1 c4"
run 29544 --config c4 --res 250 --spp 16 --scaling strong --steps 2 --warmup 1 | line "N$N c4 stripes"
run 29545 --config c4 --res 250 --spp 16 --scaling strong --sharding bands --steps 2 --warmup 1 | line "N$N c4 bands"
python bench.py --config c5 --res 122 --spp 16 --steps 2 --warmup 1 --no-cpu-baseline | line "N1 c5 (4 films)"
run 29546 --config c5 --res 122 --spp 16 --scaling strong --steps 2 --warmup 1 | line "N$N c5 stripes (4 films)"
python bench.py --config c2 --res 250 --spp 64 --steps 2 --warmup 1 --no-cpu-baseline | line "N1 c2"
run 29548 --config c2 --res 250 --spp 64 --scaling strong --steps 2 --warmup 1 | line "N$N c2 bands strong"
run 29547 --steps 2 --warmup 1 | line "N$N c2 default line (weak, with its extras)"
# the REAL backend on one GPU: a one-rank RCCL group, the frame loop issues its gather / reduce as an N > 1 run does, K steps queued on one stream and waited for once
unset DTOF_BENCH_SHARE_GPU
DTOF_BENCH_FORCE_EXCHANGE=1 python bench.py --config c4 --res 250 --spp 16 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N1 rccl reduce (stripes)', d['value'], d['config']['image_checksum'], 'pipelined', d['steps_pipelined'], d['process_group'])"
DTOF_BENCH_FORCE_EXCHANGE=1 python bench.py --config c4 --res 250 --spp 16 --sharding bands --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N1 rccl gather (bands)', d['value'], d['config']['image_checksum'], 'pipelined', d['steps_pipelined'], d['process_group'])"
