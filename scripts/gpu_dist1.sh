#!/bin/bash
# Dev aid (GPU box): the torch.distributed / RCCL code path of bench.py with a one-rank group.
MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 600 python bench.py --gpus 1 --force-dist --steps 100 --warmup 10 --no-cpu-baseline --no-extra --no-per-frame --repeats 3 --large-frames 0 --chunks 4,1,2,8 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('value %.3e' % d['value'], 'with_track_allgather', d['with_track_allgather'])"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29545 bench.py --gpus 1 --force-dist --steps 50 --warmup 5 --no-cpu-baseline --no-extra --no-per-frame --repeats 3 --large-frames 0 2>/dev/null | tail -1 | cut -c1-200
