export MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 WORLD_SIZE=2 LOCAL_RANK=0 NMP_DIST_BACKEND=gloo
RANK=1 python bench.py --gpus 2 --steps 12 --warmup 2 > gpurun_out/b2_r1.log 2>&1 &
RANK=0 python bench.py --gpus 2 --steps 12 --warmup 2 2>&1 | tail -1 | cut -c1-300
wait
