python -m pytest tests -q -x -m gpu > gpurun_out/t.txt 2>&1; tail -5 gpurun_out/t.txt
for N in 2; do
CDR_BENCH_SHARED_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2950$N bench.py --gpus $N --steps 3 --warmup 1 --users 2000001 --items-per-domain 500000 --batch 65536 > gpurun_out/b_dim$N.json 2> gpurun_out/b_dim$N.err; echo rc=$?; tail -3 gpurun_out/b_dim$N.err; cut -c1-600 gpurun_out/b_dim$N.json
done
