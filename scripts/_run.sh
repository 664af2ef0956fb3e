mkdir -p gpurun_out/ab
timeout 200 python bench.py --no-cpu-baseline --steps 20 --force-collective > gpurun_out/ab/fc.out 2> gpurun_out/ab/fc.err
echo "lines: $(wc -l < gpurun_out/ab/fc.out)"; tail -1 gpurun_out/ab/fc.out | cut -c1-120; head -3 gpurun_out/ab/fc.out | cut -c1-60
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 4 --no-cpu-baseline --force-collective > gpurun_out/ab/tr.out 2> gpurun_out/ab/tr.err
echo "lines: $(wc -l < gpurun_out/ab/tr.out)"; tail -1 gpurun_out/ab/tr.out | cut -c1-120
