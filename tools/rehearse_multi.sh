M="--workload kitti00_mono_1241x376_n1000 --no-cpu-baseline"
echo "== force collective (stdout only)"; timeout -k 10 300 python bench.py $M --force-collective 2>/dev/null > gpurun_out/fc.out; wc -l gpurun_out/fc.out; python -c "
import json; d=json.loads(open('gpurun_out/fc.out').read()); print(d['value'], d['value_host_inputs'], d.get('exchange'), d['hbm_gbps_per_rank'])"
echo "== gloo 2 ranks same gpu (driver's launch form)"; timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --dist-backend gloo --same-gpu $M --inputs device 2>/dev/null > gpurun_out/g2.out; wc -l gpurun_out/g2.out; python -c "
import json; d=json.loads(open('gpurun_out/g2.out').read()); print(d['value'], d['n_gpus'], d.get('exchange'))"
echo "== self launch form"; timeout -k 10 400 python bench.py --gpus 2 --dist-backend gloo --same-gpu $M --inputs device 2>/dev/null | tail -1 | cut -c1-200
