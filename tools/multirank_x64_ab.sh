export NB_ROOT=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
for x in 0 1; do
  if [ $x = 1 ]; then export NB_NO_X64=1; else unset NB_NO_X64; fi
  export NB_OUT=$PWD/gpurun_out/mr_$x.json
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2977$x tests/tools/multirank_worker.py > gpurun_out/mr_$x.log 2>&1
  python3 -c "
import json; d=json.load(open('$NB_OUT'))['ranks'][0]
print('NO_X64=$x', {k: (round(v['relerr_x'],12), round(v['relerr_v'],12), v['same_bits']) for k,v in d.items()})"
done
