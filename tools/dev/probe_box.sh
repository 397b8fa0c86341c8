#!/bin/bash
# what the GPU box offers (builder-side probe; not a test)
mkdir -p gpurun_out
{
echo "== nproc"; nproc
echo "== mem"; free -g | head -2
echo "== disk"; df -h . /tmp /dev/shm 2>/dev/null
echo "== cpu"; lscpu | grep -E "Model name|Socket|Thread|Core" 
echo "== gpus"; rocm-smi --showproductname 2>/dev/null | head -20
echo "== valu_peak"; ./mecat_amd/bin/valu_peak 20000
echo "== rccl dup gpu"
cat > /tmp/dup.py <<'PY'
import os, torch, torch.distributed as dist
r = int(os.environ["RANK"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    x = torch.ones(4, device="cuda") * (r + 1)
    dist.all_reduce(x)
    torch.cuda.synchronize()
    print("rank", r, "allreduce on shared GPU ok", x.tolist())
except Exception as e:
    print("rank", r, "FAILED", repr(e)[:400])
PY
timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 /tmp/dup.py 2>&1 | tail -15
echo "== torch rccl libs"; python - <<'PY'
import torch, os
d = os.path.join(os.path.dirname(torch.__file__), "lib")
print([f for f in os.listdir(d) if "rccl" in f or "nccl" in f])
PY
ls -la /opt/rocm/lib/librccl* 2>/dev/null
} > gpurun_out/probe_box.txt 2>&1
cat gpurun_out/probe_box.txt
