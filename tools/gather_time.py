import os, sys, time, torch, importlib.util
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29544"); os.environ.setdefault("RANK","0"); os.environ.setdefault("WORLD_SIZE","1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda",0))
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
spec=importlib.util.spec_from_file_location("mg", os.path.join(ROOT,"vits.cpp_amd","multi_gpu.py")); mg=importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
B,cap=64,262438
pcm=torch.randn(B,cap,device="cuda"); lengths=torch.randint(40000,75000,(B,),device="cuda",dtype=torch.int64)
for _ in range(3): mg.gather_pcm(pcm,lengths)
torch.cuda.synchronize()
for name,fn in [("gather_pcm", lambda: mg.gather_pcm(pcm,lengths))]:
    t0=time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); print(name, (time.perf_counter()-t0)/10*1e3, "ms")
import numpy as np
l=np.random.randint(40000,75000,B).astype(np.int64)
t0=time.perf_counter()
for _ in range(10): x=torch.from_numpy(l).cuda()
torch.cuda.synchronize(); print("lengths h2d", (time.perf_counter()-t0)/10*1e3)
dist.destroy_process_group()
