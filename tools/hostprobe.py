import sys, time, os
sys.path.insert(0, '/root/repo')
import torch
from bench import build_cfg
from trajsde_amd.models.model_base_mix_sde import PredictionModelSDENet
from trajsde_amd.runtime import NoiseSpec
from trajsde_amd.synth import CONFIGS, synth
spec=CONFIGS['config2']; cfg=build_cfg(spec)
dev=torch.device('cuda:0')
model=PredictionModelSDENet(**cfg, init_seed=0).eval().to(dev)
b_cpu=synth(**spec['synth'])
for ns in (1,3):
    streams=[torch.cuda.Stream() for _ in range(ns)]
    bs=[]
    for st in streams:
        with torch.cuda.stream(st): bs.append(b_cpu.to(dev))
    y0=[b.y.clone() for b in bs]
    def step(i):
        k=i%ns
        with torch.cuda.stream(streams[k]):
            bs[k].y=y0[k]; return model(bs[k], noise=NoiseSpec(seed=i))
    with torch.no_grad():
        for i in range(6): step(i)
        torch.cuda.synchronize()
        t0=time.perf_counter(); host=[]
        for i in range(30):
            a=time.perf_counter(); step(i); host.append(time.perf_counter()-a)
        torch.cuda.synchronize(); el=time.perf_counter()-t0
    host.sort()
    print(f"streams {ns}: {el/30*1e3:.3f} ms/step; host call median {host[15]*1e3:.3f} ms, p90 {host[27]*1e3:.3f}, max {host[-1]*1e3:.3f}, sum {sum(host)/30*1e3:.3f} ms/step")
