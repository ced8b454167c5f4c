import sys, os
R=os.environ.get('GRAFT_REPO_ROOT','/root/repo'); sys.path.insert(0,R); sys.path.insert(0,R+'/tools')
import torch
from sorrel_amd import _native as N
from sorrel_amd.engine import GridEngine
from tests import helpers as H
from _warm import timed_us
def tag(h,w,a,r):
    d, spec = H.load_golden("tag_9x9"); ws = H.world_spec(spec)
    ws.height, ws.width, ws.num_agents, ws.vision_radius, ws.agent_type = h, w, a, r, [ws.agent_type[0]]*a
    return ws
for name, ws in (("tag 48x48 A10 r4", tag(48,48,10,4)), ("tag 32x32 A8 r4", tag(32,32,8,4)), ("tag 40x40 A8 r3", tag(40,40,8,3))):
    for vn, o in (("auto", {}), ("whole", {"burst":1}), ("chunks", {"burst":2})):
        with N.options(**o):
            eng = GridEngine(ws, 65536, device="cuda:0")
        eng.reset(0)
        for _ in range(150): eng.step(random_actions=True)
        us = timed_us(lambda: eng.step(random_actions=True), 100)
        print(f"{name:18s} {vn:7s} {us:7.1f} us  {ws.algorithmic_bytes_per_env_step()*65536/us/1e3/8000:.2f}  {eng.launch_info().split(' group')[0]}", flush=True)
        del eng; torch.cuda.empty_cache()
