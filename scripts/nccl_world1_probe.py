#!/usr/bin/env python3
"""Timing of the one-rank RCCL path of xmca_amd/dist.py (what tests/test_gpu_rule_n.py::test_rule_n_through_an_rccl_group_of_one_rank
runs): where do the seconds go - importing torch, creating the process group, the first collective, rule_n itself?"""
import json, os, socket, sys, time
t = [time.perf_counter()]
def lap(tag):
    t.append(time.perf_counter()); out[tag] = round(t[-1] - t[-2], 3)
out = {}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch; lap("import_torch")
import torch.distributed as td
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port); os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0); lap("set_device")
td.init_process_group("nccl", rank=0, world_size=1); lap("init_process_group")
x = torch.ones(4, device="cuda"); td.broadcast(x, src=0); torch.cuda.synchronize(); lap("first_broadcast")
from golden_inputs import make_input
from xmca_amd import _hip, dist
from xmca_amd.array import MCA
m = MCA(*make_input("wide_both"), handle=_hip.Handle(0)); m.solve(complexify=True); lap("model_solve")
a = m.rule_n(5, seed=1000); lap("rule_n_first")
a = m.rule_n(5, seed=1000); lap("rule_n_second")
m.rotate(5, 2); b = m.rule_n(5, seed=77); lap("rule_n_rotated")
td.destroy_process_group(); lap("destroy")
out["backend"] = "nccl (RCCL), world size 1, one MI355X"; out["rule_n_shape"] = list(a.shape)
print(json.dumps(out))
