"""Diagnostic: time of the encoder (init embed + 6 layers + decoder cache) alone on the headline shape. Not part of the product."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
import torch
from rrnco_amd import _lib
if os.environ.get("RR_LIB"):
    _lib.LIB_PATH = _lib.LIB_PATH.replace("librrnco_hip.so", os.environ["RR_LIB"])
import bench
from rrnco_amd.envs import ATSPEnv, ATSPGenerator
from rrnco_amd.models.encoder import ATSPInitEmbedding
from rrnco_amd.models.transforms import StateAugmentation
dev = torch.device("cuda")
pol, w = bench.make_policy(dev)
env = ATSPEnv(generator_params=dict(num_loc=100, device=dev), check_solution=False, device=dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
inst_td = ATSPGenerator(num_loc=100, device=dev)(B, generator=torch.Generator(device=dev).manual_seed(1))
td = env.reset(inst_td)
td = StateAugmentation(num_augment=8)(td)
sidx = ATSPInitEmbedding.sample_indices(td["distance_matrix"], 25).contiguous()
td.set("sample_idx", sidx)
packed = pol.packed(dev)
for _ in range(2):
    pol.encoder(td, packed=packed)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(5):
    pol.encoder(td, packed=packed)
ev[1].record(); torch.cuda.synchronize()
print(f"encoder variant={os.environ.get('RR_ENC_VARIANT','1')} B'={td['distance_matrix'].shape[0]}: {ev[0].elapsed_time(ev[1])/5:.2f} ms")
