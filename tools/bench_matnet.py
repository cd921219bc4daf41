"""MatNet baseline encoder on the headline shape: ATSP n=100, B=512 x 8 augmentations, 256 / 16 heads / 5 layers
(configs/experiment/matnet.yaml).  Prints the forward time and the MFMA / VALU work it corresponds to; run under
rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
import torch
from rrnco_amd.baselines import MatNetEncoder

dev = torch.device("cuda")
Bp, N, E, H, FF, LAYERS = int(os.environ.get("BP", 4096)), int(os.environ.get("N", 100)), 256, 16, 512, 5
torch.manual_seed(1234)
enc = MatNetEncoder(embed_dim=E, num_heads=H, num_layers=LAYERS, env_name="atsp").to(dev).eval()
g = torch.Generator(device=dev).manual_seed(1)
td = {"distance_matrix": torch.rand(Bp, N, N, device=dev, generator=g)}
ridx = torch.rand(Bp, N, device=dev, generator=g).argsort(dim=1)
for _ in range(2):
    enc(td, rand_idx=ridx)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 3
a.record()
for _ in range(reps):
    enc(td, rand_idx=ridx)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / reps
gemm = 2 * N * (4 * E * E + 2 * E * FF)            # Q, K|V, out_proj, W1, W2 per instance side
attn = 2 * 2 * N * N * E                            # S = Q K^T and O = P V over all heads
mixer = N * N * H * 16 * 7                          # 2 fma + max + fma per hidden unit
tot_mfma = (gemm + attn) * 2 * Bp * LAYERS
print(f"MatNet encoder Bp={Bp} N={N}: {ms:.2f} ms per forward ({Bp / 8 / ms * 1e3:.0f} instances/s at x8 aug); "
      f"MFMA work {tot_mfma / 1e12:.2f} TFLOP -> {tot_mfma / ms / 1e9:.1f} TFLOP/s ({tot_mfma / ms / 1e9 / 157.3 * 100:.0f} % of the fp32 MFMA peak); "
      f"mixer VALU work {mixer * 2 * Bp * LAYERS / 1e12:.2f} TFLOP")

# ---- the whole baseline policy (encoder + step-wise attention-model decode), greedy, POMO N starts x 8 augmentations
if os.environ.get("POLICY", "1") != "0":
    import time
    from rrnco_amd import TensorDict
    from rrnco_amd.baselines import MatNetPolicy
    from rrnco_amd.envs import ATSPEnv, ATSPGenerator
    from rrnco_amd.models.transforms import StateAugmentation
    B = Bp // 8
    torch.manual_seed(1234)
    pol = MatNetPolicy(env_name="atsp").to(dev).eval()
    env = ATSPEnv(generator_params=dict(num_loc=N, device=dev), check_solution=False, device=dev)
    inst = ATSPGenerator(num_loc=N, device=dev)(B, generator=torch.Generator(device=dev).manual_seed(1))

    def step():
        td = StateAugmentation(augment_fn="dihedral8", no_aug_coords=False)(TensorDict(dict(inst.items()), batch_size=[B]))
        return pol(env.reset(td), env, phase="val", decode_type="multistart_greedy", num_starts=N, rand_idx=ridx)
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = step(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"MatNetPolicy ATSP n={N} B={B} x8 aug, S={N} greedy (step-wise decode): {dt * 1e3:.1f} ms per batch -> {B / dt:.0f} instances/s; "
          f"mean best cost {float(-out['reward'].view(N, -1).max(0).values.view(8, B).max(0).values.mean()):.4f}")
