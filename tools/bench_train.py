"""Throughput of BASELINE configs[4]: ATSP n=100 REINFORCE training step, 512 instances per GPU (S=100 sampled starts),
data-parallel with one flat RCCL gradient all-reduce.  A parity-test config, not the bench line.
  python tools/bench_train.py [--batch 512] [--steps 2]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_train.py
Prints one JSON line (rank 0): instances/s over all ranks, and where the step's time goes."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
import torch
import bench
from rrnco_amd.envs import ATSPEnv, ATSPGenerator
from rrnco_amd.models.rl import RRNet
from rrnco_amd.parallel import aggregate_throughput

ap = argparse.ArgumentParser()
ap.add_argument("--problem", default="atsp", choices=["atsp", "rcvrp", "rcvrptw"])
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--enc-chunk", type=int, default=512)
ap.add_argument("--dec-chunk", type=int, default=None)
args = ap.parse_args()
world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0"))
backend = os.environ.get("RR_DIST_BACKEND", "nccl")      # gloo: ranks may share a GPU (single-GPU boxes)
if backend != "nccl":
    local = local % max(torch.cuda.device_count(), 1)
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
if world > 1:
    import torch.distributed as dist
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
if args.problem == "atsp":
    pol, w = bench.make_policy(dev)
    env = ATSPEnv(generator_params=dict(num_loc=100, device=dev), check_solution=False, device=dev)
else:
    from rrnco_amd.envs import RCVRPEnv, RMTVRPEnv
    from rrnco_amd.models import RRNetPolicy
    torch.manual_seed(1234)
    pol = RRNetPolicy(env_name=args.problem, embed_dim=128, num_heads=8, num_encoder_layers=6, normalization="instance",
                      use_graph_context=False, nab_type="gating", init_embedding_kwargs=dict(sample_size=25)).to(dev)
    gp = dict(num_loc=100, device=dev)
    env = RCVRPEnv(generator_params=gp, check_solution=False, device=dev) if args.problem == "rcvrp" else RMTVRPEnv(generator_params=gp, device=dev)
pol.train()
pol.precision = os.environ.get("RR_TRAIN_PRECISION", "32")      # "16-mixed": one bf16 piece per operand in the two MLPs' backward products (opt-in)
model = RRNet(env, policy=pol)
opt = torch.optim.Adam(pol.parameters(), lr=1e-4, fused=True)
gen = torch.Generator(device=dev).manual_seed(1234 + rank)
batches = [env.generator(args.batch, generator=gen) for _ in range(args.steps + 1)]
from rrnco_amd.models import rollout as R
out = model.training_step(batches[0], optimizer=opt, world=world, enc_chunk=args.enc_chunk, dec_chunk=args.dec_chunk, seed=1, grad_clip=1.0)
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
R.TIMING = []
t0 = time.perf_counter()
for i in range(args.steps):
    out = model.training_step(batches[i + 1], optimizer=opt, world=world, enc_chunk=args.enc_chunk, dec_chunk=args.dec_chunk, seed=2 + i,
                              grad_clip=1.0)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
kern_ms = [a.elapsed_time(b) for a, b in R.TIMING]
R.TIMING = None
units, tmax = aggregate_throughput(args.batch * args.steps, dt, world > 1, dev if backend == "nccl" else torch.device("cpu"))
if rank == 0:
    line = {"config": "%s n=100 REINFORCE training step, %d instances/GPU, multistart sampling, %d GPU(s)" % (args.problem.upper(), args.batch, world),
            "metric": "trained instances/sec (REINFORCE step: sampling rollout + backward + flat all-reduce + Adam)", "value": units / tmax,
            "unit": "instances/s", "n_gpus": world, "instances_per_s": units / tmax, "ms_per_step": tmax / args.steps * 1e3, "loss": float(out["loss"]),
            "grad_norm": float(out["grad_norm"]), "peak_mem_gb": torch.cuda.max_memory_allocated() / 2**30,
            "gradient_path": "hand-written HIP backward (csrc/rr_train_dec.hip, rr_train_enc.hip, rr_train.hip) on what the sampling rollout "
                             "dumped; torch only for the init embedding and the host-side folds"}
    if args.problem == "atsp" and kern_ms:
        # the step's largest kernel is still the sampling rollout: priced like bench.py's (every product on two-piece fp16 splits)
        k_ms = sum(kern_ms) / len(kern_ms)
        steps_ = args.batch * 100 * 98          # rollouts x evaluated decode steps (the forced last move is not evaluated)
        ach = steps_ * bench.FLOP_PER_ROLLOUT_STEP / (k_ms * 1e-3) / 1e12
        peak = bench.PEAK_F16_MFMA_TFLOPS / bench.SPLIT_PRODUCTS
        line["roofline"] = {"bound": "mfma", "kernel": "k_rollout_w<7, 0, 1, true, true, false, false> (sampling rollout with the training dump)", "achieved": ach,
                            "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "kernel_ms": k_ms, "traffic": None}
    print(json.dumps(line))
if world > 1:
    dist.destroy_process_group()
