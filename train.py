"""Training driver on the MI355X engine with the hyper-parameters of the reference's configs/experiment/rrnet.yaml
(Adam lr 4e-4, weight decay 1e-6, MultiStepLR [180, 195] x 0.1, batch 64, 100 000 instances per epoch, POMO shared
baseline, sampling decode, validation with x8 augmentation) — the loop Lightning runs for `train.py experiment=rrnet`.
Instances come from the synthetic generators (or a city through RealWorldSampler when --city is given); checkpoints are
written in the layout test.py / evaluate.py read (`state_dict` with `policy.` keys).  Data-parallel under
`python -m torch.distributed.run --nproc-per-node N train.py ...` (one flat gradient all-reduce per step).

  python train.py --problem atsp --epochs 2 --train_data_size 1280 --batch_size 64
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
import torch  # noqa: E402


def instance_stream_seed(seed: int, rank: int, start_epoch: int) -> int:
    """Seed of the training-instance generator: one stream per data-parallel rank, keyed by the epoch the run (re)starts at, so a
    resumed run continues with new instances instead of replaying epoch 0's and no two ranks ever draw the same batch."""
    return int(seed) + 1000 * int(rank) + 1_000_003 * int(start_epoch)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--problem", default="atsp", choices=["atsp", "rcvrp", "rcvrptw"])
    ap.add_argument("--problem_size", type=int, default=100)
    ap.add_argument("--epochs", type=int, default=200)
    ap.add_argument("--batch_size", type=int, default=64)
    ap.add_argument("--train_data_size", type=int, default=100_000)
    ap.add_argument("--val_data_size", type=int, default=1280)
    ap.add_argument("--lr", type=float, default=4e-4)
    ap.add_argument("--weight_decay", type=float, default=1e-6)
    ap.add_argument("--milestones", type=int, nargs="*", default=[180, 195])
    ap.add_argument("--gamma", type=float, default=0.1)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--grad_clip", type=float, default=1.0, help="gradient_clip_val of configs/trainer/default.yaml:6")
    ap.add_argument("--checkpoint_dir", default="checkpoints")
    ap.add_argument("--resume", default=None, help="checkpoint written by this script (or a reference .ckpt: weights only)")
    ap.add_argument("--log_every", type=int, default=50)
    o = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise RuntimeError("train.py runs on the HIP path only (no CPU fallback)")

    world, rank, local = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    backend = os.environ.get("RR_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))

    from rrnco_amd import data
    from rrnco_amd.envs import ATSPEnv, RCVRPEnv, RMTVRPEnv
    from rrnco_amd.models import RRNetPolicy
    from rrnco_amd.models.rl import RRNet
    torch.manual_seed(o.seed)                                     # same initial weights on every rank
    n = o.problem_size
    policy = RRNetPolicy(env_name=o.problem, embed_dim=128, num_heads=8, num_encoder_layers=6, normalization="instance",
                         use_graph_context=False, nab_type="gating",
                         init_embedding_kwargs=dict(sample_size=min(25, max(1, n - 5)))).to(dev)
    gp = dict(num_loc=n, device=dev)
    env = {"atsp": lambda: ATSPEnv(check_solution=False, generator_params=gp, device=dev),
           "rcvrp": lambda: RCVRPEnv(check_solution=False, generator_params=gp, device=dev),
           "rcvrptw": lambda: RMTVRPEnv(generator_params=gp, device=dev)}[o.problem]()
    # configs/experiment/rrnet.yaml:45-51: dihedral-8 augmentation of the coordinates at validation / test time
    model = RRNet(env, policy=policy, num_augment=8, augment_fn="dihedral8", no_aug_coords=False)
    opt = torch.optim.Adam(policy.parameters(), lr=o.lr, weight_decay=o.weight_decay, fused=True)   # one launch per step, same update
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=o.milestones, gamma=o.gamma)
    start_epoch = 0
    if o.resume:
        blob = torch.load(o.resume, map_location="cpu", weights_only=False)
        policy.load_state_dict(data.load_policy_state_dict(o.resume), strict=True)
        if isinstance(blob, dict) and "optimizer_states" in blob and "rrnco_amd" in blob:
            opt.load_state_dict(blob["optimizer_states"][0]); sched.load_state_dict(blob["lr_schedulers"][0])
            start_epoch = int(blob["epoch"]) + 1
            if "cpu_rng_state" in blob:              # the encoder's neighbour-sample seeds come from torch's CPU generator
                torch.set_rng_state(blob["cpu_rng_state"])
    # a resumed run continues the instance stream instead of replaying epoch 0's data: the stream is keyed by the epoch it starts at
    gen = torch.Generator(device=dev).manual_seed(instance_stream_seed(o.seed, rank, start_epoch))
    val_gen = torch.Generator(device=dev).manual_seed(o.seed + 7)     # same validation set every epoch, on every rank
    steps_per_epoch = max(o.train_data_size // (o.batch_size * world), 1)
    val_batch = env.generator(min(o.val_data_size, 4 * o.batch_size), generator=val_gen)

    for epoch in range(start_epoch, o.epochs):
        policy.train()
        t0, seen, run_loss, run_rew = time.perf_counter(), 0, 0.0, 0.0
        for it in range(steps_per_epoch):
            out = model.training_step(env.generator(o.batch_size, generator=gen), optimizer=opt, world=world,
                                      grad_clip=o.grad_clip,
                                      seed=(o.seed + epoch * steps_per_epoch + it) * world + rank)      # every rank its own sampling noise
            seen += o.batch_size
            if (it + 1) % o.log_every == 0 or it + 1 == steps_per_epoch:
                run_loss, run_rew = float(out["loss"]), float(out["reward"].mean())
                if rank == 0:
                    from rrnco_amd.parallel import allreduce_summary
                    ar = allreduce_summary()          # the step's one collective (RCCL over xGMI): device time by events, None on a single rank
                    print(f"epoch {epoch} step {it + 1}/{steps_per_epoch}  loss {run_loss:.4f}  train/reward {run_rew:.4f}  "
                          f"grad_norm {float(out['grad_norm']):.3f}  {seen * world / (time.perf_counter() - t0):.0f} inst/s"
                          + (f"  all-reduce {ar[0]:.3f} ms / {ar[1]:.1f} MB (mean of {ar[2]})" if ar else ""), flush=True)
        sched.step()
        policy.eval()
        val = model.shared_step(val_batch, phase="val")
        val_reward = float(val["max_aug_reward"].mean())
        if rank == 0:
            print(f"epoch {epoch} done  val/max_aug_reward {val_reward:.4f}  lr {sched.get_last_lr()[0]:.2e}", flush=True)
            os.makedirs(os.path.join(o.checkpoint_dir, o.problem), exist_ok=True)
            ck = {"rrnco_amd": 1, "epoch": epoch, "state_dict": {"policy." + k: v.detach().cpu() for k, v in policy.state_dict().items()},
                  "optimizer_states": [opt.state_dict()], "lr_schedulers": [sched.state_dict()],
                  "hyper_parameters": {k: v for k, v in vars(o).items()}, "val_reward": val_reward,
                  "cpu_rng_state": torch.get_rng_state()}
            torch.save(ck, os.path.join(o.checkpoint_dir, o.problem, f"epoch_{epoch:03d}.ckpt"))
            torch.save(ck, os.path.join(o.checkpoint_dir, o.problem, "last.ckpt"))
    if world > 1:
        dist.destroy_process_group()
    return val_reward


if __name__ == "__main__":
    main()
