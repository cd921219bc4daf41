import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/real-routing-nco_amd")
import torch, train
train.main(["--problem", "rcvrptw", "--problem_size", "100", "--epochs", "1", "--batch_size", "64", "--train_data_size", str(64 * 12),
            "--checkpoint_dir", "/tmp/rr_dbg_ckpt", "--log_every", "1"])
