"""Env base mirroring the RL4COEnvBase surface the reference's policy / test.py / rl.py call
(SURVEY.md §8b): reset, step -> {"next": td}, get_reward (runs check_solution first), get_num_starts,
select_start_nodes, attributes name / normalize / generator / check_solution."""
from __future__ import annotations

import torch

from ..tensordict_lite import TensorDict


class EnvBase:
    name = "base"

    def __init__(self, *, check_solution: bool = True, device="cuda", batch_size=None, **unused):
        self.check_solution = check_solution
        self.device = torch.device(device)
        self.batch_size = torch.Size([] if batch_size is None else batch_size)

    def to(self, device):
        self.device = torch.device(device)
        return self

    def reset(self, td=None, batch_size=None) -> TensorDict:
        if batch_size is None:
            batch_size = self.batch_size if td is None else td.batch_size
        if td is None or td.is_empty():
            td = self.generator(batch_size=batch_size)
        batch_size = [batch_size] if isinstance(batch_size, int) else list(batch_size)
        out = self._reset(td, batch_size=batch_size)
        if "done" not in out:
            out.set("done", torch.zeros((*batch_size, 1), dtype=torch.bool, device=out.device))
        return out

    def step(self, td: TensorDict) -> dict:
        return {"next": self._step(td)}

    def get_reward(self, td: TensorDict, actions: torch.Tensor):
        if self.check_solution:
            self.check_solution_validity(td, actions)
        return self._get_reward(td, actions)

    def get_num_starts(self, td):
        # rl4co.utils.ops.get_num_starts: the depot is excluded only for a hard-coded name list that does
        # not contain "rcvrp" (SURVEY App. A / D-6); RMTVRPEnv overrides it (rmtvrp/env.py:566-567)
        return td["action_mask"].shape[-1]

    def select_start_nodes(self, td, num_starts):
        num_loc = getattr(self.generator, "num_loc", 0xFFFFFFFF)
        sel = torch.arange(num_starts, device=td.device).repeat_interleave(td.shape[0]) % num_loc
        return sel if self.name in ("tsp", "atsp") else sel + 1
