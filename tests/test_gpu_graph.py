"""`-m gpu`: the ATSP hot path (reset, encoder, fused rollout, reward) captures into one hipGraph and replays with identical tours:
every launcher is asynchronous on the caller's stream, none allocates through the runtime, and the deferred range guard reads
nothing back while a capture is in progress."""
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_hot_path_captures_into_a_hip_graph_and_replays_identically():
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import ATSPEnv
    from rrnco_amd.models.transforms import StateAugmentation
    fx = H.load_fixture("atsp_n20_b4_pomo")
    pol = H.make_policy(H.atsp_weights(fx)).eval()
    env = ATSPEnv(generator_params=dict(num_loc=fx["N"]), check_solution=False)
    st = {k: v.cuda() for k, v in H.fixture_state(fx).items()}
    B = st["locs"].shape[0]
    sidx = fx["sample_idx"].cuda().repeat(8, 1, 1).contiguous()

    def step():
        td = StateAugmentation(augment_fn="dihedral8", no_aug_coords=False)(TensorDict(dict(st), batch_size=[B]))
        td = env.reset(td)
        td.set("sample_idx", sidx)
        return pol(td, env, phase="val", decode_type="multistart_greedy", num_starts=fx["S"], return_actions=True)

    with torch.no_grad():
        ref = step()
        torch.cuda.synchronize()
        g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
        with torch.cuda.stream(s):
            step()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                out = step()
        torch.cuda.synchronize()
        out["actions"].zero_(); out["reward"].zero_()
        g.replay()
        torch.cuda.synchronize()
    assert torch.equal(out["actions"], ref["actions"]) and torch.equal(out["reward"], ref["reward"])
    pol.check_range()          # the word the captured call left pending is clean
