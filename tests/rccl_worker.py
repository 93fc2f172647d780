"""RCCL smoke run on a one-GPU box: a ONE-rank process group on the `nccl` backend (= RCCL on ROCm), so that the backend's
initialisation and every device-tensor collective of resource_packing_self_play_amd.distributed -- the variable-length all-gather, the
packed-replay byte exchange (all_gather_into_tensor), the flat gradient all-reduce, the parameter broadcast -- run through RCCL once
before an 8-GPU node does.  (Several ranks on one device are refused by RCCL, so more than one rank needs more GPUs.)
usage: RANK=0 WORLD_SIZE=1 RP_DIST_FORCE=1 RP_DIST_BACKEND=nccl python rccl_worker.py"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main():
    import numpy as np
    import torch
    import torch.distributed as dist
    from resource_packing_self_play_amd import distributed as rdist
    from resource_packing_self_play_amd.replay import PackedReplay
    from test_replay_cpu import H, KW, N, W, same_examples, synthetic
    rank, world, local = rdist.init_from_env()
    assert dist.is_initialized() and dist.get_backend() == "nccl" and (rank, world) == (0, 1) and rdist.collectives_on()
    dev = torch.device("cuda", local)
    t = torch.arange(10, dtype=torch.float32, device=dev).reshape(5, 2)
    assert torch.equal(rdist.all_gather_variable(t), t)
    rep = synthetic(9, 4)
    rep_dev = PackedReplay(W, H, N, *[getattr(rep, k).to(dev) for k in ("key", "wh", "value", "sp_off", "sp_n", "sp_act", "sp_cnt", "episode", "move")])
    got = rdist.all_gather_packed(rep_dev)
    want = rep.sort_by_episode_move()
    got_cpu = PackedReplay(W, H, N, *[getattr(got, k).cpu() for k in ("key", "wh", "value", "sp_off", "sp_n", "sp_act", "sp_cnt", "episode", "move")])
    assert len(got) == 9 and all(same_examples(got_cpu, k, want, k) for k in range(9))
    assert rdist.last_exchange["bytes_received"] == rdist.last_exchange["bytes_sent"] > 0
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 2)).to(dev)

    class Wrap:
        pass
    wr = Wrap(); wr.nnet = model; wr.grad_hook = None
    rdist.attach(wr)
    wr.grad_hook.timing = []
    x = torch.arange(48, dtype=torch.float32, device=dev).reshape(8, 6) / 10
    loss = (model(x) ** 2).sum() / 8
    loss.backward()
    before = [p.grad.clone() for p in model.parameters()]
    extra = wr.grad_hook(model, (loss.detach(),))
    torch.cuda.synchronize()
    assert all(torch.equal(a, p.grad) for a, p in zip(before, model.parameters())) and abs(float(extra[0]) - float(loss)) < 1e-7
    ms = [e0.elapsed_time(e1) for e0, e1 in wr.grad_hook.timing]
    print("rccl ok: backend %s, all-reduce of %d floats %.3f ms, replay exchange %d bytes %.2f ms"
          % (dist.get_backend(), wr.grad_hook.numel + 1, ms[0], rdist.last_exchange["bytes_sent"], rdist.last_exchange["ms"]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
