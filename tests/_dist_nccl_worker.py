"""Worker of tests/test_dist_nccl_gpu.py (launched by torch.distributed.run, one process per visible GPU):
UFM-tiny through ``ufm_amd.dist`` on the REAL backend ("nccl" = RCCL): the sharded + gathered result must equal
the unsharded result of the same inputs bit for bit, for the synchronous and the asynchronous ring form, and also
when there are fewer pairs than ranks."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    backend = os.environ.get("UFM_DIST_BACKEND", "nccl")
    if backend == "gloo":  # several ranks SHARING one GPU (the lease boxes have one): device tensors gathered through gloo
        local = 0
        torch.set_num_threads(2)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if backend == "gloo":
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=dev)
    import ufm_amd
    from ufm_amd.dist import ShardedPredictor, predict_sharded
    from ufm_amd.modules import init_weights_

    model = ufm_amd.UniFlowMatchConfidence(**ufm_amd.ufm_tiny_config()).eval()
    init_weights_(model, seed=0)
    model = model.to(dev)

    def predict(s, t):
        out = model.predict_correspondences_batched(s, t)
        return out.flow.flow_output, out.covisibility.mask

    g = torch.Generator().manual_seed(7)  # the same global batch on every rank
    for n_pairs in (2 * world + 1, max(1, world - 1)):
        src = torch.randint(0, 256, (n_pairs, 56, 56, 3), dtype=torch.uint8, generator=g).to(dev)
        tgt = torch.randint(0, 256, (n_pairs, 56, 56, 3), dtype=torch.uint8, generator=g).to(dev)
        want_f, want_m = predict(src, tgt)
        want_f, want_m = want_f.clone(), want_m.clone()
        got_f, got_m = predict_sharded(predict, src, tgt)
        assert torch.equal(got_f, want_f) and torch.equal(got_m, want_m), f"rank {rank}: sharded != unsharded (n={n_pairs})"
        sp = ShardedPredictor(predict, depth=2)
        tickets = [sp.submit(src.roll(k, 0), tgt.roll(k, 0)) for k in range(2)]
        for k, tk in enumerate(tickets):
            f, m = sp.result(tk)
            assert torch.equal(f, want_f.roll(k, 0)) and torch.equal(m, want_m.roll(k, 0)), f"rank {rank}: async ring step {k}"
        sp.drain()
    torch.cuda.synchronize()
    dist.barrier()
    if rank == 0:
        print(f"DIST-{backend.upper()}-OK world={world}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
