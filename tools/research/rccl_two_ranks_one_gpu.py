# two ranks sharing GPU 0: does the edge-sharded RCCL path run at all on this box?
import os, sys
sys.path.insert(0, os.getcwd())
import torch, torch.distributed as dist, torch.multiprocessing as mp
def worker(rank, world):
    os.environ["MASTER_ADDR"]="127.0.0.1"; os.environ["MASTER_PORT"]="29533"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from toyslam_amd import synth
    from toyslam_amd.optimizer import HipOptimizer
    g = synth.make(3000, 10, seed=1)
    o = HipOptimizer(device=0, rank=rank, world=world, pcg_rel_tol=1e-10)
    uid = [o.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, 0)
    try:
        o.comm_init(uid[0])
        o.set_graph(g)
        r = o.optimize(3)
        print("rank", rank, "chi2", r["chi2"], "cg", r["cg_iters"], flush=True)
    except Exception as e:
        print("rank", rank, "FAILED:", e, flush=True)
    dist.barrier()
if __name__ == "__main__":
    mp.spawn(worker, args=(2,), nprocs=2, join=True)
