"""Worker scripts shared by the CPU (gloo) and GPU (RCCL) tests of the multi-rank path."""

ONE_RANK_GROUP = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
from diart_amd import distributed as D
import torch.distributed as dist
backend = sys.argv[2]
dev = torch.device("cuda", 0) if backend == "nccl" else torch.device("cpu")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(D.free_port()))
if backend == "nccl":
    torch.cuda.set_device(0)
dist.init_process_group(backend, rank=0, world_size=1)
from diart_amd.synth import synth_segmentation_state
ref = synth_segmentation_state()
got = D.broadcast_state(ref, D.state_spec(ref), dev)          # a real collective on a group of one
assert all(torch.equal(got[k].float(), ref[k].float()) for k in ref)
el = D.timed_max_over_ranks(lambda: None, dev if backend == "nccl" else None)   # barrier + all_reduce(MAX)
assert 0 <= el < 5
assert D.gather_counts([3.0], dev) == [[3.0]]
print("group of one ok:", dist.get_backend(), flush=True)
dist.destroy_process_group()
'''
