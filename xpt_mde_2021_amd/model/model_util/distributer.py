"""Data-parallel training on one node: one process per GPU, RCCL over xGMI through torch.distributed
(backend "nccl" IS RCCL on ROCm).  Replaces the reference's tf.distribute.MirroredStrategy wrapper
(model/model_util/distributer.py:5-110).

Semantics kept from the reference: the per-example losses are summed and divided by the GLOBAL batch
(losses.py:49), so the gradient reduction across replicas is a SUM; the global batch is
replicas x PER_REPLICA_BATCH (distributer.py:12); BatchNorm statistics are frozen, so nothing else is exchanged.
The collective is an all-reduce(SUM) of the flat gradient buffer (FlatParameters.grad: 42 MB fp32 for NASNet-Mobile +
PoseNetImproved; xGMI rings are per-link bound, so few large messages): in TWO pieces when the trainer cuts the
backward pass between decoder and encoder (train_val.ModelTrainerDistrib: the decoder / PoseNet piece is exchanged
asynchronously while the encoder's backward runs, all_reduce_range), else in GRAD_BUCKETS chunks after the step.
"""
import os

import torch
import torch.distributed as dist

from ...utils.util_class import WrongInputException

from ...config import opts


class DistributionStrategy:
    """Process-group holder with the reference's accessor name (`DistributionStrategy.get_strategy()`)."""
    strategy = None

    def __init__(self, backend):
        self.backend = backend
        self.rank = dist.get_rank()
        self.num_replicas_in_sync = dist.get_world_size()
        self.local_rank = int(os.environ.get("LOCAL_RANK", self.rank))

    @classmethod
    def get_strategy(cls, backend=None):
        if cls.strategy is None:
            if not dist.is_initialized():
                if "RANK" not in os.environ:
                    return None                                # single process: no strategy, like non-distributed modes
                backend = backend or os.environ.get("XPT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
                if torch.cuda.is_available():
                    local_rank, ndev = int(os.environ.get("LOCAL_RANK", 0)), torch.cuda.device_count()
                    if backend == "gloo":                      # the rehearsal: several ranks may share the one card
                        local_rank %= ndev
                    elif int(os.environ.get("LOCAL_WORLD_SIZE", "1")) > ndev:
                        raise WrongInputException(f"{os.environ.get('LOCAL_WORLD_SIZE')} RCCL ranks on a node with {ndev} "
                                                  "GPUs: RCCL needs one card per rank")
                    torch.cuda.set_device(local_rank)
                dist.init_process_group(backend=backend)
            cls.strategy = cls(dist.get_backend())
            opts.BATCH_SIZE = cls.strategy.num_replicas_in_sync * opts.PER_REPLICA_BATCH      # distributer.py:12
            if cls.strategy.rank == 0:
                print(f"[DistributionStrategy] replicas: {cls.strategy.num_replicas_in_sync}, "
                      f"global batch size: {opts.BATCH_SIZE}, backend: {cls.strategy.backend}")
        return cls.strategy

    @classmethod
    def reset(cls):
        cls.strategy = None

    # ------------------------------------------------------------------ collectives
    def all_reduce_gradients(self, flat_grad, buckets=None):
        """SUM the flat gradient buffer over all replicas, in `buckets` contiguous chunks."""
        buckets = max(int(buckets or opts.GRAD_BUCKETS), 1)
        if self.num_replicas_in_sync == 1:
            return
        n = flat_grad.numel()
        step = (n + buckets - 1) // buckets
        for start in range(0, n, step):
            dist.all_reduce(flat_grad[start:start + step], op=dist.ReduceOp.SUM)

    def all_reduce_range(self, flat_grad, start, stop, async_op=False):
        """SUM flat_grad[start:stop] over all replicas; async_op: returns the work handle (the collective runs behind the
        kernels already queued on the current stream, concurrently with whatever is queued after it)."""
        # (XPT_DP_FORCE_COLLECTIVES=1: a one-rank group still issues its collectives -- the identity -- so that the one-GPU
        #  test box exercises RCCL itself in the graph / all-reduce / graph sequence: tests/test_rccl_single_rank_gpu.py)
        if (self.num_replicas_in_sync == 1 and os.environ.get("XPT_DP_FORCE_COLLECTIVES") != "1") or stop <= start:
            return None
        return dist.all_reduce(flat_grad[start:stop], op=dist.ReduceOp.SUM, async_op=async_op)

    def broadcast_parameters(self, flat_data):
        """All replicas start from rank 0's weights (MirroredStrategy creates mirrored variables)."""
        if self.num_replicas_in_sync > 1:
            dist.broadcast(flat_data, src=0)

    def reduce_scalars(self, tensor, op="mean"):
        """Epoch-end metric averaging: one tiny all-reduce instead of the reference's per-step host syncs."""
        if self.num_replicas_in_sync > 1:
            dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
            if op == "mean":
                tensor /= self.num_replicas_in_sync
        return tensor

    def barrier(self):
        if self.num_replicas_in_sync > 1:
            dist.barrier()
