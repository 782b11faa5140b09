"""Data-parallel training glue: one process per GPU, batch sharded across
ranks, gradients averaged with torch.distributed (backend "nccl" is RCCL over
xGMI on ROCm; "gloo" for CPU tests).

The network keeps all gradients in one flat fp32 buffer, cut into buckets at
layer boundaries and numbered in the order backward finishes them (last layer
first).  Backward records one HIP event per bucket on the compute stream; the
all-reduce of bucket b is enqueued on a separate communication stream that
waits only for event b, so reduction of the late layers' gradients overlaps
with the backward pass of the early layers.  The optimiser step waits for all
buckets.

Batch-norm statistics are per rank (each rank normalises with its local batch),
as in the usual non-synchronised data-parallel recipe.
"""
import torch
import torch.distributed as dist


class _DevicePtr:
    """Exposes a raw device pointer through __cuda_array_interface__ so that
    torch can wrap library-owned memory without copying."""

    def __init__(self, ptr, count, typestr="<f4"):
        self.__cuda_array_interface__ = {
            "shape": (int(count),), "typestr": typestr, "data": (int(ptr), False), "version": 2, "strides": None}


def tensor_from_ptr(ptr, count, device):
    return torch.as_tensor(_DevicePtr(ptr, count), device=device)


def shard_range(total, rank, world):
    """Contiguous slice [lo, hi) of a batch of `total` samples for `rank`."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_mean_(flat, buckets, group=None, streams=None, force=False):
    """In-place mean over ranks of `flat` (1-D tensor), bucket by bucket.

    buckets: [(offset, count, event_handle_or_None)], processed in list order.
    streams: None, or (comm_stream, wait_event_fn) -- each bucket's collective
      is then enqueued on comm_stream after wait_event_fn(comm_stream, event).
    force: run the collectives even in a one-rank group (exercises the transport).
    Returns the list of work handles (already waited on when streams is None).
    """
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return []
    use_avg = dist.get_backend(group) == "nccl"
    op = dist.ReduceOp.AVG if use_avg else dist.ReduceOp.SUM
    works = []
    for off, cnt, ev in buckets:
        view = flat[off:off + cnt]
        if streams is not None:
            comm, wait_event = streams
            if ev:
                wait_event(comm, ev)
            with torch.cuda.stream(comm):
                works.append(dist.all_reduce(view, op=op, group=group, async_op=True))
        else:
            dist.all_reduce(view, op=op, group=group)
            if not use_avg:
                view.div_(world)
    return works


class DataParallel:
    """Wraps a created neural_net.NeuralNet for synchronous data-parallel SGD."""

    def __init__(self, net, device, bucket_bytes=16 << 20, group=None, force_collectives=False):
        from unet_amd import capi
        self.net, self.device, self.group, self._capi = net, device, group, capi
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        net.set_bucket_bytes(int(bucket_bytes))
        self.buckets = net.gradient_buckets()
        n = net.num_parameters()
        self.grads = tensor_from_ptr(net.gradients_ptr(), n, device)
        self.params = tensor_from_ptr(net.parameters_ptr(), n, device)
        self.force = bool(force_collectives) and dist.is_initialized()
        self.comm = torch.cuda.Stream(device=device) if (self.world > 1 or self.force) else None

    def broadcast_parameters(self, src=0):
        if self.world > 1 or self.force:
            self.net.synchronize()
            dist.broadcast(self.params, src, group=self.group)
            torch.cuda.synchronize(self.device)

    def _wait_event(self, stream, event):
        self._capi.unet_stream_wait_event(stream.cuda_stream, event)

    def reduce_gradients(self):
        """Averages the gradient buffer over the ranks, bucket by bucket, each
        bucket as soon as backward has finished it."""
        if self.world == 1 and not self.force:
            return
        if dist.get_backend(self.group) == "nccl":
            works = allreduce_mean_(self.grads, self.buckets, self.group, (self.comm, self._wait_event), self.force)
            for w in works:
                w.wait()      # the compute stream waits for the reduced bucket
            return
        # Transports without device collectives (gloo, used by the single-GPU rehearsal of the
        # multi-rank path): stage each finished bucket through host memory.
        for off, cnt, ev in self.buckets:
            if ev:
                self._capi.unet_event_synchronize(ev)
            else:
                self.net.synchronize()
            host = self.grads[off:off + cnt].cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            self.grads[off:off + cnt].copy_(host.div_(self.world))
        torch.cuda.synchronize(self.device)

    def train_step(self, images_ptr, masks_ptr, batch, lr):
        """Enqueues forward, backward, the bucketed all-reduce and the Adam step
        for this rank's shard (device pointers); does not synchronise."""
        self.net.forward_backward_device(images_ptr, masks_ptr, batch)
        self.reduce_gradients()
        self.net.step(lr)
