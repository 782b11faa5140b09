"""Data-parallel training: one process per GPU, the batch sharded across ranks.

The collective itself lives behind the C ABI (include/unet_hip.h, `unet_comm_*`, `unet_allreduce_begin/wait`:
RCCL over xGMI).  The engine keeps all gradients in one flat fp32 buffer, cut into buckets at layer
boundaries and numbered in the order backward finishes them (last layer first).  With a communicator
attached (`net.set_comm`), backward records one HIP event per bucket on the compute stream and at once
enqueues that bucket's mean over the ranks on the communicator's own stream, which waits only for that
event; the reduction of the late layers' gradients therefore overlaps with the backward pass of the early
layers, and the optimiser step waits for all buckets.  This module only bootstraps the communicator (the
RCCL unique id travels over torch.distributed, whatever its backend) and shards batches.

Batch-norm statistics are per rank by default (each rank normalises with its local batch, the usual
non-synchronised recipe); `sync_bn=True` (or UNET_SYNC_BN=1) adds the per-channel sums over the ranks inside
the batch-norm entry points, which makes an N-rank step equal to the single-GPU step on the whole batch.

Where RCCL cannot connect the ranks (torch.distributed on "gloo": CPU tests, or several ranks sharing one
GPU) the buckets are staged through host memory instead; everything else is the same code.
"""
import ctypes
import os

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


def allreduce_mean_(flat, buckets, group=None):
    """In-place mean over ranks of `flat` (1-D host or device tensor), bucket by bucket, through
    torch.distributed (the host-staged transport and the CPU tests use it)."""
    world = dist.get_world_size(group)
    if world == 1:
        return []
    for off, cnt, _ in buckets:
        view = flat[off:off + cnt]
        dist.all_reduce(view, op=dist.ReduceOp.SUM, group=group)
        view.div_(world)
    return []      # no work handles: every bucket is complete on return


def init_comm(capi, group=None, device=None):
    """Creates this rank's communicator of the C ABI; rank 0's unique id reaches the others through
    torch.distributed.  Collective: every rank of `group` must call it."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    nbytes = capi.lib().unet_comm_unique_id_bytes()
    buf = ctypes.create_string_buffer(nbytes)
    if rank == 0:
        capi.unet_comm_get_unique_id(buf)
    box = [buf.raw if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group,
                                   device=device if dist.get_backend(group) == "nccl" else None)
    ident = ctypes.create_string_buffer(box[0], nbytes)
    comm = ctypes.c_void_p()
    capi.unet_comm_init_rank(ctypes.byref(comm), ident, world, rank)
    return comm


def library_transport_ready(capi, group=None, device=None):
    """Whether EVERY rank of `group` can load RCCL through the C ABI.  unet_comm_init_rank is a collective: a rank
    that cannot open the library never joins it and the others would wait for ever, so the ranks agree first
    (a MIN over torch.distributed).  Returns (ready, this rank's error message or None)."""
    ok, why = 1, None
    try:
        probe = ctypes.create_string_buffer(capi.lib().unet_comm_unique_id_bytes())
        capi.unet_comm_get_unique_id(probe)      # opens librccl.so.1 and asks it for an id (discarded)
    except RuntimeError as e:
        ok, why = 0, str(e)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        on_device = dist.get_backend(group) == "nccl"
        flag = torch.tensor([ok], dtype=torch.int32, device=device if on_device else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        ok = int(flag.item())
    return bool(ok), why


def all_ranks_agree(ok, group=None, device=None):
    """True when `ok` holds on every rank of `group` (a MIN over torch.distributed)."""
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return bool(ok)
    on_device = dist.get_backend(group) == "nccl"
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if on_device else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(int(flag.item()))


class DataParallel:
    """Wraps a created neural_net.NeuralNet for synchronous data-parallel training."""

    def __init__(self, net, device, bucket_bytes=16 << 20, group=None, force_collectives=False, sync_bn=None,
                 bf16_gradients=None):
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
        if sync_bn is None:
            sync_bn = os.environ.get("UNET_SYNC_BN", "0") not in ("", "0")
        self.sync_bn = bool(sync_bn)
        self.comm = None
        self.torch_device_reduce = False
        active = self.world > 1 or self.force
        # RCCL needs one GPU per rank; "gloo" groups (CPU tests, ranks sharing a GPU) stage through the host
        self.native = active and dist.get_backend(group) == "nccl"
        if self.native:
            if self.force:
                os.environ["UNET_COMM_FORCE"] = "1"   # run the collectives even in a one-rank communicator
            # The library's own communicator may not come up (e.g. no librccl.so.1 on the loader path):
            # torch.distributed's collectives on the same device buffers then do the same reduction, without the
            # overlap with backward.  The ranks agree on that before any of them enters the collective set-up.
            ready, why = library_transport_ready(capi, group, device)
            failure = None if ready else (why or "another rank cannot load RCCL")
            if ready:
                try:
                    self.comm = init_comm(capi, group, device)
                except RuntimeError as e:
                    failure = str(e)
                # The set-up can fail on one rank only.  A rank that fell back alone would meet the others in
                # mismatched collectives and hang them, so the ranks agree on the outcome too; where any of them
                # failed, all give up the library transport and destroy what they created.
                if not all_ranks_agree(failure is None, group, device):
                    if self.comm is not None:
                        capi.unet_comm_destroy(self.comm)
                        self.comm = None
                    failure = failure or "another rank could not set up its communicator"
            if failure is not None:
                if self.sync_bn:
                    raise RuntimeError(failure)
                print(f"unet_amd.dp: {failure}; reducing gradients through torch.distributed instead", flush=True)
                self.native = False
                self.torch_device_reduce = True
            else:
                net.set_comm(self.comm.value)
                if self.sync_bn:
                    capi.unet_bn_set_sync(self.comm)
                if bf16_gradients is None:
                    bf16_gradients = os.environ.get("UNET_GRAD_BF16", "0") not in ("", "0")
                if bf16_gradients:   # the buckets travel as bf16 (half the traffic; native transport only)
                    capi.unet_comm_set_bf16_gradients(self.comm, 1)
        elif self.sync_bn and active:
            raise RuntimeError("sync_bn needs the RCCL transport (torch.distributed backend 'nccl')")

    def close(self):
        if self.comm is not None:
            self.net.synchronize()
            self.net.set_comm(0)
            self._capi.unet_bn_set_sync(None)
            self._capi.unet_comm_destroy(self.comm)
            self.comm = None

    def broadcast_parameters(self, src=0):
        """Rank `src`'s parameters (native transport: also running statistics and optimiser state) to all."""
        if self.native:
            self.net.broadcast_state(src)
            self.net.synchronize()
        elif self.world > 1 and self.torch_device_reduce:
            self.net.synchronize()
            dist.broadcast(self.params, src, group=self.group)
            torch.cuda.synchronize(self.device)
            self.net.parameters_changed()
        elif self.world > 1:
            self.net.synchronize()
            host = self.params.cpu()
            dist.broadcast(host, src, group=self.group)
            self.params.copy_(host)
            torch.cuda.synchronize(self.device)
            self.net.parameters_changed()

    def reduce_gradients(self):
        """Host-staged transport only (the native one reduces inside backward): averages the gradient
        buffer over the ranks, bucket by bucket, each bucket as soon as backward has finished it."""
        if self.native or self.world == 1:
            return
        if self.torch_device_reduce:   # RCCL through torch.distributed, on the device buffers themselves
            self.net.synchronize()
            for off, cnt, _ in self.buckets:
                dist.all_reduce(self.grads[off:off + cnt], op=dist.ReduceOp.AVG, group=self.group)
            torch.cuda.synchronize(self.device)
            return
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
        """Enqueues forward, backward (with the bucketed all-reduce) and the Adam step for this rank's
        shard (device pointers); does not synchronise on the native transport."""
        self.net.forward_backward_device(images_ptr, masks_ptr, batch)
        self.reduce_gradients()
        self.net.step(lr)
