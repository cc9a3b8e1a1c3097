"""Data-parallel gradient exchange of the NPPC train step: ONE exchange per step, the sum of the
direction net's flat fp32 gradient over the ranks (RCCL all-reduce over xGMI on the GPU box, gloo in the CPU tests),
issued in buckets as soon as a contiguous range of the flat buffer is final so that it overlaps the rest of backward.

The reference's audio trainers have no distributed code; the precedent is the image trainer's DDP wrapper
(nppc/auxil.py:212-252, nppc/nppc.py:370-386: per-rank batch = batch_size // world, gradients averaged).  Here the mean's
1/world is NOT applied to the buffer: `finish` returns it and the optimizer kernel folds it into its gradient scale.

Bucket schedule (engine.FSNEngine.backward calls `range_ready` from the stream that produced the range):
  1. the sub-band segment (LSTM + head, the tail of the flat buffer) right after the LSTM weight-gradient GEMMs --
     ~8 ms of full-band backward are still to run;
  2. TCN blocks 7..4 of each full-band branch once block 4's weight gradients are done;
  3. whatever is left (attention layers, TCN blocks 3..0, output Linear) in `finish`.
Nothing cooperative (the CU-pair LSTM kernels) is in flight while a bucket travels: the exchange starts after the LSTM
backward kernel of the step and `finish` makes the stream wait for every bucket before the optimizer, i.e. before the
next step's first LSTM launch.
"""
import torch
import torch.distributed as dist


class FlatGradientReducer:
    def __init__(self, group=None, overlap=True):
        self.group = group
        self.overlap = overlap
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self._done = []          # [(lo, hi)] element ranges already handed to the collective this step
        self._work = []

    def range_ready(self, gflat, lo, hi):
        """gflat[lo:hi] will not be written again this step: start its all-reduce (no-op with one rank)"""
        if self.world == 1 or not self.overlap or hi <= lo:
            return
        for a, b in self._done:
            assert hi <= a or lo >= b, "gradient ranges handed to the reducer must not overlap"
        self._done.append((lo, hi))
        self._work.append(dist.all_reduce(gflat[lo:hi], group=self.group, async_op=True))

    def reset(self):
        """drop the bookkeeping of a step that did not finish (exception inside backward): wait for what is in flight"""
        for w in self._work:
            try:
                w.wait()
            except Exception:
                pass
        self._work.clear()
        self._done.clear()

    def finish(self, gflat):
        """reduce what `range_ready` has not covered, wait for every bucket; returns the scale (1/world) that turns the
        summed buffer into the mean gradient"""
        if self.world == 1:
            self._done.clear()
            return 1.0
        pos = 0
        for a, b in sorted(self._done):
            if a > pos:
                self._work.append(dist.all_reduce(gflat[pos:a], group=self.group, async_op=True))
            pos = max(pos, b)
        if pos < gflat.numel():
            self._work.append(dist.all_reduce(gflat[pos:], group=self.group, async_op=True))
        for w in self._work:
            w.wait()             # NCCL: the current stream waits for the bucket; gloo: blocks the host
        self._work.clear()
        self._done.clear()
        return 1.0 / self.world


def mean_reduce_parameter_grads(params, group=None):
    """slow path (torch optimizers that read .grad): per-tensor sum all-reduce, scaled to the mean in place"""
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    if world == 1:
        return
    for p in params:
        if p.grad is not None:
            dist.all_reduce(p.grad, group=group)
            p.grad.mul_(1.0 / world)
