"""HIP-graph capture of a whole training step (SURVEY.md section 7 "Host overhead ... HIP graphs or a single C++ step driver").

A step of the product is 130-230 C-ABI launches plus torch's own small kernels, issued from Python: 4.5 ms (seg) / 6.4 ms (reg) /
8.1 ms (joint) of host time per step whatever the volume size.  Every launcher only ENQUEUES on the stream it is given and all
scratch comes from torch's allocator, so the whole step -- forward, losses, backward on two streams, Adam -- can be captured once with
torch.cuda.CUDAGraph (hipGraph underneath) and replayed with ONE host call per step:

    g = GraphedStep(segments=[grad_fn, update_fn], between=[lambda: parallel.allreduce_gradients(opt)], optimizers=[opt])
    out = g()                  # first `warmup` calls run eagerly, the next one captures, from then on one replay per call

`segments` are callables that only enqueue device work (zero_grad / forward / loss / backward | optimizer.step()); each may return a
dict of device tensors, merged into the result.  `between[k]` runs eagerly between segment k and k + 1: that is where the
gradient all-reduce lives -- collectives are never captured.  In a single process the `between` hooks are no-ops and all segments are
captured as ONE graph; with a process group every segment is its own graph.

Rules the captured functions must obey (true for models/segmentation.py's and models/joint.py's steps):
  * inputs live in fixed device tensors (copy new data INTO them between replays); outputs are read from the returned tensors;
  * no host synchronisation inside (loss.item() belongs outside);
  * per-step host scalars come from device memory: FlatAdam switches to da_adam_step_dev (step count / lr / betas in a 6-float device
    array that GraphedStep refreshes before each replay, so learning-rate schedulers keep working).
"""
import itertools

import torch

from . import ops, parallel
from ._native import workspace

_capture_ids = itertools.count(1)


class GraphedStep:
    def __init__(self, segments, optimizers, between=None, warmup=3):
        self.segments = list(segments)
        self.between = list(between) if between is not None else [None] * (len(self.segments) - 1)
        if len(self.between) != len(self.segments) - 1:
            raise ValueError('GraphedStep: need one `between` hook per gap between segments')
        self.optimizers, self.warmup = list(optimizers), warmup
        self.calls = 0
        self.graphs = None
        self.out = {}
        self.distributed = parallel.world_size() > 1
        self.tag = None                 # key of this object's capture-time scratch buffers in _native.workspace

    def _run_eager(self):
        out = {}
        for k, seg in enumerate(self.segments):
            r = seg()
            if isinstance(r, dict):
                out.update(r)
            if k < len(self.between) and self.between[k] is not None:
                self.between[k]()
        return out

    def _capture(self):
        for o in self.optimizers:
            o.device_step = True
            o.sync_device_state()
        torch.cuda.synchronize()
        self.tag = workspace.capture_tag = 'graph%d' % next(_capture_ids)      # scratch buffers of this capture: allocated inside it, released by close()
        try:
            if self.distributed:
                graphs, pool = [], None
                for k, seg in enumerate(self.segments):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, pool=pool):
                        r = seg()
                        # weight gradients forked onto the side stream (ops.ASYNC_WGRAD) must rejoin INSIDE this capture: an unjoined
                        # fork fails hipStreamEndCapture, and the next segment's join would wait on work of a finished capture
                        ops.join_side_stream()
                    if isinstance(r, dict):
                        self.out.update(r)
                    pool = g.pool()
                    graphs.append(g)
                    if k < len(self.between) and self.between[k] is not None:
                        self.between[k]()            # keeps the replicas in step while capturing (capture itself runs nothing)
                self.graphs = graphs
            else:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self.out.update(self._run_eager())
                    ops.join_side_stream()
                self.graphs = [g]
        finally:
            workspace.capture_tag = None
            for o in self.optimizers:
                o.device_step = False

    def __call__(self):
        self.calls += 1
        if self.calls <= self.warmup:
            return self._run_eager()
        if self.graphs is None:
            self._capture()
        for o in self.optimizers:
            o.sync_device_state()
        if self.distributed:
            for k, g in enumerate(self.graphs):
                g.replay()
                if k < len(self.between) and self.between[k] is not None:
                    self.between[k]()
        else:
            self.graphs[0].replay()
        for o in self.optimizers:
            o.note_replayed_step()
        return self.out

    def close(self):
        """Drop the graphs and the scratch buffers allocated during their capture (a long-lived process that re-captures would otherwise
        keep one full set per capture: the deterministic warp scratch alone is 1.3 GB at 160 x 192 x 160 x 32)."""
        self.graphs = None
        self.out = {}
        if self.tag is not None:
            workspace.release(self.tag)
            self.tag = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
