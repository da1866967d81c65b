"""Training-step harness: the reference's ``train_step`` (src/scripts/train.py:429-441)
around the MI355X-native model, single GPU or data-parallel over one node.

The reference script itself cannot run here (Hydra / torchvision are absent and its
Python never travels to the GPU box), so the step recipe is restated:

* ``n_min = max(1, B*H*W // 16)`` per process, two ``OhemCELoss(0.7, n_min, 255)``
  (train.py:329-349), ``loss = crit_p(out, lb) + crit_16(out16, lb)``, ``loss.backward()``;
* fp32 end to end.  (The reference wraps the step in autocast; the parity contract of
  BASELINE.json is against the fp32 CPU forward/backward, so autocast is off by default.)
* under torchrun: gradients are averaged with :class:`cabinet_amd.ddp.BucketedGradReducer`;
* gradient accumulation follows train.py:435-439,478-480: ``loss / accum_steps`` per micro-step, optimizer (and,
  data-parallel, the collectives) only on the last micro-step of a window, ``flush()`` for a trailing partial window.
"""

from __future__ import annotations

import contextlib
import os

import torch

from .loss import OhemCELoss, fused_pair_finish, fused_pair_launch, ohem_upsampled_pair
from .models.cabinet import CABiNet
from .models.constants import DEFAULT_IGNORE_LABEL, DEFAULT_SCORE_THRESHOLD, MOBILENETV3_CFGS, OHEM_DIVISOR

# hipGraph captures use the THREAD-LOCAL error mode: with a process group alive, RCCL's watchdog thread polls the events of
# finished collectives (hipEventQuery); under the default global mode such a call from ANY thread while a capture is open
# is an error that terminates the process ("operation not permitted when stream is capturing", seen once in nine runs of
# tools/host_overhead.py).  Thread-local mode confines the check to the capturing thread, which is what is meant here.
_CAPTURE_MODE = "thread_local"


def build_model(mode="large", n_classes=8, device="cpu", seed=0, gamma=None, freeze_unused=True):
    """Random-init CABiNet (model seed as in BASELINE.md); ``gamma`` overrides CAB's zero-init scale so
    the attention kernels influence logits and gradients (parity trap 1)."""
    torch.manual_seed(seed)
    net = CABiNet(n_classes=n_classes, cfgs=MOBILENETV3_CFGS[mode], mode=mode)
    if gamma is not None:
        with torch.no_grad():
            net.ab.a2block.gamma.fill_(float(gamma))
    if freeze_unused:
        # mobile.classifier never runs in forward (reference mobilenetv3.py:202-205): no grads, so
        # keep it out of the gradient buckets (DDP pitfall 1)
        for p in net.mobile.classifier.parameters():
            p.requires_grad_(False)
    return net.to(device)


def make_criteria(batch, height, width, device, thresh=DEFAULT_SCORE_THRESHOLD, ignore=DEFAULT_IGNORE_LABEL):
    n_min = max(1, batch * height * width // OHEM_DIVISOR)
    return (OhemCELoss(thresh, n_min, ignore).to(device), OhemCELoss(thresh, n_min, ignore).to(device))


class TrainStep:
    """fwd + 2x OHEM-CE + bwd (+ gradient all-reduce when a reducer is given), with the reference's accumulation
    contract: call it once per micro-batch; every ``accum_steps``-th call reduces and steps the optimizer."""

    def __init__(self, net, criteria, reducer=None, optimizer=None, autocast=False, fused_loss=None, accum_steps=1,
                 before_optimizer=None, scaler=None):
        self.net, (self.crit_p, self.crit_16) = net, criteria
        self.reducer, self.optimizer, self.autocast = reducer, optimizer, autocast
        self.before_optimizer = before_optimizer  # e.g. gradient clipping (train.py:411-427), after the reduction
        # scaler: a torch.amp.GradScaler -- the reference's real step (train.py:386,411-441): autocast forward,
        # scaler.scale(loss).backward(), scaler.unscale_ before the clipping callback, scaler.step / update.  The hot-path
        # operators compute in fp32 under autocast (custom_fwd(cast_inputs=float32) on every Function); only the stock
        # convolutions around them run in the autocast dtype.
        self.scaler = scaler
        # fused_loss: run the two final x8 upsamples inside the OHEM-CE kernels (device tensors only);
        # default = on whenever the model lives on a GPU
        self.fused_loss = fused_loss
        self.accum_steps = max(1, int(accum_steps))
        self._micro = 0  # micro-steps taken inside the current accumulation window

    def zero_grad(self):
        if self.reducer is not None:
            self.reducer.zero_grad()
        else:
            for p in self.net.parameters():
                p.grad = None

    def _optimizer_step(self):
        """train.py:_optimizer_step -- here: join the collectives, then step.  Gradients are zeroed at the start of the
        next window (the reference zeroes right after the step; same values, and callers can still read .grad)."""
        if self.reducer is not None:
            self.reducer.finish()
        if self.optimizer is not None:
            if self.scaler is not None:
                if self.before_optimizer is not None:  # clipping acts on the TRUE gradients (train.py:414-416)
                    self.scaler.unscale_(self.optimizer)
                    self.before_optimizer()
                self.scaler.step(self.optimizer)  # skips the step when a gradient is inf / nan (early AMP calibration)
                self.scaler.update()
            else:
                if self.before_optimizer is not None:
                    self.before_optimizer()
                self.optimizer.step()
        self._micro = 0

    def flush(self):
        """Trailing partial accumulation window at the end of an epoch (train.py:478-480)."""
        if self._micro:
            self._optimizer_step()

    def __call__(self, im, lb):
        if self._micro == 0:
            self.zero_grad()
        last = self._micro + 1 == self.accum_steps
        fused = im.is_cuda if self.fused_loss is None else self.fused_loss
        sync = contextlib.nullcontext() if (last or self.reducer is None) else self.reducer.no_sync()
        with sync:
            with torch.amp.autocast(device_type=im.device.type, enabled=self.autocast):
                if fused:
                    low, low16 = self.net.forward_lowres(im)
                    size = im.shape[2:]
                    loss = ohem_upsampled_pair(self.crit_p, low, self.crit_16, low16, lb, size)
                else:
                    out, out16 = self.net(im)
                    loss = self.crit_p(out, lb) + self.crit_16(out16, lb)
                if self.accum_steps > 1:
                    loss = loss / self.accum_steps
            (loss if self.scaler is None else self.scaler.scale(loss)).backward()
        self._micro += 1
        if last:
            self._optimizer_step()
        return loss.detach()


def _check_static_shapes(step, im, lb):
    """A replayed graph reads the static input tensors it was captured on: ``copy_`` would silently BROADCAST a trailing
    batch of one image over the captured batch (and fail opaquely on other sizes)."""
    if tuple(im.shape) != tuple(step.s_im.shape) or tuple(lb.shape) != tuple(step.s_lb.shape) or \
            im.dtype != step.s_im.dtype or lb.dtype != step.s_lb.dtype:
        raise RuntimeError(f"{type(step).__name__}: batch {tuple(im.shape)} {im.dtype} / {tuple(lb.shape)} {lb.dtype} does "
                           f"not match the captured {tuple(step.s_im.shape)} {step.s_im.dtype} / {tuple(step.s_lb.shape)} "
                           f"{step.s_lb.dtype}; run ragged trailing batches through TrainStep (or drop_last, as the "
                           f"reference's loader does, train.py:248-256)")


class _BufferSnapshot:
    """Save / restore a module's buffers (BatchNorm running statistics and counters) with one multi-tensor copy per dtype.
    ``torch._foreach_copy_`` only takes its fused path when every tensor of the list has the same dtype; the buffers of a
    BatchNorm are two fp32 vectors and one int64 counter, and the mixed list fell back to one device copy per tensor:
    ~230 launches of 3 us each in front of every graphed step."""

    def __init__(self, module):
        groups = {}
        for b in module.buffers():
            groups.setdefault(b.dtype, []).append(b)
        self.groups = [(bufs, [b.clone() for b in bufs]) for bufs in groups.values()]

    def save(self):
        for bufs, bak in self.groups:
            torch._foreach_copy_(bak, bufs)

    def restore(self):
        for bufs, bak in self.groups:
            torch._foreach_copy_(bufs, bak)


def _hyper_snapshot(optimizer):
    """Every scalar hyper-parameter of every param group (what torch's optimizers pass to their kernels BY VALUE)."""
    return [{k: (v if isinstance(v, (int, float, bool, type(None), str)) else repr(v)) for k, v in g.items() if k != "params"}
            for g in optimizer.param_groups]


class _OptimizerSegment:
    """How a graphed step runs ``optimizer.step()``.

    Default: EAGERLY, after the backward graph (a foreach SGD is a handful of launches).  torch's optimizers pass lr,
    weight decay and momentum to their kernels as host scalars, and a Python-side schedule (the reference's
    ``Optimizer.step`` sets ``pg['lr']`` by warm-up + poly decay and counts ``self.it`` every step,
    src/utils/optimizer.py:141-155) runs in the interpreter: a captured ``step()`` would replay the capture-time values
    for ever.  ``before`` (e.g. gradient clipping, train.py:411-427) runs between backward and the step.
    ``capture=True`` records the step into its own hipGraph for a plain torch optimizer with constant hyper-parameters;
    the param-group scalars are snapshotted at capture and a replay with changed values RAISES instead of silently
    stepping with the old ones.  A wrapper object whose ``step`` does host-side work must not be captured."""

    def __init__(self, optimizer, capture=False, before=None):
        self.optimizer, self.capture, self.before = optimizer, bool(capture), before
        self.graph = self.snapshot = None
        if self.capture and optimizer is not None and not isinstance(optimizer, torch.optim.Optimizer):
            raise RuntimeError("capture_optimizer=True needs a plain torch.optim.Optimizer: a wrapper's Python-side "
                               "schedule (warm-up, decay, step counter) would be frozen at its capture-time values")
        if self.capture and optimizer is not None and before is not None:  # at construction, not `warmup` steps later
            raise RuntimeError("capture_optimizer=True cannot run a `before_optimizer` callback between the graphs")

    def record(self, pool):
        if self.optimizer is None or not self.capture:
            return
        if self.before is not None:
            raise RuntimeError("capture_optimizer=True cannot run a `before_optimizer` callback between the graphs")
        self.snapshot = _hyper_snapshot(self.optimizer)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, pool=pool, capture_error_mode=_CAPTURE_MODE):
            self.optimizer.step()

    def run(self):
        if self.optimizer is None:
            return
        if self.graph is None:
            if self.before is not None:
                self.before()
            self.optimizer.step()
            return
        if _hyper_snapshot(self.optimizer) != self.snapshot:
            raise RuntimeError("optimizer hyper-parameters changed after the optimizer step was captured in a hipGraph "
                               "(they are baked into the captured kernels); use capture_optimizer=False with schedulers")
        self.graph.replay()


class GraphedTrainStep:
    """TrainStep replayed from two hipGraphs around the step's ONE host read (single process, fused loss).

    Per step the host normally enqueues ~1,100 kernels (13 ms of Python / ctypes / autograd at config 3, for 28 ms of GPU
    work; with 8 ranks on a 16-core host that margin is gone).  Here the step is captured once:

        graph A   forward of the network + the OHEM forward kernels of both heads + their reduced statistics
        host      reads the 2 x (n_valid, n_above) counts and decides the OHEM branch      (the step's one sync)
        graph B   OHEM 'at least n_min pixels above thresh' branch, loss, backward
        eager     optimizer.step() (see _OptimizerSegment: schedules and clipping are host-side; opt-in graph for constant lr)

    and replayed with three host calls.  Nothing data-dependent is baked into a kernel argument (n_above is a device
    scalar in the loss).  If the host read says a head needs the rare top-n_min branch (or has no valid pixel), the step
    restores the BatchNorm buffers graph A advanced and runs eagerly instead -- same result as TrainStep, just slower.
    Inputs are copied into static tensors; the returned loss is a static device tensor (read it before the next step).
    Data-parallel runs keep the eager TrainStep: its reducer launches collectives from autograd hooks."""

    def __init__(self, net, criteria, optimizer=None, warmup=2, capture_optimizer=False, before_optimizer=None):
        self.net, (self.crit_p, self.crit_16) = net, criteria
        self.optimizer, self.warmup = optimizer, warmup
        self.opt_seg = _OptimizerSegment(optimizer, capture_optimizer, before_optimizer)
        self.eager = TrainStep(net, criteria, optimizer=optimizer, before_optimizer=before_optimizer)
        self.g_fwd = self.g_bwd = None
        self.fallbacks = self._calls = 0

    def _capture(self, im, lb):
        """Record the two graphs on this batch.  Nothing of the step executes here (capture only records), except one
        replay of graph A to learn the batch's OHEM branch, whose BatchNorm side effects are undone."""
        self.s_im, self.s_lb = im.clone(), lb.clone()
        self.snap = _BufferSnapshot(self.net)
        # the gradients of the eager step that just ran: the static tensors bound below have only been RECORDED into, never
        # executed, so a caller reading .grad after this step (grad-norm logging, NaN checks) would see uninitialised memory
        eager_grads = {p: p.grad for p in self.net.parameters() if p.grad is not None}
        for p in self.net.parameters():
            p.grad = None
        size = tuple(im.shape[2:])
        torch.cuda.synchronize()
        self.g_fwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_fwd, capture_error_mode=_CAPTURE_MODE):
            low, low16 = self.net.forward_lowres(self.s_im)
            self.prep = fused_pair_launch(self.crit_p, low, self.crit_16, low16, self.s_lb, size)
            self.s_stats = self.prep.stats
            if self.s_stats is None:
                raise RuntimeError("GraphedTrainStep needs the fused OHEM head (device logits, <= 32 classes, no class weights)")
        # capture does not execute: replay once to learn the capture batch's branch, then undo its BatchNorm side effects
        self.snap.save()
        self.g_fwd.replay()
        host = self.s_stats.tolist()
        self.snap.restore()
        if not all(self._selected_branch(c, h) for c, h in zip((self.crit_p, self.crit_16), host)):
            raise RuntimeError("GraphedTrainStep: capture batch does not take the OHEM 'n_min above thresh' branch; "
                               "capture on a representative batch")
        self.g_bwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_bwd, pool=self.g_fwd.pool(), capture_error_mode=_CAPTURE_MODE):
            loss = fused_pair_finish(self.prep, host)
            loss.backward()
            self.s_loss = loss.detach()
        self.opt_seg.record(self.g_fwd.pool())
        self.s_grads = [(p, p.grad) for p in self.net.parameters() if p.grad is not None]
        with torch.no_grad():
            for p, g in self.s_grads:
                if p in eager_grads:
                    g.copy_(eager_grads[p])
                else:
                    g.zero_()
        torch.cuda.synchronize()

    @staticmethod
    def _selected_branch(crit, host):
        n_valid, n_above = int(host[0]), int(host[1])
        if n_valid < 0:
            raise RuntimeError("OhemCELoss: label out of range (and != ignore_lb)")
        return n_valid > 0 and n_above >= min(crit.n_min, n_valid)

    def __call__(self, im, lb):
        if self.g_fwd is None:
            # the first max(1, warmup) steps run eagerly: every lazy initialisation happens outside capture (MIOpen solver
            # choice, kernel attributes, and the optimizer's state -- SGD creates its momentum buffers in the first step(),
            # a captured first step would re-create them on every replay); the graphs are recorded after the last of them
            self._calls += 1
            loss = self.eager(im, lb)
            if self._calls >= max(1, self.warmup):
                self._capture(im, lb)
            return loss
        _check_static_shapes(self, im, lb)
        self.s_im.copy_(im, non_blocking=True)
        self.s_lb.copy_(lb, non_blocking=True)
        self.snap.save()
        self.g_fwd.replay()
        host = self.s_stats.tolist()  # the step's one host sync
        if all(self._selected_branch(c, h) for c, h in zip((self.crit_p, self.crit_16), host)):
            for p, g in self.s_grads:  # a caller's zero_grad(set_to_none=True) must not detach the graphs' gradient tensors
                p.grad = g
            self.g_bwd.replay()
            self.opt_seg.run()
            return self.s_loss
        # rare branch (late training: fewer than n_min hard pixels): undo graph A's BatchNorm side effects, run eagerly
        self.fallbacks += 1
        self.snap.restore()
        loss = self.eager(im, lb)
        for p, g in self.s_grads:  # keep the static gradient tensors the graphs write to; a step without gradient = zeros
            if p.grad is None:
                g.zero_()
            elif p.grad is not g:
                g.copy_(p.grad)
            p.grad = g
        return loss


# ---- which replay schedule GraphedDDPStep takes (round 6) -----------------------------------------------------
# Measured on one GPU (profiles/r05_ddp_gap_probe.txt): an event record between the backward graphs is free and a side stream that
# waits for the event behind B1 (decoder) is free, but ANY stream that waits for the event between B2 (backbone) and B3 (spatial
# branch) costs the compute stream 0.3-0.4 ms -- that boundary then drains instead of letting B3's first kernels start under B2's last.
# Dropping that event exposes the backbone's buckets (12.6 MB) behind B3 instead of hiding them under the spatial branch's backward.
# Which is cheaper depends on what those buckets cost on the wire, so the choice is taken from a model of the all-reduce and printed
# (bench.py: config.ddp_schedule); CABINET_DDP_ONE_EVENT=1 / 0 overrides it for a node that can measure both.
DDP_DRAIN_MS = 0.35            # measured cost of the second event's wait (one MI355X, config 3)
XGMI_LINK_GBS_PER_DIR = 76.8   # 153.6 GB/s per link, both directions (MI355X_MICROARCH.md: 7 links x ~153 GB/s per GPU)
RCCL_LINK_EFFICIENCY = 0.5     # share of the link rate a <= 8 MB bucket reaches (latency-bound messages); conservative
RCCL_LAUNCH_MS = 0.02          # per collective
DDP_TWO_EVENT_MARGIN = 1.25    # the modelled exposure must exceed the measured drain by this factor before two events are chosen


def choose_ddp_schedule(world, exposed_bytes, n_buckets=2):
    """-> dict(one_event, exposed_ms, drain_ms, ...): ring all-reduce of `exposed_bytes` over min(world - 1, 7) rings, one xGMI link
    each (point-to-point fabric: a ring is bound by ONE link per hop), against the measured drain of the second event."""
    forced = os.environ.get("CABINET_DDP_ONE_EVENT")
    rings = max(1, min(world - 1, 7))
    if world > 1:
        wire = 2.0 * (world - 1) / world * (exposed_bytes / rings) / (RCCL_LINK_EFFICIENCY * XGMI_LINK_GBS_PER_DIR * 1e9) * 1e3
    else:
        wire = 0.0   # one-rank "collectives" are local copies
    exposed = wire + n_buckets * RCCL_LAUNCH_MS
    # the drain is MEASURED (on one GPU), the exposure is a model: the two-event schedule has to beat the one-event one by a margin
    # before it is chosen (no schedule whose benefit was never measured as the default)
    one = exposed < DDP_TWO_EVENT_MARGIN * DDP_DRAIN_MS
    why = "model"
    if forced in ("0", "1"):
        one, why = forced == "1", "CABINET_DDP_ONE_EVENT=" + forced
    return dict(one_event=bool(one), decided_by=why, world=world, exposed_mbytes=round(exposed_bytes / 2 ** 20, 2), rings=rings,
                exposed_ms_model=round(exposed, 3), drain_ms_measured=DDP_DRAIN_MS,
                what=("backbone buckets behind B3 on the compute stream (exposed), decoder buckets on the side stream behind B1"
                      if one else "decoder buckets behind B1 and backbone buckets behind B2, both on the side stream (two events)"))


class GraphedDDPStep:
    """Data-parallel step for one node of MI355X: hipGraph segments with the gradient all-reduces issued BETWEEN them.

    The eager reducer (cabinet_amd.ddp.BucketedGradReducer) launches collectives from autograd hooks, which keeps the
    whole ~1,100-launch Python / autograd enqueue path (13 ms per step at config 3) on every rank's host -- 8 ranks share
    the node's host cores.  Here the step is cut where the model cuts itself: the decoder (``conv_out, ffm, ab``: 23 MB of
    gradients) is back-propagated first, then the backbone (``mobile``: 12.6 MB), and the spatial branch LAST (``sb``: 0.36 MB
    of gradients but several milliseconds of backward on the 512^2 .. 128^2 maps -- north_star: "all-reduce ... overlapped with
    the backward of the spatial branch"):

        graph A    forward, OHEM forward kernels + statistics
        host       one read-back (OHEM branch), as in GraphedTrainStep
        graph B1   loss, backward of the decoder down to the two boundary tensors (sb output, mobile output), gradients packed
                   into the decoder's flat buckets (one multi-tensor copy)
        graph B2   backward of the backbone from its boundary gradient, packed into the backbone buckets
        graph B3   backward of the spatial branch, packed into its (single, < 0.5 MB) bucket
                   (B1, B2, B3 are enqueued back to back, an event recorded behind B1 and behind B2)
        RCCL       all-reduce(AVG) of the decoder buckets behind B1's event, on a side stream      <- overlaps graphs B2, B3
        RCCL       all-reduce of the backbone buckets behind B2's event                            <- overlaps graph B3
        RCCL       all-reduce of the spatial branch's bucket behind B3 -- the only exposed communication; join
        eager      optimizer step (_OptimizerSegment: host-side schedules / clipping; opt-in graph for constant lr)

    Collectives are ordinary eager calls between replays (nothing of RCCL is captured), always the same buckets in the
    same order on every rank, also on a rank that falls back to the eager path for this step (rare OHEM branch).
    Autograd writes each gradient into its own tensor (``.grad`` is None during backward, so nothing is accumulated); one
    ``torch._foreach_copy_`` per segment packs them into the flat fp32 buckets, and ``.grad`` then IS the bucket view the
    all-reduce averages and the optimizer reads.  (Pre-zeroed bucket views as ``.grad`` made autograd run one in-place add
    per parameter: ~180 extra launches per step.)  BatchNorm statistics and OHEM stay per rank.
    ``use_graphs=False`` (default on CPU tensors) runs the identical schedule eagerly -- that is what the gloo tests drive."""

    DECODER = ("conv_out", "ffm", "ab")
    LAST = ("sb",)  # back-propagated last: its backward hides the backbone's all-reduce

    def __init__(self, net, criteria, optimizer=None, process_group=None, bucket_mb=8.0, warmup=2, use_graphs=None,
                 always_reduce=False, broadcast_parameters=True, capture_optimizer=False, before_optimizer=None):
        import torch.distributed as dist

        from .ddp import plan_buckets

        if not dist.is_initialized():
            raise RuntimeError("GraphedDDPStep needs an initialised torch.distributed process group")
        self.dist, self.group = dist, process_group
        self.world, self.backend = dist.get_world_size(process_group), dist.get_backend(process_group)
        self.always_reduce = always_reduce
        self.net, (self.crit_p, self.crit_16) = net, criteria
        self.optimizer, self.warmup = optimizer, warmup
        self.opt_seg = _OptimizerSegment(optimizer, capture_optimizer, before_optimizer)
        dev = next(net.parameters()).device
        self.use_graphs = (dev.type == "cuda") if use_graphs is None else bool(use_graphs)
        dec, mid, last = [], [], []
        for name, child in net.named_children():
            for p in child.parameters():
                if p.requires_grad:
                    (dec if name in self.DECODER else last if name in self.LAST else mid).append(p)
        self.dec_params = dec
        cap = int(bucket_mb * 2 ** 20)
        self.segments, self.seg_views = [], []
        for params in (list(reversed(dec)), list(reversed(mid)), list(reversed(last))):  # ~ arrival order inside each part
            flats = []
            views = []
            for plan in plan_buckets([p.numel() * 4 for p in params], cap, cap, 2 ** 18):
                group = [params[i] for i in plan]
                flat = torch.zeros(sum(p.numel() for p in group), dtype=torch.float32, device=dev)
                off = 0
                for p in group:
                    views.append((p, flat[off:off + p.numel()].view_as(p)))
                    p.grad = views[-1][1]
                    off += p.numel()
                flats.append(flat)
            self.segments.append(flats)
            self.seg_views.append(views)
        if broadcast_parameters and self.world > 1:
            with torch.no_grad():
                for t in list(net.parameters()) + list(net.buffers()):
                    dist.broadcast(t, src=0, group=process_group)
        self.graphs = None
        self.fallbacks = self._calls = 0
        # side stream + events that order the collectives behind the graph segments (replay path, RCCL only);
        # CABINET_DDP_INLINE_REDUCE=1 keeps round 4's issue order (A/B timing)
        self._side, self._ev = None, None
        if self.use_graphs and dev.type == "cuda" and os.environ.get("CABINET_DDP_INLINE_REDUCE") != "1":
            self._side = torch.cuda.Stream(device=dev)
            self._ev = [torch.cuda.Event(), torch.cuda.Event()]
        self.schedule = choose_ddp_schedule(self.world, sum(f.numel() * 4 for f in self.segments[1]), len(self.segments[1]))
        self.schedule["side_stream"] = self._side is not None and self.backend == "nccl"

    # ---- the pieces of one step (run eagerly, or recorded once and replayed) ----
    def _clear(self):
        """``.grad = None`` on every parameter: autograd then stores each gradient instead of adding it to a zeroed view."""
        for views in self.seg_views:
            for p, _ in views:
                p.grad = None

    def _pack(self, seg):
        """Gradients of one segment -> its flat buckets (one multi-tensor copy); ``.grad`` becomes the bucket view.
        A parameter autograd did not reach (or a loss that does not depend on the network) contributes zeros."""
        dst, src = [], []
        for p, view in self.seg_views[seg]:
            g = p.grad
            if g is None:
                view.zero_()
            elif g is not view:
                dst.append(view)
                src.append(g)
            p.grad = view
        if src:
            torch._foreach_copy_(dst, src)

    def _forward(self, im, lb):
        """-> ((pair prep, its device statistics) | None, loss | None, boundary tensors [sb output, mobile output])"""
        boundary = []
        if im.is_cuda:
            low, low16 = self.net.forward_lowres(im, boundary)
            prep = fused_pair_launch(self.crit_p, low, self.crit_16, low16, lb, tuple(im.shape[2:]))
            if prep.stats is not None:
                return (prep, prep.stats), None, boundary
            return None, fused_pair_finish(prep, None), boundary
        final, high_up = self.net.forward_lowres(im, boundary)
        size = im.shape[2:]
        out = torch.nn.functional.interpolate(final, size=size, mode="bilinear", align_corners=False)
        out16 = torch.nn.functional.interpolate(high_up, size=size, mode="bilinear", align_corners=False)
        return None, self.crit_p(out, lb) + self.crit_16(out16, lb), boundary

    def _backward_decoder(self, loss, boundary):
        """Gradients of the decoder parameters (into their bucket views) and of the two boundary tensors (into .grad).
        Returns False when the loss does not depend on the network (constant zero: every label ignored)."""
        torch.autograd.backward(loss, inputs=self.dec_params + boundary, retain_graph=True)
        return all(self._grad_of(t) is not None for t in boundary)

    @staticmethod
    def _grad_of(t):
        import warnings

        with warnings.catch_warnings():  # .grad of a non-leaf: populated because the tensor was named in `inputs`
            warnings.simplefilter("ignore")
            return t.grad

    def _backward_from(self, t, retain=False):
        """Back-propagate one encoder from its boundary tensor (the autograd graphs of ``mobile`` and ``sb`` are disjoint)."""
        torch.autograd.backward([t], [self._grad_of(t)], retain_graph=retain)

    def _reduce(self, seg):
        if self.world == 1 and not self.always_reduce:
            return []
        op = self.dist.ReduceOp.AVG if self.backend == "nccl" else self.dist.ReduceOp.SUM
        return [(self.dist.all_reduce(f, op=op, group=self.group, async_op=True), f) for f in self.segments[seg]]

    def _join(self, works):
        for w, f in works:
            w.wait()
            if self.backend != "nccl":
                f.div_(self.world)

    def _eager_step(self, im, lb):
        self._clear()
        fused, loss, boundary = self._forward(im, lb)
        if fused is not None:
            loss = fused_pair_finish(fused[0], fused[1].tolist())
        ran = self._backward_decoder(loss, boundary)
        self._pack(0)
        works = self._reduce(0)
        feat_sb, mob = boundary
        if ran:
            self._backward_from(mob)
        self._pack(1)
        works += self._reduce(1)
        if ran:
            self._backward_from(feat_sb)
        self._pack(2)
        works += self._reduce(2)
        self._join(works)
        self._optimizer_step_eager()
        return loss.detach()

    def _optimizer_step_eager(self):
        if self.optimizer is not None:
            if self.opt_seg.before is not None:
                self.opt_seg.before()
            self.optimizer.step()

    def _capture(self, im, lb):
        self.s_im, self.s_lb = im.clone(), lb.clone()
        self.snap = _BufferSnapshot(self.net)
        torch.cuda.synchronize()
        gA, gB1, gB2, gB3 = (torch.cuda.CUDAGraph() for _ in range(4))
        self._clear()
        with torch.cuda.graph(gA, capture_error_mode=_CAPTURE_MODE):
            self.fused, _, self.boundary = self._forward(self.s_im, self.s_lb)
            if self.fused is None:
                raise RuntimeError("GraphedDDPStep with graphs needs the fused OHEM head")
            self.s_stats = self.fused[1]
        self.snap.save()
        gA.replay()
        host = self.s_stats.tolist()
        self.snap.restore()
        if not all(GraphedTrainStep._selected_branch(c, h) for c, h in zip((self.crit_p, self.crit_16), host)):
            raise RuntimeError("GraphedDDPStep: capture batch does not take the OHEM 'n_min above thresh' branch")
        feat_sb, mob = self.boundary
        with torch.cuda.graph(gB1, pool=gA.pool(), capture_error_mode=_CAPTURE_MODE):
            loss = fused_pair_finish(self.fused[0], host)
            self._backward_decoder(loss, self.boundary)
            self._pack(0)
            self.s_loss = loss.detach()
        with torch.cuda.graph(gB2, pool=gA.pool(), capture_error_mode=_CAPTURE_MODE):
            self._backward_from(mob)
            self._pack(1)
        with torch.cuda.graph(gB3, pool=gA.pool(), capture_error_mode=_CAPTURE_MODE):
            self._backward_from(feat_sb)
            self._pack(2)
        self.opt_seg.record(gA.pool())
        torch.cuda.synchronize()
        self.graphs = (gA, gB1, gB2, gB3)

    def __call__(self, im, lb):
        if not self.use_graphs:
            return self._eager_step(im, lb)
        if self.graphs is None:
            self._calls += 1
            loss = self._eager_step(im, lb)
            if self._calls >= max(1, self.warmup):
                self._capture(im, lb)
            return loss
        gA, gB1, gB2, gB3 = self.graphs
        _check_static_shapes(self, im, lb)
        self.s_im.copy_(im, non_blocking=True)
        self.s_lb.copy_(lb, non_blocking=True)
        self.snap.save()
        gA.replay()
        host = self.s_stats.tolist()  # the step's one host sync
        if not all(GraphedTrainStep._selected_branch(c, h) for c, h in zip((self.crit_p, self.crit_16), host)):
            self.fallbacks += 1       # same collectives, same order, issued by the eager path
            self.snap.restore()
            return self._eager_step(im, lb)
        for views in self.seg_views:  # a caller's zero_grad(set_to_none=True) must not detach the bucket views
            for p, view in views:
                p.grad = view
        if self.backend == "nccl" and self._side is not None:
            # Round 5: the three backward graphs are enqueued BACK TO BACK on the compute stream, with only an
            # event record between them; the collectives are issued afterwards under a side stream that waits for the event behind
            # the segment that produced the buckets.  Round 4 issued the decoder's four all_reduce calls between B1 and B2 (and
            # the backbone's two between B2 and B3): whatever the host spent in those calls sat between two graph launches of the
            # compute stream (profiles/r04_ddp_overlap_world1.md: every RCCL kernel with no compute kernel beside it, the next
            # compute kernel 10-98 us later).  Same buckets, same order on every rank.
            cur = torch.cuda.current_stream()
            # tools/ddp_gap_probe.py (profiles/r05_ddp_gap_probe.txt): event records between the graph launches are free and so is a
            # stream waiting for the event behind B1, but ANY stream waiting for the event behind B2 costs the compute stream
            # ~0.3-0.4 ms (the B2 | B3 boundary then drains instead of overlapping).  CABINET_DDP_ONE_EVENT=1 drops that event and
            # issues the backbone's buckets behind B3 on the compute stream instead (12.6 MB exposed instead of overlapped with the
            # spatial branch's backward): which one wins at 8 GPUs is for a node that has them to measure.
            one_event = self.schedule["one_event"]
            gB1.replay()
            self._ev[0].record(cur)
            gB2.replay()
            if not one_event:
                self._ev[1].record(cur)
            gB3.replay()
            with torch.cuda.stream(self._side):
                self._side.wait_event(self._ev[0])
                works = self._reduce(0)   # the decoder's buckets travel under the backbone's and the spatial branch's backward
                if not one_event:
                    self._side.wait_event(self._ev[1])
                    works += self._reduce(1)  # the backbone's under the spatial branch's
            if one_event:
                works += self._reduce(1)
            works += self._reduce(2)      # < 0.5 MB behind graph B3: the only exposed communication
        else:
            gB1.replay()
            works = self._reduce(0)       # the collective waits for graph B1, the host goes on to launch B2
            gB2.replay()
            works += self._reduce(1)      # the backbone's buckets travel under the spatial branch's backward
            gB3.replay()
            works += self._reduce(2)      # < 0.5 MB: the only exposed communication
        self._join(works)
        self.opt_seg.run()
        return self.s_loss

    @property
    def bucket_megabytes(self):
        return [f.numel() * 4 / 2 ** 20 for flats in self.segments for f in flats]


def synthetic_batch(batch, height, width, n_classes, device, seed=1):
    """Inputs of BASELINE.md section 2: images ~ N(0,1), labels uniform over classes, no ignore pixels."""
    g = torch.Generator().manual_seed(seed)
    im = torch.randn(batch, 3, height, width, generator=g)
    lb = torch.randint(0, n_classes, (batch, height, width), generator=g)
    return im.to(device), lb.to(device)
