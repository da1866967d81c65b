"""Data-parallel gradient reduction for one 8x MI355X node (RCCL over xGMI).

The reference trains single-process (no DDP anywhere), so this is
new functionality rather than a mirror.  Design, MI355X-first:

* one process per GPU, ``torch.distributed`` backend ``nccl`` (= RCCL), pure data parallel;
  BatchNorm statistics and the OHEM ``n_min`` stay per rank (the reference has no SyncBN);
* parameters that receive gradients are packed into a few flat fp32 buckets in the order
  backward PRODUCES them; ``param.grad`` is a VIEW into its bucket, so there is no
  copy-in/copy-out.  The order is not guessed: the first backward runs on a provisional plan
  (reverse registration order) while the post-accumulate hooks record the arrival order --
  for CABiNet ``conv_out, ffm, ab, mobile, sb``: the spatial branch runs first in forward
  (models/cabinet.py) so its gradients arrive last -- and the buckets are then rebuilt once in
  that order (rank 0's order is broadcast so every rank packs identically);
* a bucket's all-reduce is launched asynchronously on RCCL's own stream as soon as it AND every
  bucket before it are complete: collectives are issued strictly in bucket-index order on every
  rank, whatever each rank's autograd graph did (a rank whose OHEM loss was the constant zero
  launches nothing during backward and everything, in the same order, from ``finish()``);
* bucket sizes follow the fabric, not NVSwitch habits: xGMI is point-to-point (7 links/GPU), a
  36.7 MB gradient set is latency- not bandwidth-bound, so a small first bucket (to start early)
  and ~8 MB followers keep every launch in RCCL's low-latency regime while bounding the number of
  collectives.  A head smaller than ``min_bucket_mb`` is merged into its neighbour, a parameter
  larger than the cap closes its bucket deliberately, a small tail joins the previous bucket;
* gradient accumulation (reference train.py:435-439,478-480): micro-steps inside ``no_sync()``
  accumulate into the buckets without any collective; the final micro-step reduces the sums.

Works unchanged with the ``gloo`` backend on CPU tensors (used by the unit tests).
"""

from __future__ import annotations

import contextlib
from typing import List, Optional

import torch
import torch.distributed as dist


class _Bucket:
    __slots__ = ("flat", "params", "pending", "launched", "work")

    def __init__(self, flat, params):
        self.flat, self.params = flat, params
        self.pending, self.launched, self.work = len(params), False, None


def plan_buckets(sizes_bytes, first_cap, cap, min_bytes, last_cap=None):
    """Greedy partition of a sequence of parameter sizes (bytes, in gradient-arrival order) into buckets.

    Returns a list of index lists.  Rules: close the current bucket before a parameter that would overflow the cap,
    unless the bucket is still smaller than ``min_bytes`` (then the parameter joins it: a tiny head is merged into its
    neighbour); a bucket that reached the cap closes (so an oversize parameter ends up alone or with such a head);
    a tail smaller than ``min_bytes`` joins the previous bucket.  The first bucket uses ``first_cap``.  With
    ``last_cap`` the final bucket -- the only one whose collective cannot overlap backward -- is cut so that it holds
    at most ``last_cap`` bytes of the last-arriving parameters (at least one)."""
    plans, cur, cur_bytes, limit = [], [], 0, first_cap
    for i, nbytes in enumerate(sizes_bytes):
        if cur and cur_bytes + nbytes > limit and cur_bytes >= min_bytes:
            plans.append(cur)
            cur, cur_bytes, limit = [], 0, cap
        cur.append(i)
        cur_bytes += nbytes
        if cur_bytes >= limit:
            plans.append(cur)
            cur, cur_bytes, limit = [], 0, cap
    if cur:
        if plans and cur_bytes < min_bytes:
            plans[-1].extend(cur)
        else:
            plans.append(cur)
    if last_cap is not None and plans:
        tail, tail_bytes = [], 0
        while len(plans[-1]) > 1 and tail_bytes + sizes_bytes[plans[-1][-1]] <= last_cap:
            tail_bytes += sizes_bytes[plans[-1][-1]]
            tail.insert(0, plans[-1].pop())
        if tail and tail_bytes >= min_bytes:
            plans.append(tail)
        else:
            plans[-1].extend(tail)
    return plans


class BucketedGradReducer:
    """Bucketed, backward-overlapped gradient averaging.

    Usage::

        reducer = BucketedGradReducer(model)          # after dist.init_process_group
        loss.backward(); reducer.finish()             # grads are now the rank average
        ...optimizer.step(); reducer.zero_grad()

        with reducer.no_sync():                       # gradient accumulation: no collective
            loss_1.backward()
        loss_2.backward(); reducer.finish()           # reduces the accumulated sums
    """

    def __init__(self, module: torch.nn.Module, process_group=None, first_bucket_mb: float = 1.0,
                 bucket_mb: float = 8.0, min_bucket_mb: float = 0.25, last_bucket_mb: float = 0.5,
                 broadcast_parameters: bool = True,
                 always_reduce: bool = False, rebuild_from_arrival: bool = True):
        if not dist.is_initialized():
            raise RuntimeError("BucketedGradReducer needs an initialised torch.distributed process group")
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.backend = dist.get_backend(process_group)
        self.always_reduce = always_reduce  # issue the collectives even at world size 1 (single-GPU bring-up)
        self._caps = (int(first_bucket_mb * 2 ** 20), int(bucket_mb * 2 ** 20), int(min_bucket_mb * 2 ** 20),
                      int(last_bucket_mb * 2 ** 20) if last_bucket_mb else None)
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise RuntimeError("no trainable parameters")
        for p in self.params:
            if p.dtype != torch.float32:
                raise RuntimeError("BucketedGradReducer handles fp32 parameters only")
        self._names = {p: n for n, p in module.named_parameters()}
        self.buckets: List[_Bucket] = []
        self._bucket_of, self._index_of = {}, {}
        self._build(list(reversed(self.params)), carry_grads=False)  # provisional: ~ the order backward yields gradients
        self._sync, self._next, self._hooks_fired = True, 0, 0
        self._arrival: Optional[list] = [] if rebuild_from_arrival else None
        self.rebuilt = not rebuild_from_arrival
        self.launch_log: list = []  # (bucket index, hooks fired when it launched) of the running backward
        self.last_launch_log: list = []  # ... of the last finished step
        self.hooks_in_last_backward = 0
        if broadcast_parameters and self.world > 1:
            self.broadcast_state(module)
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    # -- construction helpers ------------------------------------------------------------------
    def _build(self, order, carry_grads):
        old = {p: p.grad for p in order} if carry_grads else {}
        plans = plan_buckets([p.numel() * 4 for p in order], *self._caps)
        self.buckets, self._bucket_of, self._index_of = [], {}, {}
        for plan in plans:
            params = [order[i] for i in plan]
            flat = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=params[0].device)
            off = 0
            for p in params:
                view = flat[off:off + p.numel()].view_as(p)
                if p in old and old[p] is not None:
                    view.copy_(old[p])
                p.grad = view  # grads live inside the bucket
                off += p.numel()
            b = _Bucket(flat, params)
            for p in params:
                self._bucket_of[p] = b
                self._index_of[p] = len(self.buckets)
            self.buckets.append(b)

    def _rebuild_from_arrival(self):
        """Re-pack the buckets in the order gradients arrived in the first synchronised backward (rank 0's order, so
        every rank issues identical collectives); parameters that produced no gradient keep their provisional place at
        the end.  Called from finish(), after the step's collectives completed: gradients are carried over."""
        seen = set()
        order_idx = []
        pos = {p: i for i, p in enumerate(self.params)}
        for p in self._arrival:
            if p not in seen:
                seen.add(p)
                order_idx.append(pos[p])
        order_idx += [pos[p] for p in reversed(self.params) if p not in seen]
        idx = torch.tensor([len(seen)] + order_idx, dtype=torch.int64, device=self.params[0].device)
        if self.world > 1:
            dist.broadcast(idx, src=0, group=self.group)
        idx = idx.tolist()
        if idx[0] == 0:  # rank 0 saw no gradient this step (constant-zero loss): keep the provisional plan, try again
            self._arrival = []
            return
        self._build([self.params[i] for i in idx[1:]], carry_grads=True)
        self._arrival, self.rebuilt = None, True

    def broadcast_state(self, module):
        """Rank 0's parameters and buffers become everyone's (done once, at wrap time)."""
        with torch.no_grad():
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t, src=0, group=self.group)

    # -- per-step machinery --------------------------------------------------------------------
    def _launch(self, b: _Bucket):
        b.launched = True
        self.launch_log.append((len(self.launch_log), self._hooks_fired))
        if self.world == 1 and not self.always_reduce:
            return
        if self.backend == "nccl":
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        else:
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _launch_ready(self):
        # strictly in index order: bucket i goes out only after buckets 0..i-1 (same sequence on every rank)
        while self._next < len(self.buckets) and self.buckets[self._next].pending == 0:
            self._launch(self.buckets[self._next])
            self._next += 1

    def _on_grad(self, p):
        b = self._bucket_of[p]
        if p.grad is None or p.grad.data_ptr() < b.flat.data_ptr() or \
                p.grad.data_ptr() >= b.flat.data_ptr() + b.flat.numel() * 4:
            raise RuntimeError("parameter .grad was detached from its bucket; use reducer.zero_grad() "
                               "instead of optimizer.zero_grad(set_to_none=True)")
        if not self._sync:
            return  # accumulation micro-step: the sum stays local
        if b.launched or b.pending == 0:
            raise RuntimeError(f"gradient of {self._names.get(p, '?')} arrived again before finish(): a second "
                               "backward in one step must run inside reducer.no_sync() (gradient accumulation)")
        self._hooks_fired += 1
        if self._arrival is not None:
            self._arrival.append(p)
        b.pending -= 1
        if b.pending == 0:
            self._launch_ready()

    @contextlib.contextmanager
    def no_sync(self):
        """Backward passes inside accumulate into the buckets; nothing is reduced (reference train.py:435-439: the
        optimizer -- here: the collective -- only runs on the last micro-step of an accumulation window)."""
        if self._hooks_fired:
            raise RuntimeError("no_sync() entered between a synchronised backward and finish()")
        prev, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = prev

    def finish(self):
        """Block the current stream until every bucket is reduced; re-arm for the next step."""
        for b in self.buckets[self._next:]:  # incomplete buckets (no gradient this step, or no backward at all on
            self._launch(b)                  # this rank): reduced all the same, in index order
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
                if self.backend != "nccl":
                    b.flat.div_(self.world)
            b.work, b.launched, b.pending = None, False, len(b.params)
        self.hooks_in_last_backward, self._hooks_fired, self._next = self._hooks_fired, 0, 0
        self.last_launch_log, self.launch_log = self.launch_log, []
        if self._arrival is not None and (self.hooks_in_last_backward or self.world > 1):
            # world > 1: every rank must take part in the order broadcast, also one whose backward never ran
            self._rebuild_from_arrival()

    def zero_grad(self):
        for b in self.buckets:
            b.flat.zero_()

    def remove(self):
        for h in self._handles:
            h.remove()

    @property
    def bucket_megabytes(self):
        return [b.flat.numel() * 4 / 2 ** 20 for b in self.buckets]

    def bucket_summary(self):
        """[(MB, first parameter name, last parameter name, #params)] per bucket, for logs and DESIGN.md."""
        return [(b.flat.numel() * 4 / 2 ** 20, self._names.get(b.params[0]), self._names.get(b.params[-1]),
                 len(b.params)) for b in self.buckets]


def _parse_cpulist(text):
    """'0-7,16-23' -> {0..7, 16..23} (the format of sysfs ``local_cpulist``)."""
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_local_cpus(local_rank, sysfs="/sys/class/drm", visible=None):
    """CPUs of the NUMA node the ``local_rank``-th visible AMD GPU hangs off, read from sysfs WITHOUT touching the HIP runtime
    (``<sysfs>/card*/device/{vendor, numa_node, local_cpulist}``; GPUs ordered by PCI address, ``HIP_VISIBLE_DEVICES`` /
    ``ROCR_VISIBLE_DEVICES`` index lists honoured).  Returns (cpu set | None, numa node | None); ``gpu_local_cpus.pci`` holds the
    PCI address of the card the answer was read from (set_rank_affinity records it for the later cross-check)."""
    import os

    cards = {}
    try:
        names = sorted(os.listdir(sysfs))
    except OSError:
        return None, None
    for name in names:
        if not name.startswith("card") or not name[4:].isdigit():
            continue
        dev = os.path.join(sysfs, name, "device")
        try:
            if open(os.path.join(dev, "vendor")).read().strip().lower() != "0x1002":
                continue
            if not os.path.exists(os.path.join(dev, "mem_info_vram_total")):
                continue  # an AMD display function without VRAM: not a compute device
            cards[os.path.basename(os.path.realpath(dev))] = dev
        except OSError:
            continue
    devs = [cards[k] for k in sorted(cards)]
    # The visibility lists COMPOSE: ROCR_VISIBLE_DEVICES filters what the runtime sees, then HIP_VISIBLE_DEVICES (or its alias
    # CUDA_VISIBLE_DEVICES, which HIP honours too) indexes into that.  `visible` (tests) = one explicit list.
    lists = [visible] if visible is not None else [os.environ.get("ROCR_VISIBLE_DEVICES"),
                                                     os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES"))]
    for vis in lists:
        if vis is not None and vis.strip() == "" and visible is None:
            return None, None   # an EMPTY visibility variable means "no devices" to the runtime: nothing to pin to
        if vis:
            try:
                devs = [devs[int(i)] for i in vis.split(",") if i.strip() != ""]
            except (ValueError, IndexError):
                return None, None  # UUID lists etc.: do not guess
    if not 0 <= local_rank < len(devs):
        return None, None
    gpu_local_cpus.pci = os.path.basename(os.path.realpath(devs[local_rank]))
    try:
        node = int(open(os.path.join(devs[local_rank], "numa_node")).read().strip())
        cpus = _parse_cpulist(open(os.path.join(devs[local_rank], "local_cpulist")).read())
    except (OSError, ValueError):
        return None, None
    return (cpus or None), (node if node >= 0 else None)


AFFINITY = {"set": False, "why": "not attempted"}  # what init_distributed did, for the bench line
_AFFINITY_PCI = None  # PCI address (sysfs) of the card the mask was taken from


def check_affinity_device(device_index):
    """After the device is up: does the runtime's device sit at the PCI address the affinity was taken from?  HIP's enumeration
    order need not be PCI order; a mismatch costs performance only (a rank pinned to the other socket), so it is RECORDED in
    AFFINITY (and thus in the bench line), never raised."""
    if not AFFINITY.get("set") or _AFFINITY_PCI is None:
        return AFFINITY
    try:
        props = torch.cuda.get_device_properties(device_index)
        dom, bus = _AFFINITY_PCI.split(":")[:2]   # "dddd:bb:dd.f"
        ok = int(props.pci_bus_id) == int(bus, 16)
        if hasattr(props, "pci_domain_id"):        # several PCI domains on one node: the bus byte alone is ambiguous
            ok = ok and int(props.pci_domain_id) == int(dom, 16)
        AFFINITY["pci_matches_runtime"] = bool(ok)
    except Exception as e:  # noqa: BLE001  (a property this build lacks: say so)
        AFFINITY["pci_matches_runtime"] = f"unknown ({type(e).__name__})"
    return AFFINITY


def set_rank_affinity(local_rank, sysfs="/sys/class/drm"):
    """Pin this process (and every thread it creates from now on: HIP runtime, RCCL proxies, autograd) to the cores of its
    GPU's NUMA node -- eight ranks' Python, MIOpen database copies and OHEM read-backs otherwise land wherever the scheduler
    puts them, across the socket from the GPU they feed (the scaling risk is host-side).  Called
    BEFORE the first GPU call so that the runtime's threads inherit the mask.  Never fails: without the sysfs data, with a
    single NUMA node, or with CABINET_NO_AFFINITY=1 the mask is left alone."""
    import os

    global AFFINITY
    if os.environ.get("CABINET_NO_AFFINITY") == "1" or not hasattr(os, "sched_setaffinity"):
        AFFINITY = {"set": False, "why": "disabled"}
        return AFFINITY
    global _AFFINITY_PCI
    gpu_local_cpus.pci = None
    cpus, node = gpu_local_cpus(local_rank, sysfs)
    _AFFINITY_PCI = gpu_local_cpus.pci   # only this caller records the card for check_affinity_device
    if not cpus:
        AFFINITY = {"set": False, "why": "no NUMA information for this GPU in sysfs"}
        return AFFINITY
    allowed = os.sched_getaffinity(0)
    target = cpus & allowed
    if not target or target == allowed:
        AFFINITY = {"set": False, "numa_node": node, "why": "the GPU's node covers every allowed core" if target else
                    "none of the node's cores is allowed for this process"}
        return AFFINITY
    try:
        os.sched_setaffinity(0, target)
    except OSError as e:
        AFFINITY = {"set": False, "why": f"sched_setaffinity: {e}"}
        return AFFINITY
    AFFINITY = {"set": True, "numa_node": node, "cpus": len(target)}
    return AFFINITY


def init_distributed(backend: Optional[str] = None):
    """Initialise from torchrun's environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  ``backend`` (or the
    environment variable CABINET_DIST_BACKEND) selects the transport: ``nccl`` (= RCCL over xGMI, the default on GPUs, one
    rank per device) or ``gloo`` (device tensors staged through the host: lets several ranks share ONE GPU, which RCCL
    refuses -- the two-rank rehearsal of ``bench.py --gpus 2`` on a single-GPU box)."""
    import os

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force = os.environ.get("CABINET_FORCE_DDP") == "1"  # exercise the RCCL path on a single GPU (tests / bring-up)
    share = os.environ.get("CABINET_SHARE_GPU") == "1"  # every rank on device 0 (gloo rehearsal on a single-GPU box)
    if world > 1:
        set_rank_affinity(0 if share else local)  # before the first GPU call below; all ranks of a shared GPU take ITS node
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = backend or os.environ.get("CABINET_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if share:
        local = 0
    if world > 1 and torch.cuda.is_available():
        check_affinity_device(local)
    return rank, local, world
