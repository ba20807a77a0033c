"""Online finetuning of DTOID on MI355X: flat parameter/gradient buffers, a one-launch AMSGrad step, and
data-parallel gradient averaging over RCCL.

Reference behaviour being reproduced (/root/reference/python/ossid/scripts/online_learning.py):
  optimizer  torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-6, amsgrad=True)          :258-263
  step       model.train(); out = model(batch); optimizer.zero_grad(); out['loss'].backward(); optimizer.step()
                                                                                                      :650-679
  multi-GPU  none in the online loop; the contract is PyTorch-Lightning DDP's (train.py:93-102): per-rank BatchNorm
             statistics, gradients averaged over ranks every step.

MI355X-first layout: all 34 M parameters live in ONE contiguous fp32 buffer (every nn.Parameter is a view into it)
and all gradients in a second one. The optimizer is then a single HBM-bound kernel over 5 x 136 MB instead of
~600 x 5 small launches, and the DDP exchange is a handful of large all-reduces on the flat gradient buffer
(xGMI rings are per-link bound: few big messages, not one per tensor), issued asynchronously and waited once.
Parameters that never receive a gradient (the SqueezeNet classifier and 3-channel stem the reference keeps but never
runs, network.py:199-209) sit in the tail of the buffers and are left untouched, exactly as torch.optim.Adam skips
parameters whose .grad is None.
"""
import torch

from .. import _lib

_UNUSED_MARKERS = (".backbone.classifier.", ".backbone.features.0.")


def _is_unused(name):
    return any(m in "." + name for m in _UNUSED_MARKERS)


class FlatParams:
    """Re-homes a module's parameters and gradients into two flat buffers (used parameters first)."""

    def __init__(self, module, unused_filter=_is_unused):
        seen, used, unused = set(), [], []
        for name, p in module.named_parameters():
            if id(p) in seen:
                continue
            seen.add(id(p))
            (unused if unused_filter(name) else used).append((name, p))
        self.entries = used + unused
        dev, dt = self.entries[0][1].device, self.entries[0][1].dtype
        n_used = sum(p.numel() for _, p in used)
        self.n_used = (n_used + 3) // 4 * 4                      # the fused step walks float4s
        total = self.n_used + sum(p.numel() for _, p in unused)
        self.total = (total + 3) // 4 * 4
        self.param = torch.zeros(self.total, dtype=dt, device=dev)
        self.grad = torch.zeros(self.total, dtype=dt, device=dev)
        off = 0
        self.offsets = {}
        for i, (name, p) in enumerate(self.entries):
            if i == len(used):
                off = self.n_used
            n = p.numel()
            view = self.param[off:off + n].view_as(p)
            view.copy_(p.data)
            p.data = view
            p.grad = self.grad[off:off + n].view_as(p)
            self.offsets[name] = (off, n)
            off += n

    def zero_grad(self):
        self.grad.zero_()

    # ---- "scatter" protocol: backward with p.grad = None, then ONE multi-tensor copy into the flat buffer -------------
    # With p.grad preset to a view of the flat buffer autograd ACCUMULATES (p.grad += g): one add kernel per parameter
    # per step (~600 launches, 2.6 ms of the 57 ms step). With p.grad = None it hands over the freshly computed tensor
    # at no cost, and torch._foreach_copy_ packs all of them into the flat buffer in a handful of launches.
    def detach_grads(self):
        for _, p in self.entries:
            p.grad = None

    def gather_grads(self):
        if not hasattr(self, "_views"):
            self._views = [self.grad[off:off + n].view_as(p) for (name, p), (off, n) in
                           zip(self.entries, (self.offsets[name] for name, _ in self.entries))]
        dst, src, missing = [], [], False
        for (_, p), v in zip(self.entries, self._views):
            if p.grad is None:
                missing = True
            else:
                dst.append(v)
                src.append(p.grad)
        if missing:
            self.grad.zero_()          # a parameter without a gradient this step contributes zero (rare: unused branches)
        if dst:
            torch._foreach_copy_(dst, src)
        for (_, p), v in zip(self.entries, self._views):
            p.grad = v

    def used_grad(self):
        return self.grad[: self.n_used]


class FusedAMSGrad:
    """torch.optim.Adam(amsgrad=True) semantics as one HIP launch over FlatParams (ossid_amsgrad_step)."""

    def __init__(self, flat, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-6):
        self.flat, self.lr, self.betas, self.eps, self.weight_decay = flat, lr, betas, eps, weight_decay
        n = flat.n_used
        self.exp_avg = torch.zeros(n, dtype=flat.param.dtype, device=flat.param.device)
        self.exp_avg_sq = torch.zeros_like(self.exp_avg)
        self.max_exp_avg_sq = torch.zeros_like(self.exp_avg)
        self.step_count = 0
        self._params = [p for _, p in flat.entries]

    def zero_grad(self, set_to_none=False):
        self.flat.zero_grad()

    def step(self):
        f = self.flat
        _lib.require_cuda(f.param)
        self.step_count += 1
        with torch.cuda.device(f.param.device):
            rc = _lib.fn("ossid_amsgrad_step")(f.param.data_ptr(), f.grad.data_ptr(), self.exp_avg.data_ptr(),
                                               self.exp_avg_sq.data_ptr(), self.max_exp_avg_sq.data_ptr(), f.n_used,
                                               self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                                               self.step_count, _lib.stream())
        _lib.check(rc, "ossid_amsgrad_step")
        # the kernel wrote the parameters behind autograd's back: bump their version counters so that anything keyed on
        # them (the packed test-time plans of dtoid.Network) notices the update
        torch.autograd.graph.increment_version(self._params)

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "max_exp_avg_sq": self.max_exp_avg_sq}


class GradSync:
    """Averages the flat gradient buffer over the ranks of a process group: a few large asynchronous all-reduces
    (RCCL when the backend is "nccl"; gloo in the CPU tests), one wait."""

    def __init__(self, flat, process_group=None, bucket_mb=32):
        import torch.distributed as dist
        self.dist, self.flat, self.group = dist, flat, process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        n = flat.n_used
        per = max(1, int(bucket_mb * (1 << 20) // 4))
        self.bounds = [(s, min(n, s + per)) for s in range(0, n, per)]

    def sync(self):
        if self.world == 1:
            return
        g = self.flat.used_grad()
        g.mul_(1.0 / self.world)          # pre-scale once; SUM of the scaled buffers is the mean
        host_staged = g.is_cuda and self.dist.get_backend(self.group) != "nccl"
        if host_staged:
            # gloo moves device tensors through host memory and makes the stream wait on a HOST-signalled event. With
            # several ranks sharing one GPU (the rehearsal set-up, never the product one) such a device-side wait can
            # hold the hardware queue the other rank needs to get its half done: seconds per step until the scheduler
            # preempts (measured: 24 s/step). Draining the device first keeps the wait on the host.
            torch.cuda.synchronize(g.device)
        works = [self.dist.all_reduce(g[a:b], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
                 for a, b in self.bounds]
        for w in works:
            w.wait()
        if host_staged:
            torch.cuda.synchronize(g.device)

    def broadcast_params(self, src=0):
        """Make every replica start from rank `src`'s weights (DDP's constructor does the same)."""
        if self.world > 1:
            self.dist.broadcast(self.flat.param, src=src, group=self.group)


class GraphedForwardBackward:
    """zero_grad + DtoidNet.forward + loss.backward() of one fixed batch shape, captured once in a hipGraph and
    replayed (the eager step is ~3 700 launches and partly host-bound). Gradients land in the FlatParams buffer, whose
    address never changes; the gradient all-reduce and the one-launch optimizer step stay outside the graph.
    BatchNorm buffers are saved and restored around the warm-up passes, so capturing changes no state."""

    def __init__(self, model, flat, example_batch, warmup=2):
        self.model, self.flat = model, flat
        self.static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in example_batch.items()}
        dev = flat.param.device
        saved = [b.detach().clone() for b in model.buffers()]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                flat.detach_grads()
                model(self.static)["loss"].backward()
                flat.gather_grads()
        torch.cuda.current_stream(dev).wait_stream(side)
        with torch.no_grad():
            for b, s0 in zip(model.buffers(), saved):
                b.copy_(s0)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            flat.detach_grads()
            out = model(self.static)
            out["loss"].backward()
            flat.gather_grads()
            self.loss = out["loss"].detach()
        flat.zero_grad()

    def __call__(self, batch):
        for k, v in batch.items():
            if torch.is_tensor(v):
                self.static[k].copy_(v, non_blocking=True)
        self.graph.replay()
        return self.loss


def finetune_step(model, batch, optimizer, sync=None, graphed=None):
    """One finetune iteration on a batch already on the device; returns the detached loss. `graphed`: a
    GraphedForwardBackward built for this batch shape (optional)."""
    if graphed is not None:
        loss = graphed(batch)
        if sync is not None:
            sync.sync()
        optimizer.step()
        return loss
    return _finetune_step_eager(model, batch, optimizer, sync)


def _finetune_step_eager(model, batch, optimizer, sync=None):
    """One finetune iteration on a batch already on the device; returns the detached loss."""
    out = model(batch)
    loss = out["loss"]
    flat = getattr(optimizer, "flat", None)
    if flat is not None:               # FusedAMSGrad over FlatParams: gradients gathered by one multi-tensor copy
        flat.detach_grads()
        loss.backward()
        flat.gather_grads()
    else:
        optimizer.zero_grad()
        loss.backward()
    if sync is not None:
        sync.sync()
    optimizer.step()
    return loss.detach()


def finetuneDtoid(model, train_set, optimizer, batch_size=8, epochs=5, num_workers=0, collate_fn=None, sync=None,
                  sampler=None):
    """Counterpart of online_learning.py:650-679: `epochs` passes over `train_set` in shuffled batches, BatchNorm in
    training mode, then back to eval. `optimizer` may be a torch optimizer (the reference's) or FusedAMSGrad."""
    from torch.utils.data import DataLoader
    loader = DataLoader(train_set, batch_size=batch_size, num_workers=num_workers, shuffle=sampler is None,
                        sampler=sampler, pin_memory=True, collate_fn=collate_fn)
    dev = next(model.parameters()).device
    model.train()
    losses = []
    for _ in range(epochs):
        for batch in loader:
            batch = {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
            losses.append(finetune_step(model, batch, optimizer, sync))
    model.eval()
    return [float(v) for v in losses]
